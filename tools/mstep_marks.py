"""Host-side stage times of the M-steps of free-running restart groups at the benchmark configuration: where do the
milliseconds between one sweep phase and the next go?  (RestartSet._mark stamps; mean over the timed EM iterations.)"""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
if os.environ.get('MARKS_LIB'):      # an alternative build of the library (A/B)
    from remixt_amd import _lib as _libmod
    _libmod.LIB_PATH = os.path.abspath(os.environ['MARKS_LIB'])
from remixt_amd.restarts import RestartGroups, RestartSet
MAXCN = int(os.environ.get('MAXCN', 8))
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=MAXCN, num_chains=23, seed=0)
R, G = int(os.environ.get('RST', 16)), int(os.environ.get('NGROUPS', 2))
ps = synthetic.make_init_params(e, R, MAXCN)
kw = {}
if os.environ.get('H_HALVES'):
    kw['h_halves'] = bool(int(os.environ['H_HALVES']))
if os.environ.get('OPTS'):           # e.g. OPTS=search_mode=5
    kw['options'] = dict((k, int(v)) for k, v in (kv.split('=') for kv in os.environ['OPTS'].split(',')))
rs = RestartGroups(e, ps, MAXCN, groups=G, num_clones=3, quiet=True, seeds=[1000 + i for i in range(R)], **kw)
for m, v in zip(rs.models, rs.calculate_elbo()):
    m.prev_elbo = float(v)
rs.run(3, 0, 5)
for s in rs.sets:
    s.marks = []
orig = RestartSet.em_iteration
def wrapped(self, i=0, n=5, **kw):
    self._mark('em:start')
    out = orig(self, i, n, **kw)
    self._mark('em:end')
    return out
RestartSet.em_iteration = wrapped
NIT = 8
t0 = time.time(); rs.run(NIT, 3, 5); rs.synchronize(); dt = time.time() - t0
print('%d groups: %.1f ms per step, %.0f EM iterations/s' % (G, dt / NIT * 1e3, R * NIT / dt))
print('device-driven search: %d blocks, one launch: %d' % (rs.sets[0].batch.info(52), rs.sets[0].batch.info(53)))
for g, s in enumerate(rs.sets):
    acc = collections.OrderedDict()
    prev = None
    seq = 0
    for label, t in s.marks:
        if label == 'em:start':
            prev = t; seq = 0
            continue
        key = '%02d %s' % (seq, label); seq += 1
        acc.setdefault(key, []).append((t - prev) * 1e3); prev = t
    print('group %d' % g)
    for key, v in acc.items():
        print('   %-22s %7.2f ms' % (key, float(np.mean(v))))

