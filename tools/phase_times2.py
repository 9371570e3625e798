"""Phase times of the EM iterations of two free-running restart groups (what does sharing the GPU cost each phase?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartGroups, RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
R, G = 16, int(os.environ.get('NGROUPS', 2))
ps = synthetic.make_init_params(e, R, 8)
rs = RestartGroups(e, ps, 8, groups=G, num_clones=3, quiet=True, seeds=[1000 + i for i in range(R)])
for m, v in zip(rs.models, rs.calculate_elbo()):
    m.prev_elbo = float(v)
log = [[] for _ in rs.sets]
orig = RestartSet.em_iteration
def wrapped(self, i=0, n=5, **kw):
    out = orig(self, i, n, **kw)
    log[rs.sets.index(self)].append([(self.phase_times[k + 1] - self.phase_times[k]) * 1e3 for k in range(4)])
    return out
RestartSet.em_iteration = wrapped
rs.run(2, 0, 5)
for l in log: l.clear()
t0 = time.time(); rs.run(6, 2, 5); rs.synchronize(); dt = time.time() - t0
print('%d groups: %.1f ms per step, %.0f EM iterations/s' % (G, dt / 6 * 1e3, R * 6 / dt))
for g, l in enumerate(log):
    a = np.mean(np.array(l), axis=0)
    print('group %d: sweeps %.1f ms, h M-step %.1f, parameter M-steps %.1f, ELBO %.1f  (sum %.1f)' % (g, a[0], a[1], a[2], a[3], a.sum()))
