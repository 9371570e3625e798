"""Profiling driver: N variational sweeps at the bench workload, nothing else."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
N = int(os.environ.get('SEG', 50000)); R = int(os.environ.get('RST', 16)); K = os.environ.get('NBRK'); it = int(os.environ.get('ITERS', 2))
mcn = int(os.environ.get('MAXCN', 8))
e = synthetic.make_experiment(N, num_clones=3, max_copy_number=mcn, num_chains=int(os.environ.get('CHAINS', 23)), seed=0, num_breakpoints=int(K) if K else None,
                              chain_fractions=synthetic.HUMAN_CHROMOSOME_MB if os.environ.get('UNEQUAL') else None)
ps = synthetic.make_init_params(e, R, mcn)
from remixt_amd import _lib
if os.environ.get('STAMPS'):      # an alternative build of the library (per-phase cycle counters, experiments): STAMPS=<name> -> tools/micro/lib_<name>.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'lib_%s.so' % os.environ['STAMPS'])
from remixt_amd import bpmodel
DEBUG = bool(os.environ.get('FB_DEBUG'))
if DEBUG:
    bpmodel.set_default_option('fb_debug', 1)
opts = {}
if os.environ.get('NV'):
    opts['fb_nv'] = int(os.environ['NV'])
if os.environ.get('ONE_STREAM'):
    opts['two_streams'] = 0
if os.environ.get('FB_KERNEL'):
    opts['fb_kernel'] = int(os.environ['FB_KERNEL'])
rs = RestartSet(e, ps, mcn, num_clones=3, quiet=True, options=opts)
b = rs.batch
print('info rpt,P,NT,BLK,lds,nfast,ngen', [b.info(i) for i in (5, 6, 7, 8, 9, 10, 11)], 'NBE', b.info(3), 'S', b.num_cn_states)
def sweep(n):
    try:
        b.variational_update(n)
    except (ValueError, AssertionError) as err:      # (experimental builds of the kernel may produce garbage: timing only)
        print('sweep raised:', str(err)[:80])
sweep(1)
b.profile_reset(); b.profile_enable(True)
b.synchronize(); t0 = time.time()
sweep(it)
b.synchronize(); dt = time.time() - t0
print('wall ms per sweep', dt / it * 1e3)
print('forward-backward kernel %d, restarts per workgroup (shapes) %d .. %d' % (b.info(12), b.info(13), b.info(15)))
for k, v in sorted(b.profile().items(), key=lambda kv: -kv[1][0]):
    print('%-28s %9.3f ms  n=%d  avg %.3f' % (k, v[0], v[1], v[0] / v[1]))

if DEBUG:
    c0, w0, c1, w1, ln = [b.info(i) for i in (20, 21, 22, 23, 24)]
    print('debug: steps', ln, 'shader cycles/step', (c1 - c0) / max(ln - 1, 1), 'wall us/step', (w1 - w0) / 100.0 / max(ln - 1, 1), 'clock GHz', (c1 - c0) / ((w1 - w0) * 10.0))
    nbe = max(b.info(25), 1)
    if os.environ.get('PSTAMPS'):      # a -DRMX_FB_PSTAMPS build: per-wave cycles of a plain step (products / tail to the barrier / in the barrier)
        for slot, wv in enumerate(('0', '4', '8', '3', '7', 'last')):
            print('plain-step stamps wave %-4s: products %6d  tail %6d  barrier %6d  (cycles per step)' % ((wv,) + tuple(round(b.info(28 + 3 * slot + i) / max(ln - 1, 1)) for i in range(3))))
        sys.exit(0)
    for wv in range(3):
        print('breakend-step stamps wave %d (cycles per breakend step of chain 0: entry wait+barrier / walk+fetch issue / products / finish):' % (4 * wv), [round(b.info(28 + 6 * wv + i) / nbe) for i in range(4)], 'breakend steps', nbe)
