"""The same measurement REPS times in ONE process WITHOUT close(): the previous restart groups are only dropped by reference.
With the process-wide stream pool (option stream_pool, default) a later pair of groups gets streams with the hardware-queue placement of the
first; STREAM_POOL=0 creates and destroys streams per batch (round 4: 106-111 instead of 148 EM it/s at 355 states in some runs).
    MAXCN=12 python tools/pool_repeat.py      (355 states, two paced groups of 8)
    MAXCN=8 RST=8 python tools/pool_repeat.py (a rank's share of the 64-restart job)"""
import sys, os, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import bpmodel, synthetic
from remixt_amd.restarts import RestartGroups
if os.environ.get('STREAM_POOL'):
    bpmodel.set_default_option('stream_pool', int(os.environ['STREAM_POOL']))
mcn, R, reps, nsteps = int(os.environ.get('MAXCN', 12)), int(os.environ.get('RST', 16)), int(os.environ.get('REPS', 5)), int(os.environ.get('STEPS', 10))
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=mcn, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, R, mcn)
vals = []
keep = []            # KEEP=1: the previous groups stay referenced (a caller that holds on to its fitted models): their batches and streams stay alive
for rep in range(reps):
    rs = RestartGroups(e, ps, mcn, groups=2, num_clones=3, quiet=True, seeds=[1000 + i for i in range(R)])
    for m, v in zip(rs.models, rs.calculate_elbo()):
        m.prev_elbo = float(v)
    rs.run(2, 0, 5); rs.synchronize()
    t0 = time.perf_counter(); rs.run(nsteps, 2, 5); rs.synchronize(); dt = time.perf_counter() - t0
    vals.append(R * nsteps / dt)
    b = rs.batches[0]
    print('run %d: %.1f EM it/s, %.1f ms per step, paced %s, pool: %d streams created, %d idle' % (rep, vals[-1], dt / nsteps * 1e3, rs.paced, b.info(16), b.info(17)), flush=True)
    if os.environ.get('KEEP'):
        keep.append(rs)
    del rs, b          # no close(), no gc.collect(): the batches go whenever the collector gets to them
print('states %d restarts %d stream_pool %s keep %s: min %.1f max %.1f spread %.1f %%' % (355 if mcn == 12 else 165, R, os.environ.get('STREAM_POOL', '1'), os.environ.get('KEEP', '0'), min(vals), max(vals), 100. * (max(vals) - min(vals)) / max(vals)))
