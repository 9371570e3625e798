"""Where the construction of a fit goes (bench `fit_from_init.construct_s`): cProfile over RestartGroups(...) for 16 restarts at the headline workload."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartGroups
mcn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=mcn, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 16, mcn)
rs = RestartGroups(e, ps, mcn, groups=2, num_clones=3, quiet=True, seeds=list(range(16))); rs.synchronize()      # warm: code objects, first allocations
del rs
import gc; gc.collect()
for rep in range(2):
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    rs = RestartGroups(e, ps, mcn, groups=2, num_clones=3, quiet=True, seeds=list(range(16))); rs.synchronize()
    pr.disable(); t1 = time.perf_counter()
    print('construct: %.1f ms' % ((t1 - t0) * 1e3))
    if rep == 1:
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(16); print(s.getvalue()[:3600])
    for s_ in rs.sets:
        s_.batch = None
        for m in s_.models: m.model = None
    del rs; gc.collect()
