import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remixt_amd import synthetic
from remixt_amd.restarts import RestartGroups
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = 16
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, R, 8, num_clones=3)
rs = RestartGroups(e, ps, 8, groups=G, num_clones=3, device=0, quiet=True, seeds=list(range(R)))
el = rs.calculate_elbo()
for m, v in zip(rs.models, el): m.prev_elbo = float(v)
rs.em_iteration(0, 5)
rs.synchronize()
for it in range(2):
    t0 = time.perf_counter()
    rs.em_iteration(1 + it, 5)
    rs.synchronize()
    t1 = time.perf_counter()
    print('wall %.1f ms' % ((t1 - t0) * 1e3))
    for g, s_ in enumerate(rs.sets):
        t = s_.phase_times
        print('  group %d: start +%.1f  sweeps %.1f  h %.1f  params %.1f  elbo %.1f' % (g, (t[0] - t0) * 1e3, (t[1] - t[0]) * 1e3, (t[2] - t[1]) * 1e3, (t[3] - t[2]) * 1e3, (t[4] - t[3]) * 1e3))
