import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
R = int(os.environ.get("RST", 16))
ps = synthetic.make_init_params(e, R, 8)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=list(range(R)))
for m, v in zip(rs.models, rs.calculate_elbo()):
    m.prev_elbo = float(v)
rs.em_iteration(0, 5)
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
rs.em_iteration(1, 5)
pr.disable(); print('wall', time.time() - t0)
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
