"""Where an EM iteration's wall time goes for one restart group: sweeps / h M-step / parameter M-steps / ELBO."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
R = int(os.environ.get("RST", 16))
ps = synthetic.make_init_params(e, R, 8)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=[1000 + i for i in range(R)])
for m, v in zip(rs.models, rs.calculate_elbo()):
    m.prev_elbo = float(v)
for it in range(int(os.environ.get("ITERS", 6))):
    rs.em_iteration(it, 5)
    rs.batch.synchronize()
    t = rs.phase_times
    print('iteration %d: sweeps %.1f ms, h M-step %.1f, parameter M-steps %.1f, ELBO %.1f, total %.1f' % (it, *[(t[i + 1] - t[i]) * 1e3 for i in range(4)], (t[4] - t[0]) * 1e3))
if os.environ.get('CPROF'):
    pr = cProfile.Profile(); pr.enable()
    rs.em_iteration(99, 5)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(30)
