"""Full-size soak: 16 restarts x 6 EM iterations in two groups; ELBO trajectories and error messages."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartGroups
R = 16
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, R, 8, num_clones=3)
rs = RestartGroups(e, ps, 8, groups=2, num_clones=3, device=0, quiet=True, seeds=list(range(R)))
el = rs.calculate_elbo()
for m, v in zip(rs.models, el): m.prev_elbo = float(v)
traj = [el.copy()]
t0 = time.time()
for it in range(6):
    traj.append(rs.run(1, it, 5).copy())
print('wall per EM iteration %.1f ms' % ((time.time() - t0) / 6 * 1e3))
traj = np.array(traj)
d = np.diff(traj, axis=0)
print('elbo diffs min per iteration:', d.min(axis=1))
print('finite:', np.isfinite(traj).all(), 'errors:', [s.error_messages for s in rs.sets])
res = rs.results()
print('ploidy', [round(r['stats']['ploidy'], 2) for r in res][:8], 'best elbo', traj[-1].max())
