import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(20000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 2, 8, num_clones=3)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=[1, 2])
rs.fit(2, 5)
p = rs.batch.get_array(0, 'posterior_marginals')
for eps in (1e-30, 1e-20, 1e-12):
    sig = p >= eps
    g = [sig[:, i:i + 64].any(axis=1) for i in (0, 64, 128)]
    print('eps', eps, 'significant states per segment: mean %.1f median %d' % (sig.sum(1).mean(), np.median(sig.sum(1))), 'groups active', [round(x.mean(), 3) for x in g])
