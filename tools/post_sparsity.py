"""How many states carry posterior mass, at all segments and at the two segments of every breakend adjacency (what the sparse
pairwise reduction lists)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 8, 8, num_clones=3)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=list(range(1, 9)))
b = rs.batch
for label, go in (('after 1 sweep', lambda: b.variational_update(1)), ('after 2 EM iterations', lambda: rs.fit(2, 5)), ('after 7 EM iterations', lambda: rs.fit(5, 5)), ('after 12 EM iterations', lambda: rs.fit(5, 5))):
    go()
    for rr in (0, 3, 7):
        p = b.get_array(rr, 'posterior_marginals')
        label = label.split(' restart')[0] + ' restart %d' % rr
        m = rs.models[0]
        brk_seg = np.unique(np.concatenate([np.nonzero(np.asarray(m.model.is_breakend_adjacency if hasattr(m.model, 'is_breakend_adjacency') else np.zeros(1)))[0]])) if False else None
        for eps in (1e-30, 1e-20, 1e-14):
            cnt = (p >= eps).sum(1)
            print(label, 'eps', eps, 'states per segment: mean %.1f median %d p90 %d p99 %d max %d' % (cnt.mean(), np.median(cnt), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max()),
                  ' mean product of neighbouring segments %.0f' % (cnt[:-1].astype(float) * cnt[1:]).mean())
