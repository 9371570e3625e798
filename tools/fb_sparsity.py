"""How many entries of the forward / backward vectors are above eps x the row maximum -- per restart and as the union over the four
restarts of a forward-backward workgroup, in states and in k-blocks of 4 states?  (What a reduction over the live rows only would cost.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
N = int(os.environ.get('NSEG', 3000))
e = synthetic.make_experiment(N, num_clones=3, max_copy_number=8, num_chains=2, seed=0)
ps = synthetic.make_init_params(e, 4, 8, num_clones=3)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=[1, 2, 3, 4])
b = rs.batch
S = b.num_cn_states
for label, go in (('after 1 sweep', lambda: b.variational_update(1)), ('after 2 EM iterations', lambda: rs.fit(2, 5))):
    go()
    b.variational_update(1)
    F = [b.get_array(r, 'framelogprob') for r in range(4)]
    n1 = F[0].shape[0]
    lt = np.zeros((n1 - 1, S, S))
    fwd = []; bwd = []
    for r in range(4):
        rs.models[r].model.calculate_log_transmat(lt)
        W = np.exp(lt)
        f = np.exp(F[r] - F[r].max(1, keepdims=True))
        a = np.zeros((n1, S)); a[0] = f[0] / f[0].sum()
        for n in range(1, n1):
            v = (a[n - 1] @ W[n - 1]) * f[n]
            a[n] = v / v.sum()
        bb = np.zeros((n1, S)); bb[-1] = 1. / S
        for n in range(n1 - 2, -1, -1):
            v = W[n] @ (bb[n + 1] * f[n + 1])
            bb[n] = v / v.sum()
        fwd.append(a); bwd.append(bb)
    for name, arrs in (('forward', fwd), ('backward x emission', [bb * np.exp(F[r] - F[r].max(1, keepdims=True)) for r, bb in enumerate(bwd)])):
        for eps in (1e-30, 1e-22, 1e-16):
            live = [x >= eps * x.max(1, keepdims=True) for x in arrs]
            per = np.mean([l.sum(1).mean() for l in live])
            uni = np.logical_or.reduce(live)
            kb = uni[:, :(S // 4) * 4].reshape(n1, S // 4, 4).any(2)
            print('%s | %s eps %.0e: live states per restart %.1f of %d; union of 4 restarts %.1f (p90 %d, max %d); live k-blocks of 4 states %.1f of %d'
                  % (label, name, eps, per, S, uni.sum(1).mean(), np.percentile(uni.sum(1), 90), uni.sum(1).max(), kb.sum(1).mean(), S // 4))
