"""Device time of the h M-step's batched objective (k_state_tables_list_v + k_ell_list_batch_sparse_grad_final) and of a round of the
shared parameter searches, alone on the GPU, by HIP events: 8 restarts of the benchmark configuration, 200-segment samples."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
if os.environ.get('STAMPS'):
    from remixt_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'lib_%s.so' % os.environ['STAMPS'])
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 8, 8)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=[1000 + i for i in range(8)])
rs.variational_update(5)
b = rs.batch
samples, lists = rs._samples_and_lists()
b.set_sample_lists([(r, -1, samples[r], lists[r]) for r in range(8)])
ev = b.h_batch_evaluator(list(range(8)))
hs = np.array([np.asarray(m.model.h, dtype=float) for m in rs.models])
ev(list(range(8)), hs)
b.profile_reset(); b.profile_enable(1)
t0 = time.perf_counter()
for i in range(50):
    ev(list(range(8)), hs * (1. + 1e-3 * (i % 3)))
dt = (time.perf_counter() - t0) / 50
b.profile_enable(0)
print('h objective round: %.1f us wall per call' % (dt * 1e6))
for k, v in sorted(b.profile().items(), key=lambda kv: -kv[1][0]):
    print('   %-24s %8.1f us per launch  n=%d' % (k, v[0] / v[1] * 1e3, v[1]))
