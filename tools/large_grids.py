"""One timed variational sweep at state grids beyond the benchmark's (M = 4 at 207 / 457 states, three clones up to 951
states), with the forward-backward / lattice kernel each grid selects (rmx_info 12 / 14): 20 000 segments, 4 restarts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
FB = {0: 'k_fb<0> (general, weights from L2)', 1: 'k_fbm', 2: 'k_fbv', 3: 'k_fbk', 4: 'k_fbq'}
VIT = {1: 'k_viterbi_reg', 2: 'k_viterbi_code', 3: 'k_viterbi', 4: 'k_viterbi_max', 5: 'k_viterbi_code_max', 6: 'k_viterbi_sad_max'}
N, R = 20000, (int(sys.argv[1]) if len(sys.argv) > 1 else 4)
ONLY_BIG = len(sys.argv) > 2 and sys.argv[2] == 'big'
OPTS = dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in sys.argv[3:])
for M, max_cn in [(3, 8), (3, 10), (3, 12), (3, 13), (3, 14), (3, 16), (3, 20), (4, 3), (4, 4), (4, 6), (4, 8)]:
    if len(sys.argv) > 2 and sys.argv[2] == 'mid' and not (M == 3 and max_cn >= 10): continue
    if ONLY_BIG and not ((M == 3 and max_cn >= 13) or (M == 4 and max_cn >= 6)): continue
    e = synthetic.make_experiment(N, num_clones=M, max_copy_number=max_cn, num_chains=23, seed=0)
    ps = synthetic.make_init_params(e, R, max_cn, num_clones=M)
    fr = (0.6, 0.4) if M == 3 else (0.5, 0.3, 0.2)
    hs = [np.array([p['h_normal']] + [p['h_tumour'] * f for f in fr]) for p in ps]
    rs = RestartSet(e, ps, max_cn, num_clones=M, quiet=True, seeds=list(range(R)), h_init=hs, options=OPTS or None)
    b = rs.batch
    b.variational_update(1)
    b.profile_reset(); b.profile_enable(True)
    b.synchronize(); t0 = time.time()
    b.variational_update(2)
    b.synchronize(); dt = (time.time() - t0) / 2
    prof = b.profile()
    b.profile_reset()
    t1 = time.time(); b.infer_cn_batch(0, R); t2 = time.time()
    pv = b.profile()
    fbms = prof['k_fb'][0] / prof['k_fb'][1]
    S = b.num_cn_states
    print('M=%d max_cn=%2d S=%4d: sweep %8.2f ms (forward-backward %8.2f ms = %5.1f TFLOP/s, %s), decode of %d restarts %7.1f ms (%s: lattice %.1f, trace-back %.1f)'
          % (M, max_cn, S, dt * 1e3, fbms, 4. * S * b.num_segments * S * R / (fbms * 1e-3) / 1e12, FB[b.info(12)], R, (t2 - t1) * 1e3, VIT[b.info(14)], pv['k_viterbi'][0], pv['k_backtrace'][0]), flush=True)
    rs.batch = None; del rs, b
