"""Where the decode + result records of a fit go (bench `fit_from_init.decode_and_results_s`): the batched lattice kernels per restart group,
then the host side of collect_fit_results, with cProfile's top entries.  Usage: python tools/decode_time.py [MAXCN [VITERBI_PLAIN [option=value ...]]]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartGroups
mcn = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:      # option viterbi_plain: 0 maxima forward + arg-maxima in the trace-back (default), 2 round 4's back-pointer lattices, 1 the plain kernel
    from remixt_amd import bpmodel
    bpmodel.set_default_option('viterbi_plain', int(sys.argv[2]))
for kv in sys.argv[3:]:      # further options as name=value (e.g. viterbi_cluster=4)
    from remixt_amd import bpmodel
    bpmodel.set_default_option(kv.split('=')[0], int(kv.split('=')[1]))
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=mcn, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 16, mcn)
rs = RestartGroups(e, ps, mcn, groups=2, num_clones=3, quiet=True, seeds=list(range(16)))
for m, v in zip(rs.models, rs.calculate_elbo()):
    m.prev_elbo = float(v)
rs.run(1, 0, 5); rs.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    outs = rs._map(lambda s: s.batch.infer_cn_batch(0, len(s.models)))
    t1 = time.perf_counter()
    print('rep %d: batched lattice + backtrace + transfer, both groups side by side: %.1f ms (lattice kernel %d)' % (rep, (t1 - t0) * 1e3, rs.batches[0].info(14)))
for b_ in rs.batches:
    b_.profile_reset(); b_.profile_enable(1)
rs._map(lambda s: s.batch.infer_cn_batch(0, len(s.models)))
for k, (ms, n) in sorted(rs.profile().items()):
    if 'viterbi' in k or 'backtrace' in k:
        print('   %-14s %8.2f ms per launch of %d restarts (%d launches)' % (k, ms / max(n, 1), len(rs.sets[0].models), n))
for b_ in rs.batches:
    b_.profile_enable(0)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = rs.results()
pr.disable()
t1 = time.perf_counter()
print('results(): %.1f ms for %d restarts' % ((t1 - t0) * 1e3, len(res)))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print(s.getvalue()[:3500])
from remixt_amd.restarts import collect_fit_results
s0 = rs.sets[0]
cn_all, _ = s0.batch.infer_cn_batch(0, len(s0.models))
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for r in range(4):
    collect_fit_results(s0.models[r], e, ps[r], cn=cn_all[r])
pr.disable(); t1 = time.perf_counter()
print('collect_fit_results, one thread: %.1f ms per restart' % ((t1 - t0) * 1e3 / 4))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
