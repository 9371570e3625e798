"""How long does the preparation of the M-step's samples take on the host (one restart group of 8 at the benchmark configuration)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 8, 8)
for threads in (8, 1):
    rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=[1000 + i for i in range(8)], mstep_threads=threads)
    rs.variational_update(2)
    rs.batch.synchronize()
    names = rs._multi_param_names()
    for rep in range(3):
        t0 = time.perf_counter(); f = rs.batch.fetch_indicators(); t1 = time.perf_counter()
        s0 = time.perf_counter(); sm = float(f[0][0][:, 1].sum()); s1 = time.perf_counter()
        a0 = time.perf_counter(); rs._draw_param_samples(names); a1 = time.perf_counter()
        h0 = time.perf_counter(); rs._samples_and_lists(); h1 = time.perf_counter()
        m = rs.models[0]
        w0 = time.perf_counter(); wc = m.get_param_sample_weight('negbin_r_1', as_column=True); w1 = time.perf_counter()
        d0 = time.perf_counter(); idx = m._draw_sample_indices(wc); d1 = time.perf_counter()
        print('threads %d: fetch %.0f us | one column sum %.0f us | _draw_param_samples %.0f us | _samples_and_lists %.0f us | weight column %.0f us | one draw %.0f us'
              % (threads, (t1 - t0) * 1e6, (s1 - s0) * 1e6, (a1 - a0) * 1e6, (h1 - h0) * 1e6, (w1 - w0) * 1e6, (d1 - d0) * 1e6))
