"""Diagnostic: smoothness / consistency of E[ll](h) and its gradient, HIP vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers as H
from remixt_amd import bpmodel as hip
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
a, h, _ = H.make_model(hip, N=N, M=3, max_cn=8, chains=4, seed=1)
b, _, _ = H.make_model(orc, N=N, M=3, max_cn=8, chains=4, seed=1)
ma, mb = H.attach(a, h), H.attach(b, h)
a.variational_update(); a.variational_update()
for name in ('posterior_marginals', 'p_outlier_total', 'p_outlier_allele', 'p_allele_swap'):
    setattr(mb, name, getattr(ma, name))
rng = np.random.RandomState(0)
sample = np.zeros(ma.num_segments, dtype=int); sample[rng.choice(ma.num_segments, 80, replace=False)] = 1
h0 = np.array(ma.h)
g = np.zeros(3); go = np.zeros(3)
ma.calculate_expected_log_likelihood_partial_h(sample, g); mb.calculate_expected_log_likelihood_partial_h(sample, go)
print('grad hip', g, 'oracle', go, 'rel', np.abs(g - go) / np.abs(go))
d = g / np.linalg.norm(g)
f0 = ma.calculate_expected_log_likelihood(sample); f0o = mb.calculate_expected_log_likelihood(sample)
print('f0', f0, f0o, (f0 - f0o) / abs(f0o))
for t in [1e-3, 1e-5, 1e-7, 1e-9, 1e-10, 1e-11, 1e-12]:
    hh = h0 + t * d * np.linalg.norm(h0)
    ma.h = hh; mb.h = hh
    f1 = ma.calculate_expected_log_likelihood(sample); f1o = mb.calculate_expected_log_likelihood(sample)
    pred = float(g @ (hh - h0))
    print('t=%.0e  hip df=%.6e  oracle df=%.6e  predicted=%.6e' % (t, f1 - f0, f1o - f0o, pred))
