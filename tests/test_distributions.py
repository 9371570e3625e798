"""CPU: NegBinDistribution / BetaBinDistribution (reference remixt/likelihood.py:569-662, 949-1084) against vectors recorded
from the reference's own classes, the survey's known answers, and -- the cross-check they exist for -- the scalar log pmfs of
the kernel restatement (oracle/remixt_oracle.c, bpmodel.pyx:238-394)."""
import os

import numpy as np

from remixt_amd.likelihood import BetaBinDistribution, NegBinDistribution

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'distributions.npz'))


def test_negbin_matches_reference_vectors():
    x, mu = G['nb_x'], G['nb_mu']
    for r in (500., 10., 1.5):
        d = NegBinDistribution(r=r)
        assert np.allclose(d.log_likelihood(x, mu), G['nb/%g/ll' % r], rtol=1e-13, atol=1e-13)
        assert np.allclose(d.log_likelihood_partial_mu(x, mu), G['nb/%g/dmu' % r], rtol=1e-13, atol=0)
        assert np.allclose(d.log_likelihood_partial_r(x, mu), G['nb/%g/dr' % r], rtol=1e-13, atol=1e-15)
    assert np.allclose(NegBinDistribution().log_likelihood(np.array([3., 4.]), np.array([-100., -600.])), G['nb/clip/ll'], rtol=1e-14)
    assert NegBinDistribution().r == 500.


def test_betabin_matches_reference_vectors():
    k, n, p = G['bb_k'], G['bb_n'], G['bb_p']
    for M in (500., 10., 2000.):
        d = BetaBinDistribution(M=M)
        assert np.allclose(d.log_likelihood(k, n, p), G['bb/%g/ll' % M], rtol=1e-13, atol=1e-12)
        assert np.allclose(d.log_likelihood_partial_p(k, n, p), G['bb/%g/dp' % M], rtol=1e-13, atol=1e-10)
        assert np.allclose(d.log_likelihood_partial_M(k, n, p), G['bb/%g/dM' % M], rtol=1e-12, atol=1e-14)
    assert BetaBinDistribution().M == 500.


def test_survey_known_answers():
    """SURVEY.md 8c KAT1 / KAT2."""
    kat1 = NegBinDistribution(r=500.).log_likelihood(np.array([1000., 0., 5.]), np.array([900., 1e-5, 7.]))
    assert np.allclose(kat1, [-6.793525188850197, -9.999999872691833e-06, -2.058967842587349], rtol=1e-12)
    kat2 = BetaBinDistribution(M=500.).log_likelihood(np.array([40., 0.]), np.array([100., 10.]), np.array([0.4, 1e-3]))
    assert np.allclose(kat2, [-2.6018530249712057, -0.00991603896773086], rtol=1e-11)


def test_cross_check_against_the_kernel_restatement(oracle_mod):
    """The vectorised distributions and the scalar functions of the kernel are two statements of the same pmfs."""
    rng = np.random.RandomState(3)
    L = oracle_mod.lib()
    for _ in range(200):
        x, mu, r = float(rng.poisson(1500)), float(rng.uniform(100, 5000)), float(rng.choice([500., 10., 37.]))
        d = NegBinDistribution(r=r)
        assert np.isclose(d.log_likelihood(np.array([x]), np.array([mu]))[0], oracle_mod.negbin_ll(x, mu, r), rtol=1e-10, atol=1e-10)
        assert np.isclose(d.log_likelihood_partial_mu(x, mu), L.rmxo_negbin_ll_partial_mu(x, mu, r), rtol=1e-12)
        n = float(rng.poisson(200) + 1); p = float(rng.uniform(0.01, 0.99)); k = float(rng.binomial(int(n), p)); M = float(rng.choice([500., 10., 1200.]))
        b = BetaBinDistribution(M=M)
        assert np.isclose(b.log_likelihood(k, n, p), oracle_mod.betabin_ll(k, n, p, M), rtol=1e-9, atol=1e-9)
        assert np.isclose(b.log_likelihood_partial_p(k, n, p), L.rmxo_betabin_ll_partial_p(k, n, p, M), rtol=1e-7, atol=1e-7)
