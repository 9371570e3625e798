"""CPU: the oracle (oracle/remixt_oracle.c) against the golden vectors recorded from the
reference -- this is what pins the oracle."""
import numpy as np
import pytest

from tests import golden_runner as GR


@pytest.mark.parametrize('name', GR.MODEL_CASES)
def test_oracle_replays_reference(oracle_mod, name):
    # same algorithm, same operation order: agreement to ~1 ulp of accumulated libm noise
    GR.replay(name, oracle_mod, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize('name', GR.GRID_CASES)
def test_oracle_replays_reference_at_bench_grids_and_dark_corners(oracle_mod, name):
    """165 / 355 states with 8-9 breakpoints (two breakends at one boundary), transition_model = 1, four clones,
    disable_breakpoints: pins the oracle where the benchmark runs and where the protocol is rarely exercised."""
    GR.replay_grid(name, oracle_mod, rtol=1e-11, atol=1e-13, mixed_elbo=True)


@pytest.mark.parametrize('name', GR.FIT_CASES)
def test_oracle_full_fit_trajectory(oracle_mod, name):
    """Seeded EM trajectories of the reference (scipy M-steps; the two no-normal-contamination cases run the ten-parameter
    M-step of cn_model.py:198-226).  Tolerances: an optimiser trajectory amplifies last-bit differences of the objective."""
    GR.replay_fit(name, oracle_mod, rtol_elbo=1e-9, rtol_h=1e-7, rtol_param=1e-6)


def test_no_fixture_records_a_failed_reference_fit():
    for name in GR.MODEL_CASES:
        assert int(GR.load(name)['fit/failed']) == 0, name


def test_oracle_chain_kats(oracle_mod):
    g = GR.load('chains')
    f, T = g['kat3_f'], g['kat3_T']
    a = np.zeros_like(f); b = np.zeros_like(f); ss = np.zeros(len(f), dtype=np.int64)
    oracle_mod.sum_product(f, T, a, b)
    assert np.allclose(a, g['kat3_alphas'], rtol=1e-14) and np.allclose(b, g['kat3_betas'], rtol=1e-14)
    assert oracle_mod.max_product(f, T, ss) == float(g['kat3_logprob']) and np.array_equal(ss, g['kat3_path'])
    # SURVEY 8c KAT3 literals
    from scipy.special import logsumexp
    assert np.isclose(logsumexp(a[-1]), 11.069206009930824, rtol=1e-13)
    assert list(ss) == [2, 3, 3, 2, 0, 2] and float(g['kat3_logprob']) == 4.030771537064376
    for i in range(3):
        f, T = g['ties%d_f' % i], g['ties%d_T' % i]
        ss = np.zeros(len(f), dtype=np.int64)
        assert oracle_mod.max_product(f, T, ss) == float(g['ties%d_logprob' % i])
        assert np.array_equal(ss, g['ties%d_path' % i])        # bit-exact, ties included
        f, T = g['rand%d_f' % i], g['rand%d_T' % i]
        a = np.zeros_like(f); b = np.zeros_like(f)
        oracle_mod.sum_product(f, T, a, b)
        assert np.allclose(a, g['rand%d_alphas' % i], rtol=1e-13, atol=1e-11) and np.allclose(b, g['rand%d_betas' % i], rtol=1e-13, atol=1e-11)
        ss = np.zeros(len(f), dtype=np.int64)
        assert oracle_mod.max_product(f, T, ss) == float(g['rand%d_logprob' % i]) and np.array_equal(ss, g['rand%d_path' % i])


def test_oracle_scalar_kats(oracle_mod):
    """SURVEY 8c KAT1 / KAT2 (values produced by the reference's likelihood.py distributions)."""
    kat1 = [oracle_mod.negbin_ll(x, mu, 500.) for x, mu in [(1000, 900), (0, 1e-5), (5, 7)]]
    assert np.allclose(kat1, [-6.793525188850197, -9.999999872691833e-06, -2.058967842587349], rtol=1e-12)
    kat2 = [oracle_mod.betabin_ll(k, n, p, 500.) for k, n, p in [(40, 100, 0.4), (0, 10, 1e-3)]]
    assert np.allclose(kat2, [-2.6018530249712057, -0.00991603896773086], rtol=1e-11)
    from scipy.special import digamma
    for x in [1e-7, 0.3, 1.0, 4.2, 8.5, 20., 1234.5]:
        assert np.isclose(oracle_mod.digamma(x), digamma(x), rtol=1e-9)


@pytest.mark.parametrize('case', ['m2', 'm3_nonormal'])
def test_oracle_kernel_object_replays_the_reference_fit_call_trace(oracle_mod, case):
    """The call trace of the reference's own BreakpointModel.fit (oracle/make_protocol_trace.py) against the oracle's kernel object:
    pins the oracle's object protocol call by call, and the replay the GPU suite runs against the HIP kernel object."""
    from .protocol_replay import replay
    replay(oracle_mod.RemixtModel, case, check_dir=False)
