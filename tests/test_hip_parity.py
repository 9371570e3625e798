"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Tolerance: 1e-6 relative for posteriors / likelihoods
(north_star), bit-exact for Viterbi paths given identical (f, T)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from remixt_amd import bpmodel
    return bpmodel


def _kat4(kernel):
    from remixt_amd.cn_model import BreakpointModel
    x = np.array([[600, 400, 10000], [900, 300, 15000], [620, 380, 10100], [500, 500, 9900], [700, 100, 8000]], dtype=float)
    l = np.ones(5) * 1e5
    m = BreakpointModel(x, l, {(0, 1), (1, 2), (2, 3)}, {'b0': frozenset([(0, 1), (2, 0)])}, max_copy_number=2, max_depth=1.0,
                        divergence_weight=1e-6, min_segment_length=0., kernel_module=kernel, quiet=True)
    m.num_em_iter = 0
    m.fit(np.array([0.02, 0.03]))
    return m


def test_kat4_known_answers(hip):
    """SURVEY.md 8c KAT4 values (produced by the reference)."""
    m = _kat4(hip)
    assert np.isclose(m.prev_elbo, -876.3848546093834, rtol=1e-9)
    m.variational_update()
    assert np.isclose(m.model.calculate_elbo(), -130.2844273873534, rtol=1e-9)
    assert np.isclose(m.model.hmm_log_norm_const, -187.10488424850666, rtol=1e-9)
    f = m.model.framelogprob
    assert np.allclose(f[0], [-281.6558766560208, -54.33495353659991, -42.709943204987134, -17.980631175517097], rtol=1e-9)
    T = m.model.log_transmat
    assert np.array_equal(T[0, 0], [-5, -15, -25, -25]) and np.array_equal(T[2, 0], [0, -10, -20, -20])
    assert np.allclose(m.model.p_breakpoint, [[0.99999999793884642, 2.0611536181901979e-09, 4.2483542465350965e-18]], rtol=1e-6, atol=1e-30)
    s = np.ones(5, dtype=int)
    assert np.isclose(m.model.calculate_expected_log_likelihood(s), -74.30206719862542, rtol=1e-9)
    g = np.zeros(2)
    m.model.calculate_expected_log_likelihood_partial_h(s, g)
    assert np.allclose(g, [174.32617346234437, -61.424026009374174], rtol=1e-8)
    cn, brk = m.optimal_cn()
    assert np.array_equal(cn[:, 1, :], [[1, 1], [2, 0], [1, 1], [1, 1], [2, 0]])
    assert np.array_equal(brk['b0'], [0, 0])


@pytest.mark.parametrize('M,max_cn,N,chains,nc', [(2, 4, 150, 3, True), (2, 6, 400, 4, True), (3, 3, 120, 4, True), (3, 4, 90, 1, True),
                                                 (2, 3, 100, 5, False), (3, 2, 80, 2, False)])
def test_coordinate_updates_match_oracle(hip, oracle_mod, M, max_cn, N, chains, nc):
    a, h, _ = H.make_model(hip, N=N, M=M, max_cn=max_cn, chains=chains, seed=M * 10 + max_cn, normal_contamination=nc)
    b, _, _ = H.make_model(oracle_mod, N=N, M=M, max_cn=max_cn, chains=chains, seed=M * 10 + max_cn, normal_contamination=nc)
    ma, mb = H.attach(a, h), H.attach(b, h)
    assert np.isclose(ma.calculate_elbo(), mb.calculate_elbo(), rtol=1e-9)
    H.compare_models(ma, mb, dense=True, tag='init')
    for it in range(2):
        for step in ('update_p_allele_swap', 'update_p_cn', 'update_p_breakpoint', 'update_p_outlier_total', 'update_p_outlier_allele'):
            getattr(ma, step)(); getattr(mb, step)()
            H.compare_models(ma, mb, dense=(it == 0), tag='%d/%s' % (it, step))
            ea, eb = ma.calculate_elbo(), mb.calculate_elbo()
            assert np.isclose(ea, eb, rtol=1e-8), (it, step, ea, eb)
        assert np.isclose(ma.hmm_log_norm_const, mb.hmm_log_norm_const, rtol=1e-9)
    # exact energy / entropy parts
    assert np.isclose(ma.calculate_variational_energy(), mb.calculate_variational_energy(), rtol=1e-8)
    assert np.isclose(ma.calculate_variational_entropy(), mb.calculate_variational_entropy(), rtol=1e-8)
    # M-step objectives on a random sample and on all segments
    rng = np.random.RandomState(3)
    for sample in (np.ones(ma.num_segments, dtype=int), (rng.rand(ma.num_segments) < 0.2).astype(int)):
        assert np.isclose(ma.calculate_expected_log_likelihood(sample), mb.calculate_expected_log_likelihood(sample), rtol=1e-9)
        if nc:
            ga, gb = np.zeros(M), np.zeros(M)
            ma.calculate_expected_log_likelihood_partial_h(sample, ga); mb.calculate_expected_log_likelihood_partial_h(sample, gb)
            assert np.allclose(ga, gb, rtol=1e-7), (ga, gb)
    # parameter change propagates
    ma.negbin_r_0 = 123.; mb.negbin_r_0 = 123.
    ma.h = np.asarray(h) * 1.1; mb.h = np.asarray(h) * 1.1
    s = np.ones(ma.num_segments, dtype=int)
    assert np.isclose(ma.calculate_expected_log_likelihood(s), mb.calculate_expected_log_likelihood(s), rtol=1e-9)
    assert np.isclose(ma.calculate_elbo(), mb.calculate_elbo(), rtol=1e-8)
    # Viterbi
    cna = np.zeros((ma.num_segments, M, 2), dtype=int); cnb = cna.copy()
    ma.infer_cn(cna); mb.infer_cn(cnb)
    assert np.array_equal(cna, cnb)


def _random_configs():
    rng = np.random.RandomState(2024)
    out = []
    for i in range(10):
        M = int(rng.choice([2, 3]))
        max_cn = int(rng.randint(2, 6 if M == 3 else 8))
        out.append((M, max_cn, int(rng.randint(30, 260)), int(rng.randint(1, 7)), bool(rng.rand() < 0.7), 500 + i))
    # three clones with grids past 32 states: strip kernels, cell cache, fused sweeps, register forward-backward
    out += [(3, 5, 150, 3, True, 601), (3, 6, 120, 2, True, 602), (3, 6, 90, 4, False, 603), (3, 5, 200, 5, False, 604)]
    return out


@pytest.mark.parametrize('M,max_cn,N,chains,nc,seed', _random_configs())
def test_randomized_configurations_match_oracle(hip, oracle_mod, M, max_cn, N, chains, nc, seed):
    """Seeded random problem shapes (clones, state grid, segments, chains, contamination mode): two full
    sweeps, ELBO, log Z, sampled E[ll] and the Viterbi decode against the CPU oracle."""
    a, h, _ = H.make_model(hip, N=N, M=M, max_cn=max_cn, chains=chains, seed=seed, normal_contamination=nc)
    b, _, _ = H.make_model(oracle_mod, N=N, M=M, max_cn=max_cn, chains=chains, seed=seed, normal_contamination=nc)
    ma, mb = H.attach(a, h), H.attach(b, h)
    for it in range(2):
        a.variational_update(); b.variational_update()
        H.compare_models(ma, mb, tag='sweep%d' % it)
        assert np.isclose(ma.calculate_elbo(), mb.calculate_elbo(), rtol=1e-8)
        assert np.isclose(ma.hmm_log_norm_const, mb.hmm_log_norm_const, rtol=1e-9)
    sample = (np.random.RandomState(seed).rand(ma.num_segments) < 0.3).astype(int)
    assert np.isclose(ma.calculate_expected_log_likelihood(sample), mb.calculate_expected_log_likelihood(sample), rtol=1e-9)
    cna = np.zeros((ma.num_segments, M, 2), dtype=int); cnb = cna.copy()
    ma.infer_cn(cna); mb.infer_cn(cnb)
    assert np.array_equal(cna, cnb)


def test_calculate_log_transmat_into_caller_buffer(hip, oracle_mod):
    """calculate_log_transmat(out) (bpmodel.pyx:639-684): dense transitions for the CURRENT p_breakpoint,
    neither snapshot touched."""
    a, h, _ = H.make_model(hip, N=60, M=3, max_cn=3, chains=3, seed=5)
    b, _, _ = H.make_model(oracle_mod, N=60, M=3, max_cn=3, chains=3, seed=5)
    ma, mb = H.attach(a, h), H.attach(b, h)
    a.variational_update(); b.variational_update()
    for m in (ma, mb):
        pb = np.asarray(m.p_breakpoint).copy()
        pb[:] = np.linspace(1., 2., pb.shape[1])[None, :]
        pb /= pb.sum(axis=1)[:, None]
        m.p_breakpoint = pb
    N, S = ma.num_segments, ma.num_cn_states
    ta, tb = np.full((N - 1, S, S), np.nan), np.full((N - 1, S, S), np.nan)
    snap = np.asarray(ma.log_transmat).copy()
    ma.calculate_log_transmat(ta); mb.calculate_log_transmat(tb)
    assert np.allclose(ta, tb, rtol=1e-12, atol=1e-12)
    assert np.array_equal(np.asarray(ma.log_transmat), snap) and not np.array_equal(ta, snap)


def test_module_sum_product_max_product(hip, oracle_mod):
    rng = np.random.RandomState(0)
    f = rng.rand(6, 5); T = -rng.rand(5, 5, 5)
    a = np.zeros((6, 5)); b = np.zeros((6, 5))
    hip.sum_product(f, T, a, b)
    from scipy.special import logsumexp
    assert np.isclose(logsumexp(a[-1]), 11.069206009930824, rtol=1e-12)   # KAT3
    assert np.allclose(b[0], [8.952942521751636, 8.654865359432627, 8.913017283777442, 9.021467603851274, 8.900114919476255], rtol=1e-12)
    ss = np.zeros(6, dtype=np.int64)
    lp = hip.max_product(f, T, ss)
    assert lp == 4.030771537064376 and list(ss) == [2, 3, 3, 2, 0, 2]
    # larger, with ties (integer-valued inputs): bit-exact against the oracle
    for seed, (N, S) in enumerate([(200, 47), (64, 165), (500, 9)]):
        rng = np.random.RandomState(seed)
        f = np.floor(rng.rand(N, S) * 8) - 4.
        T = -np.floor(rng.rand(N - 1, S, S) * 4) * 10.
        s1 = np.zeros(N, dtype=np.int64); s2 = np.zeros(N, dtype=np.int64)
        l1 = hip.max_product(f, T, s1); l2 = oracle_mod.max_product(f, T, s2)
        assert l1 == l2 and np.array_equal(s1, s2)
        f = rng.randn(N, S) * 5; T = -rng.rand(N - 1, S, S) * 30
        a1 = np.zeros((N, S)); b1 = np.zeros((N, S)); a2 = np.zeros((N, S)); b2 = np.zeros((N, S))
        hip.sum_product(f, T, a1, b1); oracle_mod.sum_product(f, T, a2, b2)
        assert np.allclose(a1, a2, rtol=1e-12, atol=1e-10) and np.allclose(b1, b2, rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize('max_cn', [3, 4])
def test_full_fit_matches_oracle(hip, oracle_mod, max_cn):
    """Seeded EM trajectory (variational sweeps + scipy M-steps) on both kernels; 20 states (row kernels) and
    47 states (strip kernels, cell cache, trial / rollback and sparse trial passes of the M-steps)."""
    res = []
    for kern in (hip, oracle_mod):
        m, h, e = H.make_model(kern, N=240, M=3, max_cn=max_cn, chains=4, seed=5)
        m.num_em_iter = 2; m.num_update_iter = 2
        np.random.seed(11)
        m.fit(h)
        cn, brk = m.optimal_cn()
        res.append((m.prev_elbo, m.h, m.get_likelihood_param_values(), cn, brk, m.p_outlier_total))
    (e1, h1, p1, cn1, b1, q1), (e2, h2, p2, cn2, b2, q2) = res
    assert np.isclose(e1, e2, rtol=1e-6), (e1, e2)
    assert np.allclose(h1, h2, rtol=1e-5)
    for k in p1:
        assert np.isclose(p1[k], p2[k], rtol=1e-4), (k, p1[k], p2[k])
    assert np.array_equal(cn1, cn2)
    assert all(np.array_equal(b1[k], b2[k]) for k in b1)
    assert np.allclose(q1, q2, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize('normal_contamination', [True, False])
def test_restart_driver_matches_oracle_driver(hip, oracle_mod, normal_contamination):
    """(normal_contamination=False adds the six hdel / LOH likelihood parameters to the M-step.)
    The batched restart driver on the device (lock-step M-steps through scipy's own optimiser steps) against
    the same driver over the CPU oracle (one model per restart, scipy per restart): same seeded trajectories --
    ELBO to 1e-6, decoded copy number identical -- and the native shared-round searches stay within the same
    bounds on this data."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(240, num_clones=3, max_copy_number=4, num_chains=4, seed=15)
    ps = synthetic.make_init_params(e, 2, 4)
    runs = []
    for kern, native in ((oracle_mod, False), (None, False), (None, True)):
        rs = RestartSet(e, ps, max_copy_number=4, num_clones=3, quiet=True, seeds=[3, 4], kernel_module=kern, native_search=native, mstep_threads=1,
                        normal_contamination=normal_contamination)
        assert len(rs.models[0].likelihood_params) == (4 if normal_contamination else 10)
        if not normal_contamination:
            # on this data fewer segments carry homozygous-deletion posterior mass than the M-step sample holds:
            # numpy's choice(replace=False, p=...) raises in the reference (cn_model.py:477), and so do both drivers
            with pytest.raises(ValueError, match='Fewer non-zero entries in p than size'):
                rs.fit(num_em_iter=2, num_update_iter=2)
            continue
        rs.fit(num_em_iter=2, num_update_iter=2)
        runs.append(rs.results())
    if not normal_contamination:
        return
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.isclose(a['stats']['elbo'], b['stats']['elbo'], rtol=1e-6), (a['stats']['elbo'], b['stats']['elbo'])
            np.testing.assert_allclose(a['h'], b['h'], rtol=1e-5)
            assert np.array_equal(a['cn'], b['cn'])
            assert all(np.array_equal(a['brk_cn'][k], b['brk_cn'][k]) for k in a['brk_cn'])
            np.testing.assert_allclose(a['p_outlier_total'], b['p_outlier_total'], rtol=1e-5, atol=1e-8)


def test_batch_equals_single(hip):
    """R restarts in one batch give the same numbers as R separate models."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(200, num_clones=3, max_copy_number=3, num_chains=4, seed=2)
    ps = synthetic.make_init_params(e, 3, 3)
    from remixt_amd.restarts import RestartSet
    rs = RestartSet(e, ps, max_copy_number=3, num_clones=3, quiet=True)
    rs.batch.variational_update(2)
    elbo_b = rs.batch.calculate_elbo()
    for r, p in enumerate(ps):
        m, h, _ = H.make_model(hip, N=200, M=3, max_cn=3, chains=4, seed=2, restart=r)
        mm = H.attach(m, h)
        for _ in range(2):
            m.variational_update()
        assert np.isclose(mm.calculate_elbo(), elbo_b[r], rtol=1e-12), (r, mm.calculate_elbo(), elbo_b[r])
        assert np.allclose(mm.posterior_marginals, rs.batch.get_array(r, 'posterior_marginals'), rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize('M,max_cn', [(2, 4), (3, 3), (3, 6), (3, 8)])
def test_batched_decode_matches_oracle_paths(hip, oracle_mod, M, max_cn):
    """rmx_infer_cn_batch on grids of 9 / 20 / 97 / 165 states (different register-tile widths of the
    lattice kernel): every restart's path equals the oracle's Viterbi path of the same model, bit for bit;
    before the first update_p_cn every comparison ties and the first state wins (bpmodel.pyx:557-558)."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(260, num_clones=M, max_copy_number=max_cn, num_chains=4, seed=17)
    ps = synthetic.make_init_params(e, 3, max_cn, num_clones=M)
    sets = [RestartSet(e, ps, max_copy_number=max_cn, num_clones=M, quiet=True, kernel_module=k) for k in (None, oracle_mod)]
    b = sets[0].batch
    cn0, lp0 = b.infer_cn_batch(0, 3)
    for r in range(3):
        ref = np.zeros((b.num_segments, M, 2), dtype=int); sets[1].models[r].model.infer_cn(ref)
        assert np.array_equal(cn0[r], ref)
    for rs in sets:
        rs.variational_update(2)
    cn, lp = b.infer_cn_batch(0, 3)
    for r in range(3):
        ref = np.zeros((b.num_segments, M, 2), dtype=int); sets[1].models[r].model.infer_cn(ref)
        assert np.array_equal(cn[r], ref), r
        one, lp1 = b.infer_cn(r)
        assert np.array_equal(one, cn[r]) and lp1 == lp[r]
    cn12, _ = b.infer_cn_batch(1, 2)
    assert np.array_equal(cn12, cn[1:3])
    for r0, nr in ((0, 4), (-1, 1), (2, 0), (3, 1)):
        with pytest.raises(ValueError):
            b.infer_cn_batch(r0, nr)


def test_decode_code_table_lattice_equals_plain_lattice(hip):
    """355 states (max_cn = 12): the lattice kernel that keeps 8-bit codes of the S x S transition values
    in LDS against the plain kernel that reads the tabulated doubles (option viterbi_plain), breakend and
    telomere steps included: same paths and path log-probabilities, bit for bit."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(900, num_clones=3, max_copy_number=12, num_chains=5, seed=23, num_breakpoints=30)
    ps = synthetic.make_init_params(e, 2, 12)
    rs = RestartSet(e, ps, max_copy_number=12, num_clones=3, quiet=True)
    b = rs.batch
    assert b.num_cn_states == 355
    b.variational_update(2)
    cn, lp = b.infer_cn_batch(0, 2)
    assert b.info(14) == 6 and b.info(18) == 8                               # k_viterbi_sad_max: transition values from the packed copies, eight workgroups per restart
    for w in (1, 2, 4):                                                      # one workgroup per restart is k_viterbi_code_max (8-bit codes in LDS)
        b.set_option('viterbi_cluster', w)
        cn_w, lp_w = b.infer_cn_batch(0, 2)
        assert b.info(14) == (5 if w == 1 else 6) and b.info(18) == w and np.array_equal(cn, cn_w) and np.array_equal(lp, lp_w), w
    b.set_option('viterbi_cluster', 0)
    b.set_option('viterbi_plain', 1)
    cn_plain, lp_plain = b.infer_cn_batch(0, 2)
    assert np.array_equal(cn, cn_plain) and np.array_equal(lp, lp_plain)
    b.set_option('viterbi_plain', 2)                                         # round 4's code-table lattice with back-pointers
    cn_bp, lp_bp = b.infer_cn_batch(0, 2)
    assert b.info(14) == 2 and np.array_equal(cn, cn_bp) and np.array_equal(lp, lp_bp)
    assert len(np.unique(cn[0].reshape(len(cn[0]), -1), axis=0)) > 3        # a non-trivial path


@pytest.mark.parametrize('max_cn,N,R', [(4, 2, 2), (6, 3, 2), (8, 129, 3), (8, 130, 3), (8, 1500, 3), (10, 700, 2), (12, 400, 2), (14, 300, 2)])
def test_parallel_traceback_equals_the_sequential_walk(hip, max_cn, N, R):
    """The trace-back as arg-maxima of every target state (k_bp_all) + composed maps (k_chase_compose / _ends / _fill; blocks of up to 128
    rows: two segments, three, a block exactly and one row past it, many blocks) against the sequential walk of bpmodel.pyx:1320-1331 on one wave
    (k_backtrace_max / k_backtrace_sad, option traceback = 1): same paths and log-probabilities, bit for bit, breakend and telomere steps
    included -- at 45 / 84 / 165 states (k_viterbi_max), 251 / 355 (workgroup clusters of 4 / 8) and 477 states."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(N, num_clones=3, max_copy_number=max_cn, num_chains=min(3, N), seed=31 + N, num_breakpoints=min(20, N // 2))
    rs = RestartSet(e, synthetic.make_init_params(e, R, max_cn), max_cn, num_clones=3, quiet=True)
    b = rs.batch
    b.variational_update(2)
    cn, lp = b.infer_cn_batch(0, R)
    assert b.info(19) == 1
    b.set_option('traceback', 1)
    cn_seq, lp_seq = b.infer_cn_batch(0, R)
    assert b.info(19) == 0
    assert np.array_equal(cn, cn_seq) and np.array_equal(lp, lp_seq)
    b.set_option('viterbi_plain', 1)                                        # and the table-reading lattice with back-pointers
    cn_p, lp_p = b.infer_cn_batch(0, R)
    assert np.array_equal(cn, cn_p) and np.array_equal(lp, lp_p)


def test_decode_of_many_restarts_at_once_takes_one_workgroup_per_restart(hip):
    """Lattice clusters wait for each other, so a launch holds at most 64 of their workgroups: 33 restarts of 355 states in one decode get one
    workgroup each (the code-table lattice), 16 of them clusters of 4, 8 of them clusters of 8 -- the same paths."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(60, num_clones=3, max_copy_number=12, num_chains=2, seed=77, num_breakpoints=6)
    rs = RestartSet(e, synthetic.make_init_params(e, 33, 12), max_copy_number=12, num_clones=3, quiet=True)
    b = rs.batch
    b.variational_update(1)
    cn, lp = b.infer_cn_batch(0, 33)
    assert (b.info(14), b.info(18)) == (5, 1)
    cn16, lp16 = b.infer_cn_batch(0, 16)
    assert (b.info(14), b.info(18)) == (6, 4) and np.array_equal(cn16, cn[:16]) and np.array_equal(lp16, lp[:16])
    cn8, lp8 = b.infer_cn_batch(25, 8)
    assert (b.info(14), b.info(18)) == (6, 8) and np.array_equal(cn8, cn[25:]) and np.array_equal(lp8, lp[25:])


def test_lattice_cluster_watchdog_falls_back_to_one_workgroup_per_restart(hip):
    """A member of a lattice cluster that never publishes a row (test hook: viterbi_cluster = 100 + W) must not hold its partners for ever: their waits run
    out (seconds), the launch raises its flag, and the decode is repeated with one workgroup per restart -- same paths as the undisturbed decode."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(50, num_clones=3, max_copy_number=12, num_chains=2, seed=78, num_breakpoints=5)
    rs = RestartSet(e, synthetic.make_init_params(e, 3, 12), max_copy_number=12, num_clones=3, quiet=True)
    b = rs.batch
    b.variational_update(1)
    cn, lp = b.infer_cn_batch(0, 3)
    assert (b.info(14), b.info(18), b.info(54)) == (6, 8, 0)
    b.set_option('viterbi_cluster', 104)
    cn_w, lp_w = b.infer_cn_batch(0, 3)
    assert b.info(54) == 1 and b.info(18) == 1            # the watchdog fired once; the repeat ran one workgroup per restart
    assert np.array_equal(cn_w, cn) and np.array_equal(lp_w, lp)
    b.set_option('viterbi_cluster', 0)
    cn2, lp2 = b.infer_cn_batch(0, 3)
    assert (b.info(18), b.info(54)) == (8, 1) and np.array_equal(cn2, cn) and np.array_equal(lp2, lp)


@pytest.mark.parametrize('max_cn,vit', [(4, 4), (8, 4), (12, 6)])
def test_decode_with_exact_ties_everywhere_matches_oracle(hip, oracle_mod, max_cn, vit):
    """Both likelihood masks off: a segment's frame log-probability is the subclonality prior alone (bpmodel.pyx:746-749, 898-919) -- the same
    number for every state with the same count of subclonal alleles -- and the transition values are integer multiples of the penalty, so
    the lattice is full of EXACT ties (the two sides compute identical doubles).  The reference breaks every one of them towards the lower
    state index (_max / _argmax, bpmodel.pyx:21-75); the maxima-forward lattice with the arg-maxima recomputed in the trace-back (round 5),
    round 4's back-pointer lattices and the plain kernel must all return the oracle's path, bit for bit."""
    dev_m, h, e = H.make_model(hip, N=150, M=3, max_cn=max_cn, chains=3, seed=29)
    ora_m, _, _ = H.make_model(oracle_mod, N=150, M=3, max_cn=max_cn, chains=3, seed=29, experiment=e)
    dev, ora = H.attach(dev_m, h), H.attach(ora_m, h)
    for mdl in (dev, ora):
        mdl.total_likelihood_mask = np.zeros(mdl.num_segments, dtype=int)
        mdl.allele_likelihood_mask = np.zeros(mdl.num_segments, dtype=int)
    for _ in range(2):
        dev_m.variational_update(); ora_m.variational_update()
    assert np.array_equal(np.asarray(dev.framelogprob), np.asarray(ora.framelogprob))      # identical inputs: ties are ties on both sides
    f = np.asarray(ora.framelogprob)
    assert (np.array([len(np.unique(row)) for row in f]) <= 5).all()                        # one value per subclonal-allele count
    want = np.zeros((ora.num_segments, 3, 2), dtype=np.int64); ora.infer_cn(want)
    for opt, kern in ((0, vit), (2, 1 if vit == 4 else 2), (1, 3)):
        dev._batch.set_option('viterbi_plain', opt)
        got = np.zeros_like(want); dev.infer_cn(got)
        assert dev._batch.info(14) == kern, (opt, dev._batch.info(14))
        assert np.array_equal(got, want), ('viterbi_plain %d' % opt, int((got != want).any(axis=(1, 2)).sum()))
    dev._batch.set_option('viterbi_plain', 0); dev._batch.set_option('traceback', 1)      # the sequential walk (the loop above ran the parallel trace-back for option 0)
    got = np.zeros_like(want); dev.infer_cn(got)
    assert dev._batch.info(19) == 0 and np.array_equal(got, want)
    dev._batch.set_option('traceback', 0)
    if vit == 6:      # the code-table lattice (one workgroup per restart) as well
        dev._batch.set_option('viterbi_plain', 0); dev._batch.set_option('viterbi_cluster', 1)
        got = np.zeros_like(want); dev.infer_cn(got)
        assert dev._batch.info(14) == 5 and np.array_equal(got, want)


def test_s165_matches_oracle(hip, oracle_mod):
    """BASELINE state grid (3 clones, max_cn=8 -> 165 states): register-stationary FB path vs oracle."""
    a, h, _ = H.make_model(hip, N=96, M=3, max_cn=8, chains=3, seed=7)
    b, _, _ = H.make_model(oracle_mod, N=96, M=3, max_cn=8, chains=3, seed=7)
    ma, mb = H.attach(a, h), H.attach(b, h)
    assert ma.num_cn_states == 165
    for it in range(2):
        a.variational_update(); b.variational_update()
        H.compare_models(ma, mb, tag='sweep%d' % it)
        assert np.isclose(ma.calculate_elbo(), mb.calculate_elbo(), rtol=1e-9)
        assert np.isclose(ma.hmm_log_norm_const, mb.hmm_log_norm_const, rtol=1e-10)
    cna = np.zeros((ma.num_segments, 3, 2), dtype=int); cnb = cna.copy()
    ma.infer_cn(cna); mb.infer_cn(cnb)
    assert np.array_equal(cna, cnb)


def test_fb_register_and_generic_paths_agree(hip):
    """Long chains: the register-stationary kernel and the general kernel (option fb_kernel = 1) must produce the
    same posteriors; repeated runs are bit-identical."""
    outs = []
    for generic in (False, True, False):
        hip.set_default_option('fb_kernel', 1 if generic else 0)
        try:
            m, h, _ = H.make_model(hip, N=6000, M=3, max_cn=8, chains=5, seed=3)
            mm = H.attach(m, h)
        finally:
            hip.set_default_option('fb_kernel', 0)
        m.variational_update(); m.variational_update()
        outs.append((mm.posterior_marginals, mm.hmm_log_norm_const, mm.calculate_elbo(), mm.p_breakpoint))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-9, atol=1e-13)
    assert np.isclose(outs[0][1], outs[1][1], rtol=1e-12) and np.isclose(outs[0][2], outs[1][2], rtol=1e-12)
    assert np.array_equal(outs[0][0], outs[2][0]) and outs[0][1] == outs[2][1] and outs[0][2] == outs[2][2]
    assert np.array_equal(outs[0][3], outs[2][3])


def test_s355_on_the_fly_weights_match_tabulated(hip):
    """max_cn = 12 (355 states): the S x S weights do not fit the register file; k_fbk rebuilds them from
    byte-packed copy numbers (two SADs and a min per pair).  Must agree with the general kernel that reads
    the tabulated weights (option fb_kernel = 2), breakend steps included, and be repeatable bit for bit.
    (Both against the oracle: tests/test_hip_bench_shapes.py.)"""
    outs = []
    for mode in ('fbk', 'generic', 'fbk'):
        hip.set_default_option('fb_kernel', 2 if mode == 'generic' else 0)
        try:
            m, h, _ = H.make_model(hip, N=1500, M=3, max_cn=12, chains=4, seed=13)
            mm = H.attach(m, h)
        finally:
            hip.set_default_option('fb_kernel', 0)
        assert mm.num_cn_states == 355
        m.variational_update(); m.variational_update()
        outs.append((mm.posterior_marginals, mm.hmm_log_norm_const, mm.calculate_elbo(), mm.p_breakpoint))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-9, atol=1e-13)
    assert np.isclose(outs[0][1], outs[1][1], rtol=1e-12) and np.isclose(outs[0][2], outs[1][2], rtol=1e-12)
    assert np.allclose(outs[0][3], outs[1][3], rtol=1e-9, atol=1e-13)
    assert np.array_equal(outs[0][0], outs[2][0]) and outs[0][1] == outs[2][1] and outs[0][2] == outs[2][2]


def test_fused_sweeps_equal_separate_updates(hip):
    """rmx_variational_update fuses marginals + outlier / allele-swap updates + the next sweep's frame
    pass into one kernel between sweeps; the result must equal the separate coordinate updates bit
    for bit (S = 165: strip kernels with the cell cache)."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(700, num_clones=3, max_copy_number=8, num_chains=4, seed=11)
    ps = synthetic.make_init_params(e, 3, 8)
    outs = []
    for knob in (None, 'fuse_sweeps', 'two_streams'):      # default: fused passes, breakend branch on its own stream
        rs = RestartSet(e, ps, max_copy_number=8, num_clones=3, quiet=True, options=({knob: 0} if knob else None))
        assert rs.batch.num_cn_states == 165
        rs.batch.variational_update(3)
        el = rs.batch.calculate_elbo()
        outs.append((el, [rs.batch.get_array(r, 'posterior_marginals') for r in range(3)],
                     [rs.batch.get_array(r, 'p_outlier_total') for r in range(3)], [rs.batch.get_array(r, 'p_outlier_allele') for r in range(3)],
                     [rs.batch.get_array(r, 'p_allele_swap') for r in range(3)], [rs.batch.get_array(r, 'p_breakpoint') for r in range(3)]))
    for other in outs[1:]:
        assert np.array_equal(outs[0][0], other[0])
        for k in range(1, 6):
            for x, y in zip(outs[0][k], other[k]):
                assert np.array_equal(x, y), k


def test_lockstep_mstep_equals_per_restart_mstep(hip):
    """The batched lock-step parameter search gives every restart exactly what its own sequential
    brute + fmin search gives (same evaluation sequence per restart)."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(600, num_clones=3, max_copy_number=4, num_chains=5, seed=6)      # 47 states: strip kernels, lists of states with posterior mass
    ps = synthetic.make_init_params(e, 4, 4)
    configs = [
        (True, True, 0),      # 0: native, the four standard searches in shared rounds
        (True, False, 0),     # 1: python lock-step
        (False, False, 0),    # 2: per-restart scipy
        (True, True, 2),      # 3: native, one parameter at a time, table-rebuilding evaluation rounds
        (True, True, 3),      # 4: native, one at a time, rounds that also evaluate the optimisers' possible next points
        (True, True, 1),      # 5: native, one at a time, table-free rounds
        (True, True, 0),      # 6: as 0, but one accept pass over the cells per parameter instead of one for all four
        (True, True, 5),      # 7: the shared rounds driven by the device (cells laid out flat over the threads)
        (True, True, 7),      # 8: the same as ONE launch of resident blocks (partial sums published and polled, an optimiser copy per block)
    ]
    out = []
    for ci, (lock, native, mode) in enumerate(configs):
        rs = RestartSet(e, ps, max_copy_number=4, num_clones=3, quiet=True, seeds=[5, 6, 7, 8], lockstep=lock,
                        native_search=native, mstep_threads=1, options={'search_mode': mode}, joint_accept=ci != 6)
        rs.fit(num_em_iter=2, num_update_iter=2)
        out.append([(m.prev_elbo, np.array(m.h), m.get_likelihood_param_values()) for m in rs.models])
    # the table-free search kernel evaluates exactly what the table-rebuilding rounds evaluate, and the
    # optional look-ahead evaluations change nothing any optimiser sees
    for (e1, h1, p1), (e2, h2, p2) in list(zip(out[5], out[3])) + list(zip(out[5], out[4])):
        assert e1 == e2 and np.array_equal(h1, h2) and p1 == p2
    # the joint accept test decides what the four sequential ones decide (its E[ll] values differ from theirs by rounding only),
    # and the decisions are all that reaches the model
    for (e1, h1, p1), (e2, h2, p2) in zip(out[0], out[6]):
        assert e1 == e2 and np.array_equal(h1, h2) and p1 == p2
    # the one-launch search evaluates the same cells in the same blocks and adds the same partial sums in the same order as the kernel pairs
    for (e1, h1, p1), (e2, h2, p2) in zip(out[7], out[8]):
        assert e1 == e2 and np.array_equal(h1, h2) and p1 == p2
    # python lock-step == per-restart scipy path, bit for bit
    for (e1, h1, p1), (e2, h2, p2) in zip(out[1], out[2]):
        assert e1 == e2 and np.array_equal(h1, h2) and p1 == p2
    # the native searches evaluate only the likelihood component the parameter moves (plus, one at a time, the
    # rest as a constant from one full evaluation; none in the shared rounds): same objective up to rounding
    for a_, b_ in ((0, 1), (5, 1), (0, 5), (7, 0), (7, 5)):
        for (e1, h1, p1), (e2, h2, p2) in zip(out[a_], out[b_]):
            assert abs(e1 - e2) <= 1e-8 * abs(e2)
            np.testing.assert_allclose(h1, h2, rtol=1e-6)
            for k in p1:
                assert abs(p1[k] - p2[k]) <= 1e-3 + 1e-5 * abs(p2[k]), k


def test_param_search_multi_argument_checks(hip):
    """rmx_param_search_multi: what does not qualify is refused (NotImplementedError -> the caller falls back
    to rmx_param_search), bad arguments raise ValueError, and a qualifying call leaves the model untouched."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(500, num_clones=3, max_copy_number=6, num_chains=4, seed=3)
    ps = synthetic.make_init_params(e, 2, 6)
    rs = RestartSet(e, ps, max_copy_number=6, num_clones=3, quiet=True, seeds=[1, 2])
    b = rs.batch
    assert b.num_cn_states == 97       # (grids of <= 32 states have no lists of states with posterior mass: never qualify)
    b.variational_update(1)
    grid = np.mgrid[10.:2000.:complex(20)]
    with pytest.raises(NotImplementedError):              # no slot sample was ever set
        b.param_search_multi([0, 1], ['negbin_r_0'], [10.], [2000.], grid[None, :])
    sample = np.zeros(b.num_segments, dtype=int); sample[::7] = 1
    b.set_sample_slot(0, 0, sample)
    with pytest.raises(ValueError):                       # restart 1 has no sample in slot 0
        b.param_search_multi([0, 1], ['negbin_r_0'], [10.], [2000.], grid[None, :])
    for r in range(2):
        for slot in range(2):
            b.set_sample_slot(r, slot, sample)
    with pytest.raises(NotImplementedError):              # not one of the four standard parameters
        b.param_search_multi([0, 1], ['negbin_hdel_mu'], [1e-9], [1e-4], np.mgrid[1e-9:1e-4:complex(20)][None, :])
    with pytest.raises(NotImplementedError):              # listed twice
        b.param_search_multi([0, 1], ['negbin_r_0', 'negbin_r_0'], [10., 10.], [2000., 2000.], np.stack([grid, grid]))
    with pytest.raises(ValueError):
        b.set_sample_slot(0, 4, sample)
    before = [(b.get_param(r, 'negbin_r_0'), b.get_param(r, 'betabin_M_0')) for r in range(2)]
    xopt, last = b.param_search_multi([0, 1], ['negbin_r_0', 'betabin_M_0'], [10., 10.], [2000., 2000.], np.stack([grid, grid]))
    assert xopt.shape == (2, 2) and last.shape == (2, 2) and np.all((xopt >= 10.) & (xopt <= 2000.))
    assert before == [(b.get_param(r, 'negbin_r_0'), b.get_param(r, 'betabin_M_0')) for r in range(2)]
    # the same searches one at a time (full-sum objective constant included): same optimum up to rounding
    for j, name in enumerate(['negbin_r_0', 'betabin_M_0']):
        for r in range(2):
            b._use_sample(r, sample)
        one = b.param_search([0, 1], name, 10., 2000., grid)
        np.testing.assert_allclose(one, xopt[j], rtol=1e-5, atol=1e-3)


def test_sample_lists_equal_dense_masks(hip):
    """rmx_set_sample_lists (all samples of an M-step in one transfer) leaves the device exactly where the per-restart dense-mask
    uploads leave it: sampled objectives, gradients and a shared-round search are bit-identical; bad lists are refused."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(500, num_clones=3, max_copy_number=6, num_chains=4, seed=3)
    ps = synthetic.make_init_params(e, 3, 6)
    grid = np.mgrid[10.:2000.:complex(20)]
    rng = np.random.RandomState(4)
    masks = []
    outs = []
    for mode in ('dense', 'lists'):
        rs = RestartSet(e, ps, max_copy_number=6, num_clones=3, quiet=True, seeds=[1, 2, 3])
        b = rs.batch
        N = b.num_segments
        if mode == 'dense':
            for r in range(3):
                m = np.zeros(N, dtype=int); m[rng.choice(N, size=40 + 7 * r, replace=False)] = 1
                masks.append(m)
        b.variational_update(1)
        if mode == 'dense':
            for r in range(3):
                b._use_sample(r, masks[r])
                for slot in range(2):
                    b.set_sample_slot(r, slot, masks[(r + slot + 1) % 3])
        else:
            b.set_sample_lists([(r, -1, masks[r], np.flatnonzero(masks[r])) for r in range(3)] +
                               [(r, slot, masks[(r + slot + 1) % 3], np.flatnonzero(masks[(r + slot + 1) % 3])) for r in range(3) for slot in range(2)])
        f, g = b.expected_log_likelihood_h_batch([0, 1, 2], np.stack([np.array(b.get_array(r, 'h')) * 1.01 for r in range(3)]))
        xopt, last = b.param_search_multi([0, 1, 2], ['negbin_r_0', 'betabin_M_0'], [10., 10.], [2000., 2000.], np.stack([grid, grid]))
        single = [b.expected_log_likelihood(r, masks[r], want_grad=True) for r in range(3)]      # (identity-cached after the list upload)
        outs.append((f, g, xopt, last, single))
        if mode == 'lists':
            with pytest.raises(ValueError):
                b.set_sample_lists([(0, -1, masks[0], np.array([5, 3]))])                      # not ascending
            with pytest.raises(ValueError):
                b.set_sample_lists([(0, 4, masks[0], np.array([1, 2]))])                       # no such slot
            with pytest.raises(ValueError):
                b.set_sample_lists([(0, 0, masks[0], np.array([1, N]))])                       # past the last segment
    a, c = outs
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2]) and np.array_equal(a[3], c[3])
    for (e1, g1), (e2, g2) in zip(a[4], c[4]):
        assert e1 == e2 and np.array_equal(g1, g2)


@pytest.mark.parametrize('search_mode', [0, 5, 7])
def test_restart_groups_do_not_change_results(hip, search_mode):
    """Restarts split into groups (own batch, stream and host thread each) give every restart the
    same fit as one batch of all restarts."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    e = synthetic.make_experiment(500, num_clones=3, max_copy_number=4, num_chains=4, seed=9)      # 47 states: strip kernels, two streams per sweep, shared search rounds
    ps = synthetic.make_init_params(e, 4, 4)
    out = []
    for groups in (1, 2):
        # (the search driver is pinned: RestartGroups picks it by grouping, and bit-identity across groupings holds per driver)
        rs = RestartGroups(e, ps, 4, groups=groups, num_clones=3, quiet=True, seeds=[1, 2, 3, 4], options={'search_mode': search_mode})
        el = rs.calculate_elbo()
        for m, v in zip(rs.models, el):
            m.prev_elbo = float(v)
        elbo = rs.run(2, 0, 2)
        out.append((elbo, [np.array(m.h) for m in rs.models], [m.get_likelihood_param_values() for m in rs.models]))
    assert np.array_equal(out[0][0], out[1][0])
    for h1, h2 in zip(out[0][1], out[1][1]):
        assert np.array_equal(h1, h2)
    assert out[0][2] == out[1][2]


def test_pipeline_run_experiment_to_best_solution(hip, tmp_path):
    """init grid -> all restarts in one device batch -> collation / best-ELBO selection (the reference's
    init / fit_task / collate chain, workflow.py:307-354) on one GPU."""
    from remixt_amd import synthetic
    from remixt_amd.analysis import pipeline
    e = synthetic.make_experiment(800, num_clones=3, max_copy_number=4, num_chains=6, seed=12)
    config = {'max_copy_number': 4, 'tumour_mix_fractions': [0.45, 0.3, 0.2], 'divergence_weights': [1e-6, 1e-7], 'num_em_iter': 1, 'num_update_iter': 2,
              'min_ploidy': None, 'max_ploidy': None, 'h_normal': 0.04, 'h_tumour': 0.06}
    init_params, results, best = pipeline.run(e, config)
    assert sorted(results) == sorted(init_params) and best in results
    elig = [i for i in results if results[i]['stats']['proportion_divergent'] < 0.5] or list(results)
    assert results[best]['stats']['elbo'] == max(results[i]['stats']['elbo'] for i in elig)
    with pipeline._Store(str(tmp_path / 'collated.store'), 'w') as st:
        assert pipeline.collate_results(st, e, results, config) == best
        assert len(st['/cn']) == 800 and 'major_2' in st['/cn'].columns
