"""GPU parity on the shapes the benchmark runs ("what's weak" 1-3): the HIP path against the CPU oracle at
165 and 355 states, several restarts per forward-backward workgroup, dozens of breakend adjacencies (two of them at one
boundary) -- posteriors, p_breakpoint, log Z and ELBO after EVERY coordinate update, not only decoded paths -- and the
protocol's rarely used corners: transition_model = 1, four clones, disable_breakpoints, breakpoint_init, check_elbo.

Tolerance: 1e-6 relative is the requirement (north_star); asserted at 1e-8 / 1e-9.  Viterbi paths bit-exact."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

STEPS = ('update_p_allele_swap', 'update_p_cn', 'update_p_breakpoint', 'update_p_outlier_total', 'update_p_outlier_allele')
ARRAYS = ('posterior_marginals', 'p_breakpoint', 'p_outlier_total', 'p_outlier_allele', 'p_allele_swap', 'framelogprob')


@pytest.fixture(scope='module')
def hip():
    from remixt_amd import bpmodel
    return bpmodel


def _two_sets(oracle_mod, e, ps, max_cn, M, options=None, **kw):
    from remixt_amd.restarts import RestartSet
    dev = RestartSet(e, ps, max_copy_number=max_cn, num_clones=M, quiet=True, options=options, **kw)
    ora = RestartSet(e, ps, max_copy_number=max_cn, num_clones=M, quiet=True, kernel_module=oracle_mod, **kw)
    return dev, ora


def _path_scores(model, paths, table):
    """log-probability of each decoded copy-number path under `model`'s own frame log-probabilities and transition matrices
    (table: cn_states [N][S][M][2])"""
    f = np.asarray(model.framelogprob)
    n1, S = f.shape
    # the lattice runs on the log_transmat SNAPSHOT of the last update_p_cn (bpmodel.pyx:939, 1201), not on calculate_log_transmat's matrix of the
    # current p_breakpoint: at breakend adjacencies the two differ after update_p_breakpoint (found by fuzz seed 2041: 30 breakpoints on 90 segments)
    lt = np.asarray(model.log_transmat)
    out = []
    for cn in paths:
        st = [int(np.nonzero((table[n] == cn[n][None]).all(axis=(1, 2)))[0][0]) for n in range(n1)]
        out.append(float(sum(f[n, st[n]] for n in range(n1)) + sum(lt[n, st[n], st[n + 1]] for n in range(n1 - 1))))
    return out


def _compare_after_every_update(dev, ora, sweeps=2, elbo_rtol=1e-8, rtol=1e-8, ties_ok=False):
    b = dev.batch
    R = len(dev.models)
    np.testing.assert_allclose(b.calculate_elbo(), [m.model.calculate_elbo() for m in ora.models], rtol=1e-9)
    for sweep in range(sweeps):
        for step in STEPS:
            getattr(b, step)()                                 # all restarts in ONE launch sequence
            for m in ora.models:
                getattr(m.model, step)()
            tag = 'sweep %d %s' % (sweep, step)
            for r in range(R):
                for name in ARRAYS:
                    got, want = b.get_array(r, name), np.asarray(getattr(ora.models[r].model, name))
                    assert got.shape == want.shape
                    assert H.close(got, want, rtol=rtol, atol=1e-11), '%s restart %d %s: max rel err %.3e' % (tag, r, name, H.maxerr(got, want))
                assert np.isclose(b.get_param(r, 'hmm_log_norm_const'), ora.models[r].model.hmm_log_norm_const, rtol=1e-10), tag
            np.testing.assert_allclose(b.calculate_elbo(), [m.model.calculate_elbo() for m in ora.models], rtol=elbo_rtol, err_msg=tag)
    cn, _ = b.infer_cn_batch(0, R)
    for r in range(R):
        ref = np.zeros_like(cn[r]); ora.models[r].model.infer_cn(ref)
        if ties_ok and not np.array_equal(cn[r], ref):
            # the two lattices see inputs that differ in their last bits (lgamma of two libraries): paths may differ where two paths TIE --
            # both must then have the same log-probability, under either side's arrays, to rounding
            for mdl in (dev.models[r].model, ora.models[r].model):
                a, b_ = _path_scores(mdl, [cn[r], ref], np.asarray(dev.models[r].model.cn_states))
                assert abs(a - b_) <= 1e-11 * abs(b_), 'Viterbi path of restart %d differs and is not a tie: %.17g vs %.17g' % (r, a, b_)
            continue
        assert np.array_equal(cn[r], ref), 'Viterbi path of restart %d differs' % r


@pytest.mark.parametrize('R,NV,two_phase', [(4, 4, 0), (8, 4, 0), (8, 2, 0), (3, 2, 0), (7, 2, 0), (5, 1, 0), (3, 0, 0), (8, 2, 1), (3, 2, 1)])
def test_s165_restart_batch_with_dense_breakends_matches_oracle(hip, oracle_mod, R, NV, two_phase):
    """165 states, R restarts in one batch, 24 breakpoints + two sharing a boundary on 110 segments -> ~50 breakend adjacencies, i.e.
    every other step takes the breakend branch, and k_pairwise_be2 / k_brk_lut see all of them.  The production kernel k_fbm in its three
    workgroup shapes -- NV = 4: FP64 matrix cores, quads of restarts (R = 4 one full quad, R = 8 two); NV = 2 / 1: the same step on the
    vector ALU, two / one restarts per workgroup ((3, 2), (7, 2): a ragged last unit inside a quad; (5, 1): every slot of a quad and a
    quad with one restart); NV = 0: what a launch of three restarts selects by itself -- and the two-phase vector kernel k_fbv
    (fb_kernel = 3).  Which one ran is asserted through rmx_info(12 / 13)."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(110, num_clones=3, max_copy_number=8, num_chains=2, seed=31, num_breakpoints=24)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, R, 8)
    dev, ora = _two_sets(oracle_mod, e, ps, 8, 3, options={'fb_nv': NV, 'fb_kernel': 3 if two_phase else 0})
    b = dev.batch
    assert b.num_cn_states == 165 and b.info(3) >= 48 and b.info(10) == 2 and b.info(11) == 0      # breakend adjacencies; both chains on the register-resident kernels
    _compare_after_every_update(dev, ora)
    assert (b.info(12), b.info(13)) == ((2, 2) if two_phase else (1, NV or 1))                     # k_fbv<., 2> / k_fbm<., NV> (3 restarts x 2 chains x 2 directions: one per workgroup)


def test_s165_unequal_chains_get_mixed_workgroup_shapes_and_match_oracle(hip, oracle_mod):
    """Chains of unequal length (chromosomes): k_fbm gives the long chains fewer restarts per workgroup than the short ones, all shapes in
    ONE launch (rmx_api.hip fb_items_for).  A workgroup budget of 40 puts this small problem (5 chains, 6 restarts, breakends on every
    chain) on the mix a genome gets on 256 CUs -- asserted through rmx_info(13 / 15) -- and every coordinate update agrees with the oracle."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(200, num_clones=3, max_copy_number=8, num_chains=5, seed=37, num_breakpoints=14, chain_fractions=(10, 6, 3, 2, 1))
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, 6, 8)
    dev, ora = _two_sets(oracle_mod, e, ps, 8, 3, options={'fb_wg_budget': 40})
    b = dev.batch
    assert b.num_cn_states == 165 and b.info(10) == 5 and b.info(11) == 0
    _compare_after_every_update(dev, ora)
    assert b.info(12) == 1 and (b.info(13), b.info(15)) == (1, 4), (b.info(13), b.info(15))      # one restart per workgroup on the longest chain, four on the shortest
    # the launch's bits do not depend on how the chip is shared out, as long as a chain keeps its side of the matrix / vector divide:
    # pinned shapes reproduce the unpinned equal-budget run chain by chain (vector shapes 1 and 2 are bit-identical)
    post = [b.get_array(r, 'posterior_marginals') for r in range(6)]
    dev2, _ = _two_sets(oracle_mod, e, ps, 8, 3, options={'fb_wg_budget': 40})
    for step in STEPS * 2:
        getattr(dev2.batch, step)()
    for r in range(6):
        assert np.array_equal(dev2.batch.get_array(r, 'posterior_marginals'), post[r]), r


def test_s355_unequal_chains_get_mixed_workgroup_shapes_and_match_oracle(hip, oracle_mod):
    """The same at 355 states (k_fbq: one restart per workgroup on the long chains, four on the short ones, in one launch)."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(70, num_clones=3, max_copy_number=12, num_chains=4, seed=43, num_breakpoints=8, chain_fractions=(8, 4, 2, 1))
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, 5, 12)
    dev, ora = _two_sets(oracle_mod, e, ps, 12, 3, options={'fb_wg_budget': 24})
    b = dev.batch
    assert b.num_cn_states == 355 and b.info(10) == 4 and b.info(11) == 0
    _compare_after_every_update(dev, ora)
    assert b.info(12) == 4 and (b.info(13), b.info(15)) == (1, 4), (b.info(13), b.info(15))


def test_s165_workgroup_shapes_agree_and_subranges_are_bit_identical(hip):
    """k_fbm<., 4> (matrix cores) and k_fbm<., 2> / <., 1> (vector ALU) sum a column in different orders: posteriors agree to 1e-10, not
    to the bit.  Inside ONE shape a restart's result does not depend on the range of restarts a launch covers (units are absolute:
    restarts NV u .. NV u + NV - 1), which is what keeps restart groups and shards bit-identical."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(300, num_clones=3, max_copy_number=8, num_chains=3, seed=35, num_breakpoints=10)
    ps = synthetic.make_init_params(e, 7, 8)
    post = {}
    for nv in (4, 2, 1):
        rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, options={'fb_nv': nv})
        b = rs.batch
        b.variational_update(2)
        assert (b.info(12), b.info(13)) == (1, nv)
        post[nv] = [b.get_array(r, 'posterior_marginals') for r in range(7)]
        # the same sweeps over sub-ranges that cut units and quads differently
        rs2 = RestartSet(e, ps, 8, num_clones=3, quiet=True, options={'fb_nv': nv})
        for r0, r1 in ((0, 3), (3, 6), (6, 7)):
            rs2.batch.variational_update(2, r0, r1)
        for r in range(7):
            assert np.array_equal(rs2.batch.get_array(r, 'posterior_marginals'), post[nv][r]), (nv, r)
    for nv in (2, 1):
        for r in range(7):
            assert H.close(post[nv][r], post[4][r], rtol=1e-10, atol=1e-13), (nv, r, H.maxerr(post[nv][r], post[4][r]))
    for r in range(7):
        assert np.array_equal(post[2][r], post[1][r]), ('vector shapes', r)      # two and one restart per workgroup: the same arithmetic per restart


def test_s165_generic_kernel_and_plain_breakend_tables_match_oracle(hip, oracle_mod):
    """The same problem through the kernels the defaults do not select: general forward-backward kernel, breakend steps
    from per-clone distance tables, the general pairwise kernel."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(90, num_clones=3, max_copy_number=8, num_chains=2, seed=33, num_breakpoints=12)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, 2, 8)
    for options in ({'fb_kernel': 1}, {'fb_breakend_codes': 0, 'fb_nv': 2}, {'fb_kernel': 3, 'fb_nv': 1}, {'pairwise_kernel': 1}, {'pairwise_kernel': 3}, {'pairwise_kernel': 4}):      # (3 / 4: the sparse pairwise kernel, a block of 256 threads / of one wave per adjacency)
        dev, ora = _two_sets(oracle_mod, e, ps, 8, 3, options=options)
        _compare_after_every_update(dev, ora, sweeps=1)


@pytest.mark.parametrize('options,fb', [({}, 4), ({'fb_nv': 4}, 4), ({'fb_nv': 1}, 4), ({'fb_kernel': 3, 'fb_nv': 2}, 3), ({'fb_kernel': 3, 'fb_nv': 1}, 3), ({'fb_kernel': 2}, 0), ({'viterbi_plain': 1}, 4), ({'viterbi_plain': 2}, 4),
                                        ({'pairwise_kernel': 2}, 4), ({'viterbi_cluster': 1}, 4), ({'viterbi_cluster': 4}, 4)])
def test_s355_matches_oracle(hip, oracle_mod, options, fb):
    """355 states (max_cn = 12, the "~400 states" of BASELINE's metric): k_fbq (B operands looked up from 8-bit distances; four restarts per
    workgroup on the FP64 matrix cores -- a quad with two and with four restarts present -- and one per workgroup on the vector ALU
    (two per workgroup exist up to 256 states: test_grids_between_the_benchmark_grids_match_oracle)), k_fbk (two-phase vector FMA, weights rebuilt from packed copy numbers), the general kernel on tabulated weights,
    the lattice from the packed copies in clusters of 8 / 4 workgroups per restart (k_viterbi_sad_max), the code-table lattices and the plain one, each against the oracle -- not against each other."""
    from remixt_amd import synthetic
    R = 4 if options.get('fb_nv') == 4 else (3 if options.get('fb_nv') == 1 else 2)
    e = synthetic.make_experiment(44, num_clones=3, max_copy_number=12, num_chains=2, seed=41, num_breakpoints=8)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, R, 12)
    dev, ora = _two_sets(oracle_mod, e, ps, 12, 3, options=options)
    assert dev.batch.num_cn_states == 355 and dev.batch.info(3) >= 16
    _compare_after_every_update(dev, ora)
    assert dev.batch.info(12) == fb, ('forward-backward kernel', dev.batch.info(12), dev.batch.info(13))
    if fb == 4:
        assert dev.batch.info(13) == (options.get('fb_nv') or 1)      # (two restarts x two chains x two directions: the automatic choice is one per workgroup)


@pytest.mark.parametrize('max_cn,S,R,nv', [(9, 205, 5, 4), (10, 251, 4, 4), (11, 300, 6, 4), (9, 205, 5, 2), (10, 251, 3, 1), (11, 300, 6, 0), (11, 300, 5, 1)])
def test_grids_between_the_benchmark_grids_match_oracle(hip, oracle_mod, max_cn, S, R, nv):
    """k_fbq's other instantiations and ragged shapes: 205 / 251 states (64 k-blocks), 300 states (90 k-blocks, 10 of 12 column
    waves), restart counts that leave the last quad with one or two restarts."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(40, num_clones=3, max_copy_number=max_cn, num_chains=2, seed=80 + max_cn, num_breakpoints=6)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, R, max_cn)
    dev, ora = _two_sets(oracle_mod, e, ps, max_cn, 3, options={'fb_nv': nv})
    assert dev.batch.num_cn_states == S
    _compare_after_every_update(dev, ora)
    assert (dev.batch.info(12), dev.batch.info(13)) == (4, nv or 1)


@pytest.mark.parametrize('nv', [0, 4, 2])
def test_two_clones_at_169_states_match_oracle(hip, oracle_mod, nv):
    """Two clones at max_cn = 24: 169 states -- between the three-clone grids (165, 205).  The copy-number range is beyond the breakend tables of
    k_fbm (2 max_cn + 3 = 51 > 31 table entries per clone), so with breakends this grid runs the two-phase vector kernel k_fbv with per-clone tables."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(36, num_clones=2, max_copy_number=24, num_chains=2, seed=71, num_breakpoints=6)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, 5, 24, num_clones=2)
    dev, ora = _two_sets(oracle_mod, e, ps, 24, 2, options={'fb_nv': nv})
    assert dev.batch.num_cn_states == 169
    _compare_after_every_update(dev, ora)
    assert dev.batch.info(12) == 2 and dev.batch.info(13) == (nv or 1)


@pytest.mark.parametrize('max_cn', [3, 6])
def test_transition_model_1_matches_oracle(hip, oracle_mod, max_cn):
    """transition_model = 1 (bpmodel.pyx:606-616: 0/1 cost per changed copy number instead of |d|), set after construction
    like cn_model.py:404: the plain tables are rebuilt, the SAD closed form of k_fbk no longer applies, breakend tables use
    the other cost.  ELBO is compared where the reference's two snapshots belong to the same model."""
    a, h, _ = H.make_model(hip, N=80, M=3, max_cn=max_cn, chains=3, seed=50 + max_cn, transition_model=1)
    b, _, _ = H.make_model(oracle_mod, N=80, M=3, max_cn=max_cn, chains=3, seed=50 + max_cn, transition_model=1)
    ma, mb = H.attach(a, h), H.attach(b, h)
    assert ma.transition_model == 1 and mb.transition_model == 1
    assert np.isclose(ma.calculate_elbo(), mb.calculate_elbo(), rtol=1e-9)          # still the constructor's model-0 snapshot
    for it in range(2):
        for step in STEPS:
            getattr(ma, step)(); getattr(mb, step)()
            H.compare_models(ma, mb, dense=(it == 0 and max_cn == 3), tag='%d/%s' % (it, step))
            # (first sweep, after update_p_cn: log_transmat under model 1, cached_log_transmat still the constructor's model-0
            # tables -- the one state in which the plain-adjacency terms of energy and entropy do not cancel)
            assert np.isclose(ma.calculate_elbo(), mb.calculate_elbo(), rtol=1e-8), (it, step)
        assert np.isclose(ma.hmm_log_norm_const, mb.hmm_log_norm_const, rtol=1e-10)
    if max_cn == 3:
        # energy and entropy individually in that mixed state, and the decode when the model changes AFTER update_p_cn
        # (the lattice runs on the log_transmat snapshot, bpmodel.pyx:1197-1204)
        c, hc, _ = H.make_model(hip, N=80, M=3, max_cn=max_cn, chains=3, seed=50 + max_cn, transition_model=1)
        o, _, _ = H.make_model(oracle_mod, N=80, M=3, max_cn=max_cn, chains=3, seed=50 + max_cn, transition_model=1)
        mc, mo = H.attach(c, hc), H.attach(o, hc)
        for m_ in (mc, mo):
            m_.update_p_allele_swap(); m_.update_p_cn()
        assert np.isclose(mc.calculate_variational_energy(), mo.calculate_variational_energy(), rtol=1e-8)
        assert np.isclose(mc.calculate_variational_entropy(), mo.calculate_variational_entropy(), rtol=1e-8)
        for m_ in (mc, mo):
            m_.transition_model = 0
        assert np.allclose(mc.log_transmat, mo.log_transmat, rtol=1e-12, atol=1e-12)
        c1 = np.zeros((mc.num_segments, 3, 2), dtype=int); c2 = c1.copy()
        mc.infer_cn(c1); mo.infer_cn(c2)
        assert np.array_equal(c1, c2)
    cna = np.zeros((ma.num_segments, 3, 2), dtype=int); cnb = cna.copy()
    ma.infer_cn(cna); mb.infer_cn(cnb)
    assert np.array_equal(cna, cnb)


def test_transition_model_1_at_355_states_deselects_the_closed_form_kernel(hip, oracle_mod):
    from remixt_amd import synthetic
    e = synthetic.make_experiment(40, num_clones=3, max_copy_number=12, num_chains=2, seed=43, num_breakpoints=6)
    ps = synthetic.make_init_params(e, 2, 12)
    dev, ora = _two_sets(oracle_mod, e, ps, 12, 3, transition_model=1)
    for m in ora.models:
        assert m.model.transition_model == 1
    b = dev.batch
    assert b.transition_model == 1
    for step in STEPS:
        getattr(b, step)()
        for m in ora.models:
            getattr(m.model, step)()
    for r in range(2):
        for name in ARRAYS:
            assert H.close(b.get_array(r, name), np.asarray(getattr(ora.models[r].model, name)), rtol=1e-8, atol=1e-11), name
    np.testing.assert_allclose(b.calculate_elbo(), [m.model.calculate_elbo() for m in ora.models], rtol=1e-8)


@pytest.mark.parametrize('max_cn,N', [(2, 90), (3, 60)])
def test_four_clones_match_oracle(hip, oracle_mod, max_cn, N):
    """num_clones = 4 (RMX_MAX_CLONES): 3 tumour clones, 117 states at max_cn = 3."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(N, num_clones=4, max_copy_number=max_cn, num_chains=2, seed=60 + max_cn, num_breakpoints=8)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, 2, max_cn, num_clones=4)
    hs = [np.array([p['h_normal']] + [p['h_tumour'] * f for f in (0.5, 0.3, 0.2)]) for p in ps]
    from remixt_amd.restarts import RestartSet
    sets = []
    for kern in (None, oracle_mod):
        rs = RestartSet(e, ps, max_copy_number=max_cn, num_clones=4, quiet=True, kernel_module=kern, h_init=hs)
        sets.append(rs)
    dev, ora = sets
    assert dev.batch.num_clones == 4
    _compare_after_every_update(dev, ora)
    # the M-step objectives with a 4-vector gradient
    sample = (np.random.RandomState(1).rand(dev.batch.num_segments) < 0.4).astype(int)
    for r in range(2):
        ma, mb = dev.models[r].model, ora.models[r].model
        assert np.isclose(ma.calculate_expected_log_likelihood(sample), mb.calculate_expected_log_likelihood(sample), rtol=1e-9)
        ga, gb = np.zeros(4), np.zeros(4)
        ma.calculate_expected_log_likelihood_partial_h(sample, ga); mb.calculate_expected_log_likelihood_partial_h(sample, gb)
        assert np.allclose(ga, gb, rtol=1e-7, atol=1e-6)


def test_disable_breakpoints_fit_matches_oracle(hip, oracle_mod):
    """disable_breakpoints=True (cn_model.py:187-190): num_breakpoints = 0, no breakend adjacency anywhere, naive breakpoint
    decoding afterwards (analysis/pipeline.py:205-206)."""
    from remixt_amd.cn_model import decode_breakpoints_naive
    res = []
    for kern in (hip, oracle_mod):
        m, h, e = H.make_model(kern, N=200, M=3, max_cn=4, chains=4, seed=71, disable_breakpoints=True)
        m.num_em_iter = 2; m.num_update_iter = 2
        np.random.seed(5)
        m.fit(h)
        assert m.model.num_breakpoints == 0 and np.asarray(m.model.p_breakpoint).shape[0] == 0
        cn, brk = m.optimal_cn()
        assert brk == {}
        res.append((m.prev_elbo, np.array(m.h), cn, decode_breakpoints_naive(cn, e.adjacencies, e.breakpoints), np.array(m.p_outlier_total)))
    (e1, h1, c1, b1, q1), (e2, h2, c2, b2, q2) = res
    assert np.isclose(e1, e2, rtol=1e-6) and np.allclose(h1, h2, rtol=1e-5) and np.array_equal(c1, c2)
    # (after two EM iterations the scipy optimisers have amplified the kernels' rounding differences: the bound of golden_runner.replay_fit)
    assert all(np.array_equal(b1[k], b2[k]) for k in b1) and np.allclose(q1, q2, rtol=1e-4, atol=1e-7)


def test_breakpoint_init_and_check_elbo_match_oracle(hip, oracle_mod):
    """breakpoint_init (cn_model.py:389-402: p_breakpoint seeded before the first sweep -- the device tables that depend on
    it must follow) and check_elbo=True (:430-442: an ELBO before and after every coordinate update and M-step; raises when
    one decreases it)."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(160, num_clones=3, max_copy_number=4, num_chains=3, seed=81, num_breakpoints=10)
    init = dict((bp, np.array([0, 1, 1])) for bp in e.breakpoints.values())
    res = []
    for kern in (hip, oracle_mod):
        m, h, _ = H.make_model(kern, N=160, M=3, max_cn=4, chains=3, seed=81, breakpoint_init=init, experiment=e)
        mod = H.attach(m, h)
        p0 = np.asarray(mod.p_breakpoint).copy()
        assert np.allclose(p0.max(axis=1), 1000. / (1000. + p0.shape[1] - 1))
        elbo0 = mod.calculate_elbo()
        m.check_elbo = True
        m.variational_update()
        m.prev_elbo = mod.calculate_elbo()
        np.random.seed(3)
        m.num_update_iter = 1
        m.em_iteration(0)
        res.append((elbo0, p0, np.asarray(mod.p_breakpoint).copy(), np.asarray(mod.posterior_marginals).copy(), m.prev_elbo, np.array(m.h)))
    (ea, pa, qa, post_a, fa, ha), (eb, pb, qb, post_b, fb, hb) = res
    assert np.isclose(ea, eb, rtol=1e-9) and np.array_equal(pa, pb)
    assert H.close(qa, qb, rtol=1e-8, atol=1e-11) and H.close(post_a, post_b, rtol=1e-7, atol=1e-10)
    assert np.isclose(fa, fb, rtol=1e-6) and np.allclose(ha, hb, rtol=1e-5)


def test_per_cell_accessors_match_oracle(hip, oracle_mod):
    """The per-cell cpdef methods of RemixtModel (bpmodel.pyx:686-749, 778-807, 855-896)."""
    a, h, _ = H.make_model(hip, N=60, M=3, max_cn=4, chains=2, seed=91)
    b, _, _ = H.make_model(oracle_mod, N=60, M=3, max_cn=4, chains=2, seed=91)
    ma, mb = H.attach(a, h), H.attach(b, h)
    rng = np.random.RandomState(0)
    for _ in range(40):
        n, s = int(rng.randint(0, ma.num_segments)), int(rng.randint(0, ma.num_cn_states))
        assert np.isclose(ma.calculate_expected_total_reads(n, s), mb.calculate_expected_total_reads(n, s), rtol=1e-14)
        assert np.isclose(ma.calculate_expected_allele_ratio(n, s), mb.calculate_expected_allele_ratio(n, s), rtol=1e-14)
        assert ma.calculate_log_prior_cn(n, s) == mb.calculate_log_prior_cn(n, s)
        ga, gb = np.zeros(3), np.zeros(3)
        ma.calculate_expected_total_reads_partial_h(n, s, ga); mb.calculate_expected_total_reads_partial_h(n, s, gb)
        assert np.array_equal(ga, gb)
        ma.calculate_expected_allele_ratio_partial_h(n, s, ga); mb.calculate_expected_allele_ratio_partial_h(n, s, gb)
        assert np.allclose(ga, gb, rtol=1e-12, atol=1e-15)
        for u in range(2):
            ga, gb = np.zeros(3), np.zeros(3)
            ma.calculate_log_likelihood_total_partial_h(n, s, u, ga); mb.calculate_log_likelihood_total_partial_h(n, s, u, gb)
            assert np.allclose(ga, gb, rtol=1e-9, atol=1e-9), (n, s, u, ga, gb)
            for w in range(2):
                ga, gb = np.zeros(3), np.zeros(3)
                ma.calculate_log_likelihood_allele_partial_h(n, s, u, w, ga); mb.calculate_log_likelihood_allele_partial_h(n, s, u, w, gb)
                assert np.allclose(ga, gb, rtol=1e-8, atol=1e-8), (n, s, u, w, ga, gb)


def test_two_datasets_on_one_gpu_equal_their_single_dataset_fits(hip):
    """BASELINE configs[4]: two tumour samples on one segmentation and breakpoint set, fitted independently side by side
    (reference workflow.py:472-485) -- each dataset's results must equal its own single-dataset fit bit for bit."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import DatasetGroups, RestartGroups
    e0 = synthetic.make_experiment(400, num_clones=3, max_copy_number=4, num_chains=4, seed=14, num_breakpoints=12)
    e1 = synthetic.resample_counts(e0, seed=101)
    assert e1.breakpoints is e0.breakpoints and not np.array_equal(e0.x, e1.x)
    ps = [synthetic.make_init_params(e, 4, 4) for e in (e0, e1)]
    seeds = [[1, 2, 3, 4], [11, 12, 13, 14]]
    both = DatasetGroups([e0, e1], ps, 4, groups=2, num_clones=3, quiet=True, seeds=seeds)
    elbo = both.fit(num_em_iter=2, num_update_iter=2)
    res = both.results_by_dataset()
    for i, e in enumerate((e0, e1)):
        one = RestartGroups(e, ps[i], 4, groups=2, num_clones=3, quiet=True, seeds=seeds[i])
        el = one.fit(num_em_iter=2, num_update_iter=2)
        assert np.array_equal(el, elbo[4 * i:4 * i + 4])
        for a, b in zip(one.results(), res[i]):
            assert a['stats']['elbo'] == b['stats']['elbo'] and np.array_equal(a['h'], b['h']) and np.array_equal(a['cn'], b['cn'])
            assert all(np.array_equal(a['brk_cn'][k], b['brk_cn'][k]) for k in a['brk_cn'])
            assert np.array_equal(a['p_outlier_total'], b['p_outlier_total'])


def test_datasets_keep_two_restart_groups_on_the_device_by_default(hip):
    """DatasetGroups without groups=: one restart group per dataset from two datasets on, at most two groups running at once (three datasets: two at a
    time) -- more than two groups share the runtime's four hardware queues.  Results equal the single-dataset fits with the same grouping, bit for bit."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import DatasetGroups, RestartGroups
    e0 = synthetic.make_experiment(300, num_clones=3, max_copy_number=4, num_chains=3, seed=15, num_breakpoints=8)
    es = [e0, synthetic.resample_counts(e0, seed=102), synthetic.resample_counts(e0, seed=103)]
    ps = [synthetic.make_init_params(e, 3, 4) for e in es]
    seeds = [[1, 2, 3], [11, 12, 13], [21, 22, 23]]
    one = DatasetGroups(es[:1], ps[:1], 4, num_clones=3, quiet=True, seeds=seeds[:1])
    assert (one.groups_per_dataset, one._workers, len(one.sets)) == (2, 1, 2)
    one.close()
    three = DatasetGroups(es, ps, 4, num_clones=3, quiet=True, seeds=seeds)
    assert (three.groups_per_dataset, three._workers, len(three.sets)) == (1, 2, 3)
    elbo = three.fit(num_em_iter=2, num_update_iter=2)
    res = three.results_by_dataset()
    for i, e in enumerate(es):
        ref = RestartGroups(e, ps[i], 4, groups=1, num_clones=3, quiet=True, seeds=seeds[i])
        el = ref.fit(num_em_iter=2, num_update_iter=2)
        assert np.array_equal(el, elbo[3 * i:3 * i + 3])
        for a, b in zip(ref.results(), res[i]):
            assert a['stats']['elbo'] == b['stats']['elbo'] and np.array_equal(a['h'], b['h']) and np.array_equal(a['cn'], b['cn'])
        ref.close()


def test_every_flagged_restart_is_reported_and_cleared(hip):
    """a batched call that flags several restarts used to clear only the first one's error word; the others
    surfaced in later, unrelated calls."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(150, num_clones=3, max_copy_number=4, num_chains=3, seed=19)
    ps = synthetic.make_init_params(e, 4, 4)
    rs = RestartSet(e, ps, 4, num_clones=3, quiet=True, seeds=[1, 2, 3, 4])
    b = rs.batch
    h = [b.get_array(r, 'h') for r in range(4)]
    for r in (1, 3):
        b.set_array(r, 'h', -h[r])                       # total_depth <= 0 in every state (bpmodel.pyx:721)
    with pytest.raises(ValueError, match='total_depth <= 0') as info:
        b.update_p_cn()
    assert info.value.restarts == [1, 3] and 'restart 1' in str(info.value)
    for r in (1, 3):
        b.set_array(r, 'h', h[r])
    b.update_p_cn(3, 4)                                   # restart 3's word was cleared together with restart 1's
    b.update_p_cn()
    assert np.all(np.isfinite(b.calculate_elbo()))


def test_lockstep_h_step_fails_only_the_flagged_restart(hip, monkeypatch):
    """A ValueError of the reference raised by ONE restart's trial h during the lock-step h M-step ends that restart's h
    update only; the others get exactly the h they get when that restart is not there."""
    from remixt_amd import synthetic, lockstep
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(300, num_clones=3, max_copy_number=4, num_chains=3, seed=23)
    ps = synthetic.make_init_params(e, 3, 4)

    def run(poison):
        rs = RestartSet(e, ps, 4, num_clones=3, quiet=True, seeds=[5, 6, 7])
        el = rs.calculate_elbo()
        for m, v in zip(rs.models, el):
            m.prev_elbo = float(v)
        if poison:
            make = rs.batch.h_batch_evaluator
            state = {'n': 0}

            def poisoned_factory(restarts):
                restarts = list(restarts)
                evaluate = make(restarts)

                def poisoned(ids, xs):
                    xs = [np.array(x, dtype=float) for x in xs]
                    state['n'] += 1
                    listed = [restarts[i] for i in ids]
                    if state['n'] == 2 and 1 in listed:
                        xs[listed.index(1)] *= -1.          # the optimiser of restart 1 "proposes" a negative depth
                    return evaluate(ids, xs)
                return poisoned
            monkeypatch.setattr(rs.batch, 'h_batch_evaluator', poisoned_factory)
        rs.em_iteration(0, 2)
        return rs
    bad, good = run(True), run(False)
    assert list(bad.error_messages) == [1] and 'total_depth' in bad.error_messages[1]
    for r in (0, 2):
        assert np.array_equal(bad.models[r].h, good.models[r].h)
    assert np.all(np.isfinite([m.prev_elbo for m in bad.models]))


def test_baseline_config0_shape_full_fit_matches_oracle(hip, oracle_mod):
    """BASELINE configs[0]: 1 000 segments, 2 clones, max_cn = 4, a single h initialisation -- the reference's own CPU-runnable case,
    as a seeded EM fit (sweeps + scipy M-steps) on the device and on the oracle: ELBO to 1e-6, decoded copy number identical."""
    res = []
    for kern in (hip, oracle_mod):
        m, h, e = H.make_model(kern, N=1000, M=2, max_cn=4, chains=23, seed=101)
        m.num_em_iter = 3; m.num_update_iter = 5
        np.random.seed(7)
        m.fit(h)
        cn, brk = m.optimal_cn()
        res.append((m.prev_elbo, np.array(m.h), m.get_likelihood_param_values(), cn, brk))
    (e1, h1, p1, cn1, b1), (e2, h2, p2, cn2, b2) = res
    assert np.isclose(e1, e2, rtol=1e-6), (e1, e2)
    np.testing.assert_allclose(h1, h2, rtol=1e-5)
    for k in p1:
        assert np.isclose(p1[k], p2[k], rtol=1e-4), (k, p1[k], p2[k])
    assert np.array_equal(cn1, cn2) and all(np.array_equal(b1[k], b2[k]) for k in b1)


def test_baseline_config1_shape_matches_oracle(hip, oracle_mod):
    """BASELINE configs[1]: 10 000 segments, 2 clones, max_cn = 6, a single h initialisation: the likelihood fill, the
    forward-backward sweep, the ELBO and the Viterbi decode of the device against the oracle at the full size of the configuration."""
    a, h, e = H.make_model(hip, N=10000, M=2, max_cn=6, chains=23, seed=202)
    b, _, _ = H.make_model(oracle_mod, experiment=e, M=2, max_cn=6)
    ma, mb = H.attach(a, h), H.attach(b, h)
    assert ma.num_cn_states == mb.num_cn_states
    for sweep in range(2):
        for step in ('update_p_allele_swap', 'update_p_cn', 'update_p_breakpoint', 'update_p_outlier_total', 'update_p_outlier_allele'):
            getattr(ma, step)(); getattr(mb, step)()
        H.compare_models(ma, mb, tag='configs[1] sweep %d' % sweep)
        assert ma.calculate_elbo() == pytest.approx(mb.calculate_elbo(), rel=1e-9)
    s = np.ones(ma.num_segments, dtype=np.int64)
    assert ma.calculate_expected_log_likelihood(s) == pytest.approx(mb.calculate_expected_log_likelihood(s), rel=1e-9)
    cna = np.zeros((ma.num_segments, 2, 2), dtype=int); cnb = cna.copy()
    ma.infer_cn(cna); mb.infer_cn(cnb)
    assert np.array_equal(cna, cnb)


# ---- state grids beyond the benchmark's (kernels generic in S <= 1024, M <= 4) ------------------------------------
@pytest.mark.parametrize('M,max_cn,S,N,fb,vit', [
    (4, 4, 207, 30, 3, 6),         # four clones with breakends above 176 states: k_fbk (round 4: the third tumour clone in a second packed word, clone-product tables of D^3 entries)
    (4, 6, 457, 28, 3, 6),
    (3, 13, 413, 30, 3, 6),        # three clones above 355 states: k_fbk (weights from packed copy numbers; blocks of 896 threads); lattice from the packed copies (round 5; the plain lattice before)
    (3, 14, 477, 26, 3, 6),        # ... of 1 024 threads: the largest grid with four row slices per column pair
    (3, 16, 617, 26, 3, 6),        # above 512 states (round 5): k_fbk with two row slices per column pair (blocks of 640 threads); the general kernel k_fb<0> before
    (3, 20, 951, 22, 3, 6),        # the largest three-clone grid below 1 024 states (960 threads; allele distances up to 80: 128 table entries)
    (4, 8, 805, 20, 3, 6),         # four clones at max_cn 8 (clone-product tables of 19^3 entries, one vector per workgroup)
])
def test_large_state_grids_match_oracle(hip, oracle_mod, M, max_cn, S, N, fb, vit):
    """every coordinate update of two sweeps and the decode against the oracle at the four-clone grids of
    cn_model.py:228-253 (207 / 457 states) and at three-clone grids up to 951 states, with a record of the forward-backward /
    lattice kernel each one selects (rmx_info 12 / 14)."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(N, num_clones=M, max_copy_number=max_cn, num_chains=2, seed=60 + max_cn, num_breakpoints=5)
    e.breakpoints = H.add_shared_boundary_breakpoints(e)
    ps = synthetic.make_init_params(e, 2, max_cn, num_clones=M)
    fr = (0.6, 0.4) if M == 3 else (0.5, 0.3, 0.2)
    hs = [np.array([p['h_normal']] + [p['h_tumour'] * f for f in fr]) for p in ps]
    dev, ora = _two_sets(oracle_mod, e, ps, max_cn, M, h_init=hs)
    b = dev.batch
    assert b.num_cn_states == S and b.info(3) >= 10
    _compare_after_every_update(dev, ora, sweeps=2)
    got = (b.info(12), b.info(14))
    print('state grid M=%d max_cn=%d S=%d: forward-backward kernel %d (restarts per workgroup %d), lattice kernel %d' % (M, max_cn, S, got[0], b.info(13), got[1]))
    if fb is not None:
        assert got == (fb, vit), ('kernel selection changed', got)


@pytest.mark.parametrize('R', [1, 2, 3])
def test_four_clones_ragged_unit_of_four_vectors_matches_oracle(hip, oracle_mod, R):
    """k_fbk pinned to four vectors per workgroup with fewer restarts than that (207 states: blocks of 512 threads publish the vectors in two passes of
    two): the absent vectors' emission requests must stay inside the arrays (a --big fuzz sequence faulted on a read two restarts past the end of the
    emission array, round 5) and the present ones must match the oracle."""
    from remixt_amd import synthetic
    e = synthetic.make_experiment(33, num_clones=4, max_copy_number=4, num_chains=2, seed=4037, num_breakpoints=8, chain_fractions=[1., 2.])
    ps = synthetic.make_init_params(e, R, 4, num_clones=4)
    hs = [np.array([p['h_normal']] + [p['h_tumour'] * f for f in (0.5, 0.3, 0.2)]) for p in ps]
    dev, ora = _two_sets(oracle_mod, e, ps, 4, 4, options={'fb_nv': 4, 'fb_wg_budget': 12}, h_init=hs)
    assert dev.batch.num_cn_states == 207
    _compare_after_every_update(dev, ora, sweeps=2)
    assert (dev.batch.info(12), dev.batch.info(13)) == (3, 4)


def test_kernel_selection_at_the_benchmark_grids(hip):
    """Which kernels the parametrisations above really run (docstrings named k_fbv where k_fbm runs)."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    # (max_cn, fb_nv, fb_kernel) -> (forward-backward kernel, restarts per workgroup, lattice kernel); 4 restarts x 2 chains x 2 directions
    # leave room for one restart per workgroup, which is what the automatic choice takes at 165 states
    # lattice kernel (rmx_info 14): 4 = k_viterbi_max (maxima forward, transition values in registers), 5 = k_viterbi_code_max (8-bit codes in LDS; option viterbi_cluster = 1), 6 = k_viterbi_sad_max (above 176 states: values from the packed copies, workgroup clusters); the
    # round-4 back-pointer forms 1 / 2 and the plain kernel 3 are reachable through option viterbi_plain = 2 / 1
    want = {(8, None, 0): (1, 1, 4), (8, 4, 0): (1, 4, 4), (8, 2, 0): (1, 2, 4), (8, 2, 3): (2, 2, 4), (8, 1, 3): (2, 1, 4),
            (12, None, 0): (4, 1, 6), (12, 4, 0): (4, 4, 6), (12, 2, 0): (4, 4, 6), (12, 2, 3): (3, 2, 6)}
    for (max_cn, nv, fk), (fb, nvx, vit) in want.items():
        e = synthetic.make_experiment(60, num_clones=3, max_copy_number=max_cn, num_chains=2, seed=3, num_breakpoints=4)
        rs = RestartSet(e, synthetic.make_init_params(e, 4, max_cn), max_cn, num_clones=3, quiet=True, options={'fb_nv': nv or 0, 'fb_kernel': fk})
        rs.batch.variational_update(1)
        rs.batch.infer_cn_batch(0, 2)
        got = (rs.batch.info(12), rs.batch.info(13), rs.batch.info(14))
        assert got[0] == fb and got[2] == vit and (nvx is None or got[1] == nvx), (max_cn, nv, fk, got)


# ---- ADVICE r2 ------------------------------------------------------------------------------------------------------------------
def test_an_unflagged_failure_does_not_inherit_the_previous_calls_restart_list(hip):
    """rmx_last_error_restarts describes the calling thread's LAST failing call only: a device-flagged failure followed by a
    failure that flags no restart (bad argument) must report an empty list (it used to keep the earlier one, and the batched
    driver would have failed the wrong restarts)."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(200, num_clones=3, max_copy_number=4, num_chains=3, seed=5)
    rs = RestartSet(e, synthetic.make_init_params(e, 3, 4), 4, num_clones=3, quiet=True, seeds=[1, 2, 3])
    b = rs.batch
    b.variational_update(1)
    smp = np.zeros(b.num_segments, dtype=np.int64); smp[::3] = 1
    for r in range(3):
        b._use_sample(r, smp)
    hs = np.array([np.asarray(m.model.h, dtype=float) for m in rs.models])
    hs[1] *= -1.                                              # total_depth <= 0 for restart 1 only
    with pytest.raises(ValueError) as flagged:
        b.expected_log_likelihood_h_batch([0, 1, 2], hs)
    assert flagged.value.restarts == [1]
    for r, m in enumerate(rs.models):
        m.model.h = np.abs(hs[r])
    with pytest.raises(ValueError) as plain:
        b.set_option('fb_nv', 16)                             # no such workgroup shape any more: RMX_EARG, flags nothing
    assert getattr(plain.value, 'restarts', []) == [] and hip.last_error_restarts() == []


def test_a_restart_whose_h_step_failed_is_not_retried_in_later_iterations(hip, monkeypatch):
    """a restart flagged during the lock-step h M-step stays out of later h M-steps (its message is never
    cleared, it can never be selected), so the next EM iteration runs the lock-step rounds exactly once."""
    from remixt_amd import synthetic, lockstep
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(300, num_clones=3, max_copy_number=4, num_chains=3, seed=23)
    ps = synthetic.make_init_params(e, 3, 4)
    rs = RestartSet(e, ps, 4, num_clones=3, quiet=True, seeds=[5, 6, 7])
    for m, v in zip(rs.models, rs.calculate_elbo()):
        m.prev_elbo = float(v)
    make = rs.batch.h_batch_evaluator
    poison = {'on': True, 'n': 0}

    def factory(restarts):
        restarts = list(restarts)
        evaluate = make(restarts)

        def wrapped(ids, xs):
            xs = [np.array(x, dtype=float) for x in xs]
            poison['n'] += 1
            listed = [restarts[i] for i in ids]
            if poison['on'] and poison['n'] == 2 and 1 in listed:
                xs[listed.index(1)] *= -1.
            return evaluate(ids, xs)
        return wrapped
    monkeypatch.setattr(rs.batch, 'h_batch_evaluator', factory)
    rs.em_iteration(0, 2)
    assert list(rs.error_messages) == [1]
    poison['on'] = False
    runs = []
    real = lockstep.lbfgsb_lockstep
    monkeypatch.setattr(lockstep, 'lbfgsb_lockstep', lambda x0s, bounds, ev: (runs.append(len(x0s)), real(x0s, bounds, ev))[1])
    h1 = np.array(rs.models[1].h, dtype=float)
    rs.em_iteration(1, 2)
    assert runs == [2]                                        # one lock-step run, without restart 1
    assert np.array_equal(np.array(rs.models[1].h, dtype=float), h1) and list(rs.error_messages) == [1]


@pytest.mark.parametrize('seed,N,max_cn,nc', [(71, 400, 4, True), (72, 500, 3, True), (73, 300, 4, True), (34, 48, 2, False), (33, 48, 2, False)])
def test_joint_accept_decides_like_the_sequential_accept_tests(hip, seed, N, max_cn, nc):
    """the accept tests of the four standard parameters from ONE pass over the cells (component sums, the default)
    against the reference's four sequential full-data tests (RestartSet(joint_accept=False)) on several seeds and shapes,
    rejections included (the second EM iteration starts near the optimum, where trial values are often rejected; without
    normal contamination six more parameters follow sequentially): same decisions, hence bit-identical h, parameters, ELBO."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    # (without normal contamination the weighted samples of the hdel / LOH parameters need min(200, N / 10) segments with
    # posterior mass on such states -- numpy's "Fewer non-zero entries in p than size" otherwise, in the reference too: short genomes)
    e = synthetic.make_experiment(N, num_clones=3, max_copy_number=max_cn, num_chains=(3 if nc else 2), seed=seed)
    ps = synthetic.make_init_params(e, 4, max_cn)
    out = []
    rejected = []
    for joint in (True, False):
        rs = RestartSet(e, ps, max_cn, num_clones=3, quiet=False, seeds=[seed + i for i in range(4)], joint_accept=joint, normal_contamination=nc)
        logs = []
        for m in rs.models:
            m._log = lambda msg, logs=logs: logs.append(msg)
        rs.fit(num_em_iter=3, num_update_iter=2)
        out.append([(m.prev_elbo, np.array(m.h), sorted(m.get_likelihood_param_values().items())) for m in rs.models])
        rejected.append(sorted(l.split(' rejected')[0] for l in logs if 'rejected' in l))
    assert rejected[0] == rejected[1]
    for a, b_ in zip(*out):
        assert a[0] == b_[0] and np.array_equal(a[1], b_[1]) and a[2] == b_[2]


def test_components_after_a_mixed_h_accept_need_no_refresh_pass(hip):
    """rmx_expected_ll_components(trial=3): after the h accept test of a batch in which some restarts keep their trial h and others are
    rolled back, the "before" values of the parameter accept tests come from the trial pass's scratch expectations (kept) and from the
    restart's own, still current ones (rolled back) -- equal, up to the rounding between the sparse trial pass and the dense refresh, to
    what a refresh pass over the cells gives; and refused when the last pass over the range was not that trial pass."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(400, num_clones=3, max_copy_number=6, num_chains=4, seed=21)
    ps = synthetic.make_init_params(e, 4, 6)
    rs = RestartSet(e, ps, max_copy_number=6, num_clones=3, quiet=True, seeds=[1, 2, 3, 4])
    b = rs.batch
    R = 4
    b.variational_update(2)
    with pytest.raises(NotImplementedError):
        b.expected_log_likelihood_components(0, R, trial=3)             # no trial pass yet
    before = b.expected_log_likelihood_full(0, R)
    own = b.expected_log_likelihood_components(0, R)
    h0 = [np.array(b.get_array(r, 'h')) for r in range(R)]
    for r in range(R):
        rs.models[r].model.h = h0[r] * (1.02 + 0.01 * r)
    after = b.expected_log_likelihood_full_trial(0, R)
    assert np.all(np.isfinite(after)) and not np.allclose(after, before, rtol=1e-9)
    for r in (1, 3):
        b.rollback_h(r, h0[r])                                          # restarts 1 and 3 reject, 0 and 2 keep the trial h
    mixed = b.expected_log_likelihood_components(0, R, trial=3)
    for r in (1, 3):
        assert np.array_equal(mixed[r], own[r])                         # their own expectations, untouched by the trial
    for r in (0, 2):
        assert abs(mixed[r].sum() - after[r]) <= 1e-9 * abs(after[r])   # the trial pass's E[ll], split into components
    dense = b.expected_log_likelihood_components(0, R)                  # the refresh pass the mixed call saves
    np.testing.assert_allclose(mixed, dense, rtol=1e-10)
    tried = b.expected_log_likelihood_components(0, R, trial=True)      # a new trial pass over the range ...
    with pytest.raises(NotImplementedError):
        b.expected_log_likelihood_components(0, R, trial=3)             # ... ends the validity of the first one's sums
    np.testing.assert_allclose(tried, dense, rtol=1e-10)                # (nothing on trial: the same expectations)


@pytest.mark.parametrize('N,max_cn,R', [(2400, 8, 5), (700, 6, 3), (420, 12, 2)])
def test_h_round_kernels_give_the_same_bits(hip, N, max_cn, R):
    """Round 5: the objective + gradient rounds of the lock-step h M-step with the lane chains laid out flat over the threads
    (k_gradflat_round, option grad_kernel 0, the default; layout and per-segment constants made once per M-step) against half a wave
    per sampled segment with the final sums folded in (1, round 4's form) and with the sums as a kernel of their own (2): the flat kernel
    adds a segment's units with the half-wave kernel's own reduction, so every value and every gradient component is BIT-identical --
    over several rounds on one layout, on a subset of the restarts, after the lists of states with posterior mass changed (another sweep)
    and after new samples; and the per-restart entry point (scipy's driver) still returns the same bits as the batched one."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(N, num_clones=3, max_copy_number=max_cn, num_chains=3, seed=N + 1)
    ps = synthetic.make_init_params(e, R, max_cn)
    rs = RestartSet(e, ps, max_copy_number=max_cn, num_clones=3, quiet=True, seeds=list(range(10, 10 + R)))
    b = rs.batch
    rng = np.random.RandomState(5)

    def rounds(live, hs):
        out = {}
        for kern in (0, 1, 2, 0):
            b.set_option('grad_kernel', kern)
            res = [b.expected_log_likelihood_h_batch(live, h) for h in hs]
            key = kern if kern not in out else 'again'
            out[key] = res
        for kern in (1, 2, 'again'):
            for (f0, g0), (f1, g1) in zip(out[0], out[kern]):
                assert np.array_equal(f0, f1) and np.array_equal(g0, g1), (kern, f0, f1, g0, g1)
        return out[0]

    for sweep in range(2):
        b.variational_update(1 + sweep)
        h0 = np.array([b.get_array(r, 'h') for r in range(R)])
        samples, lists = rs._samples_and_lists()
        b.set_sample_lists([(r, -1, samples[r], lists[r]) for r in range(R)])
        hs = [h0 * (1. + 0.02 * rng.rand(R, 3)) for _ in range(3)]
        full = rounds(list(range(R)), hs)
        assert all(np.all(np.isfinite(f)) and np.all(np.isfinite(g)) for f, g in full)
        if R > 2:
            live = [0, R - 1]
            sub = rounds(live, [h[live] for h in hs])
            for (f0, g0), (f1, g1) in zip(full, sub):      # a request's sums do not depend on which other requests travel with it
                assert np.array_equal(f0[live], f1) and np.array_equal(g0[live], g1)
        # the per-restart entry point (what scipy's own driver calls): the same bits as the batched round
        b.set_option('grad_kernel', 0)
        f_b, g_b = b.expected_log_likelihood_h_batch(list(range(R)), hs[1])
        for r in range(R):
            rs.models[r].model.h = hs[1][r]
            f1 = rs.models[r].model.calculate_expected_log_likelihood(samples[r])
            g1 = np.zeros(3); rs.models[r].model.calculate_expected_log_likelihood_partial_h(samples[r], g1)
            assert f1 == f_b[r] and np.array_equal(g1, g_b[r]), (r, f1, f_b[r], g1, g_b[r])
        for r in range(R):
            rs.models[r].model.h = h0[r]


@pytest.mark.parametrize('N,max_cn', [(403, 6), (90, 8), (1000, 4)])
def test_trial_pass_kernels_agree_with_each_other_and_with_the_dense_refresh(hip, N, max_cn):
    """Round 5: the M-step's trial passes with the (segment, listed state) cells laid out flat over the threads (k_trial_flat, option
    trial_kernel 0, the default: a segmented sum in list order) against the quarter-wave-per-segment kernel (k_trial_sparse, 1) and against
    the dense refresh of the same expectations: for h on trial (all four components), for one and for two likelihood parameters on trial
    (component masks 1 / 4 / 12 ...), on segment counts that leave a ragged last block -- the same sums to rounding (1e-12 of the restart's
    E[ll]; the kernels add a segment's products in different orders), bit-identical when repeated and across batch ranges."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartSet
    e = synthetic.make_experiment(N, num_clones=3, max_copy_number=max_cn, num_chains=3, seed=N)
    ps = synthetic.make_init_params(e, 3, max_cn)
    R = 3
    trials = [('h', None), ('negbin_r_0', 310.), ('betabin_M_0', 1200.), (('betabin_M_0', 'betabin_M_1'), (900., 14.)), (('negbin_r_0', 'negbin_r_1', 'betabin_M_1'), (250., 7., 33.))]
    got = {}
    for kern in (0, 1):
        rs = RestartSet(e, ps, max_copy_number=max_cn, num_clones=3, quiet=True, seeds=[1, 2, 3], options={'trial_kernel': kern})
        b = rs.batch
        b.variational_update(2)
        assert b.get_option('trial_kernel') == kern
        h0 = [np.array(b.get_array(r, 'h')) for r in range(R)]
        for name, value in trials:
            if name == 'h':
                for r in range(R):
                    rs.models[r].model.h = h0[r] * (1.03 + 0.01 * r)
                full = b.expected_log_likelihood_full_trial(0, R)
                comp = b.expected_log_likelihood_components(0, R, trial=2)
                again = b.expected_log_likelihood_full_trial(0, R)
                part = b.expected_log_likelihood_full_trial(1, R)              # a sub-range of the batch: the same bits
                assert np.array_equal(full, again) and np.array_equal(full[1:], part)
                for r in range(R):
                    b.rollback_h(r, h0[r])
                dense = []
                for r in range(R):
                    rs.models[r].model.h = h0[r] * (1.03 + 0.01 * r)
                dense = b.expected_log_likelihood_components(0, R)             # the dense refresh pass at the same h
                np.testing.assert_allclose(comp, dense, rtol=1e-10)
                for r in range(R):
                    rs.models[r].model.h = h0[r]
                b.expected_log_likelihood_components(0, R)
                got[kern, name] = (full, comp)
            else:
                names = (name,) if isinstance(name, str) else name
                values = (value,) if isinstance(name, str) else value
                before = [[b.get_param(r, nm) for r in range(R)] for nm in names]
                for nm, v in zip(names, values):
                    for r in range(R):
                        b.set_param(r, nm, float(v) * (1. + 0.02 * r))
                tried = b.expected_log_likelihood_components(0, R, trial=True)
                again = b.expected_log_likelihood_components(0, R, trial=True)
                assert np.array_equal(tried, again)
                for nm, bv in zip(names, before):
                    for r in range(R):
                        b.rollback_param(r, nm, bv[r])
                got[kern, name] = (tried,)
    for name, _ in trials:
        for x, y in zip(got[0, name], got[1, name]):
            scale = np.abs(np.asarray(y)).sum(axis=-1, keepdims=True) if np.ndim(y) > 1 else np.abs(y)
            assert np.all(np.abs(np.asarray(x) - np.asarray(y)) <= 1e-12 * scale), (name, x, y)


@pytest.mark.parametrize('max_cn,kernel', [(12, 4), (8, 1)])
def test_paced_and_free_running_groups_equal_one_group(hip, max_cn, kernel):
    """Restart groups of a GPU (own stream, own host thread), free-running and paced (pace_sweeps: a group reaches a sweep's forward-backward
    point only after its previous forward-backward launch has finished): the posteriors of every restart equal the one-group run's bit for bit."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    e = synthetic.make_experiment(2400, num_clones=3, max_copy_number=max_cn, num_chains=3, seed=12, num_breakpoints=6)
    ps = synthetic.make_init_params(e, 8, max_cn)
    out = {}
    for groups, paced in ((1, False), (2, False), (2, True)):
        rs = RestartGroups(e, ps, max_cn, groups=groups, num_clones=3, quiet=True, seeds=list(range(8)), paced=paced, options={'fb_nv': 4})
        rs.variational_update(3)                      # both groups' threads enter their sweeps together
        b = rs.batches[0]
        assert b.info(12) == kernel and b.info(13) == 4
        assert rs.paced == paced and b.get_option('pace_sweeps') == int(paced)
        out[groups, paced] = [(np.array(m.model.posterior_marginals), np.array(m.model.p_breakpoint), np.array(m.model.p_allele_swap)) for m in rs.models]
    for key in ((2, False), (2, True)):
        for r in range(8):
            for x, y in zip(out[key][r], out[1, False][r]):
                assert np.array_equal(x, y), (key, r)
