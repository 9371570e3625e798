"""GPU: BASELINE.json configs[2] / [3] / [4] AT THEIR WORKLOAD (50 000 segments x 165 states), through size-independent
properties (no -m gpu test ran 64 restarts, two datasets, or an EM iteration with M-steps at this size).

The oracle needs 33 GB and ~15 minutes per restart and sweep here, so what is asserted is what the domain guarantees at any
size: an EM iteration (coordinate-ascent sweeps + M-steps whose accept tests compare full-data E[ll], cn_model.py:497-505,
563-569) never lowers a restart's ELBO; restarts are independent (reference: one process per init_id, workflow.py:329-340), so
a restart's trajectory is BIT-IDENTICAL whatever batch, restart group, dataset group or GPU share it runs in; posterior rows
are distributions; the batched decode equals the plain lattice.  Equality with the oracle at these state grids is the
business of tests/test_hip_bench_shapes.py (small N)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEG, MAX_CN, M = 50000, 8, 3


@pytest.fixture(scope='module')
def hip():
    from remixt_amd import bpmodel
    return bpmodel


@pytest.fixture(scope='module')
def workload(hip):
    from remixt_amd import synthetic
    e = synthetic.make_experiment(SEG, num_clones=M, max_copy_number=MAX_CN, num_chains=23, seed=0)
    e2 = synthetic.resample_counts(e, seed=101)           # configs[4]: second tumour sample, same segmentation and breakpoints
    params64 = synthetic.make_init_params(e, 64, MAX_CN, num_clones=M)
    return e, e2, params64


def _seeds(ids, base=1000):
    return [base + i for i in ids]


def _state(rs):
    """What a restart's EM trajectory leaves behind: (ELBO, h, likelihood parameters) per restart."""
    out = []
    for m in rs.models:
        pv = m.get_likelihood_param_values()
        out.append((float(m.prev_elbo), np.array(m.h, dtype=float), np.array([pv[k] for k in sorted(pv)])))
    return out


def _run(rs, iters=1):
    e0 = np.asarray(rs.calculate_elbo(), dtype=float)
    for m, v in zip(rs.models, e0):
        m.prev_elbo = float(v)
    elbo = e0
    for i in range(iters):
        new = np.asarray(rs.run(1, i, 5), dtype=float)
        assert np.all(np.isfinite(new))
        assert np.all(new >= elbo - 1e-9 * np.abs(elbo)), ('an EM iteration lowered an ELBO', i, elbo, new)
        elbo = new
    rs.synchronize()
    return e0, elbo


def _same(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def _release(rs):
    import gc
    for s_ in rs.sets:
        s_.batch = None
        for m in s_.models:
            m.model = None
    gc.collect()


def test_config2_bench_shape_em_iterations_with_msteps(workload):
    """The benchmark's own shape: 16 restarts as 2 restart groups of 8, EM iterations WITH M-steps (5 sweeps, lock-step h
    M-step, shared-round parameter searches, joint accept, ELBO): ELBO never decreases, and every restart's trajectory equals
    the one-group run (forward-backward launches of 16) bit for bit."""
    from remixt_amd.restarts import RestartGroups
    e, _, p64 = workload
    ids = list(range(16))
    out = {}
    for groups, paced in ((2, True), (2, False), (1, False)):
        # two groups launch 8 restarts at a time: two per forward-backward workgroup (k_fbm<., 2>, vector ALU; 184 workgroups), chosen by the
        # library; the one-group run (16 per launch: four per workgroup by itself) is pinned to the same shape -- bit-identity across launch
        # sizes holds per workgroup shape (test_s165_workgroup_shapes_agree_and_subranges_are_bit_identical)
        # The parameter-search driver is pinned too: RestartGroups gives a single group and paced groups the device-driven rounds
        # (search_mode 5: device log(), per-block cell sums) and free-running groups the host-driven ones (0); the two agree to rounding, not to
        # the bit, so bit-identity across groupings holds per search mode and workgroup shape.
        opts = {'search_mode': 0}
        if groups == 1:
            opts['fb_nv'] = 2
        rs = RestartGroups(e, [p64[i] for i in ids], MAX_CN, groups=groups, num_clones=M, quiet=True, seeds=_seeds(ids), paced=paced, options=opts)
        b = rs.batches[0]
        assert b.num_cn_states == 165 and b.num_segments >= SEG
        e0, e2 = _run(rs, iters=2)
        assert b.info(12) == 1 and b.info(13) == 2          # k_fbm, two restarts per workgroup
        assert rs.paced == paced and b.get_option('pace_sweeps') == int(paced) and b.get_option('search_mode') == 0
        out[groups, paced] = _state(rs)
        _release(rs)
    for r in ids:
        assert _same(out[2, False][r], out[1, False][r]), ('restart %d: 2 groups vs 1 group' % r, out[2, False][r], out[1, False][r])
        assert _same(out[2, True][r], out[1, False][r]), ('restart %d: 2 paced groups vs 1 group' % r, out[2, True][r], out[1, False][r])
    # the automatic choice for one group (device-driven search rounds, four restarts per workgroup) against the pinned runs: at tolerance
    rs = RestartGroups(e, [p64[i] for i in ids], MAX_CN, groups=1, num_clones=M, quiet=True, seeds=_seeds(ids))
    _run(rs, iters=2)
    assert rs.batches[0].get_option('search_mode') == 5 and rs.batches[0].info(13) == 4
    auto = _state(rs)
    _release(rs)
    for r in ids:
        assert abs(auto[r][0] - out[1, False][r][0]) <= 1e-7 * abs(auto[r][0]), (r, auto[r][0], out[1, False][r][0])
        np.testing.assert_allclose(auto[r][1], out[1, False][r][1], rtol=1e-5)
        np.testing.assert_allclose(auto[r][2], out[1, False][r][2], rtol=1e-3)


def test_config3_per_gpu_share_and_the_whole_job_on_one_gpu(workload):
    """configs[3]: 64 restarts sharded over 8 GPUs, restart i on rank i mod 8.  Rank 0's share (8 restarts, 2 groups of 4) and
    the WHOLE 64-restart job on this one GPU (4 groups of 16: the `bench.py --total-restarts 64` shape): ELBO monotone for all
    64; the share's restarts come out bit-identical inside the whole; posteriors are distributions; the batched decode of a
    group equals the plain lattice."""
    from remixt_amd.restarts import RestartGroups, shard_indices
    e, _, p64 = workload
    share = shard_indices(64, 8, 0)
    assert share == list(range(0, 64, 8))
    # the share as a rank runs it: 4 restarts per launch leave room for one restart per forward-backward workgroup (k_fbm<., 1>, vector ALU) ...
    rs = RestartGroups(e, [p64[i] for i in share], MAX_CN, groups=2, num_clones=M, quiet=True, seeds=_seeds(share))
    _run(rs, iters=1)
    assert rs.batches[0].info(12) == 1 and rs.batches[0].info(13) == 1
    part_auto = dict(zip(share, _state(rs)))
    _release(rs)
    # ... and with the workgroup shape of the whole job's 16-restart launches (four per workgroup, matrix cores): the shapes sum a column in
    # different orders, so bit-identity across launch sizes holds per shape -- and per search driver: two paced groups of 4 get the device-driven
    # rounds (search_mode 5) by themselves, the whole job's free-running groups of 16 the host-driven ones (0)
    rs = RestartGroups(e, [p64[i] for i in share], MAX_CN, groups=2, num_clones=M, quiet=True, seeds=_seeds(share), options={'fb_nv': 4, 'search_mode': 0})
    _run(rs, iters=1)
    part = dict(zip(share, _state(rs)))
    _release(rs)

    ids = list(range(64))
    rs = RestartGroups(e, p64, MAX_CN, groups=4, num_clones=M, quiet=True, seeds=_seeds(ids), options={'search_mode': 0})
    assert len(rs.batches) == 4 and all(b.num_restarts == 16 for b in rs.batches)
    _run(rs, iters=1)
    whole = _state(rs)
    assert rs.batches[0].info(13) == 4
    for i in share:
        assert _same(part[i], whole[i]), ('restart %d: alone on a rank vs inside the 64-restart job' % i, part[i], whole[i])
        # one EM iteration (sweeps + scipy / Nelder-Mead M-steps) on the other workgroup shape: rounding-level differences of the posteriors, amplified by the optimisers
        assert abs(part_auto[i][0] - whole[i][0]) <= 1e-7 * abs(whole[i][0]), (i, part_auto[i][0], whole[i][0])
        np.testing.assert_allclose(part_auto[i][1], whole[i][1], rtol=1e-5)
        np.testing.assert_allclose(part_auto[i][2], whole[i][2], rtol=1e-3)
    b = rs.batches[3]
    for r in (0, 15):
        post = b.get_array(r, 'posterior_marginals')
        assert post.min() >= 0. and np.allclose(post.sum(axis=1), 1., rtol=0, atol=1e-12)
        pb = b.get_array(r, 'p_breakpoint')
        assert pb.min() >= 0. and np.allclose(pb.sum(axis=1), 1., rtol=0, atol=1e-12)
    cn_all, lp_all = b.infer_cn_batch(0, 16)
    assert b.info(14) == 4                                   # k_viterbi_max + k_backtrace_max (the reference's formulation: maxima forward, arg-maxima in the trace-back)
    b.set_option('viterbi_plain', 1)
    for r in (0, 7, 15):
        cn, lp = b.infer_cn(r)
        assert b.info(14) == 3 and np.array_equal(cn, cn_all[r]) and lp == lp_all[r]
    b.set_option('viterbi_plain', 2)                         # round 4's lattice with back-pointers, all restarts side by side
    cn_bp, lp_bp = b.infer_cn_batch(0, 16)
    assert b.info(14) == 1 and np.array_equal(cn_bp, cn_all) and np.array_equal(lp_bp, lp_all)
    b.set_option('viterbi_plain', 0)
    _release(rs)


def test_config4_two_datasets_at_the_workload(workload):
    """configs[4]: two tumour samples on one segmentation and breakpoint set, fitted independently (workflow.py:472-485): a
    GPU's share of the 64-restart job (8 units = 4 restarts per dataset) and 8 restarts per dataset, through DatasetGroups at
    50 000 x 165.  ELBO monotone; dataset 1's restarts equal their single-dataset fit bit for bit; the two datasets differ."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import DatasetGroups, RestartGroups
    e, e2, p64 = workload
    p2 = synthetic.make_init_params(e2, 64, MAX_CN, num_clones=M)
    for per_ds in (4, 8):
        ids = list(range(per_ds))
        dg = DatasetGroups([e, e2], [[p64[i] for i in ids], [p2[i] for i in ids]], MAX_CN, groups=2, num_clones=M, quiet=True,
                           seeds=[_seeds(ids), _seeds(ids, 8919)])
        assert len(dg.batches) == 4
        _run(dg, iters=1)
        both = _state(dg)
        res = dg.results_by_dataset() if per_ds == 4 else None
        _release(dg)
        alone = RestartGroups(e2, [p2[i] for i in ids], MAX_CN, groups=2, num_clones=M, quiet=True, seeds=_seeds(ids, 8919))
        _run(alone, iters=1)
        single = _state(alone)
        for r in ids:
            assert _same(both[per_ds + r], single[r]), ('dataset 1 restart %d: next to dataset 0 vs alone' % r)
            assert both[r][0] != both[per_ds + r][0]                       # different read counts, different fits
        if res is not None:
            sres = alone.results()
            for r in ids:
                assert np.array_equal(res[1][r]['cn'], sres[r]['cn'])
                assert all(np.array_equal(res[1][r]['brk_cn'][k], sres[r]['brk_cn'][k]) for k in sres[r]['brk_cn'])
        _release(alone)


# ---- the metric string's own configuration: 50 000 segments x 355 states (max_cn = 12, remixt/defaults.py:117) -----------------------------
@pytest.fixture(scope='module')
def workload355(hip):
    from remixt_amd import synthetic
    e = synthetic.make_experiment(SEG, num_clones=M, max_copy_number=12, num_chains=23, seed=0)
    return e, synthetic.make_init_params(e, 16, 12, num_clones=M)


def test_states355_bench_shape_em_iterations_with_msteps(workload355):
    """The `states_355` line of the bench at its workload: 16 restarts as two paced restart groups of 8, EM iterations
    WITH M-steps, 50 000 x 355 -- k_fbq, the sparse pairwise kernel, paced groups.  An EM iteration never lowers a restart's ELBO; every
    restart's trajectory (ELBO, h, parameters) equals the one-group run's bit for bit; posteriors and breakpoint probabilities are
    distributions; and the batched decode (k_viterbi_code) equals the plain lattice."""
    from remixt_amd.restarts import RestartGroups
    e, p16 = workload355
    ids = list(range(16))
    out = {}
    for groups in (2, 1):
        rs = RestartGroups(e, p16, 12, groups=groups, num_clones=M, quiet=True, seeds=_seeds(ids), options={'fb_nv': 4, 'search_mode': 5})      # (both pinned: bit-identity is per workgroup shape and search driver)
        b = rs.batches[0]
        assert b.num_cn_states == 355 and b.num_segments >= SEG
        assert rs.paced == (groups == 2)                      # what RestartGroups chooses above 200 states
        _run(rs, iters=2)
        assert b.info(12) == 4                                # k_fbq
        out[groups] = _state(rs)
        if groups == 2:
            b = rs.batches[1]
            for r in (0, 7):
                post = b.get_array(r, 'posterior_marginals')
                assert post.shape[1] == 355 and post.min() >= 0. and np.allclose(post.sum(axis=1), 1., rtol=0, atol=1e-12)
                pb = b.get_array(r, 'p_breakpoint')
                assert pb.min() >= 0. and np.allclose(pb.sum(axis=1), 1., rtol=0, atol=1e-12)
            cn_all, lp_all = b.infer_cn_batch(0, 8)
            assert b.info(14) == 6 and b.info(18) == 8        # k_viterbi_sad_max, eight workgroups per restart (the rows exchanged through memory) + k_backtrace_sad
            b.set_option('viterbi_cluster', 1)
            cn_c, lp_c = b.infer_cn_batch(0, 8)
            assert b.info(14) == 5 and np.array_equal(cn_c, cn_all) and np.array_equal(lp_c, lp_all)      # k_viterbi_code_max + k_backtrace_max (one workgroup per restart)
            b.set_option('viterbi_cluster', 0)
            b.set_option('viterbi_plain', 1)
            for r in (0, 5):
                cn, lp = b.infer_cn(r)
                assert b.info(14) == 3 and np.array_equal(cn, cn_all[r]) and lp == lp_all[r]
            b.set_option('viterbi_plain', 2)
            cn_bp, lp_bp = b.infer_cn_batch(0, 8)
            assert b.info(14) == 2 and np.array_equal(cn_bp, cn_all) and np.array_equal(lp_bp, lp_all)
            b.set_option('viterbi_plain', 0)
        _release(rs)
    for r in ids:
        assert _same(out[2][r], out[1][r]), ('restart %d: 2 paced groups vs 1 group at 355 states' % r, out[2][r], out[1][r])
