"""CPU: this project's host logic (state grids, breakend remap, masks, restart selection,
brute-force search) against golden vectors recorded from the reference's cn_model.py."""
import numpy as np
import pytest
import scipy.optimize

from remixt_amd import cn_model, restarts
from tests import golden_runner as GR


def test_state_grids():
    g = GR.load('state_grids')
    for key in g.files:
        kind, M, cn = key.split('_')
        if kind == 'cn':
            mine = cn_model.create_cn_states(int(M), 2, int(cn), 1)
        else:
            mine = cn_model.create_brk_states(int(M), int(cn), 1)
        assert mine.dtype == np.int64 and np.array_equal(mine, g[key]), key
    # grid sizes quoted in SURVEY.md 0.3
    assert len(cn_model.create_cn_states(3, 2, 8, 1)) == 165 and len(cn_model.create_brk_states(3, 8, 1)) == 25
    assert len(cn_model.create_cn_states(3, 2, 12, 1)) == 355


@pytest.mark.parametrize('layout', ['interior', 'telomere', 'two_at_one_boundary', 'left_edge', 'mixed'])
def test_breakend_remap(layout):
    g = GR.load('remap')
    adj = set((int(a), int(b)) for a, b in g['adjacencies'])
    brk = {}
    for k, be in zip(g[layout + '/ids'], g[layout + '/breakends']):
        brk[str(k)] = frozenset([(int(be[0][0]), int(be[0][1])), (int(be[1][0]), int(be[1][1]))])
    m = cn_model.BreakpointModel(g['x'], g['l'], adj, brk, max_copy_number=2, max_depth=1.0, min_segment_length=0., quiet=True)
    assert m.N1 == int(g[layout + '/N1'])
    for a in ['seg_fwd_remap', 'seg_rev_remap', 'seg_is_original', 'is_telomere', 'breakpoint_idx', 'breakpoint_orient', 'x1', 'l1']:
        assert np.array_equal(np.asarray(getattr(m, a)), g[layout + '/' + a]), (layout, a)


def _remap_walk(N, adjacencies, breakpoints):
    """The reference's walk over the segment boundaries (cn_model.py:96-147), one Python iteration per segment: what BreakpointModel._remap_segments
    computed until round 5 and now computes without the per-segment loop."""
    import collections
    at_boundary = collections.defaultdict(set)
    for bp_idx, breakpoint in enumerate(breakpoints):
        for be_idx, breakend in enumerate(breakpoint):
            n, orient = cn_model._get_brkend_seg_orient(breakend)
            at_boundary[n].add((bp_idx, be_idx, orient))
    seg_rev, is_orig, is_tel, bidx, borient = [], [], [], [], []
    fwd = np.zeros(N, dtype=int)

    def push(n, original, telomere, bp=-1, orient=0):
        seg_rev.append(n); is_orig.append(original); is_tel.append(telomere); bidx.append(bp); borient.append(orient)
    for n in range(-1, N):
        adjacent = (n, n + 1) in adjacencies
        if n in at_boundary:
            first = True
            for bp_idx, be_idx, orient in at_boundary[n]:
                if first and n >= 0:
                    fwd[n] = len(seg_rev)
                push(n, first and n >= 0, 0, bp_idx, orient)
                first = False
            if not adjacent:
                push(n, False, 1)
        elif n >= 0:
            fwd[n] = len(seg_rev)
            push(n, True, 0 if adjacent else 1)
    return fwd, np.array(seg_rev), np.array(is_orig, dtype=bool), np.array(is_tel), np.array(bidx), np.array(borient)


@pytest.mark.parametrize('seed', range(12))
def test_breakend_remap_equals_the_walk_over_every_boundary(seed):
    """Random layouts -- chains of one to many segments, breakends at the left edge (boundary -1), at chain ends, several at one boundary,
    both ends of a breakpoint at the same boundary -- against the per-segment walk."""
    rng = np.random.RandomState(seed)
    N = int(rng.choice([1, 2, 3, 7, 40, 300]))
    cuts = set(int(c) for c in rng.choice(N, size=min(N, int(rng.randint(0, 5))), replace=False)) if N > 1 else set()
    adj = set((n, n + 1) for n in range(N - 1) if n not in cuts)
    brk = []
    for k in range(int(rng.randint(1, max(2, N // 2 + 1)))):
        ends = []
        for _ in range(2):
            n = int(rng.randint(0, N)); side = int(rng.randint(0, 2))
            ends.append((n, side))
        if ends[0] == ends[1]:
            ends[1] = (ends[1][0], 1 - ends[1][1])
        brk.append(frozenset(ends))
    x = np.tile(np.array([[6., 4., 100.]]), (N, 1)); l = np.ones(N) * 1e5
    m = cn_model.BreakpointModel(x, l, adj, dict((str(i), b) for i, b in enumerate(brk)), max_copy_number=2, max_depth=1.0, min_segment_length=0., quiet=True)
    fwd, rev, orig, tel, bidx, borient = _remap_walk(N, adj, list(m.breakpoints))
    assert m.N1 == len(rev)
    for mine, want in ((m.seg_fwd_remap, fwd), (m.seg_rev_remap, rev), (m.seg_is_original, orig), (m.is_telomere, tel), (m.breakpoint_idx, bidx), (m.breakpoint_orient, borient)):
        assert np.asarray(mine).dtype == want.dtype and np.array_equal(np.asarray(mine), want)


def test_state_tables_are_the_row_wise_unique_of_the_normal_copies():
    """_state_tables finds the classes of segments (distinct rows of normal copies, lexicographic order) through an integer key per row; the
    definition is numpy's row-wise unique (cn_model.py:359-364 builds the full [N1, S, M, 2] array the classes compress)."""
    rng = np.random.RandomState(4)
    N = 500
    x = np.tile(np.array([[6., 4., 100.]]), (N, 1)); l = np.ones(N) * 1e5
    adj = set((n, n + 1) for n in range(N - 1) if n % 97 != 96)
    m = cn_model.BreakpointModel(x, l, adj, {'a': frozenset([(3, 1), (250, 0)]), 'b': frozenset([(0, 0), (499, 1)])}, max_copy_number=3, max_depth=1.0,
                                 min_segment_length=0., quiet=True, normal_copies=rng.randint(0, 3, size=(N, 2)))
    classes, seg_class = m._state_tables(3)
    rows = np.asarray(m.normal_copies)[m.seg_rev_remap]
    uniq, inv = np.unique(rows, axis=0, return_inverse=True)
    assert classes.dtype == np.int64 and seg_class.dtype == np.int32 and len(classes) == len(uniq)
    assert np.array_equal(classes[:, :, 0, :], np.repeat(uniq[:, None, :], classes.shape[1], axis=1)) and np.array_equal(seg_class, np.asarray(inv).reshape(-1))
    full = classes[seg_class]                      # the reference's cn_states
    assert np.array_equal(full[:, :, 0, :], np.repeat(rows[:, None, :], classes.shape[1], axis=1))


def test_constructor_errors():
    x = np.array([[6., 4., 100.]] * 3); l = np.ones(3) * 1e5
    with pytest.raises(ValueError):
        cn_model.BreakpointModel(x, l, {(0, 1)}, {'a': frozenset([(0, 1), (2, 0)])})          # max_depth is mandatory
    with pytest.raises(ValueError):
        cn_model.BreakpointModel(x, l, {(0, 1)}, {}, max_depth=1.)                            # empty breakpoints (reference quirk)
    with pytest.raises(AssertionError):
        cn_model.BreakpointModel(x[:, [1, 0, 2]], l, {(0, 1)}, {'a': frozenset([(0, 1), (2, 0)])}, max_depth=1.)   # minor > major


def test_brute_1d_is_scipy_brute():
    calls = []

    def f(v):
        assert v.shape == (1,)
        calls.append(float(v[0]))
        x = float(v[0])
        if x < 10 or x > 2000:
            return np.inf
        return (np.log(x) - 5.3) ** 2 + 0.01 * np.sin(x / 50.)

    ref = scipy.optimize.brute(f, ranges=[(10., 2000.)], full_output=True)
    ref_calls = list(calls); calls.clear()

    class Dummy(object):
        pass
    bm = cn_model.BreakpointModel.__new__(cn_model.BreakpointModel)
    bm.model = Dummy()
    mine = bm._brute_1d(f, 'x', (10., 2000.), None)
    assert mine == float(ref[0][0]) and calls == ref_calls


def test_decode_breakpoints_naive():
    cn = np.array([[[1, 1], [2, 1]], [[1, 1], [1, 1]], [[1, 1], [1, 1]], [[1, 1], [3, 1]]])
    adj = [(0, 1), (1, 2), (2, 3)]
    brk = {'a': frozenset([(0, 1), (3, 0)])}
    out = cn_model.decode_breakpoints_naive(cn, adj, brk)
    assert np.array_equal(out['a'], [0, 1])


def test_select_optimal():
    res = {i: {'stats': {'elbo': e, 'proportion_divergent': p}} for i, (e, p) in enumerate([(-10., 0.1), (-5., 0.9), (-7., 0.2), (-7., 0.3)])}
    assert restarts.select_optimal(res, 0.5) == 2          # best ELBO among the admissible, first on ties
    assert restarts.select_optimal(res, 0.05) == 1         # nothing admissible: best overall
    assert restarts.shard_indices(10, 4, 1) == [1, 5, 9]


def test_ploidy_and_divergence_statistics_equal_the_reference_expressions_bit_for_bit():
    """restarts.tumour_ploidy_and_divergence against the literal expressions of analysis/pipeline.py:215-217 (reductions over the clone
    axis written clone by clone: numpy's reductions over an axis of 2-3 elements took 7 ms per restart of a fit's result records)."""
    from remixt_amd.restarts import tumour_ploidy_and_divergence
    rng = np.random.RandomState(3)
    for N, M in ((5000, 3), (777, 4), (9, 2), (1, 3)):
        cn = rng.randint(0, 9, size=(N, M, 2)).astype(np.int64); l = rng.uniform(1e5, 1e6, size=N)
        ploidy = (cn[:, 1:, :].mean(axis=1).T * l).sum() / l.sum()
        divergent = (cn[:, 1:, :].max(axis=1) != cn[:, 1:, :].min(axis=1)) * 1.
        p2, d2 = tumour_ploidy_and_divergence(cn, l)
        assert p2 == ploidy and np.array_equal(d2, divergent) and (d2.T * l).sum() == (divergent.T * l).sum()
