"""count / breakpoint tables -> Experiment (adjacencies, breakpoint -> segment-end
mapping, chains) against vectors recorded from the reference's Experiment -- CPU only."""
import os

import numpy as np
import pandas as pd

from remixt_amd.analysis import experiment as ex

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'experiment_tables.npz')


def _tables():
    d = np.load(GOLD, allow_pickle=False)
    counts = pd.DataFrame(dict((k[len('counts/'):], d[k]) for k in d.files if k.startswith('counts/')))
    brk = pd.DataFrame(dict((k[len('brk/'):], d[k]) for k in d.files if k.startswith('brk/')))
    brk = brk[ex.BREAKPOINT_COLUMNS]
    return d, counts, brk


def test_find_closest():
    d, _, _ = _tables()
    idx, dist = ex.find_closest(d['fc/a'], d['fc/v'])
    assert np.array_equal(idx, d['fc/idx']) and np.array_equal(dist, d['fc/dist'])


def test_experiment_matches_reference():
    d, counts, brk = _tables()
    e = ex.Experiment(counts, brk)
    assert np.array_equal(np.array(sorted(e.adjacencies), dtype=np.int64).reshape(-1, 2), d['adjacencies'])
    bsd = e.breakpoint_segment_data
    for c in ('prediction_id', 'n_1', 'side_1', 'n_2', 'side_2'):
        assert np.array_equal(bsd[c].values.astype(np.int64), d['bsd/' + c]), c
    assert np.array_equal(np.array(e.chains, dtype=np.int64), d['chains'])
    assert np.array_equal(e.x, d['x']) and np.array_equal(e.l, d['l'])
    closest = ex.find_closest_segment_end(e.count_data, e.breakpoint_data).sort_values(['prediction_id', 'prediction_side'])
    for c in ('prediction_id', 'prediction_side', 'dist', 'segment_idx', 'segment_side'):
        assert np.array_equal(closest[c].values.astype(np.int64), d['closest/' + c]), c
    # what the model consumes: id -> frozenset of (segment, side), integers
    bp = e.breakpoints
    assert sorted(bp) == sorted(d['bsd/prediction_id'].tolist())
    for pid, n1, s1, n2, s2 in zip(d['bsd/prediction_id'], d['bsd/n_1'], d['bsd/side_1'], d['bsd/n_2'], d['bsd/side_2']):
        assert bp[pid] == frozenset([(int(n1), int(s1)), (int(n2), int(s2))])
        assert all(isinstance(v, int) for end in bp[pid] for v in end)
    # the merged table keeps the breakpoint columns the result tables join on
    assert set(ex.BREAKPOINT_COLUMNS) <= set(bsd.columns)


def test_create_experiment_from_tsv_and_model_construction(tmp_path):
    """counts.tsv + breakpoints.tsv -> pickled Experiment -> BreakpointModel host-side remap (no kernel)."""
    import pickle
    d, counts, brk = _tables()
    counts.to_csv(tmp_path / 'counts.tsv', sep='\t', index=False)
    brk.to_csv(tmp_path / 'breakpoints.tsv', sep='\t', index=False)
    e = ex.create_experiment(str(tmp_path / 'counts.tsv'), str(tmp_path / 'breakpoints.tsv'), str(tmp_path / 'experiment.pickle'))
    with open(tmp_path / 'experiment.pickle', 'rb') as f:
        e2 = pickle.load(f)
    assert e2.breakpoints == e.breakpoints and e2.adjacencies == e.adjacencies
    assert np.array_equal(e.breakpoint_segment_data['n_1'].values.astype(np.int64), d['bsd/n_1'])
    e3 = ex.create_experiment(str(tmp_path / 'counts.tsv'), str(tmp_path / 'breakpoints.tsv'), str(tmp_path / 'e3.pickle'), min_length=float(np.median(counts['length'])))
    assert len(e3.l) < len(e.l) and all(n < len(e3.l) for bp in e3.breakpoints.values() for n, _ in bp)
    # no breakpoints at all
    e4 = ex.Experiment(counts)
    assert len(e4.breakpoint_segment_data) == 0 and e4.breakpoints == {}
    seg = ex.create_segment_table(e)
    assert len(seg) == len(counts) and np.all(np.isfinite(seg['total_depth']))
    # the three count columns keep experiment.x's dtype (the reference builds them from experiment.x[:, i] unchanged, analysis/experiment.py:333-342)
    for col, i in (('major_readcount', 0), ('minor_readcount', 1), ('readcount', 2)):
        assert seg[col].dtype == np.asarray(e.x).dtype and np.array_equal(seg[col].values, np.asarray(e.x)[:, i]), col

    class IntCounts(object):
        pass
    ei = IntCounts()
    for name in ('segment_chromosome_id', 'segment_start', 'segment_end', 'segment_major_is_allele_a', 'l'):
        setattr(ei, name, getattr(e, name))
    ei.x = np.asarray(e.x).astype(np.int64)
    segi = ex.create_segment_table(ei)
    assert segi['readcount'].dtype == np.int64 and segi['major_readcount'].dtype == np.int64 and segi['allele_ratio'].dtype == np.float64
    np.testing.assert_allclose(segi['total_depth'].values, ei.x[:, 2] / np.asarray(e.l, dtype=float))
