"""Accuracy statistics (remixt_amd/evaluate.py) on hand-computed cases -- CPU only."""
import numpy as np

from remixt_amd import evaluate


def test_hand_computed_case():
    normal = [1, 1]
    cn_true = np.array([[normal, [2, 1], [2, 1]],      # clonal
                        [normal, [3, 1], [2, 1]],      # subclonal (major differs)
                        [normal, [1, 0], [1, 0]],
                        [normal, [2, 2], [2, 2]]])
    cn_pred = np.array([[normal, [1, 2], [1, 2]],      # same up to major/minor order -> correct
                        [normal, [3, 1], [3, 1]],      # dominant right, second clone wrong, clonal status wrong
                        [normal, [1, 1], [1, 0]],      # dominant wrong
                        [normal, [2, 2], [2, 2]]])
    l = np.array([1., 2., 3., 4.])
    ev = evaluate.evaluate_cn(cn_true, cn_pred, l)
    assert np.isclose(ev['proportion_cn_correct'], (1 + 4) / 10.)
    assert np.isclose(ev['proportion_dom_cn_correct'], (1 + 2 + 4) / 10.)
    assert np.isclose(ev['proportion_clonal_correct'], (1 + 4) / 10.)       # segments 0, 3 agree on clonality; 1 and 2 do not
    assert np.isclose(ev['true_ploidy'], (3 * 1 + 3.5 * 2 + 1 * 3 + 4 * 4) / 10.)       # mean over clones, summed over alleles
    assert np.isclose(ev['true_ploidy_1'], (3 * 1 + 4 * 2 + 1 * 3 + 4 * 4) / 10.)
    assert np.isclose(ev['true_proportion_divergent'], (1 * 2.) / 20.)      # one allele of segment 1 differs between clones
    assert np.isclose(ev['pred_proportion_divergent'], (1 * 3.) / 20.)


def test_clone_order_and_swap():
    normal = [1, 1]
    cn_true = np.array([[normal, [2, 1], [3, 1]], [normal, [1, 1], [2, 1]]])
    cn_pred = cn_true[:, [0, 2, 1], :]                  # tumour clones listed the other way round
    l = np.array([1., 1.])
    assert evaluate.evaluate_cn(cn_true, cn_pred, l)['proportion_cn_correct'] == 0.
    assert evaluate.evaluate_cn(cn_true, cn_pred, l, allow_swap=True)['proportion_cn_correct'] == 1.
    # ordering by prevalence undoes the permutation
    assert evaluate.evaluate_cn(cn_true, cn_pred, l, h_true=[0.1, 0.3, 0.2], h_pred=[0.1, 0.2, 0.3])['proportion_cn_correct'] == 1.
    assert list(evaluate.clone_order([0.1, 0.2, 0.5, 0.3])) == [1, 2, 0]
    # different number of clones
    assert evaluate.evaluate_cn(cn_true, cn_pred[:, :2, :], l)['proportion_cn_correct'] == -1.


# ---------------------------------------------------------------------------------
# table-level entry points against vectors recorded from the reference's evaluate_results
# (oracle/make_golden.py `evaluation_case`)
# ---------------------------------------------------------------------------------
import os

import pandas as pd
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'evaluation.npz')


@pytest.fixture(scope='module')
def golden():
    return np.load(GOLDEN)


def _truth_collection(g):
    from remixt_amd import synthetic
    order = [frozenset((tuple(r[0]), tuple(r[1]))) for r in g['true_breakpoints'].tolist()]
    adjacencies = set((int(a), int(b)) for a, b in g['adjacencies'])
    return synthetic.GenomeCollection(
        g['l'], g['cn'], adjacencies, set(order), g['chromosome'], g['segment_start'], g['segment_end'],
        breakpoint_copy_number=dict(zip(order, g['brk_cn'])), minimal_breakpoint_copy_number=dict(zip(order, g['min_brk_cn'])),
        balanced_breakpoints=set(order[i] for i in g['balanced']))


def _tables(g, name):
    cn = pd.DataFrame(g[name + '_cn_values'], columns=list(g[name + '_cn_columns']))
    cn.insert(0, 'chromosome', list(g[name + '_cn_chromosome']))
    brk = pd.DataFrame(g[name + '_brk_values'], columns=list(g[name + '_brk_columns']))
    return cn, brk


@pytest.mark.parametrize('index', range(4))
def test_evaluate_results_reproduces_the_reference(golden, index):
    from remixt_amd import simulations
    name = str(golden['case_names'][index])
    gc = _truth_collection(golden)
    np.random.seed(2000 + index)        # the mixture of the recorded case (sampler pinned in tests/test_simulations.py)
    gm = simulations.GenomeMixtureSampler({'frac_normal': 0.4, 'frac_clone_1': float(golden[name + '_frac_clone_1']),
                                           'num_false_breakpoints': 6}).sample_genome_mixture(gc)
    cn, brk = _tables(golden, name)
    mix_pred = np.array(golden[name + '_mix_pred'])
    res = evaluate.evaluate_results(gm, cn, brk, mix_pred.copy())
    for key in ('cn_evaluation', 'brk_cn_evaluation', 'mix_results'):
        assert list(res[key].index) == list(golden[name + '_' + key + '_keys']), key
        np.testing.assert_allclose(res[key].values.astype(float), golden[name + '_' + key + '_values'], rtol=1e-13, atol=0, err_msg=key)
    cols = ['prediction_id', 'cn_correct', 'true_present', 'pred_present', 'true_subclonal', 'pred_subclonal']
    assert np.array_equal(res['brk_cn_table'][cols].values.astype(np.int64), golden[name + '_brk_table'])


def test_reindex_segments_reproduces_the_reference(golden):
    name = str(golden['case_names'][-1])
    truth = pd.DataFrame({'chromosome': list(golden['chromosome']), 'start': golden['segment_start'], 'end': golden['segment_end']})
    cn, _ = _tables(golden, name)
    out = evaluate.reindex_segments(truth, cn)
    assert np.array_equal(out[['start', 'end', 'idx_1', 'idx_2']].values.astype(np.int64), golden['reindex_last'])
    assert list(out['chromosome']) == list(golden['reindex_last_chromosome'])
    empty = evaluate.reindex_segments(truth.iloc[0:0], cn)
    assert len(empty.index) == 0 and list(empty.columns) == ['chromosome', 'start', 'end', 'idx_1', 'idx_2']


def test_evaluate_results_of_an_empty_prediction(golden):
    gc = _truth_collection(golden)
    from remixt_amd import simulations
    np.random.seed(1)
    gm = simulations.GenomeMixtureSampler({'frac_clone_1': 0.4, 'num_false_breakpoints': 2}).sample_genome_mixture(gc)
    res = evaluate.evaluate_results(gm, pd.DataFrame(columns=['chromosome', 'start', 'end']), pd.DataFrame(columns=['prediction_id', 'cn_1']), np.array([0.4, 0.6]))
    assert sorted(res) == ['brk_cn_evaluation', 'brk_cn_table', 'cn_evaluation', 'mix_results'] and all(len(v.index) == 0 for v in res.values())


def test_the_truth_scores_perfectly_through_the_result_tables():
    """Sampler -> experiment -> result tables of analysis.experiment -> evaluate_results: the true copy
    number and the breakend-step breakpoint copies are 100 % correct against their own mixture."""
    from remixt_amd import simulations, synthetic
    from remixt_amd.analysis import experiment as ex
    from remixt_amd.cn_model import decode_breakpoints_naive
    gc = synthetic.collection(600, num_clones=3, max_copy_number=6, num_chains=5, seed=31)
    np.random.seed(77)
    gm = simulations.GenomeMixtureSampler({'frac_normal': 0.4, 'frac_clone_1': 0.4, 'num_false_breakpoints': 10}).sample_genome_mixture(gc)
    e = simulations.ExperimentSampler({}).sample_experiment(gm)
    cn_table = ex.create_cn_table(e, e.cn, e.h)
    brk_table = ex.create_brk_cn_table(decode_breakpoints_naive(e.cn, e.adjacencies, e.breakpoints), e.breakpoint_segment_data)
    res = evaluate.evaluate_results(gm, cn_table, brk_table, e.h / e.h.sum())
    ev = res['cn_evaluation']
    assert ev['proportion_cn_correct'] == 1. and ev['proportion_dom_cn_correct'] == 1. and ev['pred_ploidy'] == ev['true_ploidy']
    assert res['brk_cn_evaluation']['brk_cn_correct_proportion'] == 1.
    assert res['brk_cn_evaluation']['brk_cn_present_num_true_pos'] == res['brk_cn_evaluation']['brk_cn_present_num_true']
