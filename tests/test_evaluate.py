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
