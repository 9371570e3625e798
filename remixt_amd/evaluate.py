"""Accuracy of a fit against a known mixture: the copy-number statistics of the
reference's `evaluate_cn_results` (remixt/simulations/pipeline.py:343-453) for predictions on the same
segmentation as the truth.

PARITY UNPINNED: the reference module does not import in the build container (it needs `blossomv` and
`scipy.misc.logsumexp`), so these statistics are checked by hand-computed cases only
(tests/test_evaluate.py), not against vectors from the reference."""
import numpy as np


def clone_order(h):
    """Tumour clone indices (0-based among tumour clones) by decreasing prevalence, as the reference
    orders true and predicted clones before comparing them (simulations/pipeline.py:599-611)."""
    h = np.asarray(h, dtype=float)
    return np.argsort(-h[1:], kind='stable')


def evaluate_cn(cn_true, cn_pred, lengths, h_true=None, h_pred=None, allow_swap=False):
    """Length-weighted agreement between true and predicted clone copy number.

    cn_true, cn_pred: (N, M, 2) including the normal clone at index 0; lengths: (N,).
    Returns a dict with the reference's keys: proportion_cn_correct, proportion_dom_cn_correct,
    proportion_clonal_correct, proportion_subclonal_correct, pred/true_ploidy(_1, _2),
    pred/true_proportion_divergent."""
    cn_true = np.asarray(cn_true)[:, 1:, :]
    cn_pred = np.asarray(cn_pred)[:, 1:, :]
    w = np.asarray(lengths, dtype=float)
    if h_true is not None:
        cn_true = cn_true[:, clone_order(h_true), :]
    if h_pred is not None:
        cn_pred = cn_pred[:, clone_order(h_pred), :]
    cn_true = np.sort(cn_true, axis=2)       # major / minor order is not identifiable
    cn_pred = np.sort(cn_pred, axis=2)
    tot = w.sum()
    out = {}
    if cn_true.shape[1] != cn_pred.shape[1]:
        out['proportion_cn_correct'] = -1.
    else:
        ok = (cn_true == cn_pred).all(axis=(1, 2))
        if allow_swap:
            ok = ok | (cn_true == cn_pred[:, ::-1, :]).all(axis=(1, 2))
        out['proportion_cn_correct'] = float((ok * w).sum()) / float(tot)
    dom = np.all(cn_true[:, 0, :] == cn_pred[:, 0, :], axis=1)
    out['proportion_dom_cn_correct'] = float((dom * w).sum()) / float(tot)
    clonal_true = np.all(cn_true[:, 0:1, :] == cn_true, axis=(1, 2))
    clonal_pred = np.all(cn_pred[:, 0:1, :] == cn_pred, axis=(1, 2))
    out['proportion_clonal_correct'] = float(((clonal_true == clonal_pred) * w).sum()) / float(tot)
    out['proportion_subclonal_correct'] = out['proportion_clonal_correct']       # (~a == ~b) == (a == b), as in the reference
    for name, cn in (('pred', cn_pred), ('true', cn_true)):
        out[name + '_ploidy'] = float((cn.mean(axis=1) * w[:, None]).sum() / tot)
        for m in range(min(2, cn.shape[1])):
            out['%s_ploidy_%d' % (name, m + 1)] = float((cn[:, m, :] * w[:, None]).sum() / tot)
        div = (cn.max(axis=1) != cn.min(axis=1)) * 1.
        out[name + '_proportion_divergent'] = float((div * w[:, None]).sum() / (2. * tot))
    return out
