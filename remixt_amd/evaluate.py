"""Accuracy of a fit against a known mixture: the copy-number statistics of the
reference's `evaluate_cn_results` (remixt/simulations/pipeline.py:343-453) for predictions on the same
segmentation as the truth.

`evaluate_cn` works on arrays on a common segmentation (hand-computed cases in tests/test_evaluate.py);
the table-level functions below it (`evaluate_results` and what it calls) take the reference's arguments
and are pinned by vectors recorded from the reference's own functions (tests/golden/evaluation.npz)."""
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


# ---------------------------------------------------------------------------------
# The reference's table-level entry points (remixt/simulations/pipeline.py:343-647,
# remixt/segalg.py:260-336), restated.  PINNED: tests/test_evaluate.py replays
# tests/golden/evaluation.npz, recorded from the reference's own functions by
# oracle/make_golden.py `evaluation_case`.
# ---------------------------------------------------------------------------------
def reindex_segments(cn_1, cn_2):
    """Common refinement of two segmentations (segalg.py:260-336): rows (chromosome, start, end, idx_1,
    idx_2) for every interval between consecutive boundaries of either table that lies inside exactly one
    segment of each; idx_* are index labels of the covering rows."""
    import pandas as pd
    if len(cn_1.index) == 0 or len(cn_2.index) == 0:
        empty = pd.DataFrame(columns=['chromosome', 'start', 'end', 'idx_1', 'idx_2'], dtype=int)
        empty['chromosome'] = empty['chromosome'].astype(str)
        return empty
    parts = []
    for chromosome, a in cn_1.groupby('chromosome'):
        b = cn_2[cn_2['chromosome'] == chromosome]
        if len(b.index) == 0:
            continue
        cuts = np.sort(np.unique(np.concatenate([a['start'].values, a['end'].values, b['start'].values, b['end'].values])))
        start, end = cuts[:-1], cuts[1:]
        keep = np.ones(len(start), dtype=bool)
        cover = []
        for tab in (a, b):
            first = np.searchsorted(tab['start'].values, start, side='right') - 1      # last segment starting at or before
            last = np.searchsorted(tab['end'].values, end, side='left')                # first segment ending at or after
            keep &= ~((first != last) | (first < 0) | (first >= len(end)))
            cover.append(first)
        part = pd.DataFrame({'start': start[keep], 'end': end[keep]})
        part['idx_1'] = a.index.values[cover[0][keep]]
        part['idx_2'] = b.index.values[cover[1][keep]]
        part['chromosome'] = chromosome
        parts.append(part)
    return pd.concat(parts, ignore_index=True)


def _weighted(flags, lengths):
    return float((flags * lengths).sum()) / float(lengths.sum())


def evaluate_cn_results(genome_mixture, cn_data_table, order_true, order_pred, allow_swap):
    """Copy-number accuracy of a predicted table against a known mixture (simulations/pipeline.py:343-463).

    `cn_data_table`: columns chromosome, start, end and major_m / minor_m (or total_m) for tumour clones
    m = 1, 2, on any segmentation; `order_*`: tumour clone orders (largest first).  Returns
    {'cn_evaluation': pandas.Series} with the reference's twelve statistics."""
    import pandas as pd
    truth = pd.DataFrame({'chromosome': genome_mixture.segment_chromosome_id, 'start': genome_mixture.segment_start,
                          'end': genome_mixture.segment_end})
    if 'major_1' in cn_data_table:
        cn_true = genome_mixture.cn[:, 1:, :]
        cn_pred = np.stack([np.stack([cn_data_table['major_%d' % m].values, cn_data_table['minor_%d' % m].values], axis=-1)
                            for m in (1, 2)], axis=1)
    else:
        cn_true = genome_mixture.cn[:, 1:, :].sum(axis=2)[:, :, None].astype(float)
        cn_pred = np.stack([cn_data_table['total_%d' % m].values[:, None] for m in (1, 2)], axis=1)
    cn_true = np.sort(cn_true[:, order_true, :], axis=2)      # clones largest first; major / minor not identifiable
    cn_pred = np.sort(cn_pred[:, order_pred, :], axis=2)

    common = reindex_segments(truth, cn_data_table)
    cn_true = cn_true[common['idx_1'].values, :, :]
    cn_pred = cn_pred[common['idx_2'].values, :, :]
    w = (common['end'] - common['start']).values

    out = dict()
    if cn_true.shape[1] != cn_pred.shape[1]:
        out['proportion_cn_correct'] = -1.
    else:
        same = (cn_true == cn_pred).all(axis=(1, 2))
        if allow_swap:
            same = same | (cn_true == cn_pred[:, ::-1, :]).all(axis=(1, 2))
        out['proportion_cn_correct'] = _weighted(same, w)
    out['proportion_dom_cn_correct'] = _weighted(np.all(cn_true[:, 0, :] == cn_pred[:, 0, :], axis=1), w)
    clonal_true = np.all(cn_true[:, 0:1, :] == cn_true, axis=(1, 2))
    clonal_pred = np.all(cn_pred[:, 0:1, :] == cn_pred, axis=(1, 2))
    out['proportion_clonal_correct'] = _weighted(clonal_true == clonal_pred, w)
    out['proportion_subclonal_correct'] = _weighted(~clonal_true == ~clonal_pred, w)
    wsum = float(w.sum())
    both = (('pred', cn_pred), ('true', cn_true))
    for name, cn in both:
        out[name + '_ploidy'] = (cn.mean(axis=1) * w[:, np.newaxis]).sum() / wsum
    for k in (0, 1):
        for name, cn in both:
            out['%s_ploidy_%d' % (name, k + 1)] = (cn[:, k, :] * w[:, np.newaxis]).sum() / wsum
    for name, cn in both:
        divergent = (cn.max(axis=1) != cn.min(axis=1)) * 1.
        out[name + '_proportion_divergent'] = (divergent * w[:, np.newaxis]).sum() / (2. * w.sum())
    return {'cn_evaluation': pd.Series(out)}


def evaluate_brk_cn_results(genome_mixture, brk_cn_table, order_true, order_pred, allow_swap):
    """Breakpoint copy-number accuracy (simulations/pipeline.py:466-572).  `brk_cn_table`: prediction_id and
    cn_m per tumour clone.  The truth comes from `genome_mixture.genome_collection`'s
    `collapsed_breakpoint_copy_number()`, `collapsed_minimal_breakpoint_copy_number()` (dicts breakpoint ->
    per-clone copies, normal first) and `collapsed_balanced_breakpoints()`; balanced breakpoints are left
    out, undetected truth counts as zero copies.  Returns {'brk_cn_table', 'brk_cn_evaluation'}."""
    import itertools
    import pandas as pd
    M = genome_mixture.M
    true_cols = ['true_cn_%d' % m for m in range(1, M)]
    min_true_cols = ['min_true_cn_%d' % m for m in range(1, M)]
    pred_cols = list(itertools.takewhile(lambda c: c in brk_cn_table, ('cn_%d' % m for m in itertools.count(1))))

    data = genome_mixture.breakpoint_segment_data.set_index('prediction_id')
    for col in true_cols + min_true_cols:
        data[col] = 0
    data['is_balanced'] = False
    gc = genome_mixture.genome_collection
    true_brk_cn = gc.collapsed_breakpoint_copy_number()
    min_true_brk_cn = gc.collapsed_minimal_breakpoint_copy_number()
    balanced = gc.collapsed_balanced_breakpoints()
    for prediction_id, breakpoint in genome_mixture.detected_breakpoints.items():
        if breakpoint not in true_brk_cn:
            continue
        data.loc[prediction_id, true_cols] = true_brk_cn[breakpoint][1:]
        data.loc[prediction_id, min_true_cols] = min_true_brk_cn[breakpoint][1:]
        if breakpoint in balanced:
            data.loc[prediction_id, 'is_balanced'] = True
    data.reset_index(inplace=True)
    data = data.merge(brk_cn_table[['prediction_id'] + pred_cols], on='prediction_id', how='left').fillna(0.0)
    data = data[~data['is_balanced']]

    cn_true = data[min_true_cols].values[:, order_true]
    cn_pred = data[pred_cols].values[:, order_pred]
    if cn_true.shape[1] != cn_pred.shape[1]:
        cn_correct = -1.
    else:
        cn_correct = (cn_true == cn_pred).all(axis=(1,))
        if allow_swap:
            cn_correct = cn_correct | (cn_true == cn_pred[:, ::-1]).all(axis=(1,))
    data['cn_correct'] = cn_correct
    data['true_present'] = (data[min_true_cols] > 0).any(axis=1)
    data['pred_present'] = (data[pred_cols] > 0).any(axis=1)
    data['true_subclonal'] = (data[min_true_cols] == 0).any(axis=1) & data['true_present']
    data['pred_subclonal'] = (data[pred_cols] == 0).any(axis=1) & data['pred_present']

    ev = dict()
    ev['brk_cn_correct_proportion'] = float(data['cn_correct'].sum()) / float(len(data.index))
    ev['brk_cn_present_num_true'] = float(data['true_present'].sum())
    ev['brk_cn_present_num_pos'] = float(data['pred_present'].sum())
    ev['brk_cn_present_num_true_pos'] = float((data['pred_present'] & data['true_present']).sum())
    ev['brk_cn_subclonal_num_true'] = float(data['true_subclonal'].sum())
    ev['brk_cn_subclonal_num_pos'] = float(data['pred_subclonal'].sum())
    ev['brk_cn_subclonal_num_true_pos'] = float((data['pred_subclonal'] & data['true_subclonal']).sum())
    return {'brk_cn_table': data, 'brk_cn_evaluation': pd.Series(ev)}


def evaluate_results(genome_mixture, cn_table, brk_cn_table, mix_pred):
    """All accuracy statistics of one prediction (simulations/pipeline.py:575-647): clone orders by
    decreasing prevalence, swapping of the tumour clones tolerated when the true prevalences are within
    75 % of each other; single-clone predictions are duplicated into the second clone.  Returns the
    reference's dict: cn_evaluation, brk_cn_table, brk_cn_evaluation, mix_results."""
    import pandas as pd
    if len(cn_table.index) == 0 or np.asarray(mix_pred).shape[0] == 0:
        return {'brk_cn_evaluation': pd.Series(dtype=float), 'brk_cn_table': pd.DataFrame(), 'cn_evaluation': pd.Series(dtype=float),
                'mix_results': pd.Series(dtype=float)}
    cn_table = cn_table.copy()
    brk_cn_table = brk_cn_table.copy()
    mix_true = genome_mixture.frac.copy()
    if 'major_1' in cn_table and 'major_2' not in cn_table:
        cn_table['major_2'] = cn_table['major_1']
        cn_table['minor_2'] = cn_table['minor_1']
    if 'total_1' in cn_table and 'total_2' not in cn_table:
        cn_table['total_2'] = cn_table['total_1']
    if 'cn_2' not in brk_cn_table:
        brk_cn_table['cn_2'] = brk_cn_table['cn_1']
    assert isinstance(mix_pred, np.ndarray) and isinstance(mix_true, np.ndarray)
    mix_pred = np.concatenate([mix_pred, [0.]]) if len(mix_pred) == 2 else mix_pred.copy()

    order_true = np.argsort(mix_true[1:])[::-1]
    mix_true[1:] = mix_true[1:][order_true]
    order_pred = np.argsort(mix_pred[1:])[::-1]
    mix_pred[1:] = mix_pred[1:][order_pred]
    allow_swap = mix_true[1:].min() / mix_true[1:].max() > 0.75

    results = evaluate_cn_results(genome_mixture, cn_table, order_true, order_pred, allow_swap)
    results.update(evaluate_brk_cn_results(genome_mixture, brk_cn_table, order_true, order_pred, allow_swap))
    mix = {}
    for i, f in enumerate(mix_true):
        mix['mix_true_' + str(i)] = f
    for i, f in enumerate(mix_pred):
        mix['mix_pred_' + str(i)] = f
    results['mix_results'] = pd.Series(mix)
    return results
