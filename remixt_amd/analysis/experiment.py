"""Result / segment tables either side of the hot path (reference remixt/analysis/experiment.py:323-422).

`experiment` is any object with the reference Experiment's read-only properties used here:
segment_chromosome_id, segment_start, segment_end, segment_major_is_allele_a, x, l and (for
breakpoint tables) breakpoint_segment_data."""
import numpy as np
import pandas as pd


def create_segment_table(experiment):
    """experiment.py:323-351: per-segment read counts, allele ratio and depths."""
    x = np.asarray(experiment.x)
    data = pd.DataFrame({
        'chromosome': experiment.segment_chromosome_id,
        'start': experiment.segment_start,
        'end': experiment.segment_end,
        'major_is_allele_a': experiment.segment_major_is_allele_a,
        'length': experiment.l,
        'major_readcount': x[:, 0],
        'minor_readcount': x[:, 1],
        'readcount': x[:, 2],
    })
    data['allele_ratio'] = data['minor_readcount'] / (data['major_readcount'] + data['minor_readcount'])
    data['allele_ratio'] = data['allele_ratio'].fillna(0)
    data['major_depth'] = data['readcount'] * (1. - data['allele_ratio']) / data['length']
    data['minor_depth'] = data['readcount'] * data['allele_ratio'] / data['length']
    data['total_depth'] = data['readcount'] / data['length']
    return data


def create_cn_table(experiment, cn, h, phi=None):
    """experiment.py:354-394: segment table + clone copy number, raw and expected depths."""
    cn = np.asarray(cn); h = np.asarray(h)
    data = create_segment_table(experiment)
    for m in range(0, cn.shape[1]):
        data['major_{0}'.format(m)] = cn[:, m, 0]
        data['minor_{0}'.format(m)] = cn[:, m, 1]
    data['major_raw'] = (data['major_depth'] - data['major_0'] * h[0]) / h[1:].sum()
    data['minor_raw'] = (data['minor_depth'] - data['minor_0'] * h[0]) / h[1:].sum()
    data['major_depth_e'] = (cn[:, :, 0] * h[np.newaxis, :]).sum(axis=-1)
    data['minor_depth_e'] = (cn[:, :, 1] * h[np.newaxis, :]).sum(axis=-1)
    data['total_depth_e'] = (cn.sum(axis=-1) * h[np.newaxis, :]).sum(axis=-1)
    data['major_e'] = data['major_depth_e'] * experiment.l
    data['minor_e'] = data['minor_depth_e'] * experiment.l
    data['total_e'] = data['total_depth_e'] * experiment.l
    data['major_raw_e'] = (data['major_depth_e'] - data['major_0'] * h[0]) / h[1:].sum()
    data['minor_raw_e'] = (data['minor_depth_e'] - data['minor_0'] * h[0]) / h[1:].sum()
    if 'major_2' in data:
        data['major_diff'] = np.absolute(data['major_1'] - data['major_2'])
        data['minor_diff'] = np.absolute(data['minor_1'] - data['minor_2'])
    return data


def create_brk_cn_table(brk_cn, breakpoint_segment_data):
    """experiment.py:397-422: breakpoint copy number joined to the breakpoint / segment mapping
    (`breakpoint_segment_data` needs a 'prediction_id' column)."""
    if len(brk_cn) == 0:
        return pd.DataFrame(columns=['prediction_id'])
    brk_cn_table = pd.DataFrame(list(brk_cn.values()), index=list(brk_cn.keys()))
    brk_cn_table.columns = ['cn_{}'.format(m) for m in brk_cn_table.columns]
    brk_cn_table.index.name = 'prediction_id'
    brk_cn_table = brk_cn_table.reset_index()
    brk_cn_table = brk_cn_table.merge(breakpoint_segment_data, on='prediction_id').fillna(0.)
    return brk_cn_table
