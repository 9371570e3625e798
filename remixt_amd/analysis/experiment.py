"""Result / segment tables either side of the hot path (reference remixt/analysis/experiment.py:323-422).

`experiment` is any object with the reference Experiment's read-only properties used here:
segment_chromosome_id, segment_start, segment_end, segment_major_is_allele_a, x, l and (for
breakpoint tables) breakpoint_segment_data."""
import numpy as np
import pandas as pd


def _segment_columns(experiment):
    """The per-segment columns of the result tables as an ordered list of (name, array): identification, length, the three read counts,
    the minor-allele ratio (0 where no allele reads were counted) and the depths that follow from it.  The names and formulas are the
    reference's table schema (remixt/analysis/experiment.py:323-351); the arrays are computed once, in numpy."""
    counts = np.asarray(experiment.x)                 # the three count columns keep experiment.x's own dtype, as in the reference (:333-342)
    x = counts.astype(float)
    length = np.asarray(experiment.l, dtype=float)
    major, minor, total = x[:, 0], x[:, 1], x[:, 2]
    allele_reads = major + minor
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = np.where(allele_reads != 0, minor / allele_reads, np.nan)
    ratio = np.where(np.isnan(ratio), 0., ratio)
    with np.errstate(divide='ignore', invalid='ignore'):
        depth = total / length
        depths = (total * (1. - ratio) / length, total * ratio / length, depth)
    return [
        ('chromosome', experiment.segment_chromosome_id), ('start', experiment.segment_start), ('end', experiment.segment_end),
        ('major_is_allele_a', experiment.segment_major_is_allele_a), ('length', experiment.l),
        ('major_readcount', counts[:, 0]), ('minor_readcount', counts[:, 1]), ('readcount', counts[:, 2]), ('allele_ratio', ratio),
        ('major_depth', depths[0]), ('minor_depth', depths[1]), ('total_depth', depths[2]),
    ]


def _frame(columns):
    return pd.DataFrame({name: np.asarray(values) for name, values in columns}, columns=[name for name, _ in columns])


def create_segment_table(experiment):
    """Per-segment read counts, allele ratio and depths (schema of experiment.py:323-351)."""
    return _frame(_segment_columns(experiment))


def create_cn_table(experiment, cn, h, phi=None):
    """Segment table + clone copy number, tumour-only ("raw") depths and the depths / read counts the fitted mixture expects
    (schema of experiment.py:354-394).  cn: (N, M, 2) [segment, clone, allele]; h: (M,) haploid depths, clone 0 = normal."""
    cn = np.asarray(cn); h = np.asarray(h, dtype=float)
    cols = _segment_columns(experiment)
    seg = dict(cols)
    length = np.asarray(experiment.l, dtype=float)
    tumour_depth = h[1:].sum()
    allele_names = ('major', 'minor')
    for m in range(cn.shape[1]):
        cols += [('%s_%d' % (allele_names[a], m), cn[:, m, a]) for a in range(2)]
    # observed depth less the normal clone's share, per unit of tumour depth
    cols += [('%s_raw' % allele_names[a], (seg['%s_depth' % allele_names[a]] - cn[:, 0, a] * h[0]) / tumour_depth) for a in range(2)]
    # what the mixture h expects: depths, then read counts, then the tumour-only form of the expected depths
    expected = [(cn[:, :, a] * h[np.newaxis, :]).sum(axis=-1) for a in range(2)] + [(cn.sum(axis=-1) * h[np.newaxis, :]).sum(axis=-1)]
    kinds = allele_names + ('total',)
    cols += [('%s_depth_e' % kinds[i], expected[i]) for i in range(3)]
    cols += [('%s_e' % kinds[i], expected[i] * length) for i in range(3)]
    cols += [('%s_raw_e' % allele_names[a], (expected[a] - cn[:, 0, a] * h[0]) / tumour_depth) for a in range(2)]
    if cn.shape[1] > 2:      # two tumour clones: where they differ
        cols += [('%s_diff' % allele_names[a], np.absolute(cn[:, 1, a] - cn[:, 2, a])) for a in range(2)]
    return _frame(cols)


def create_brk_cn_table(brk_cn, breakpoint_segment_data):
    """Breakpoint copy number (dict prediction_id -> (M,) per-clone copies) as columns cn_0 .. cn_{M-1} beside the breakpoint / segment
    mapping, one row per breakpoint that has both (schema of experiment.py:397-422; `breakpoint_segment_data` needs 'prediction_id')."""
    if len(brk_cn) == 0:
        return pd.DataFrame(columns=['prediction_id'])
    ids = list(brk_cn.keys())
    copies = np.asarray([np.asarray(brk_cn[i]) for i in ids])
    table = pd.DataFrame({'prediction_id': ids})
    for m in range(copies.shape[1]):
        table['cn_%d' % m] = copies[:, m]
    return table.merge(breakpoint_segment_data, on='prediction_id').fillna(0.)


# ---------------------------------------------------------------------------------
# Experiment: count / breakpoint tables -> the arrays the hot path consumes
# (reference remixt/analysis/experiment.py:8-320)
# ---------------------------------------------------------------------------------
BREAKPOINT_COLUMNS = ['prediction_id', 'chromosome_1', 'strand_1', 'position_1', 'chromosome_2', 'strand_2', 'position_2']


def find_closest(a, v):
    """Index into the sorted array `a` of the element closest to each target in `v`, and the distance
    (experiment.py:8-35).  A tie goes to the right neighbour (strict `<` on the left distance)."""
    a = np.asarray(a); v = np.asarray(v)
    right = np.minimum(np.searchsorted(a, v), len(a) - 1)
    left = np.maximum(right - 1, 0)
    d_left, d_right = v - a[left], a[right] - v
    return np.where(d_left < d_right, left, right), np.minimum(d_left, d_right)


def find_closest_segment_end(segment_data, breakpoint_data):
    """Closest segment extremity of matching chromosome and strand for every break end
    (experiment.py:38-121): a '+' break end pairs with a segment END (side 1), a '-' break end with a
    segment START (side 0).  Columns: prediction_id, prediction_side (0 / 1 for the _1 / _2 columns),
    dist, segment_idx, segment_side.  Break ends on chromosomes without segments are absent."""
    chrom = np.asarray(segment_data['chromosome'].values)
    pos = {0: np.asarray(segment_data['start'].values), 1: np.asarray(segment_data['end'].values)}
    seg_index = np.asarray(segment_data.index.values)
    rows = []
    chromosomes = list(pd.unique(chrom))
    for side_p, suffix in ((0, '1'), (1, '2')):
        b_chrom = np.asarray(breakpoint_data['chromosome_' + suffix].values)
        b_strand = np.asarray(breakpoint_data['strand_' + suffix].values)
        b_pos = np.asarray(breakpoint_data['position_' + suffix].values)
        b_id = np.asarray(breakpoint_data['prediction_id'].values)
        for c in chromosomes:
            in_c = np.nonzero(chrom == c)[0]
            for strand, seg_side in (('+', 1), ('-', 0)):
                sel = np.nonzero((b_chrom == c) & (b_strand == strand))[0]
                if len(sel) == 0:
                    continue
                order = np.argsort(pos[seg_side][in_c], kind='stable')
                idx, dist = find_closest(pos[seg_side][in_c][order], b_pos[sel])
                for k, i_, d_ in zip(sel, idx, dist):
                    rows.append((b_id[k], side_p, d_, seg_index[in_c[order[i_]]], seg_side))
    return pd.DataFrame(rows, columns=['prediction_id', 'prediction_side', 'dist', 'segment_idx', 'segment_side'])


def get_wild_type_adjacencies(segment_data, max_seg_gap):
    """Pairs (i, i+1) of consecutive segments on one chromosome whose gap is at most max_seg_gap
    (experiment.py:124-143)."""
    chrom = np.asarray(segment_data['chromosome'].values)
    start = np.asarray(segment_data['start'].values); end = np.asarray(segment_data['end'].values)
    return set((int(i), int(i) + 1) for i in range(len(chrom) - 1)
               if chrom[i] == chrom[i + 1] and start[i + 1] - end[i] <= max_seg_gap)


def create_breakpoint_segment_table(segment_data, breakpoint_data, adjacencies, max_brk_dist=2000):
    """Breakpoints as pairs of segment extremities (experiment.py:146-216): both break ends must have a
    closest extremity, their distances must sum to at most max_brk_dist; events that look like a
    wild-type adjacency and loop-back inversions are dropped.  Rows in prediction_id order; n_* / side_*
    are integers."""
    closest = find_closest_segment_end(segment_data, breakpoint_data)
    out = []
    by_id = {}
    for pid, side, dist, n, s in closest.itertuples(index=False):
        by_id.setdefault(pid, {})[int(side)] = (dist, int(n), int(s))
    for pid in sorted(by_id):
        ends = by_id[pid]
        if 0 not in ends or 1 not in ends:
            continue
        (d1, n1, s1), (d2, n2, s2) = ends[0], ends[1]
        if d1 + d2 > max_brk_dist:
            continue
        if (n1, n2) in adjacencies and s1 == 1 and s2 == 0:
            continue
        if (n2, n1) in adjacencies and s2 == 1 and s1 == 0:
            continue
        if (n1, s1) == (n2, s2):
            continue
        out.append((pid, n1, s1, n2, s2))
    return pd.DataFrame(out, columns=['prediction_id', 'n_1', 'side_1', 'n_2', 'side_2'])


def convert_breakpoints_to_dict(breakpoint_segment_data):
    """prediction_id -> frozenset of the two (segment, side) break ends (experiment.py:219-225)."""
    breakpoints = dict()
    for pid, n1, s1, n2, s2 in breakpoint_segment_data[['prediction_id', 'n_1', 'side_1', 'n_2', 'side_2']].itertuples(index=False):
        breakpoints[pid] = frozenset([(int(n1), int(s1)), (int(n2), int(s2))])
    return breakpoints


class Experiment(object):
    """Segment counts + breakpoints as the model consumes them (experiment.py:244-320).

    count_data columns: chromosome, start, end, length, major_readcount, minor_readcount, readcount,
    major_is_allele_a; breakpoint_data columns: BREAKPOINT_COLUMNS."""

    def __init__(self, count_data, breakpoint_data=None, max_brk_dist=2000, max_seg_gap=int(3e6)):
        if breakpoint_data is not None:
            breakpoint_data = breakpoint_data[BREAKPOINT_COLUMNS]
        else:
            breakpoint_data = pd.DataFrame(columns=BREAKPOINT_COLUMNS)
        chromosomes = count_data['chromosome'].unique()
        # breakpoints touching a chromosome without count data cannot be modelled
        self.breakpoint_data = breakpoint_data[breakpoint_data['chromosome_1'].isin(chromosomes) &
                                               breakpoint_data['chromosome_2'].isin(chromosomes)]
        self.count_data = count_data.reset_index(drop=True).reset_index()      # 0..n-1, also as column 'index'
        self.adjacencies = get_wild_type_adjacencies(self.count_data, max_seg_gap)
        self.breakpoint_segment_data = create_breakpoint_segment_table(self.count_data, self.breakpoint_data, self.adjacencies, max_brk_dist=max_brk_dist)
        self.breakpoint_segment_data = self.breakpoint_segment_data.merge(self.breakpoint_data, on='prediction_id')

    @property
    def segment_chromosome_id(self):
        return self.count_data['chromosome'].values

    @property
    def segment_start(self):
        return self.count_data['start'].values

    @property
    def segment_end(self):
        return self.count_data['end'].values

    @property
    def segment_major_is_allele_a(self):
        return self.count_data['major_is_allele_a'].values

    @property
    def x(self):
        return self.count_data[['major_readcount', 'minor_readcount', 'readcount']].values

    @property
    def l(self):
        return self.count_data['length'].values

    @property
    def breakpoints(self):
        return convert_breakpoints_to_dict(self.breakpoint_segment_data)

    @property
    def chains(self):
        """Half-open [start, end) runs of segments joined by wild-type adjacencies."""
        n = len(self.count_data.index)
        cuts = [i + 1 for i in range(n - 1) if (i, i + 1) not in self.adjacencies]
        return list(zip([0] + cuts, cuts + [n]))


def create_experiment(count_filename, breakpoint_filename, experiment_filename, max_brk_dist=2000, min_length=None):
    """counts.tsv + breakpoints.tsv -> pickled Experiment (experiment.py:228-241)."""
    import pickle
    count_data = pd.read_csv(count_filename, sep='\t', converters={'chromosome': str})
    if min_length is not None:
        count_data = count_data[count_data['length'] > min_length]
    breakpoint_data = pd.read_csv(breakpoint_filename, sep='\t', converters={'chromosome_1': str, 'chromosome_2': str})
    experiment = Experiment(count_data, breakpoint_data, max_brk_dist=max_brk_dist)
    with open(experiment_filename, 'wb') as f:
        pickle.dump(experiment, f)
    return experiment
