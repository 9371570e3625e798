"""Read-depth initialisation of the restart grid: per-segment allele depths -> modes of the minor depth ->
candidate (normal, tumour) haploid depths -> ploidy of a candidate.

Behaviour follows the reference's `remixt/analysis/readdepth.py` (calculate_depth :12-57,
calculate_minor_modes :60-90, calculate_candidate_h_monoclonal :93-126, estimate_ploidy :129-147) and is
pinned by tests/golden/pipeline_init*.npz, recorded from the reference's own functions.  The
implementation is array-first: the depths are computed once into a `_Depths` record straight from the
experiment's count matrix, and the pandas table the reference's callers expect is only a view of it.
"""
import collections

import numpy as np
import pandas as pd

from .. import likelihood

_Depths = collections.namedtuple('_Depths', 'length major minor total high_quality measurable')

_KMEANS_CLUSTERS = 5          # readdepth.py:75
_MIN_CLUSTER_SHARE = 0.01     # readdepth.py:83
_TRIM_PERCENTILE = 95         # readdepth.py:70
_RESAMPLE_DRAWS = 10000       # remixt/utils.py:24


def _above_lowest_decile(values):
    return values > np.percentile(values, 10)


def _depths(experiment):
    """Allele-specific read depth per segment and the two per-segment flags of the reference's table."""
    counts = np.asarray(experiment.x, dtype=float)
    major_reads, minor_reads, reads = counts[:, 0], counts[:, 1], counts[:, 2]
    length = np.asarray(experiment.l, dtype=float)
    phased = minor_reads + major_reads
    with np.errstate(divide='ignore', invalid='ignore'):
        minor_share = minor_reads / (major_reads + minor_reads)
        minor_share = np.where(np.isnan(minor_share), 0., minor_share)          # no phased read at all: all depth to the major allele
        major = reads * (1. - minor_share) / length
        minor = reads * minor_share / length
        total = reads / length
        # share of the annotated segment that is mappable ("length" is the mappable length)
        span = np.asarray(experiment.segment_end) - np.asarray(experiment.segment_start) + 1
        covered = length / span
    high_quality = _above_lowest_decile(length) & _above_lowest_decile(phased) & _above_lowest_decile(covered)
    # a depth is only defined where the segment has length and both the phased and the total read
    # fraction can be measured (likelihood.py:71-90)
    p = likelihood.proportion_measureable_matrix(likelihood.estimate_phi(experiment.x))
    measurable = (length > 0) & np.all(p > 0, axis=1)
    return _Depths(length, major, minor, total, high_quality, measurable)


def calculate_depth(experiment):
    """Table of the measurable segments (original index kept) with columns chromosome, start, end, length,
    major, minor, total, high_quality."""
    d = _depths(experiment)
    keep = np.nonzero(d.measurable)[0]
    table = collections.OrderedDict()
    table['chromosome'] = np.asarray(experiment.segment_chromosome_id)[keep]
    table['start'] = np.asarray(experiment.segment_start)[keep]
    table['end'] = np.asarray(experiment.segment_end)[keep]
    for name in ('length', 'major', 'minor', 'total', 'high_quality'):
        table[name] = getattr(d, name)[keep]
    return pd.DataFrame(table, index=keep)


def weighted_resample(data, weights, num_samples=_RESAMPLE_DRAWS):
    """Multinomial bootstrap of `data` with probabilities proportional to `weights` (remixt/utils.py:24-29).
    The reference wraps the draw in a context manager that saves and restores the global generator's state
    without reseeding: the draw uses the state as it stands and leaves it untouched."""
    share = weights.astype(float) / float(weights.sum())
    saved = np.random.get_state()
    try:
        times = np.random.multinomial(num_samples, share)
    finally:
        np.random.set_state(saved)
    return np.repeat(data, times)


def calculate_minor_modes(read_depth):
    """Modes of the minor-allele depth: centres of a 5-means clustering (scikit-learn defaults, global numpy
    generator) of a length-weighted bootstrap of the depths under their 95th percentile, without the clusters
    holding less than 1 % of the bootstrap."""
    from sklearn.cluster import KMeans
    minor = np.asarray(read_depth['minor'])
    length = np.asarray(read_depth['length'])
    body = minor < np.percentile(minor, _TRIM_PERCENTILE)
    draws = weighted_resample(minor[body], length[body]).reshape(-1, 1)
    model = KMeans(n_clusters=_KMEANS_CLUSTERS)
    model.fit(draws)
    membership = np.bincount(model.predict(draws))
    share = membership.astype(float) / membership.sum()
    return model.cluster_centers_[:, 0][share >= _MIN_CLUSTER_SHARE]


def calculate_candidate_h_monoclonal(minor_modes, h_normal=None, h_tumour=None):
    """(normal, tumour) haploid depth candidates.  The lowest mode is the normal depth unless given; each higher
    mode is read once as one tumour minor copy and once as two.  With both depths given there is one candidate."""
    if h_normal is None:
        h_normal = minor_modes.min()
    if h_tumour is not None:
        return np.array([[h_normal, h_tumour]])
    tumour_steps = [mode - h_normal for mode in minor_modes if mode > h_normal]
    return [np.array([h_normal, step * copies_share]) for step in tumour_steps for copies_share in (1., 0.5)]


def estimate_ploidy(h, experiment):
    """Length-weighted mean tumour copy number implied by haploid depths `h` = (normal, tumour clones...): each
    allele's depth less one normal copy, in units of the summed tumour depth.  Segments whose depths are
    undefined or +inf do not count."""
    d = _depths(experiment)
    tumour_depth = h[1:].sum()
    rows = d.measurable
    major_copies = (d.major[rows] - h[0]) / tumour_depth
    minor_copies = (d.minor[rows] - h[0]) / tumour_depth
    length = d.length[rows]
    columns = (length, d.major[rows], d.minor[rows], d.total[rows], major_copies, minor_copies)
    usable = np.ones(length.shape, dtype=bool)
    for c in columns:
        usable &= ~(np.isnan(c) | (c == np.inf))
    return ((major_copies[usable] + minor_copies[usable]) * length[usable]).sum() / length[usable].sum()
