"""Candidate haploid depths from read depth (reference remixt/analysis/readdepth.py:12-147): the
initialisations whose grid with the tumour mix fractions and divergence weights is the restart set
the GPUs shard."""
import numpy as np

from .. import likelihood
from . import experiment as _experiment


def calculate_depth(experiment):
    """readdepth.py:12-57: table with columns chromosome, start, end, length, major, minor, total,
    high_quality; segments with zero length or no genotypable reads are dropped."""
    data = _experiment.create_segment_table(experiment)
    data['segment_length'] = data['end'] - data['start'] + 1
    data['length_ratio'] = data['length'] / data['segment_length']
    data['allele_readcount'] = data['minor_readcount'] + data['major_readcount']
    data['high_quality'] = (
        (data['length'] > np.percentile(data['length'].values, 10)) &
        (data['allele_readcount'] > np.percentile(data['allele_readcount'].values, 10)) &
        (data['length_ratio'] > np.percentile(data['length_ratio'].values, 10)))
    phi = likelihood.estimate_phi(experiment.x)
    p = likelihood.proportion_measureable_matrix(phi)
    data = data[(data['length'] > 0) & np.all(p > 0, axis=1)]
    data = data.rename(columns={'major_depth': 'major', 'minor_depth': 'minor', 'total_depth': 'total'})
    return data[['chromosome', 'start', 'end', 'length', 'major', 'minor', 'total', 'high_quality']]


def weighted_resample(data, weights, num_samples=10000):
    """remixt/utils.py:24-29: multinomial resample; the global numpy RNG state is restored afterwards
    (TempRandomSeed saves and restores it -- the draw itself uses the state as it is)."""
    norm_weights = weights.astype(float) / float(weights.sum())
    state = np.random.get_state()
    counts = np.random.multinomial(num_samples, norm_weights)
    np.random.set_state(state)
    return np.repeat(data, counts)


def calculate_minor_modes(read_depth):
    """readdepth.py:60-90: k-means (k = 5, sklearn defaults, global numpy RNG) on a length-weighted
    resample of the minor depths below their 95th percentile; clusters under 1 % are dropped."""
    import sklearn.cluster
    amp_rd = np.percentile(read_depth['minor'], 95)
    read_depth = read_depth[read_depth['minor'] < amp_rd]
    rd_samples = weighted_resample(read_depth['minor'].values, read_depth['length'].values)
    kmm = sklearn.cluster.KMeans(n_clusters=5)
    kmm.fit(rd_samples.reshape((rd_samples.size, 1)))
    means = kmm.cluster_centers_[:, 0]
    cluster_idx = kmm.predict(rd_samples.reshape((rd_samples.size, 1)))
    cluster_counts = np.bincount(cluster_idx)
    cluster_prop = cluster_counts.astype(float) / cluster_counts.sum()
    return means[cluster_prop >= 0.01]


def calculate_candidate_h_monoclonal(minor_modes, h_normal=None, h_tumour=None):
    """readdepth.py:93-126: (h_normal, h_tumour) candidates; every mode above the normal depth is
    tried as one and as two minor copies."""
    if h_normal is None:
        h_normal = minor_modes.min()
    if h_tumour is not None:
        return np.array([[h_normal, h_tumour]])
    h_candidates = list()
    for h_t in minor_modes:
        if h_t <= h_normal:
            continue
        h_t = h_t - h_normal
        for scale in (1., 0.5):
            h_candidates.append(np.array([h_normal, h_t * scale]))
    return h_candidates


def estimate_ploidy(h, experiment):
    """readdepth.py:129-147: length-weighted mean of the raw total copy number."""
    read_depth = calculate_depth(experiment).copy()
    read_depth['major_raw'] = (read_depth['major'] - h[0]) / h[1:].sum()
    read_depth['minor_raw'] = (read_depth['minor'] - h[0]) / h[1:].sum()
    major, minor, length = read_depth.replace(np.inf, np.nan).dropna()[['major_raw', 'minor_raw', 'length']].values.T
    return ((major + minor) * length).sum() / length.sum()
