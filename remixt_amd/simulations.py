"""Genome-mixture and read-count samplers of the reference's simulator:
`GenomeMixture`, `sample_random_breakpoints`, `GenomeMixtureSampler`, `Experiment` and
`ExperimentSampler` of remixt/simulations/experiment.py:965-1399, restated.

These produce the inputs of the hot path (x, l, adjacencies, breakpoints) from a collection of clone
genomes, with the truth (cn, h, outlier flags) kept next to them for accuracy regressions
(remixt_amd/evaluate.py).  They draw from numpy's GLOBAL generator with the reference's calls in the
reference's order, so a seeded run gives the reference's numbers (tests/test_simulations.py checks
this against vectors recorded from the reference).  The clone genomes themselves come from a duck-typed
`genome_collection` (attributes N, M, l, cn, adjacencies, breakpoints, segment_chromosome_id,
segment_start, segment_end): the reference's rearrangement-history sampler (:16-963) is out of scope
(it does not run on current scipy), remixt_amd.synthetic.collection() gives a simple one.

Host-side numpy; nothing here touches the device.
"""
import numpy as np
import pandas as pd

from . import likelihood


class GenomeMixture(object):
    """Normal + tumour clone genomes with mixing fractions and the detected breakpoints
    (simulations/experiment.py:965-1032)."""

    def __init__(self, genome_collection, frac, detected_breakpoints):
        self.genome_collection = genome_collection
        self.frac = frac
        self.detected_breakpoints = detected_breakpoints
        rows = []
        for prediction_id, breakpoint in self.detected_breakpoints.items():
            row = {'prediction_id': prediction_id}
            for i, (n, side) in enumerate(breakpoint):
                if side not in (0, 1):
                    raise Exception('unexpected side value')
                # a breakend on the left side of a segment points away from it: '-' at the segment start
                row['n_%d' % (i + 1)] = n
                row['side_%d' % (i + 1)] = side
                row['chromosome_%d' % (i + 1)] = self.segment_chromosome_id[n]
                row['position_%d' % (i + 1)] = self.segment_end[n] if side == 1 else self.segment_start[n]
                row['strand_%d' % (i + 1)] = '+' if side == 1 else '-'
            rows.append(row)
        self.breakpoint_segment_data = pd.DataFrame(rows)

    N = property(lambda self: self.genome_collection.N)
    M = property(lambda self: self.genome_collection.M)
    l = property(lambda self: self.genome_collection.l)
    segment_chromosome_id = property(lambda self: self.genome_collection.segment_chromosome_id)
    segment_start = property(lambda self: self.genome_collection.segment_start)
    segment_end = property(lambda self: self.genome_collection.segment_end)
    cn = property(lambda self: self.genome_collection.cn)
    adjacencies = property(lambda self: self.genome_collection.adjacencies)
    breakpoints = property(lambda self: self.genome_collection.breakpoints)


def sample_random_breakpoints(N, num_breakpoints, adjacencies, excluded_breakpoints=None):
    """`num_breakpoints` distinct random breakpoints that are neither reference adjacencies, nor a
    breakend paired with itself, nor in `excluded_breakpoints` (simulations/experiment.py:1035-1063).
    Four randint draws per attempt: segment, segment, side, side."""
    found = set()
    while len(found) < num_breakpoints:
        n_1 = np.random.randint(N)
        n_2 = np.random.randint(N)
        side_1 = np.random.randint(2)
        side_2 = np.random.randint(2)
        wild_type = (((n_1, n_2) in adjacencies and side_1 == 1 and side_2 == 0) or
                     ((n_2, n_1) in adjacencies and side_2 == 1 and side_1 == 0))
        if wild_type or (n_1, side_1) == (n_2, side_2):
            continue
        candidate = frozenset([(n_1, side_1), (n_2, side_2)])
        if excluded_breakpoints is not None and candidate in excluded_breakpoints:
            continue
        found.add(candidate)
    return found


class GenomeMixtureSampler(object):
    """simulations/experiment.py:1066-1125."""

    def __init__(self, params):
        self.frac_normal = params.get('frac_normal', 0.4)
        self.frac_clone_concentration = params.get('frac_clone_concentration', 1.)
        self.frac_clone_1 = params.get('frac_clone_1', None)
        self.num_false_breakpoints = params.get('num_false_breakpoints', 50)
        self.proportion_breakpoints_detected = params.get('proportion_breakpoints_detected', 0.9)

    def sample_genome_mixture(self, genome_collection):
        M = genome_collection.M
        tumour = 1 - self.frac_normal
        frac = np.zeros((M,))
        frac[0] = self.frac_normal
        if self.frac_clone_1 is None:
            frac[1:] = np.random.dirichlet([self.frac_clone_concentration] * (M - 1)) * tumour
        elif M == 3:
            frac[1:] = np.array([self.frac_clone_1, 1. - self.frac_normal - self.frac_clone_1])
        elif M == 4:
            rest = 1. - self.frac_normal - self.frac_clone_1
            rest = np.random.dirichlet([self.frac_clone_concentration] * (M - 2)) * rest
            frac[1:] = np.array([self.frac_clone_1] + list(rest))
        else:
            raise Exception('Case not handled')
        assert abs(1. - np.sum(frac)) < 1e-8

        # a random subset of the true breakpoints is "detected", then false ones are added
        num_detected = int(self.proportion_breakpoints_detected * len(genome_collection.breakpoints))
        detected = list(genome_collection.breakpoints)
        np.random.shuffle(detected)
        detected = detected[:num_detected]
        detected.extend(sample_random_breakpoints(
            genome_collection.N, self.num_false_breakpoints, genome_collection.adjacencies,
            excluded_breakpoints=genome_collection.breakpoints))
        return GenomeMixture(genome_collection, frac, dict(enumerate(detected)))


class Experiment(object):
    """Read counts of one sequencing experiment with the truth they were drawn from
    (simulations/experiment.py:1128-1190); exposes what BreakpointModel / analysis.pipeline read."""

    def __init__(self, genome_mixture, h, phi, x, h_pred, **kwargs):
        self.genome_mixture = genome_mixture
        self.h = h
        self.phi = phi
        self.x = x
        self.h_pred = h_pred
        self.__dict__.update(kwargs)

    N = property(lambda self: self.genome_mixture.N)
    M = property(lambda self: self.genome_mixture.M)
    l = property(lambda self: self.genome_mixture.l)
    segment_chromosome_id = property(lambda self: self.genome_mixture.segment_chromosome_id)
    segment_start = property(lambda self: self.genome_mixture.segment_start)
    segment_end = property(lambda self: self.genome_mixture.segment_end)
    cn = property(lambda self: self.genome_mixture.cn)
    adjacencies = property(lambda self: self.genome_mixture.adjacencies)
    breakpoints = property(lambda self: self.genome_mixture.detected_breakpoints)
    breakpoint_segment_data = property(lambda self: self.genome_mixture.breakpoint_segment_data)

    @property
    def chains(self):
        """Half-open [start, end) runs of reference-adjacent segments."""
        cuts = [i + 1 for i in range(self.N - 1) if (i, i + 1) not in self.adjacencies]
        return zip(sorted([0] + cuts), sorted([self.N] + cuts))


def _negbin_draws(mu, r):
    """One scalar negative_binomial draw per segment.  The mean is nudged IN PLACE, as in the reference
    (:1193-1197): the second mixture component therefore sees the first one's nudge as well."""
    mu += 1e-16
    success = r / (r + mu)
    return np.array([np.random.negative_binomial(r, a) for a in success]).reshape(mu.shape)


def _negbin_mixture(mu, r_0, r_1, mix):
    x_0 = _negbin_draws(mu, r_0)
    x_1 = _negbin_draws(mu, r_1)
    from_0 = np.random.random(size=x_0.shape) > mix
    return np.where(from_0, x_0, x_1), from_0


def _betabin_draws(n, p, M):
    return np.random.binomial(n, np.random.beta(M * p, M * (1 - p)))


def _betabin_mixture(n, p, M_0, M_1, mix):
    x_0 = _betabin_draws(n, p, M_0)
    x_1 = _betabin_draws(n, p, M_1)
    from_0 = np.random.random(size=x_0.shape) > mix
    return np.where(from_0, x_0, x_1), from_0


_NORMAL_VAR_TOTAL = (0.14514556880927346, 1.3745893696636038)      # variance = a * mu**b (:1304, :1345)
_NORMAL_VAR_ALLELE = (0.040819090849598873, 1.4981089638117262)


def _power_variance(mu, ab):
    var = ab[0] * mu ** ab[1]
    var[var == 0] = 50.
    return var


class ExperimentSampler(object):
    """Read counts for a genome mixture (simulations/experiment.py:1222-1399).

    params: h_total (0.1), phi_min / phi_max (0.05 / 0.2), emission_model ('negbin_betabin'),
    frac_beta_noise_stddev (None) and, for 'negbin_betabin', negbin_r_0 / negbin_r_1 / negbin_mix
    (1000 / 10 / 0.01) and betabin_M_0 / betabin_M_1 / betabin_mix (2000 / 10 / 0.01).
    The 'negbin' and 'normal' models read `self.negbin_r` / `self.noise_prior`, which the reference's
    constructor never sets either: assign them on the instance first (AttributeError otherwise)."""

    def __init__(self, params):
        self.h_total = params.get('h_total', 0.1)
        self.phi_min = params.get('phi_min', 0.05)
        self.phi_max = params.get('phi_max', 0.2)
        self.emission_model = params.get('emission_model', 'negbin_betabin')
        if self.emission_model not in ('poisson', 'negbin', 'normal', 'full', 'negbin_betabin'):
            raise ValueError('emission_model must be one of "poisson", "negbin", "normal", "full"')
        self.frac_beta_noise_stddev = params.get('frac_beta_noise_stddev', None)
        self.params = params.copy()

    def sample_experiment(self, genome_mixture):
        N = genome_mixture.N
        h = genome_mixture.frac * self.h_total
        phi = np.random.uniform(low=self.phi_min, high=self.phi_max, size=N)
        mu = likelihood.expected_read_count(genome_mixture.l, genome_mixture.cn, h, phi)
        extra = dict()
        model = self.emission_model

        if model == 'poisson':
            rates = mu + 1e-16
            x = np.array([np.random.poisson(row) for row in rates]).reshape(rates.shape)

        elif model == 'negbin':
            means = mu + 1e-16
            success = self.negbin_r / (self.negbin_r + means)
            x = np.array([np.random.negative_binomial(self.negbin_r, row) for row in success]).reshape(means.shape)
            extra['negbin_r'] = self.negbin_r

        elif model == 'negbin_betabin':
            get = self.params.get
            total, total_from_0 = _negbin_mixture(
                mu[:, 2] + 1e-16, get('negbin_r_0', 1000.), get('negbin_r_1', 10.), get('negbin_mix', 0.01))
            allele_total = (phi * total).astype(int)
            p_true = mu[:, 0] / (mu[:, 0:2].sum(axis=1) + 1e-16)
            allele_1, allele_from_0 = _betabin_mixture(
                allele_total, p_true, get('betabin_M_0', 2000.), get('betabin_M_1', 10.), get('betabin_mix', 0.01))
            x = np.zeros(mu.shape)
            x[:, 2] = total
            x[:, 0] = allele_1
            x[:, 1] = allele_total - allele_1
            extra['is_outlier_total'] = ~total_from_0
            extra['is_outlier_allele'] = ~allele_from_0

        elif model == 'normal':
            x = np.zeros(mu.shape)
            x[:, 2] = np.random.normal(loc=mu[:, 2], scale=_power_variance(mu[:, 2], _NORMAL_VAR_TOTAL) ** 0.5)
            x[:, 0:2] = np.random.normal(loc=mu[:, 0:2], scale=_power_variance(mu[:, 0:2], _NORMAL_VAR_ALLELE) ** 0.5)
            x[x < 0] = 0
            x = x.round().astype(int)
            if self.noise_prior is not None:
                noise_range_total = mu[:, 2].max() * 1.25
                weights = [self.noise_prior, 1. - self.noise_prior]
                is_outlier_total = np.random.choice([True, False], size=mu[:, 2].shape, p=weights)
                x[is_outlier_total, 2] = (np.random.randint(noise_range_total, size=mu[:, 2].shape))[is_outlier_total]
                is_outlier_allele = np.random.choice([True, False], size=mu[:, 0].shape, p=weights)
                ratio = np.random.beta(2, 2, size=mu[:, 0].shape)
                x[is_outlier_allele, 0] = (ratio * x[:, 0:2].sum(axis=1))[is_outlier_allele]
                x[is_outlier_allele, 1] = ((1. - ratio) * x[:, 0:2].sum(axis=1))[is_outlier_allele]

        else:   # 'full': normal totals, dispersed binomial alleles with uniform-noise outliers
            x = np.zeros(mu.shape)
            x[:, 2] = np.random.normal(loc=mu[:, 2], scale=_power_variance(mu[:, 2], _NORMAL_VAR_TOTAL) ** 0.5)
            M, loh_p, noise_prior = 1200, 0.01, 0.03
            p_true = mu[:, 0] / mu[:, 0:2].sum(axis=1)
            p_true[p_true == 0] = loh_p
            p_true[p_true == 1] = loh_p
            dispersed = np.random.beta(M * p_true, M * (1 - p_true))
            noise = np.random.random(size=p_true.shape)
            is_noise = np.random.random(size=p_true.shape) <= noise_prior
            allele_reads = (x[:, 2] * phi).astype(int)
            x[:, 0] = np.random.binomial(allele_reads, np.where(is_noise, noise, dispersed))
            x[:, 1] = allele_reads.astype(float) - x[:, 0]

        # columns become (major, minor, total); remember which allele the major one was
        major_is_allele_a = x[:, 0] > x[:, 1]
        flip = ~major_is_allele_a
        x[flip, 0], x[flip, 1] = x[flip, 1].copy(), x[flip, 0].copy()
        extra['segment_major_is_allele_a'] = major_is_allele_a * 1

        if self.frac_beta_noise_stddev is not None:
            mean, var = genome_mixture.frac, self.frac_beta_noise_stddev ** 2.
            if np.any(var >= mean * (1. - mean)):
                raise ValueError('var >= mu * (1. - mu)')
            nu = mean * (1. - mean) / var - 1.
            frac = np.array([np.random.beta(a, b) for a, b in zip(mean * nu, (1 - mean) * nu)])
        else:
            frac = genome_mixture.frac
        return Experiment(genome_mixture, h, phi, x, frac * self.h_total, **extra)
