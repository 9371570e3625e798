"""Synthetic segment / read-count / breakpoint data for benchmarks and tests.

The reference's simulator (remixt/simulations/experiment.py) does not run on
current scipy (`scipy.misc.logsumexp`, experiment.py:698).  This generator
follows what `ExperimentSampler.sample_experiment` (experiment.py:1243-1399)
and benchmark/sim_defs.yaml produce, at the level the hot path consumes:

* N segments in `num_chains` chromosomes (reference adjacencies = consecutive
  pairs inside a chromosome), lengths U(1e5, 1e6);
* piecewise-constant true clone copy number (runs of 5-50 segments; tumour clones
  differ by <= 1 copy of one allele in ~30 % of runs);
* total reads ~ negative-binomial mixture around l * sum_m h_m * cn (r = 1000, with
  a 1 % outlier component r = 10); a fraction phi ~ U(0.05, 0.2) of reads is
  allele-informative and split by a beta-binomial mixture (M = 2000 / outliers 10);
* x = [major, minor, total] (experiment.py:1262-1333 ordering convention);
* K = N/100 breakpoints whose breakends sit on copy-number change points where
  possible, never on a reference adjacency partner pair, every (segment, side) used
  at most once (experiment.py:1035-1063).
"""
import numpy as np


class SyntheticExperiment(object):
    """The four attributes the hot path reads from remixt.analysis.experiment.Experiment
    (x, l, adjacencies, breakpoints) + segment_chromosome_id + the simulated truth."""

    def __init__(self, x, l, adjacencies, breakpoints, segment_chromosome_id, h, cn):
        self.x = x
        self.l = l
        self.adjacencies = adjacencies
        self.breakpoints = breakpoints
        self.segment_chromosome_id = segment_chromosome_id
        self.h = h
        self.cn = cn
        # the remaining read-only properties of the reference Experiment that the tables either side of
        # the hot path use (analysis/experiment.py:284-310): coordinates laid end to end per chromosome
        import pandas as pd
        l_int = np.maximum(1, np.round(np.asarray(l)).astype(np.int64))
        start = np.zeros(len(l_int), dtype=np.int64)
        chrom = np.asarray(segment_chromosome_id)
        for c in pd.unique(chrom):
            idx = np.nonzero(chrom == c)[0]
            ends = np.cumsum(l_int[idx])
            start[idx] = np.concatenate([[0], ends[:-1]]) + 1
        self.segment_start = start
        self.segment_end = start + l_int - 1
        self.segment_major_is_allele_a = np.ones(len(l_int), dtype=np.int64)
        self.breakpoint_segment_data = pd.DataFrame({'prediction_id': list(breakpoints.keys())})


def _true_copy_number(rng, N, M, max_cn, chain_of):
    cn = np.zeros((N, M, 2), dtype=np.int64)
    cn[:, 0, :] = 1
    n = 0
    while n < N:
        run = int(rng.integers(5, 51))
        end = min(N, n + run)
        # do not let a run cross a chromosome boundary
        same = np.nonzero(chain_of[n:end] != chain_of[n])[0]
        if len(same):
            end = n + int(same[0])
        tot = int(min(max_cn, rng.choice([1, 2, 2, 2, 3, 3, 4, 5, 6, 8])))
        major = int(rng.integers((tot + 1) // 2, tot + 1))
        base = np.array([major, tot - major])
        for m in range(1, M):
            cn[n:end, m, :] = base
        if M > 2 and rng.random() < 0.3:
            m = int(rng.integers(1, M)); a = int(rng.integers(0, 2))
            delta = 1 if (rng.random() < 0.5 and cn[n, m, :].sum() < max_cn) else -1
            if cn[n, m, a] + delta >= 0 and cn[n, m, :].sum() + delta <= max_cn:
                cn[n:end, m, a] += delta
        n = end
    return cn


# lengths of the chromosomes 1 .. 22, X of the human genome in Mb (GRCh37): the proportions of the chains a real genome gives (5 : 1)
HUMAN_CHROMOSOME_MB = (249, 243, 198, 191, 181, 171, 159, 146, 141, 136, 135, 134, 115, 107, 103, 90, 81, 78, 59, 63, 48, 51, 155)


def make_experiment(num_segments, num_clones=3, max_copy_number=8, num_chains=23, seed=0,
                    h_total=0.1, num_breakpoints=None, chain_fractions=None):
    """`chain_fractions`: relative lengths of the chains (default: equal chains; HUMAN_CHROMOSOME_MB: the chromosomes
    of a genome).  The random draws do not depend on it."""
    rng = np.random.default_rng(seed)
    N, M = int(num_segments), int(num_clones)
    num_chains = max(1, min(num_chains, N // 2))
    if chain_fractions is not None:
        fr = np.asarray(chain_fractions, dtype=float)[:num_chains]
        num_chains = len(fr)
        bounds = np.concatenate([[0], np.round(np.cumsum(fr) / fr.sum() * N)]).astype(int)
        for c in range(1, num_chains + 1):      # at least two segments per chain
            bounds[c] = max(bounds[c], bounds[c - 1] + 2)
        bounds[-1] = N
    else:
        bounds = np.linspace(0, N, num_chains + 1).astype(int)
    chain_of = np.zeros(N, dtype=int)
    for c in range(num_chains):
        chain_of[bounds[c]:bounds[c + 1]] = c
    chrom = np.array([str(c + 1) for c in chain_of])
    adjacencies = set((n, n + 1) for n in range(N - 1) if chain_of[n] == chain_of[n + 1])

    l = rng.uniform(1e5, 1e6, size=N)
    frac = {1: [1.0], 2: [0.4, 0.6], 3: [0.4, 0.4, 0.2], 4: [0.3, 0.3, 0.25, 0.15]}[M]
    h = h_total * np.array(frac)
    cn = _true_copy_number(rng, N, M, max_copy_number, chain_of)

    tot = cn.sum(axis=2)                                  # (N, M)
    mu = l * (tot * h[None, :]).sum(axis=1) + 1e-16
    outlier = rng.random(N) < 0.01
    r = np.where(outlier, 10., 1000.)
    x_total = rng.negative_binomial(r, r / (r + mu)).astype(float)

    phi = rng.uniform(0.05, 0.2, size=N)
    n_allele = np.floor(phi * x_total)
    depth0 = (cn[:, :, 0] * h[None, :]).sum(axis=1)
    p_true = np.clip(depth0 / np.maximum((tot * h[None, :]).sum(axis=1), 1e-12), 1e-3, 1 - 1e-3)
    Mdisp = np.where(rng.random(N) < 0.01, 10., 2000.)
    pb = rng.beta(Mdisp * p_true, Mdisp * (1 - p_true))
    a0 = rng.binomial(n_allele.astype(np.int64), pb).astype(float)
    a1 = n_allele - a0
    x = np.stack([np.maximum(a0, a1), np.minimum(a0, a1), x_total], axis=1)

    # breakpoints: breakends at copy-number change points where available
    K = max(1, N // 100) if num_breakpoints is None else max(1, int(num_breakpoints))
    change = [n for n in range(N - 1) if (n, n + 1) in adjacencies and np.any(tot[n] != tot[n + 1])]
    rng.shuffle(change)
    ends = []
    for n in change:
        # the side of the boundary with more copies carries the breakend
        ends.append((n, 1) if tot[n].sum() >= tot[n + 1].sum() else (n + 1, 0))
    extra = [(int(n), int(s)) for n, s in zip(rng.integers(0, N, size=4 * K + 8), rng.integers(0, 2, size=4 * K + 8))]
    ends.extend(extra)
    used = set(); breakpoints = {}
    partner = {}
    for (a, b_) in adjacencies:
        partner[(a, 1)] = (b_, 0); partner[(b_, 0)] = (a, 1)
    pool = []
    for e in ends:
        if e not in used:
            used.add(e); pool.append(e)
    i = 0
    while len(breakpoints) < K and i + 1 < len(pool):
        e1, e2 = pool[i], pool[i + 1]
        i += 2
        if e1 == e2 or partner.get(e1) == e2:
            continue
        breakpoints['bp%d' % len(breakpoints)] = frozenset([e1, e2])
    if not breakpoints:
        breakpoints['bp0'] = frozenset([(0, 1), (N - 1, 0)])
    return SyntheticExperiment(x, l, adjacencies, breakpoints, chrom, h, cn)


def resample_counts(experiment, seed, h=None):
    """A second tumour sample of the same patient: the segmentation, reference adjacencies, breakpoints and
    clone genomes of `experiment`, its own clone mixture `h` (default: the tumour clones' shares exchanged)
    and its own read counts, drawn like make_experiment's.  The reference fits the samples of a patient
    independently on one segmentation (remixt/workflow.py:472-485)."""
    rng = np.random.default_rng(seed)
    l, cn = np.asarray(experiment.l), np.asarray(experiment.cn)
    N = len(l)
    if h is None:
        h = np.asarray(experiment.h, dtype=float).copy()
        if len(h) > 2:
            h[1:] = h[1:][::-1]
    tot = cn.sum(axis=2)
    mu = l * (tot * h[None, :]).sum(axis=1) + 1e-16
    r = np.where(rng.random(N) < 0.01, 10., 1000.)
    x_total = rng.negative_binomial(r, r / (r + mu)).astype(float)
    phi = rng.uniform(0.05, 0.2, size=N)
    n_allele = np.floor(phi * x_total)
    depth0 = (cn[:, :, 0] * h[None, :]).sum(axis=1)
    p_true = np.clip(depth0 / np.maximum((tot * h[None, :]).sum(axis=1), 1e-12), 1e-3, 1 - 1e-3)
    Mdisp = np.where(rng.random(N) < 0.01, 10., 2000.)
    pb = rng.beta(Mdisp * p_true, Mdisp * (1 - p_true))
    a0 = rng.binomial(n_allele.astype(np.int64), pb).astype(float)
    a1 = n_allele - a0
    x = np.stack([np.maximum(a0, a1), np.minimum(a0, a1), x_total], axis=1)
    return SyntheticExperiment(x, l, experiment.adjacencies, experiment.breakpoints, experiment.segment_chromosome_id, h, cn)


def make_init_params(experiment, num_restarts, max_copy_number, num_clones=3):
    """Restart grid in the shape of analysis/pipeline.py:42-58, 96-103: depth modes x
    tumour_mix_fractions x divergence_weights, all with one common max_depth."""
    h = experiment.h
    h_normal, h_tumour = float(h[0]), float(h[1:].sum())
    mode_scales = [1.0, 0.5, 2.0, 0.75, 1.5, 0.6, 1.25, 0.9]
    mix_fracs = [0.45, 0.3, 0.2, 0.1]
    weights = [1e-6, 1e-7, 1e-8]
    max_depth = min(2. * h_normal + (max_copy_number + 0.25) * h_tumour * s for s in mode_scales[:max(1, (num_restarts + 11) // 12)])
    params = []
    mode_idx = 0
    while len(params) < num_restarts:
        s = mode_scales[mode_idx % len(mode_scales)]
        for mix in mix_fracs:
            for w in weights:
                params.append({'mode_idx': mode_idx, 'h_normal': h_normal, 'h_tumour': h_tumour * s, 'mix_frac': mix,
                               'divergence_weight': w, 'max_depth': max_depth})
        mode_idx += 1
    return params[:num_restarts]


def h_init_from_params(p, num_clones=3):
    """analysis/pipeline.py:128-132 (three clones); two-clone variant for M = 2."""
    if num_clones == 2:
        return np.array([p['h_normal'], p['h_tumour']])
    return np.array([p['h_normal'], p['h_tumour'] * p['mix_frac'], p['h_tumour'] * (1. - p['mix_frac'])])


class GenomeCollection(object):
    """Clone genomes in the shape remixt_amd.simulations' samplers read (the reference's
    GenomeCollection, simulations/experiment.py:776-866, without its rearrangement history): segment
    lengths and coordinates, per-clone copy number, reference adjacencies and the set of true
    breakpoints."""

    def __init__(self, l, cn, adjacencies, breakpoints, segment_chromosome_id, segment_start, segment_end,
                 breakpoint_copy_number=None, minimal_breakpoint_copy_number=None, balanced_breakpoints=None):
        self.l = np.asarray(l)
        self.cn = np.asarray(cn)
        self.adjacencies = adjacencies
        self.breakpoints = breakpoints
        self.segment_chromosome_id = segment_chromosome_id
        self.segment_start = segment_start
        self.segment_end = segment_end
        # truth for remixt_amd.evaluate.evaluate_brk_cn_results: breakpoint -> copies per clone (normal first)
        self._brk_cn = breakpoint_copy_number
        self._min_brk_cn = minimal_breakpoint_copy_number if minimal_breakpoint_copy_number is not None else breakpoint_copy_number
        self._balanced = balanced_breakpoints if balanced_breakpoints is not None else set()

    def collapsed_breakpoint_copy_number(self):
        """Per-clone copies of every true breakpoint (simulations/experiment.py:856-857).  Without a
        recorded history the copies are the copy-number steps at the breakends (decode_breakpoints_naive)."""
        if self._brk_cn is None:
            from .cn_model import decode_breakpoints_naive
            self._brk_cn = decode_breakpoints_naive(self.cn, self.adjacencies, dict((b, b) for b in self.breakpoints))
            if self._min_brk_cn is None:
                self._min_brk_cn = self._brk_cn
        return self._brk_cn

    def collapsed_minimal_breakpoint_copy_number(self):
        """simulations/experiment.py:859-862 (here: the same table unless one was given)."""
        self.collapsed_breakpoint_copy_number()
        return self._min_brk_cn

    def collapsed_balanced_breakpoints(self):
        """simulations/experiment.py:864-865 (here: the set given at construction, empty by default)."""
        return self._balanced

    @property
    def N(self):
        return self.cn.shape[0]

    @property
    def M(self):
        return self.cn.shape[1]


def collection(num_segments, num_clones=3, max_copy_number=8, num_chains=23, seed=0, num_breakpoints=None):
    """A GenomeCollection with make_experiment's piecewise-constant clone copy number and change-point
    breakpoints: the input of simulations.GenomeMixtureSampler / ExperimentSampler."""
    e = make_experiment(num_segments, num_clones=num_clones, max_copy_number=max_copy_number, num_chains=num_chains,
                        seed=seed, num_breakpoints=num_breakpoints)
    return GenomeCollection(e.l, e.cn, e.adjacencies, set(e.breakpoints.values()), e.segment_chromosome_id,
                            e.segment_start, e.segment_end)
