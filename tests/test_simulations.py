"""CPU: the genome-mixture and read-count samplers (remixt_amd/simulations.py) against vectors recorded
from the reference's GenomeMixtureSampler / ExperimentSampler (simulations/experiment.py:1066-1399) by
oracle/make_golden.py `sampler_cases`: same seed of numpy's global generator -> same numbers, bit for bit
(integers, and floats that come from the same numpy calls in the same order)."""
import os

import numpy as np
import pytest

from remixt_amd import simulations as sim
from remixt_amd import synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'simulations.npz')

# (mixture-sampler params, experiment-sampler params, attributes set on the sampler) -- the table of
# oracle/make_golden.py SAMPLER_CASES, in its order (the per-case seed is 1000 + index)
CASES = [
    ('default', {}, {}, {}),
    ('custom', {'frac_normal': 0.3, 'frac_clone_1': 0.5, 'num_false_breakpoints': 7, 'proportion_breakpoints_detected': 0.5},
     {'h_total': 0.05, 'negbin_r_0': 300., 'negbin_mix': 0.1, 'betabin_M_0': 500., 'betabin_mix': 0.2, 'frac_beta_noise_stddev': 0.05}, {}),
    ('poisson', {'num_false_breakpoints': 3}, {'emission_model': 'poisson'}, {}),
    ('full', {'num_false_breakpoints': 3}, {'emission_model': 'full'}, {}),
    ('negbin', {'num_false_breakpoints': 3}, {'emission_model': 'negbin'}, {'negbin_r': 200.}),
    ('normal', {'num_false_breakpoints': 3}, {'emission_model': 'normal'}, {'noise_prior': None}),
    ('normal_noise', {'num_false_breakpoints': 3}, {'emission_model': 'normal'}, {'noise_prior': 0.05}),
]


@pytest.fixture(scope='module')
def golden():
    return np.load(GOLDEN)


def _collection(g):
    adjacencies = set((int(a), int(b)) for a, b in g['adjacencies'])
    breakpoints = set(frozenset((tuple(r[0]), tuple(r[1]))) for r in g['true_breakpoints'].tolist())
    return synthetic.GenomeCollection(g['l'], g['cn'], adjacencies, breakpoints, g['chromosome'], g['segment_start'], g['segment_end'])


def test_case_table_matches_the_recorded_one(golden):
    assert [c[0] for c in CASES] == list(golden['case_names'])


@pytest.mark.parametrize('index', range(len(CASES)))
def test_samplers_reproduce_the_reference_draws(golden, index):
    name, mix_params, exp_params, attrs = CASES[index]
    gc = _collection(golden)
    np.random.seed(1000 + index)
    gm = sim.GenomeMixtureSampler(mix_params).sample_genome_mixture(gc)
    sampler = sim.ExperimentSampler(exp_params)
    for k, v in attrs.items():
        setattr(sampler, k, v)
    e = sampler.sample_experiment(gm)

    assert np.array_equal(gm.frac, golden[name + '_frac'])
    detected = np.array([[list(be) for be in sorted(gm.detected_breakpoints[k])] for k in sorted(gm.detected_breakpoints)], dtype=np.int64)
    assert np.array_equal(detected, golden[name + '_detected'])
    assert np.array_equal(np.asarray(e.x, dtype=float), golden[name + '_x'])
    assert np.array_equal(e.h, golden[name + '_h']) and np.array_equal(e.phi, golden[name + '_phi'])
    assert np.array_equal(e.h_pred, golden[name + '_h_pred'])
    assert np.array_equal(np.asarray(e.segment_major_is_allele_a), golden[name + '_major_is_a'])
    assert np.array_equal(np.array(list(e.chains), dtype=np.int64), golden[name + '_chains'])
    for k in ('is_outlier_total', 'is_outlier_allele'):
        if name + '_' + k in golden.files:
            assert np.array_equal(np.asarray(getattr(e, k)), golden[name + '_' + k])
        else:
            assert not hasattr(e, k)
    bsd = gm.breakpoint_segment_data
    assert np.array_equal(bsd[['position_1', 'position_2']].values.astype(np.int64), golden[name + '_bsd_position'])
    assert np.array_equal(np.array(bsd[['strand_1', 'strand_2']].values.tolist()), golden[name + '_bsd_strand'])
    # what the hot path reads from the experiment
    assert e.N == len(golden['l']) and e.M == 3 and e.breakpoints is gm.detected_breakpoints
    assert np.all(e.x[:, 1] <= e.x[:, 0])


def test_missing_sampler_attributes_fail_like_the_reference(golden):
    """'negbin' and 'normal' read attributes the constructor never sets (simulations/experiment.py:1272,
    :1318): AttributeError unless the caller assigned them."""
    gc = _collection(golden)
    np.random.seed(0)
    gm = sim.GenomeMixtureSampler({'num_false_breakpoints': 2}).sample_genome_mixture(gc)
    for model in ('negbin', 'normal'):
        with pytest.raises(AttributeError):
            sim.ExperimentSampler({'emission_model': model}).sample_experiment(gm)
    with pytest.raises(ValueError):
        sim.ExperimentSampler({'emission_model': 'gaussian'})
    with pytest.raises(ValueError):      # beta noise wider than the fractions allow
        sim.ExperimentSampler({'frac_beta_noise_stddev': 0.6}).sample_experiment(gm)


def test_random_breakpoints_respect_the_exclusions():
    adjacencies = set((n, n + 1) for n in range(9))
    np.random.seed(3)
    excluded = set([frozenset([(0, 1), (5, 0)])])
    found = sim.sample_random_breakpoints(10, 40, adjacencies, excluded_breakpoints=excluded)
    assert len(found) == 40 and not (found & excluded)
    for b in found:
        assert len(b) == 2                                     # never a breakend paired with itself
        (n1, s1), (n2, s2) = sorted(b)
        assert not (n2 == n1 + 1 and s1 == 1 and s2 == 0)       # never a reference adjacency


def test_sampled_experiment_feeds_the_model_host_side(golden):
    """The sampled experiment has what BreakpointModel's constructor needs (x ordered major, minor, total;
    breakpoints dict; adjacencies) -- host-side construction only, no device."""
    from remixt_amd.cn_model import BreakpointModel
    gc = _collection(golden)
    np.random.seed(5)
    gm = sim.GenomeMixtureSampler({'num_false_breakpoints': 4}).sample_genome_mixture(gc)
    e = sim.ExperimentSampler({}).sample_experiment(gm)
    m = BreakpointModel(e.x, e.l, e.adjacencies, e.breakpoints, max_copy_number=6, max_depth=1e9, quiet=True)
    assert m.N1 >= e.N and m.num_breakpoints == len(e.breakpoints)
