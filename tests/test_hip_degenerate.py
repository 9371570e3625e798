"""Degenerate inputs through the whole boundary (BreakpointModel -> C ABI -> kernels) against the CPU oracle: one segment, chains of ONE
segment (no adjacency at all), a breakpoint whose two ends are the two ends of the only segment, a breakpoint between two one-segment
chains, every segment its own chain but two.  (An EMPTY breakpoint dict is not an input: the reference's constructor raises on it,
cn_model.py:59, and so does the mirror -- asserted here.)"""
import copy

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _variant(e, keep, adjacencies, breakpoints):
    v = copy.copy(e)
    v.x = np.asarray(e.x)[keep].copy(); v.l = np.asarray(e.l)[keep].copy()
    v.adjacencies = set(adjacencies); v.breakpoints = dict(breakpoints)
    return v


CASES = {
    'one segment': lambda e: _variant(e, [0], [], {'b0': frozenset([(0, 0), (0, 1)])}),
    'three one-segment chains': lambda e: _variant(e, [0, 1, 2], [], {'b0': frozenset([(0, 1), (2, 0)])}),
    'one chain, breakpoint inside one adjacency': lambda e: _variant(e, list(range(6)), [(n, n + 1) for n in range(5)], {'b0': frozenset([(2, 1), (3, 0)])}),
    'breakpoint between two one-segment chains': lambda e: _variant(e, [0, 1], [], {'b0': frozenset([(0, 1), (1, 0)])}),
    'two-segment chain among one-segment chains': lambda e: _variant(e, list(range(5)), [(2, 3)], {'b0': frozenset([(0, 1), (4, 0)]), 'b1': frozenset([(2, 1), (3, 0)])}),
}


@pytest.mark.parametrize('name', sorted(CASES))
@pytest.mark.parametrize('max_cn,M', [(3, 3), (8, 3), (4, 2)])
def test_degenerate_input_matches_oracle(oracle_mod, name, max_cn, M):
    from remixt_amd import bpmodel, synthetic
    base = synthetic.make_experiment(12, num_clones=M, max_copy_number=max_cn, num_chains=2, seed=5)
    e = CASES[name](base)
    models = []
    for kern in (bpmodel, oracle_mod):
        m, h, _ = H.make_model(kern, M=M, max_cn=max_cn, experiment=e)
        models.append(H.attach(m, h))
    dev, ora = models
    assert np.isclose(dev.calculate_elbo(), ora.calculate_elbo(), rtol=1e-9)
    for sweep in range(2):
        for step in ('update_p_allele_swap', 'update_p_cn', 'update_p_breakpoint', 'update_p_outlier_total', 'update_p_outlier_allele'):
            getattr(dev, step)(); getattr(ora, step)()
            H.compare_models(dev, ora, tag='%s sweep %d %s' % (name, sweep, step))
        assert np.isclose(dev.calculate_elbo(), ora.calculate_elbo(), rtol=1e-8), (name, sweep)
    N1 = dev.num_segments
    a = np.zeros((N1, M, 2), dtype=np.int64); b = np.zeros((N1, M, 2), dtype=np.int64)
    dev.infer_cn(a); ora.infer_cn(b)
    assert np.array_equal(a, b), name


def test_empty_breakpoint_dict_raises_like_the_reference():
    from remixt_amd import bpmodel, synthetic
    e = _variant(synthetic.make_experiment(12, num_clones=3, max_copy_number=3, num_chains=2, seed=5), [0, 1, 2], [(0, 1), (1, 2)], {})
    with pytest.raises(ValueError):
        H.make_model(bpmodel, M=3, max_cn=3, experiment=e)
