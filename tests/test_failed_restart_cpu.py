"""CPU: a restart whose h M-step fails.  The reference raises inside that restart's job (cn_model.py:510-521) and the whole
workflow stops; the batched driver keeps the restart's previous h, records the failure -- and the record must survive the
fixed-size gather and keep the restart from ever being selected."""
import numpy as np
import pytest

from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet, _pack, _unpack, select_optimal, _failure_code


def _fit_with_failure(oracle_mod, strict=False):
    e = synthetic.make_experiment(90, num_clones=3, max_copy_number=2, num_chains=3, seed=4)
    ps = synthetic.make_init_params(e, 3, 2)
    rs = RestartSet(e, ps, 2, num_clones=3, quiet=True, kernel_module=oracle_mod, seeds=[1, 2, 3], strict=strict, mstep_threads=1)

    def broken():
        raise ValueError('optimization failed\n  message: ABNORMAL')
    rs.models[1].em_update_h = broken
    return e, ps, rs


def test_failed_h_step_is_recorded_carried_and_never_selected(oracle_mod):
    e, ps, rs = _fit_with_failure(oracle_mod)
    h1 = np.array(rs.models[1].model.h)
    rs.fit(num_em_iter=1, num_update_iter=1)
    assert np.array_equal(rs.models[1].model.h, h1)                       # h kept
    res = rs.results()
    assert [bool(r['stats']['error_message']) for r in res] == [False, True, False]
    assert res[1]['stats']['error_message'].startswith('optimization failed')
    # through the fixed-size records of the multi-GPU gather
    ids = list(e.breakpoints.keys()); names = list(rs.models[0].likelihood_params)
    back = {}
    for i, r in enumerate(res):
        f, i8 = _pack(r, len(e.x), 3, len(ids), len(names), ids, names)
        back[i] = _unpack(f, i8, len(e.x), 3, len(ids), len(names), ids, names, ps[i])
    assert back[1]['stats']['error_message'].startswith('optimization failed') and not back[0]['stats']['error_message']
    assert np.array_equal(back[2]['cn'], res[2]['cn']) and back[2]['stats']['elbo'] == res[2]['stats']['elbo']
    # the failed restart cannot win, whatever its ELBO
    back[1]['stats']['elbo'] = 1e30
    assert select_optimal(back) in (0, 2)
    from remixt_amd.analysis import pipeline
    import pandas as pd

    class Store(dict):
        pass
    st = Store()
    for i in back:
        st['/solutions/solution_%d/cn' % i] = i; st['/solutions/solution_%d/mix' % i] = i; st['/solutions/solution_%d/brk_cn' % i] = i
    table = pd.DataFrame([dict(back[i]['stats'], init_id=i) for i in back])
    assert pipeline.store_optimal_solution(table, st, {}) in (0, 2)
    # all failed: nothing to select
    for i in back:
        back[i]['stats']['error_message'] = 'optimization failed (h kept)'
    with pytest.raises(ValueError, match='every restart failed'):
        select_optimal(back)


def test_strict_mode_raises_like_the_reference(oracle_mod):
    e, ps, rs = _fit_with_failure(oracle_mod, strict=True)
    with pytest.raises(ValueError, match='optimization failed'):
        rs.fit(num_em_iter=1, num_update_iter=1)


def test_select_optimal_ignores_nan_elbo():
    """pandas' sort_values(ascending=False) puts NaN last (analysis/pipeline.py:257): a degenerate restart never wins."""
    def r(elbo, div=0.1):
        return {'stats': {'elbo': elbo, 'proportion_divergent': div, 'error_message': ''}}
    assert select_optimal({0: r(float('nan')), 1: r(-5.), 2: r(-3.)}) == 2
    assert select_optimal({0: r(float('nan')), 1: r(float('nan'))}) == 0
    assert select_optimal({0: r(-1., div=0.9), 1: r(-5.)}) == 1
    assert _failure_code('') == 0 and _failure_code('gradiant error, analytic: ...') == 2 and _failure_code('anything else') == 3
