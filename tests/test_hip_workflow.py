"""GPU: the fit workflow (remixt/workflow.py:307-354 create_fit_model_workflow: init -> fit per init_id -> collate) on the device, through
the reference's argument list, and the file-level tasks of its DAG."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REFERENCE_KEYS = ('stats', 'read_depth', 'minor_modes', 'cn', 'mix', 'brk_cn')      # analysis/pipeline.py:105-107, 253-293
SOLUTION_TABLES = ('cn', 'brk_cn', 'h', 'mix')                                      # :231-250


@pytest.fixture(scope='module')
def hip():
    from remixt_amd import bpmodel
    return bpmodel


@pytest.fixture(scope='module')
def case(tmp_path_factory, hip):
    from remixt_amd import synthetic
    tmp = tmp_path_factory.mktemp('workflow')
    e = synthetic.make_experiment(600, num_clones=3, max_copy_number=4, num_chains=5, seed=14)
    exp_file = str(tmp / 'experiment.pickle')
    with open(exp_file, 'wb') as f:
        pickle.dump(e, f)
    config = {'max_copy_number': 4, 'h_normal': float(e.h[0]), 'h_tumour': float(e.h[1:].sum()), 'tumour_mix_fractions': [0.45, 0.3, 0.2],
              'divergence_weights': [1e-6, 1e-8], 'num_em_iter': 2, 'num_update_iter': 3, 'min_ploidy': None, 'max_ploidy': None,
              'sample_specific': {'tumour_b': {'divergence_weights': [1e-7]}}}
    return tmp, e, exp_file, config


def test_fit_model_on_the_device_writes_every_reference_key(case):
    """workflow.fit_model / create_fit_model_workflow(experiment, results, config, ref_data_dir, tumour_id) on the GPU: every key the
    reference's collate writes is in the store, for every init_id, and the optimal solution is the best ELBO under max_prop_diverge."""
    from remixt_amd import workflow, bpmodel
    from remixt_amd.analysis import pipeline
    tmp, e, exp_file, config = case
    wf = workflow.create_fit_model_workflow(exp_file, str(tmp / 'results.store'), config, '/unused/ref_data', tumour_id='tumour_a')
    best = wf.run()
    with pipeline._Store(str(tmp / 'results.store'), 'r') as st:
        keys = set(k.lstrip('/') for k in st.keys())
        stats = st['stats']
        ids = sorted(stats['init_id'])
        assert ids == list(range(6))                                                 # 3 mix fractions x 2 divergence weights
        for k in REFERENCE_KEYS:
            assert k in keys, k
        for i in ids:
            for t in SOLUTION_TABLES:
                assert 'solutions/solution_%d/%s' % (i, t) in keys
            cn = st['solutions/solution_%d/cn' % i]
            assert len(cn) == len(e.l) and {'major_0', 'minor_0', 'major_1', 'minor_1', 'major_2', 'minor_2', 'major_raw', 'total_depth_e', 'major_diff',
                                            'prob_is_outlier_total', 'prob_is_outlier_allele', 'total_likelihood_mask', 'allele_likelihood_mask'} <= set(cn.columns)
            assert np.isclose(st['solutions/solution_%d/mix' % i].values.sum(), 1.)
        for col in ('elbo', 'elbo_diff', 'error_message', 'num_clones', 'num_segments', 'ploidy', 'proportion_divergent', 'mode_idx', 'divergence_weight',
                    'negbin_r_0', 'negbin_r_1', 'betabin_M_0', 'betabin_M_1', 'init_id'):      # analysis/pipeline.py:209-226, 272-277
            assert col in stats.columns, col
        ok = stats[stats['proportion_divergent'] < 0.5]
        pool = ok if len(ok) else stats
        assert best == int(pool.sort_values('elbo', ascending=False)['init_id'].iloc[0])
        assert st['/cn'].equals(st['/solutions/solution_%d/cn' % best]) and st['/brk_cn'].equals(st['/solutions/solution_%d/brk_cn' % best])
        assert np.all(np.isfinite(stats['elbo'])) and not stats['error_message'].astype(str).str.len().any()
    # the per-sample overlay decides the grid (remixt/config.py:56-59)
    assert workflow.fit_model(exp_file, str(tmp / 'b.store'), config, None, 'tumour_b') in (0, 1, 2)
    with pipeline._Store(str(tmp / 'b.store'), 'r') as st:
        assert sorted(st['stats']['init_id']) == [0, 1, 2] and set(st['stats']['divergence_weight']) == {1e-7}


def test_file_level_tasks_reproduce_the_one_call_form(case):
    """init -> fit_task per init_id (analysis/pipeline.py:112-124: one restart, pickled results) -> collate, and the batched
    fit_restarts_task writing the same per-init_id pickles: the same keys, and solutions that agree as far as two M-step drivers that draw
    their 200-segment samples differently can (fit_task: numpy's global generator and choice(), like the reference; the batched driver:
    a generator per restart)."""
    from remixt_amd.analysis import pipeline
    tmp, e, exp_file, config = case
    from remixt_amd import defaults
    config = defaults.get_sample_config(config, 'tumour_b')                          # three init_ids
    init_params = pipeline.init(str(tmp / 'init.store'), exp_file, config)
    single, batched = {}, {}
    for init_id, p in init_params.items():
        single[init_id] = str(tmp / ('single_%d.pickle' % init_id)); batched[init_id] = str(tmp / ('batched_%d.pickle' % init_id))
        np.random.seed(100 + init_id)                                                # fit_task samples from the global generator, like the reference
        pipeline.fit_task(single[init_id], exp_file, p, config)
    pipeline.fit_restarts_task(batched, exp_file, init_params, config, seeds=[100 + i for i in sorted(init_params)])
    for files, name in ((single, 'single.store'), (batched, 'batched.store')):
        pipeline.collate(str(tmp / name), exp_file, str(tmp / 'init.store'), files, config)
    with pipeline._Store(str(tmp / 'single.store'), 'r') as a, pipeline._Store(str(tmp / 'batched.store'), 'r') as b:
        sa, sb = a['stats'].sort_values('init_id'), b['stats'].sort_values('init_id')
        np.testing.assert_allclose(sa['elbo'].values, sb['elbo'].values, rtol=1e-2)
        for i in sorted(init_params):      # (an h M-step accepted on one sample and rejected on another moves h by tens of per cent: shapes only)
            assert a['solutions/solution_%d/h' % i].shape == b['solutions/solution_%d/h' % i].shape == (3,)
            assert list(a['solutions/solution_%d/cn' % i].columns) == list(b['solutions/solution_%d/cn' % i].columns)
        assert set(k.lstrip('/') for k in a.keys()) == set(k.lstrip('/') for k in b.keys())


def test_fit_model_on_the_device_equals_fit_model_over_the_oracle(case, oracle_mod):
    """The workflow-level parity check (remixt/workflow.py:307-354 → analysis/pipeline.py:231-293): `workflow.fit_model` on the HIP library
    against the SAME call over the CPU oracle kernel (`kernel_module=oracle`: one model per restart, scipy per restart) on the same experiment
    pickle, config and seeds -- the same optimal init_id, every solution's ELBO to 1e-6 and h to 1e-4, and the `cn` / `brk_cn` tables of every
    solution equal (copy numbers, their differences and masks exactly, the float columns derived from h to 1e-4)."""
    from remixt_amd import workflow
    from remixt_amd.analysis import pipeline
    tmp, e, exp_file, config = case
    seeds = [41, 42, 43]
    best_hip = workflow.fit_model(exp_file, str(tmp / 'parity_hip.store'), config, None, 'tumour_b', seeds=seeds)
    best_cpu = workflow.fit_model(exp_file, str(tmp / 'parity_cpu.store'), config, None, 'tumour_b', seeds=seeds, kernel_module=oracle_mod)
    assert best_hip == best_cpu
    with pipeline._Store(str(tmp / 'parity_hip.store'), 'r') as a, pipeline._Store(str(tmp / 'parity_cpu.store'), 'r') as b:
        assert set(k.lstrip('/') for k in a.keys()) == set(k.lstrip('/') for k in b.keys())
        sa, sb = a['stats'].sort_values('init_id').reset_index(drop=True), b['stats'].sort_values('init_id').reset_index(drop=True)
        assert list(sa['init_id']) == list(sb['init_id']) == [0, 1, 2]
        np.testing.assert_allclose(sa['elbo'].values, sb['elbo'].values, rtol=1e-6)
        np.testing.assert_allclose(sa['elbo_diff'].values, sb['elbo_diff'].values, rtol=1e-4, atol=1e-6 * np.abs(sb['elbo'].values).max())
        for col in ('ploidy', 'proportion_divergent', 'negbin_r_0', 'negbin_r_1', 'betabin_M_0', 'betabin_M_1'):
            np.testing.assert_allclose(sa[col].values, sb[col].values, rtol=1e-3, err_msg=col)
        assert list(sa['error_message'].astype(str)) == list(sb['error_message'].astype(str))
        tables = ['cn', 'brk_cn', 'mix'] + ['solutions/solution_%d/%s' % (i, t) for i in range(3) for t in SOLUTION_TABLES]
        for key in tables:
            ta, tb = a[key], b[key]
            if not hasattr(ta, 'columns'):                      # h, mix: series
                np.testing.assert_allclose(np.asarray(ta, dtype=float), np.asarray(tb, dtype=float), rtol=1e-4, err_msg=key)
                continue
            assert list(ta.columns) == list(tb.columns) and len(ta) == len(tb), key
            for col in ta.columns:
                va, vb = ta[col].values, tb[col].values
                if va.dtype.kind == 'f':                        # depths, raw copies, outlier probabilities: functions of h and the posteriors
                    np.testing.assert_allclose(va, vb, rtol=1e-4, atol=1e-7, err_msg='%s.%s' % (key, col))
                else:                                           # decoded copy numbers (int64), masks, major/minor differences, ids, positions: exact
                    assert np.array_equal(va, vb), '%s.%s' % (key, col)
        np.testing.assert_allclose(a['read_depth'].values.astype(float), b['read_depth'].values.astype(float), rtol=1e-12)
