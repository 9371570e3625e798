"""CPU: the fit workflow (init -> restarts -> collate -> results store) end to end over the CPU oracle
kernel -- the control flow the GPU runs, without a GPU."""
import pickle

import numpy as np

from remixt_amd import synthetic, workflow
from remixt_amd.analysis import pipeline


def test_fit_model_writes_the_reference_keys(tmp_path, oracle_mod):
    e = synthetic.make_experiment(120, num_clones=3, max_copy_number=2, num_chains=3, seed=4)
    with open(tmp_path / 'experiment.pickle', 'wb') as f:
        pickle.dump(e, f)
    config = {'max_copy_number': 2, 'h_normal': float(e.h[0]), 'h_tumour': float(e.h[1:].sum()), 'tumour_mix_fractions': [0.45, 0.2],
              'divergence_weights': [1e-6], 'num_em_iter': 1, 'num_update_iter': 2, 'min_ploidy': None, 'max_ploidy': None}
    best = workflow.fit_model(str(tmp_path / 'experiment.pickle'), str(tmp_path / 'results.store'), config, kernel_module=oracle_mod)
    with pipeline._Store(str(tmp_path / 'results.store'), 'r') as st:
        keys = set(k.lstrip('/') for k in st.keys())
        for k in ('stats', 'read_depth', 'minor_modes', 'cn', 'mix', 'brk_cn', 'solutions/solution_0/cn', 'solutions/solution_0/h',
                  'solutions/solution_1/mix', 'solutions/solution_1/brk_cn'):
            assert k in keys, k
        stats = st['stats']
        assert list(stats['init_id']) == [0, 1] and best in (0, 1)
        assert np.isclose(st['/mix'].values.sum(), 1.0)
        assert st['/cn'].equals(st['/solutions/solution_%d/cn' % best])
        assert len(st['/cn']) == 120 and {'major_1', 'minor_2', 'prob_is_outlier_total'} <= set(st['/cn'].columns)


def test_workflow_object_sample_config_and_file_tasks(tmp_path, oracle_mod):
    """create_fit_model_workflow with the reference's argument list (workflow.py:307-315): the per-sample overlay of the config
    (config.py:56-59) decides the restart grid; and the file-level tasks of the reference's DAG -- init -> fit_task per init_id ->
    collate (analysis/pipeline.py:12, 112-124, 267-293) -- give the same store as the one-call form."""
    from remixt_amd import defaults
    e = synthetic.make_experiment(100, num_clones=3, max_copy_number=2, num_chains=3, seed=6)
    exp_file = str(tmp_path / 'experiment.pickle')
    with open(exp_file, 'wb') as f:
        pickle.dump(e, f)
    base = {'max_copy_number': 2, 'h_normal': float(e.h[0]), 'h_tumour': float(e.h[1:].sum()), 'tumour_mix_fractions': [0.45, 0.2, 0.1],
            'divergence_weights': [1e-6], 'num_em_iter': 1, 'num_update_iter': 1, 'min_ploidy': None, 'max_ploidy': None,
            'sample_specific': {'tumour_a': {'tumour_mix_fractions': [0.3]}}}
    assert defaults.get_sample_config(base, 'tumour_a')['tumour_mix_fractions'] == [0.3]
    assert defaults.get_sample_config(base, 'other')['tumour_mix_fractions'] == [0.45, 0.2, 0.1] and defaults.get_sample_config(base, None) is not base
    wf = workflow.create_fit_model_workflow(exp_file, str(tmp_path / 'a.store'), base, '/no/ref/data', tumour_id='tumour_a', kernel_module=oracle_mod)
    assert wf.run() == 0
    with pipeline._Store(str(tmp_path / 'a.store'), 'r') as st:
        assert list(st['stats']['init_id']) == [0]                       # one mix fraction x one divergence weight for this sample
    # the reference's three file-level stages, one fit_task per init_id
    config = defaults.get_sample_config(base, 'other')
    init_params = pipeline.init(str(tmp_path / 'init.store'), exp_file, config)
    assert sorted(init_params) == [0, 1, 2]
    files = {}
    seeds = {0: 0, 1: 1, 2: 2}
    for init_id, p in init_params.items():
        files[init_id] = str(tmp_path / ('fit_%d.pickle' % init_id))
        np.random.seed(seeds[init_id])
        _fit_task_over(oracle_mod, files[init_id], exp_file, p, config)
    pipeline.collate(str(tmp_path / 'collated.store'), exp_file, str(tmp_path / 'init.store'), files, config)
    with pipeline._Store(str(tmp_path / 'collated.store'), 'r') as st:
        keys = set(k.lstrip('/') for k in st.keys())
        assert {'stats', 'read_depth', 'minor_modes', 'cn', 'mix', 'brk_cn'} <= keys
        assert all('solutions/solution_%d/%s' % (i, t) in keys for i in range(3) for t in ('cn', 'brk_cn', 'h', 'mix'))
        assert sorted(st['stats']['init_id']) == [0, 1, 2]


def _fit_task_over(kernel_module, results_filename, experiment_filename, init_params, config):
    """pipeline.fit_task with the model built over `kernel_module` (the CPU suite has no device): same file protocol."""
    from unittest import mock
    from remixt_amd import cn_model
    real = cn_model.BreakpointModel
    def over(*a, **kw):
        kw.pop('device', None)
        return real(*a, kernel_module=kernel_module, **kw)
    with mock.patch.object(pipeline, 'BreakpointModel', over):
        pipeline.fit_task(results_filename, experiment_filename, init_params, config)
