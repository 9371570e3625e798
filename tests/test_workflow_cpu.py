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
