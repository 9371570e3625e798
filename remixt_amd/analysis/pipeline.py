"""Drop-in for the restart-level entry points of the reference's
remixt/analysis/pipeline.py: `fit` (:127-228) with the same arguments and result
dictionary, plus the batched / multi-GPU forms the restart axis maps to here."""
import numpy as np

from .. import defaults
from ..cn_model import BreakpointModel
from ..restarts import RestartSet, collect_fit_results, fit_restarts_distributed, select_optimal  # noqa: F401


def _model_kwargs(experiment, config):
    get = lambda k: defaults.get_param(config, k)
    normal_copies = np.array([[1, 1]] * experiment.l.shape[0])
    if not get('is_female'):
        chrom = np.asarray(experiment.segment_chromosome_id)
        normal_copies[chrom == 'X', :] = np.array([1, 0])
        if np.any(experiment.x[chrom == 'X', 0:2] > 0):
            raise Exception('inconsistent allele read counts for chromosome X')
    return dict(
        normal_contamination=get('normal_contamination'),
        min_segment_length=get('likelihood_min_segment_length'),
        min_proportion_genotyped=get('likelihood_min_proportion_genotyped'),
        normal_copies=normal_copies,
        disable_breakpoints=get('disable_breakpoints'),
        do_h_update=get('do_h_update'),
    )


def fit(experiment, init_params, config, device=0, quiet=False):
    """analysis/pipeline.py:127-228 (one restart)."""
    h_init = np.array([
        init_params['h_normal'],
        init_params['h_tumour'] * init_params['mix_frac'],
        init_params['h_tumour'] * (1. - init_params['mix_frac']),
    ])
    breakpoint_init = None
    if config.get('optimal_initialization', False):
        breakpoint_init = experiment.genome_mixture.genome_collection.collapsed_breakpoint_copy_number()
        for bp in experiment.genome_mixture.detected_breakpoints.values():
            if bp not in breakpoint_init:
                breakpoint_init[bp] = np.zeros((experiment.genome_mixture.M,))
        swap = (experiment.h[1] < experiment.h[2]) != (h_init[1] < h_init[2])
        if swap:
            for bp, cn in list(breakpoint_init.items()):
                cn = cn.copy()
                cn[1:] = cn[1:][::-1]
                breakpoint_init[bp] = cn
    model = BreakpointModel(
        experiment.x, experiment.l, experiment.adjacencies, experiment.breakpoints,
        max_copy_number=defaults.get_param(config, 'max_copy_number'),
        divergence_weight=init_params['divergence_weight'], max_depth=init_params['max_depth'],
        breakpoint_init=breakpoint_init, device=device, quiet=quiet, **_model_kwargs(experiment, config))
    model.num_em_iter = defaults.get_param(config, 'num_em_iter')
    model.num_update_iter = defaults.get_param(config, 'num_update_iter')
    model.fit(h_init)
    return collect_fit_results(model, experiment, init_params)


def fit_restarts(experiment, init_params_by_id, config, device=0, quiet=True, seeds=None):
    """All restarts of one GPU in lockstep; returns {init_id: fit_results} like one
    `fit_task` per init_id (workflow.py:329-340)."""
    ids = sorted(init_params_by_id)
    rs = RestartSet(experiment, [init_params_by_id[i] for i in ids], defaults.get_param(config, 'max_copy_number'),
                    num_clones=3, device=device, quiet=quiet, seeds=seeds, **_model_kwargs(experiment, config))
    rs.fit(defaults.get_param(config, 'num_em_iter'), defaults.get_param(config, 'num_update_iter'))
    return dict(zip(ids, rs.results()))
