"""Drop-in for the restart-level entry points of the reference's
remixt/analysis/pipeline.py: `fit` (:127-228) with the same arguments and result
dictionary, plus the batched / multi-GPU forms the restart axis maps to here."""
import itertools
import pickle

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


def fit_task(results_filename, experiment_filename, init_params, config, device=0, quiet=True):
    """analysis/pipeline.py:112-124 with the reference's arguments: one restart, experiment pickle in, pickled fit results out (the
    file `collate` reads).  The reference runs one such job per init_id; `fit_restarts_task` below is the batched form."""
    with open(experiment_filename, 'rb') as f:
        experiment = pickle.load(f)
    fit_results = fit(experiment, init_params, config, device=device, quiet=quiet)
    with open(results_filename, 'wb') as f:
        pickle.dump(fit_results, f)


def fit_restarts_task(results_filenames, experiment_filename, init_params_by_id, config, device=0, quiet=True, seeds=None, groups=2):
    """Every init_id of the `fit` axis (workflow.py:329-340) in one device batch: `results_filenames` maps init_id -> the pickle
    `fit_task` would have written for it, so that `collate` reads them unchanged."""
    with open(experiment_filename, 'rb') as f:
        experiment = pickle.load(f)
    results = fit_restarts(experiment, init_params_by_id, config, device=device, quiet=quiet, seeds=seeds, groups=groups)
    for init_id, filename in results_filenames.items():
        with open(filename, 'wb') as f:
            pickle.dump(results[init_id], f)


def fit_restarts(experiment, init_params_by_id, config, device=0, quiet=True, seeds=None, groups=2):
    """All restarts of one GPU in lockstep; returns {init_id: fit_results} like one
    `fit_task` per init_id (workflow.py:329-340).  With per-restart seeds the restarts run as
    `groups` RestartSets on their own streams and host threads (same per-restart results)."""
    ids = sorted(init_params_by_id)
    params = [init_params_by_id[i] for i in ids]
    max_cn = defaults.get_param(config, 'max_copy_number')
    if seeds is not None and groups > 1 and len(ids) >= 2 * groups:
        from ..restarts import RestartGroups
        rs = RestartGroups(experiment, params, max_cn, groups=groups, num_clones=3, device=device, quiet=quiet,
                           seeds=seeds, **_model_kwargs(experiment, config))
    else:
        rs = RestartSet(experiment, params, max_cn, num_clones=3, device=device, quiet=quiet, seeds=seeds,
                        **_model_kwargs(experiment, config))
    rs.fit(defaults.get_param(config, 'num_em_iter'), defaults.get_param(config, 'num_update_iter'))
    out = dict(zip(ids, rs.results()))
    rs.close()      # (the batches' device memory and streams now, not when the collector gets to them: DESIGN 4.6)
    return out


# ---------------------------------------------------------------------------------
# initialisation grid and collation (reference analysis/pipeline.py:12-109, 231-293)
# ---------------------------------------------------------------------------------
def generate_init_params(experiment, config):
    """The body of the reference's `init` (analysis/pipeline.py:16-109) on an experiment object:
    candidate (h_normal, h_tumour) from the minor-depth modes x tumour mix fractions, ploidy filter,
    one common max_depth, x divergence weights.  Returns (init_params dict keyed by init_id,
    read_depth table, minor_modes).  Seeds the global numpy RNG like the reference (:29)."""
    from . import readdepth
    get = lambda k: defaults.get_param(config, k)
    min_ploidy, max_ploidy = get('min_ploidy'), get('max_ploidy')
    h_normal, h_tumour = get('h_normal'), get('h_tumour')
    tumour_mix_fractions, divergence_weights = get('tumour_mix_fractions'), get('divergence_weights')
    max_copy_number = get('max_copy_number')
    np.random.seed(config.get('random_seed', 1234))

    read_depth = readdepth.calculate_depth(experiment)
    minor_modes = readdepth.calculate_minor_modes(read_depth)
    init_h_mono = readdepth.calculate_candidate_h_monoclonal(minor_modes, h_normal=h_normal, h_tumour=h_tumour)

    init_h_params, ploidy_estimates, max_depths = [], [], []
    for mode_idx, h_mono in enumerate(init_h_mono):
        estimated_ploidy = readdepth.estimate_ploidy(h_mono, experiment)
        assert not np.isinf(estimated_ploidy) and not np.isnan(estimated_ploidy)
        max_depth = 2. * h_mono[0] + (max_copy_number + 0.25) * h_mono[1]
        for mix_frac in tumour_mix_fractions:
            init_h_params.append({'mode_idx': mode_idx, 'h_normal': h_mono[0], 'h_tumour': h_mono[1], 'mix_frac': mix_frac})
            ploidy_estimates.append(estimated_ploidy)
            max_depths.append(max_depth)

    def ploidy_filter_dist(ploidy):
        if min_ploidy is not None and ploidy < min_ploidy:
            return min_ploidy - ploidy
        if max_ploidy is not None and ploidy > max_ploidy:
            return ploidy - max_ploidy
        return 0.

    keep = [ploidy_filter_dist(a) == 0. for a in ploidy_estimates]
    if not any(keep):
        # limits too strict: the single closest ploidy (all candidates at that distance)
        dists = [ploidy_filter_dist(a) for a in ploidy_estimates]
        keep = [a == min(dists) for a in dists]
    init_h_params = [a for i, a in enumerate(init_h_params) if keep[i]]
    max_depths = [a for i, a in enumerate(max_depths) if keep[i]]

    # one common max depth so that the objective is comparable between initialisations
    max_depth = min(max_depths)
    depth = experiment.x[:, 2] / experiment.l
    proportion_below_max_depth = np.sum((depth <= max_depth) * experiment.l) / np.sum(experiment.l)
    if proportion_below_max_depth < 0.75:
        raise ValueError('Unable to model {} of the genome, consider reducing max ploidy or increasing max copy number'.format(1. - proportion_below_max_depth))

    init_params = []
    for h_p, w in itertools.product(init_h_params, divergence_weights):
        params = h_p.copy()
        params['divergence_weight'] = w
        params['max_depth'] = max_depth
        init_params.append(params)
    return dict(enumerate(init_params)), read_depth, minor_modes


class _Store(object):
    """pandas.HDFStore when PyTables is installed (the reference's container), otherwise the same keys
    in a dict that is pickled to the file name on close."""

    def __init__(self, filename, mode):
        import pandas as pd
        self.filename, self.mode, self.hdf, self.data = filename, mode, None, {}
        try:
            import tables  # noqa: F401
            self.hdf = pd.HDFStore(filename, mode)
        except ImportError:
            if mode == 'r':
                with open(filename, 'rb') as f:
                    self.data = pickle.load(f)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.hdf is not None:
            self.hdf.close()
        elif self.mode != 'r':
            with open(self.filename, 'wb') as f:
                pickle.dump(self.data, f)

    @staticmethod
    def _key(k):
        return '/' + k.lstrip('/')

    def __setitem__(self, k, v):
        if self.hdf is not None:
            self.hdf[k] = v
        else:
            self.data[self._key(k)] = v

    def __getitem__(self, k):
        return self.hdf[k] if self.hdf is not None else self.data[self._key(k)]

    def keys(self):
        return list(self.hdf.keys()) if self.hdf is not None else list(self.data.keys())


def init(init_results_filename, experiment_filename, config):
    """analysis/pipeline.py:12-109 with the reference's arguments: experiment pickle in, dict
    init_id -> init_params out, read depth + minor modes stored under the reference's keys."""
    import pandas as pd
    with open(experiment_filename, 'rb') as f:
        experiment = pickle.load(f)
    init_params, read_depth, minor_modes = generate_init_params(experiment, config)
    with _Store(init_results_filename, 'w') as store:
        store['read_depth'] = read_depth
        store['minor_modes'] = pd.Series(minor_modes, index=range(len(minor_modes)))
    return init_params


def store_fit_results(store, experiment, fit_results, key_prefix):
    """analysis/pipeline.py:231-250."""
    import pandas as pd
    from . import experiment as _experiment
    h, cn, brk_cn = fit_results['h'], fit_results['cn'], fit_results['brk_cn']
    cn_table = _experiment.create_cn_table(experiment, cn, h)
    cn_table['prob_is_outlier_total'] = fit_results['p_outlier_total'][:, 1]
    cn_table['prob_is_outlier_allele'] = fit_results['p_outlier_allele'][:, 1]
    cn_table['total_likelihood_mask'] = fit_results['total_likelihood_mask']
    cn_table['allele_likelihood_mask'] = fit_results['allele_likelihood_mask']
    brk_cn_table = _experiment.create_brk_cn_table(brk_cn, experiment.breakpoint_segment_data)
    store[key_prefix + '/h'] = pd.Series(h, index=range(len(h)))
    store[key_prefix + '/cn'] = cn_table
    store[key_prefix + '/mix'] = pd.Series(h / h.sum(), index=range(len(h)))
    store[key_prefix + '/brk_cn'] = brk_cn_table


def store_optimal_solution(stats, store, config):
    """analysis/pipeline.py:253-264: best ELBO among the solutions under max_prop_diverge."""
    max_prop_diverge = defaults.get_param(config, 'max_prop_diverge')
    # a restart whose h M-step failed ends the reference's whole workflow (cn_model.py:510-521 raises inside its
    # job); the batched driver records the failure instead, and such a restart is never the optimal solution
    if 'error_message' in stats.columns:
        failed = stats['error_message'].fillna('').astype(str) != ''
        if failed.all():
            raise ValueError('every restart failed: ' + '; '.join(sorted(set(stats['error_message'].astype(str)))))
        stats = stats[~failed]
    if (stats['proportion_divergent'] < max_prop_diverge).any():
        stats = stats[stats['proportion_divergent'] < max_prop_diverge].copy()
    stats = stats.sort_values('elbo', ascending=False)
    solution_idx = stats.loc[stats.index[0], 'init_id']
    key_prefix = '/solutions/solution_{}'.format(solution_idx)
    store['/cn'] = store[key_prefix + '/cn']
    store['/mix'] = store[key_prefix + '/mix']
    store['/brk_cn'] = store[key_prefix + '/brk_cn']
    return solution_idx


def collate_results(store, experiment, fit_results_by_id, config, init_store=None):
    """The body of `collate` (analysis/pipeline.py:267-293) on in-memory fit results:
    stats table, every solution's tables, the optimal solution at the top level."""
    import pandas as pd
    stats_table = []
    for init_id, results in fit_results_by_id.items():
        stats = dict(results['stats'])
        stats['init_id'] = init_id
        stats_table.append(stats)
    stats_table = pd.DataFrame(stats_table)
    store['stats'] = stats_table
    if init_store is not None:
        for key in init_store.keys():
            store[key] = init_store[key]
    for init_id, results in fit_results_by_id.items():
        store_fit_results(store, experiment, results, 'solutions/solution_{0}'.format(init_id))
    return store_optimal_solution(stats_table, store, config)


def collate(collate_filename, experiment_filename, init_results_filename, fit_results_filenames, config):
    """analysis/pipeline.py:267-293 with the reference's arguments (pickled per-restart results)."""
    with open(experiment_filename, 'rb') as f:
        experiment = pickle.load(f)
    fit_results = {}
    for init_id, results_filename in fit_results_filenames.items():
        with open(results_filename, 'rb') as f:
            fit_results[init_id] = pickle.load(f)
    with _Store(collate_filename, 'w') as collated:
        with _Store(init_results_filename, 'r') as init_store:
            collate_results(collated, experiment, fit_results, config, init_store=init_store)


def run(experiment, config, device=0, seeds=None, quiet=True):
    """Experiment in -> best solution out on one GPU, without pypeliner: the reference's
    init -> fit_task per init_id -> collate chain (workflow.py:307-354) with every restart of the grid
    in one device batch.  Returns (init_params, fit_results by init_id, optimal init_id)."""
    init_params, _, _ = generate_init_params(experiment, config)
    if seeds is None:
        seeds = list(range(len(init_params)))
    results = fit_restarts(experiment, init_params, config, device=device, quiet=quiet, seeds=seeds)
    best = select_optimal(results, defaults.get_param(config, 'max_prop_diverge'))
    return init_params, results, best
