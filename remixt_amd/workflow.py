"""Single-process-per-GPU replacement of the reference's fit workflow (`create_fit_model_workflow`,
remixt/workflow.py:307-354): experiment pickle in -> results store out, with the reference's keys
(`stats`, `solutions/solution_{id}/{cn,brk_cn,h,mix}`, `/cn`, `/mix`, `/brk_cn`, `read_depth`,
`minor_modes`).  The three pypeliner stages map to

    init      -> analysis.pipeline.generate_init_params           (every rank, deterministic)
    fit       -> restarts.fit_restarts_distributed                 (restart i on rank i mod world, one gather)
    collate   -> analysis.pipeline.collate_results                 (rank 0)

Run it under `python -m torch.distributed.run --nproc-per-node <GPUs>` for several GPUs, or plainly for
one.  `create_fit_model_workflow` has the reference's argument list and returns an object whose `run()`
does what the pypeliner scheduler does with the reference's workflow object; `fit_model` is the same
call as a function."""
import pickle

from . import defaults
from .analysis import pipeline
from .restarts import fit_restarts_distributed


def _rank():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except ImportError:
        pass
    return 0


def fit_model(experiment_filename, results_filename, config, ref_data_dir=None, tumour_id=None, seeds=None, device=None, kernel_module=None,
              quiet=True):
    """The transform chain of create_fit_model_workflow (workflow.py:307-354) with its arguments: `config` is overlaid with
    `config['sample_specific'][tumour_id]` (remixt/config.py:56-59); `ref_data_dir` is accepted for signature compatibility -- the
    reference passes it through to the workflow and none of the three fit stages reads it.  Returns the optimal init_id on rank 0
    (None elsewhere)."""
    import pandas as pd
    config = defaults.get_sample_config(config, tumour_id)
    with open(experiment_filename, 'rb') as f:
        experiment = pickle.load(f)
    init_params, read_depth, minor_modes = pipeline.generate_init_params(experiment, config)
    ids = sorted(init_params)
    if seeds is None:
        seeds = list(range(len(ids)))
    get = lambda k: defaults.get_param(config, k)
    results = fit_restarts_distributed(
        experiment, [init_params[i] for i in ids], get('max_copy_number'), num_clones=3,
        num_em_iter=get('num_em_iter'), num_update_iter=get('num_update_iter'), device=device, kernel_module=kernel_module,
        seeds=seeds, quiet=quiet, **pipeline._model_kwargs(experiment, config))
    results = dict((ids[k], results[k]) for k in results)
    if _rank() != 0:
        return None
    with pipeline._Store(results_filename, 'w') as store:
        store['read_depth'] = read_depth
        store['minor_modes'] = pd.Series(minor_modes, index=range(len(minor_modes)))
        return pipeline.collate_results(store, experiment, results, config)


class FitModelWorkflow(object):
    """What create_fit_model_workflow returns here: the call, kept until `run()`.  (The reference returns a pypeliner Workflow that a
    scheduler runs as one job per stage and init_id; this one runs the stages in the calling process, the `fit` axis as device batches
    sharded over the process group's ranks.)"""

    def __init__(self, experiment_filename, results_filename, config, ref_data_dir, tumour_id=None, **run_kwargs):
        self.args = (experiment_filename, results_filename, config, ref_data_dir, tumour_id)
        self.run_kwargs = run_kwargs
        self.optimal_init_id = None

    def run(self):
        self.optimal_init_id = fit_model(*self.args, **self.run_kwargs)
        return self.optimal_init_id


def create_fit_model_workflow(experiment_filename, results_filename, config, ref_data_dir, tumour_id=None, **run_kwargs):
    """remixt/workflow.py:307-354, same positional arguments.  `run_kwargs` (seeds, device, kernel_module, quiet) go to `fit_model`."""
    return FitModelWorkflow(experiment_filename, results_filename, config, ref_data_dir, tumour_id=tumour_id, **run_kwargs)
