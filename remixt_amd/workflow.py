"""Single-process-per-GPU replacement of the reference's fit workflow (`create_fit_model_workflow`,
remixt/workflow.py:307-354): experiment pickle in -> results store out, with the reference's keys
(`stats`, `solutions/solution_{id}/{cn,brk_cn,h,mix}`, `/cn`, `/mix`, `/brk_cn`, `read_depth`,
`minor_modes`).  The three pypeliner stages map to

    init      -> analysis.pipeline.generate_init_params           (every rank, deterministic)
    fit       -> restarts.fit_restarts_distributed                 (restart i on rank i mod world, one gather)
    collate   -> analysis.pipeline.collate_results                 (rank 0)

Run it under `python -m torch.distributed.run --nproc-per-node <GPUs>` for several GPUs, or plainly for
one."""
import pickle

from . import defaults
from .analysis import pipeline
from .restarts import fit_restarts_distributed


def fit_model(experiment_filename, results_filename, config, seeds=None, device=None, kernel_module=None, quiet=True):
    """Returns the optimal init_id on rank 0 (None elsewhere)."""
    import pandas as pd
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
    rank = 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except ImportError:
        pass
    if rank != 0:
        return None
    with pipeline._Store(results_filename, 'w') as store:
        store['read_depth'] = read_depth
        store['minor_modes'] = pd.Series(minor_modes, index=range(len(minor_modes)))
        return pipeline.collate_results(store, experiment, results, config)
