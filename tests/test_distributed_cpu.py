"""CPU: the multi-GPU restart sharding + gather logic over a world of 2 gloo ranks (the CPU
oracle stands in for the HIP kernel, which needs a GPU)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, pickle
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from oracle import oracle
    from remixt_amd import synthetic
    from remixt_amd.restarts import fit_restarts_distributed, select_optimal
    dist.init_process_group('gloo')
    e = synthetic.make_experiment(80, num_clones=3, max_copy_number=2, num_chains=3, seed=4)
    ps = synthetic.make_init_params(e, 5, 2)
    res = fit_restarts_distributed(e, ps, 2, num_clones=3, num_em_iter=1, num_update_iter=2, kernel_module=oracle,
                                   seeds=[11, 12, 13, 14, 15], quiet=True)
    assert sorted(res) == [0, 1, 2, 3, 4]
    if dist.get_rank() == 0:
        out = dict((i, (r['stats']['elbo'], r['h'], r['cn'], r['brk_cn'], r['p_outlier_total'])) for i, r in res.items())
        pickle.dump((out, select_optimal(res)), open(sys.argv[1], 'wb'))
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_gloo_matches_single_process(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    out2 = tmp_path / 'two.pkl'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                           '--master-port', '29533', str(script), str(out2)], env=env, timeout=600)
    import pickle
    two, best2 = pickle.load(open(out2, 'rb'))
    # single process, no process group
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from remixt_amd import synthetic
    from remixt_amd.restarts import fit_restarts_distributed, select_optimal
    e = synthetic.make_experiment(80, num_clones=3, max_copy_number=2, num_chains=3, seed=4)
    ps = synthetic.make_init_params(e, 5, 2)
    one = fit_restarts_distributed(e, ps, 2, num_clones=3, num_em_iter=1, num_update_iter=2, kernel_module=oracle,
                                   seeds=[11, 12, 13, 14, 15], quiet=True)
    assert select_optimal(one) == best2
    for i in range(5):
        elbo, h, cn, brk, q = two[i]
        assert elbo == one[i]['stats']['elbo']                      # restarts are independent: sharding changes nothing
        assert np.array_equal(h, one[i]['h']) and np.array_equal(cn, one[i]['cn'])
        assert all(np.array_equal(brk[k], one[i]['brk_cn'][k]) for k in brk)
        assert np.array_equal(q, one[i]['p_outlier_total'])
