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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bench(extra, env_extra=None):
    """Run bench.main() through tests/bench_rehearsal.py (CPU oracle kernel, gloo: launch / shard / gather logic only) and
    return its JSON line."""
    import json
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'bench_rehearsal.py'), '--segments', '80', '--max-cn', '2', '--steps', '1', '--warmup', '0',
                          '--no-cpu-baseline'] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` outside torchrun starts the two ranks itself (--gpus was ignored)."""
    line = _bench(['--gpus', '2', '--restarts', '2', '--strong-total', '5'])
    assert line['n_gpus'] == 2 and line['config']['world_size_observed'] == 2
    # ... and the fixed job of BASELINE configs[3] beside the weak-scaling line (5 restarts here: 3 + 2)
    s64 = line['configs3_strong_64']
    assert s64['scaling'] == 'strong' and s64['restarts_total'] == 5 and s64['restarts_this_rank'] == 3 and s64['value'] > 0
    assert line['scaling'] == 'weak' and line['config']['restarts_total'] == 4 and line['config']['restarts_this_rank'] == 2
    assert line['final_gather']['records'] == 4
    # the gather is the real message: a float64 and an int8 record per restart
    assert line['final_gather']['int8_record_bytes'] > 0 and line['final_gather']['bytes_per_rank'] == 2 * (
        line['final_gather']['float_record_bytes'] + line['final_gather']['int8_record_bytes'])
    assert line['value'] > 0 and np.isfinite(line['elbo_best'])


def test_bench_strong_scaling_and_two_datasets():
    """BASELINE configs[3] / configs[4] shapes: a fixed total sharded over the ranks, two datasets side by side."""
    line = _bench(['--gpus', '2', '--total-restarts', '6', '--datasets', '2'])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong'
    assert line['config']['restarts_total'] == 6 and line['config']['datasets'] == 2 and line['final_gather']['records'] == 6
    assert 'configs[4]' in line['config']['workload']
    one = _bench(['--gpus', '1', '--total-restarts', '6'])
    assert one['n_gpus'] == 1 and one['scaling'] == 'strong' and 'configs[3]' in one['config']['workload']


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'WORLD_SIZE' in (out.stderr + out.stdout)


def test_bench_has_no_switch_that_times_the_checker():
    """outside the CPU-baseline leg bench.py never names the oracle, and reads no environment variable
    that could select a kernel module or a process-group backend."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main_part = src[src.index('def build_datasets'):]
    assert 'oracle' not in main_part.replace('oracle/', '')
    assert 'BENCH_KERNEL' not in src and 'BENCH_DIST_BACKEND' not in src


def test_two_rank_gloo_matches_single_process(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    out2 = tmp_path / 'two.pkl'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                           '--master-port', str(_free_port()), str(script), str(out2)], env=env, timeout=600)
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
