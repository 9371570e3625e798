#!/usr/bin/env python3
"""Benchmark of the ReMixT variational-EM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = ONE EM iteration (reference remixt/cn_model.py:409-418: 5
variational sweeps + the h M-step + the likelihood-parameter M-steps + the ELBO)
for EVERY restart resident on the GPU.  Workload at N=1 = BASELINE.json
configs[2]: 50k segments, 3 clones (normal + 2 tumour), max_cn = 8 (165 states),
16 (h, divergence-weight) restarts.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself
(`python -m torch.distributed.run`, one process per GPU, RCCL) as a child process
BEFORE anything touches the GPU and exits with the child's code; under the
driver's own torchrun launch the ranks find WORLD_SIZE set and just run.

Scaling modes (restarts are independent units, reference remixt/workflow.py:329-340;
no collective during EM, ONE all-gather of the per-restart result records at the end):
  default                 weak: every rank fits its own `--restarts` (16) restarts
  --total-restarts T      strong: T restarts in total, restart i on rank i mod N
                          (BASELINE configs[3]: T = 64 -> 8 per GPU at N = 8)
  --datasets 2            BASELINE configs[4]: two tumour samples sharing segmentation and
                          breakpoints, fitted independently (reference workflow.py:472-485),
                          `--restarts` / `--total-restarts` split evenly between them

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dominant kernel, HIP-event timed on the batch stream) and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # dense FP64 rate of MI355X (vector = matrix on gfx950): 256 CUs x 4 SIMDs x 16 FMA lanes x 2.4 GHz
# algorithmic HBM bytes per (segment,state) cell per variational update, split by kernel (this design's
# own passes); SURVEY.md 8(d)'s model for the whole update is 88 B per cell
ALG_BYTES_PER_CELL = {
    'k_framelogprob': 64.0,      # read the 6 cached likelihood components, write f and exp(f - rowmax)
    'k_fb': 32.0,                # read exp(f - rowmax) (fwd) + write alpha + read it again (bwd) + write beta
    'k_marginals<true>': 72.0,   # read alpha, beta and the 6 cached components, write the posterior
}
SURVEY_BYTES_PER_CELL = 88.0
SWEEP_KERNELS = ('k_framelogprob', 'k_fb', 'k_marginals<true>', 'k_pairwise', 'k_brk_update', 'k_brk_lut',
                 'k_update_outlier_total', 'k_update_outlier_allele', 'k_update_allele_swap')


def alg_flops_per_cell(name, S):
    """FP64 flops per (segment, state) cell: the forward and the backward recursion are S x S
    matrix-vector products per segment, i.e. S FMAs per cell and direction."""
    return 4.0 * S if name == 'k_fb' else None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--segments', type=int, default=50000)
    ap.add_argument('--clones', type=int, default=3)
    ap.add_argument('--max-cn', type=int, default=8)
    ap.add_argument('--restarts', type=int, default=16, help='restarts per GPU (weak scaling, the default)')
    ap.add_argument('--total-restarts', type=int, default=0, help='strong scaling: this many restarts in total, sharded over the GPUs (BASELINE configs[3]: 64)')
    ap.add_argument('--strong-total', type=int, default=64, help='N > 1 in the default (weak) mode also times the fixed job of BASELINE configs[3] with this many restarts in total; 0 = skip')
    ap.add_argument('--datasets', type=int, default=1, help='2 = BASELINE configs[4]: two tumour samples on the same segmentation / breakpoints, fitted independently')
    ap.add_argument('--update-iters', type=int, default=5)
    ap.add_argument('--groups', type=int, default=2, help='restart groups per GPU and dataset (default 2; with --datasets 2 and no --groups: one per dataset, i.e. two on the device; own stream + host thread each; a restart\'s result is bit-identical across groupings per search mode and forward-backward workgroup shape, which RestartGroups picks by group size: DESIGN 4.6)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-segments', type=int, default=800, help='segments of the CPU baseline sample at the headline grid (about 10 s of one core)')
    ap.add_argument('--cpu-sample-segments-355', type=int, default=120, help='segments of the CPU baseline sample at 355 states (0 = skip)')
    ap.add_argument('--profile-all', action='store_true', help='HIP-event every kernel (default: only the variational-sweep kernels; the M-step objective kernels are ~3000 tiny launches per step)')
    ap.add_argument('--no-extra-states', action='store_true', help='skip the additional 355-state (max_cn = 12) measurement at N = 1')
    ap.add_argument('--no-fit-from-init', action='store_true', help='skip the construct -> 5 EM iterations -> decode wall-clock measurement at N = 1')
    ap.add_argument('--no-mstep', action='store_true', help='diagnostic only: variational sweeps without M-steps (NOT the reported metric)')
    ap.add_argument('--master-port', type=int, default=0, help='rendezvous port when bench.py starts the ranks itself (0 = pick a free one)')
    ap.add_argument('--option', action='append', default=[], metavar='NAME=VALUE',
                    help='A/B measurements: a tuning option of the library (include/remixt_amd.h, enum rmx_option_id) for every batch of this run')
    ap.add_argument('--host-option', action='append', default=[], metavar='NAME=VALUE',
                    help='A/B measurements: a keyword of remixt_amd.restarts.RestartSet (sample_prep, mstep_threads, joint_accept, lockstep, native_search) for this run')
    ap.add_argument('--lib', default=None, help='A/B measurements: an alternative build of libremixt_hip.so for this run')
    ap.add_argument('--switch-interval-us', type=float, default=0., help='A/B measurements: sys.setswitchinterval for the restart groups\' host threads (0 = leave Python\'s 5 ms)')
    ap.add_argument('--cpu-leg', action='store_true', help=argparse.SUPPRESS)       # internal: the CPU baseline child process
    ap.add_argument('--sub-run', default=None, help=argparse.SUPPRESS)              # internal: one additional measurement in a process of its own (restarts,groups,max_cn,nsteps,warm,unequal)
    args = ap.parse_args(argv)
    # (--groups given on the command line: honoured for several datasets too; otherwise DatasetGroups keeps two groups on the device at a time)
    args.groups_given = any(a == '--groups' or a.startswith('--groups=') for a in (argv if argv is not None else sys.argv[1:]))
    return args


# ---------------------------------------------------------------------------------------------------
# launching N ranks
# ---------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, script=None):
    """`bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process (never an exec of this one)
    before torch is imported here, and return its exit code.  `script`: the program every rank runs (this file)."""
    port = args.master_port or _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), script or os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


# ---------------------------------------------------------------------------------------------------
# CPU baseline: the C restatement (oracle/remixt_oracle.c, "port") on the box's host cores.  Runs in a
# child process started before this process touches the GPU.
# ---------------------------------------------------------------------------------------------------
def _cpu_one(job):
    """One EM iteration of one restart on the CPU restatement; returns the wall time split into the part that
    grows with N (sweeps, full-data objectives, ELBO) and the sampled M-step objectives (constant in N once
    N >= 2000: the reference's sample is min(200, N / 10) segments, cn_model.py:476)."""
    ns, clones, max_cn, update_iters, restart = job
    import types
    from oracle import oracle
    from remixt_amd import synthetic
    from remixt_amd.cn_model import BreakpointModel
    oracle.build()
    acc = {'sample_s': 0.0, 'sample_size': 0}

    class Timed(oracle.RemixtModel):
        def _timed(self, fn, sample, *a):
            part = int(np.count_nonzero(sample)) < self.num_segments
            t0 = time.perf_counter()
            out = fn(self, sample, *a)
            if part:
                acc['sample_s'] += time.perf_counter() - t0
                acc['sample_size'] = int(np.count_nonzero(sample))
            return out

        def calculate_expected_log_likelihood(self, sample):
            return self._timed(oracle.RemixtModel.calculate_expected_log_likelihood, sample)

        def calculate_expected_log_likelihood_partial_h(self, sample, out):
            return self._timed(oracle.RemixtModel.calculate_expected_log_likelihood_partial_h, sample, out)

    kern = types.SimpleNamespace(RemixtModel=Timed)
    e = synthetic.make_experiment(ns, num_clones=clones, max_copy_number=max_cn, num_chains=4, seed=123)
    p = synthetic.make_init_params(e, restart + 1, max_cn, num_clones=clones)[restart]
    m = BreakpointModel(e.x, e.l, e.adjacencies, e.breakpoints, max_copy_number=max_cn,
                        divergence_weight=p['divergence_weight'], max_depth=p['max_depth'], kernel_module=kern, quiet=True,
                        rng=np.random.RandomState(1000 + restart))
    m.num_update_iter = update_iters
    m._attach_model(m._build_model(synthetic.h_init_from_params(p, clones)))
    m.prev_elbo = m.model.calculate_elbo()
    t0 = time.perf_counter()
    try:
        m.em_iteration(0)
    except ValueError:          # an unsuccessful L-BFGS-B run ends the reference's restart; the time spent still counts
        pass
    dt = time.perf_counter() - t0
    return {'total_s': dt, 'sample_s': acc['sample_s'], 'sample_size': acc['sample_size'], 'N1': int(m.model.num_segments), 'S': int(m.model.num_cn_states)}


def cpu_leg(args):
    """Child process: (1) one restart on one core, (2) one restart per host core side by side (the reference's
    own parallelism model: one process per init_id, workflow.py:329-340)."""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # the GPU box's CPU share for one GPU
    ns = args.cpu_sample_segments
    job = (ns, args.clones, args.max_cn, args.update_iters)
    one = _cpu_one(job + (0,))
    scale = float(args.segments) / float(ns)
    full_sample = min(200, args.segments // 10)

    def full_time(r):
        # linear part scaled by N_full / N_sample, sampled objectives by the ratio of the sample sizes
        return (r['total_s'] - r['sample_s']) * scale + r['sample_s'] * (float(full_sample) / max(1, r['sample_size']))
    t_one = full_time(one)
    multi = None
    if cores > 1:
        ctx = mp.get_context('fork')
        t0 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            rs = pool.map(_cpu_one, [job + (i,) for i in range(cores)])
        wall = time.perf_counter() - t0
        # all cores busy for the slowest worker's time; throughput = restarts / scaled time of the slowest
        multi = {'cores': cores, 'value': cores / max(full_time(r) for r in rs), 'wall_s': wall}
    out = {'value': 1.0 / t_one, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
           'sample': 'one EM iteration of one restart of oracle/remixt_oracle.c on %d segments (%d after breakend remap) x %d states: %.2f s, of which %.2f s '
                     'sampled M-step objectives (%d-segment samples); scaled to %d segments (linear part x %.0f, sampled part x %.1f)'
                     % (ns, one['N1'], one['S'], one['total_s'], one['sample_s'], one['sample_size'], args.segments, scale,
                        float(full_sample) / max(1, one['sample_size'])),
           'seconds_per_em_iteration_full_size': t_one}
    if multi:
        out['all_cores'] = {'value': multi['value'], 'unit': 'EM iterations/s', 'cores': multi['cores'],
                            'sample': 'the same sample, one restart per core on %d cores at once (%.1f s wall)' % (multi['cores'], multi['wall_s'])}
    if args.max_cn == 8 and args.cpu_sample_segments_355 > 0:
        # the 355-state configuration of the metric string (max_cn = 12, remixt/defaults.py:117): its own sample (the dense S x S work per
        # segment is 4.6 x the 165-state grid's), one core and one restart per core
        ns5 = args.cpu_sample_segments_355
        job5 = (ns5, args.clones, 12, args.update_iters)
        scale5 = float(args.segments) / float(ns5)
        ft5 = lambda r: (r['total_s'] - r['sample_s']) * scale5 + r['sample_s'] * (float(full_sample) / max(1, r['sample_size']))
        if cores > 1:
            ctx = mp.get_context('fork')
            t0 = time.perf_counter()
            with ctx.Pool(cores) as pool:
                rs5 = pool.map(_cpu_one, [job5 + (i,) for i in range(cores)])
            wall5 = time.perf_counter() - t0
        else:
            t0 = time.perf_counter(); rs5 = [_cpu_one(job5 + (0,))]; wall5 = time.perf_counter() - t0
        one5 = rs5[0]
        out['states_355'] = {'value': 1.0 / ft5(one5), 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
                             'sample': 'one EM iteration of one restart of oracle/remixt_oracle.c on %d segments (%d after breakend remap) x %d states: %.2f s (timed with '
                                       'one restart on each of %d cores at once, %.1f s wall); scaled to %d segments (linear part x %.0f)'
                                       % (ns5, one5['N1'], one5['S'], one5['total_s'], len(rs5), wall5, args.segments, scale5),
                             'seconds_per_em_iteration_full_size': ft5(one5),
                             'all_cores': {'value': len(rs5) / max(ft5(r) for r in rs5), 'unit': 'EM iterations/s', 'cores': len(rs5)}}
    print('CPU_LEG ' + json.dumps(out), flush=True)


def cpu_baseline(args):
    """Run the CPU leg as a child process (no GPU in it) and attach the port-vs-reference calibration measured in
    the build container by oracle/calibrate.py (the reference itself never travels to the GPU box)."""
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-leg', '--segments', str(args.segments), '--clones', str(args.clones),
           '--max-cn', str(args.max_cn), '--update-iters', str(args.update_iters), '--cpu-sample-segments', str(args.cpu_sample_segments),
           '--cpu-sample-segments-355', str(0 if args.no_extra_states else args.cpu_sample_segments_355)]
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in res.stdout.splitlines() if l.startswith('CPU_LEG ')]
        if not line:
            return {'error': 'cpu leg failed: ' + (res.stderr or res.stdout)[-400:]}
        out = json.loads(line[-1][len('CPU_LEG '):])
    except Exception as err:
        return {'error': 'cpu leg failed: %s' % err}
    try:
        cal = json.load(open(os.path.join(ROOT, 'profiles', 'cpu_calibration.json')))
        out['calibration'] = {'port_seconds_over_reference_seconds': cal['port_over_reference'], 'where': cal['where'],
                              'note': 'compiled reference kernel (remixt/bpmodel.pyx) vs this port on the same sample in the build container; '
                                      'reference-equivalent value = value x this ratio'}
        out['reference_equivalent_value'] = out['value'] * cal['port_over_reference']
        if isinstance(out.get('states_355'), dict):
            out['states_355']['reference_equivalent_value'] = out['states_355']['value'] * cal['port_over_reference']
            out['states_355']['calibration_note'] = 'the ratio port / reference was measured at 165 states (profiles/cpu_calibration.json); both are dominated by the same dense S x S loops'
    except Exception:
        out['calibration'] = None
    return out


# ---------------------------------------------------------------------------------------------------
def build_datasets(args):
    """The experiment(s) of the run: dataset 0 is the headline synthetic experiment; with --datasets 2 the second
    tumour sample has the same segments, adjacencies and breakpoints and its own read counts (another seed's
    counts on dataset 0's segmentation), SURVEY.md 8(d)."""
    from remixt_amd import synthetic
    e0 = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=23, seed=0)
    out = [e0]
    for d in range(1, args.datasets):
        out.append(synthetic.resample_counts(e0, seed=100 + d))
    return out


def main(argv=None, kernel_module=None, dist_backend='nccl', script=None):
    """The benchmark.  The keyword arguments exist for tests/bench_rehearsal.py ONLY (a world of gloo ranks on a
    machine without GPUs rehearsing the launch / shard / gather logic over a stand-in kernel module): this program
    itself always runs the HIP library over RCCL and has no switch, flag or environment variable that selects anything else."""
    args = parse(argv)
    if args.cpu_leg:
        return cpu_leg(args)
    if args.sub_run:
        return sub_run(args)
    if args.switch_interval_us > 0:
        sys.setswitchinterval(args.switch_interval_us * 1e-6)
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, script))
    world = int(env_world or '1')
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    # the CPU leg runs first, in a child process, before this process initialises the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    import torch
    import torch.distributed as dist
    backend = None
    if world > 1:
        backend = dist_backend          # RCCL over xGMI
        if backend != 'nccl':
            local_rank = local_rank % max(1, torch.cuda.device_count())
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    device = local_rank if world > 1 else 0
    if kernel_module is None:
        torch.cuda.set_device(device)

    if args.lib:
        from remixt_amd import _lib as _libmod
        _libmod.LIB_PATH = os.path.abspath(args.lib)
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups, DatasetGroups, gather_result_records
    if args.option and kernel_module is None:
        from remixt_amd import bpmodel
        for item in args.option:
            name, value = item.split('=')
            bpmodel.set_default_option(name, int(value))

    # ---- which restarts does this rank fit -----------------------------------------------------------
    strong = args.total_restarts > 0
    total = args.total_restarts if strong else args.restarts * world
    ND = max(1, args.datasets)
    datasets = build_datasets(args)
    per_ds = total // ND
    units = []                                                  # (dataset, restart id within the dataset), in global order
    for dsi in range(ND):
        units.extend((dsi, i) for i in range(per_ds))
    mine = units[rank::world]
    all_params = [synthetic.make_init_params(datasets[dsi], per_ds, args.max_cn, num_clones=args.clones) for dsi in range(ND)]
    sets = []
    for dsi in range(ND):
        ids = [i for (d_, i) in mine if d_ == dsi]
        if ids:
            sets.append((datasets[dsi], [all_params[dsi][i] for i in ids], [1000 + 7919 * dsi + i for i in ids]))
    R = len(mine)
    if R == 0:
        raise SystemExit('bench.py: rank %d has no restart (total %d over %d ranks)' % (rank, total, world))
    host_kw = dict((item.split('=')[0], int(item.split('=')[1])) for item in args.host_option)
    if len(sets) == 1:
        e, params, seeds = sets[0]
        rs = RestartGroups(e, params, args.max_cn, groups=args.groups, num_clones=args.clones, device=device, quiet=True, seeds=seeds,
                           kernel_module=kernel_module, **host_kw)
    else:
        rs = DatasetGroups([s[0] for s in sets], [s[1] for s in sets], args.max_cn, groups=(args.groups if args.groups_given else None), num_clones=args.clones, device=device,
                           quiet=True, seeds=[s[2] for s in sets], kernel_module=kernel_module, **host_kw)
    e = sets[0][0]
    on_gpu = kernel_module is None
    m0 = rs.models[0]
    N1, S = int(m0.model.num_segments), int(m0.model.num_cn_states)
    elbo0 = rs.calculate_elbo()
    for m, v in zip(rs.models, elbo0):
        m.prev_elbo = float(v)

    def steps(first, count):
        """`count` EM iterations of every restart on this GPU (restart groups free-run inside)."""
        if args.no_mstep:
            for _ in range(count):
                rs.variational_update(args.update_iters)
            return rs.calculate_elbo()
        return rs.run(count, first, args.update_iters)

    if args.warmup:
        steps(0, args.warmup)

    def fence():
        rs.synchronize()
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    if on_gpu:
        for b_ in rs.batches:
            b_.profile_reset(); b_.profile_enable(1 if args.profile_all else 2)
    fence()
    t0 = time.perf_counter()
    elbo = steps(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    prof = {}
    if on_gpu:
        for b_ in rs.batches:
            b_.profile_enable(0)
        prof = rs.profile()
    observed_world = 1
    gathered = None
    if world > 1:
        observed_world = dist.get_world_size()
        cdev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([float(R)], dtype=torch.float64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_fitted = int(round(float(cnt.item())))
        # final gather of the per-restart results (outside the timed region: it happens once per fit): the same
        # function and the same two records (float64 + int8) per restart as fit_restarts_distributed.  In the
        # weak / strong single-dataset modes unit j of rank g is restart g + j * world, i.e. shard_indices' order.
        res = rs.results()               # this rank's units in the order of `mine`: dataset after dataset, ascending restart id
        names = list(rs.models[0].likelihood_params)
        gathered = {'records': 0, 'seconds': 0.0, 'bytes_per_rank': 0}
        pos = 0
        for dsi in range(ND):
            ids = [i for (d_, i) in mine if d_ == dsi]
            tm = {}
            allres = gather_result_records(res[pos:pos + len(ids)], datasets[dsi], all_params[dsi], args.clones, names, device=device, timing=tm,
                                           local_ids=ids)
            pos += len(ids)
            gathered['records'] += int(sum(1 for r_ in allres.values() if np.isfinite(r_['stats']['elbo'])))
            gathered['seconds'] += tm['seconds']; gathered['bytes_per_rank'] += tm['bytes_per_rank']
            gathered['float_record_bytes'] = tm['float_record_bytes']; gathered['int8_record_bytes'] = tm['int8_record_bytes']
        gathered['note'] = 'per dataset two all_gathers (float64 record + int8 record per restart) through restarts.gather_result_records'
    else:
        total_fitted = R

    # N > 1: the weak-scaling line above keeps BASELINE configs[2]'s 16 restarts on every GPU, so that the driver's N = 1, 2, 4, 8
    # values are the same per-GPU workload; the north-star's multi-GPU target is configs[3], a FIXED 64-restart job cut over the
    # GPUs -- measured here as well, in the same process group, and reported beside it (`configs3_strong_64`)
    strong64 = None
    if world > 1 and not strong and ND == 1 and args.strong_total >= world:
        try:
            strong64 = strong_64_over_ranks(args, rs, device, rank, world, dist, torch, on_gpu, kernel_module, host_kw)
        except Exception as err:
            strong64 = {'error': str(err)}

    if rank == 0:
        total_ms = sum(v[0] for v in prof.values())
        # dominant kernel of the hot path = the data-parallel (segment x state) kernel with the largest
        # device time (the M-step's sampled-objective kernels are ~100 us host round trips over
        # <= 200 segments: latency, not a roofline subject; they are listed under "kernels")
        hot = [(k, prof[k]) for k in ALG_BYTES_PER_CELL if k in prof]
        dom = max(hot, key=lambda kv: kv[1][0]) if hot else (None, (0., 0))
        cells_total = float(N1) * S * R
        nbatches = max(1, len(rs.batches)) if on_gpu else 1
        cells_per_launch = cells_total / nbatches        # every group launches over its own restarts
        roof = roofline_object(dom, cells_per_launch, S, args, R // nbatches)
        others = []
        for name in ('k_framelogprob', 'k_marginals<true>'):
            if name in prof and prof[name][1]:
                avg = prof[name][0] / prof[name][1]
                gbs = ALG_BYTES_PER_CELL[name] * cells_per_launch / (avg * 1e-3) / 1e9
                others.append({'kernel': name, 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                               'avg_launch_ms': avg, 'launches': prof[name][1]})
        upd = sum(prof.get(k, (0., 0))[0] for k in SWEEP_KERNELS)
        nupd = prof.get('k_fb', (0., 1))[1]
        sweep_ms = upd / max(nupd, 1)

        def hbm_frac(bytes_per_cell):
            return (bytes_per_cell * cells_per_launch / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sweep_ms else None
        if ND > 1:
            wl = ('BASELINE configs[4]: %d tumour samples on one segmentation (%d segments, %d after breakend remap) and breakpoint set, fitted independently, '
                  '%d clones, max_cn=%d (%d states), %d restarts in total' % (ND, args.segments, N1, args.clones, args.max_cn, S, total))
        elif strong:
            wl = ('BASELINE configs[3]: %d segments (%d after breakend remap), %d clones, max_cn=%d (%d states), %d restarts in total sharded over %d GPU(s)'
                  % (args.segments, N1, args.clones, args.max_cn, S, total, world))
        else:
            wl = ('BASELINE configs[2]: %d segments (%d after breakend remap), %d clones, max_cn=%d (%d states), %d restarts/GPU'
                  % (args.segments, N1, args.clones, args.max_cn, S, args.restarts))
        wl += ', %d variational sweeps + M-steps per EM iteration' % args.update_iters
        line = {
            'metric': 'EM iterations/sec (%dk seg x %d states)' % (args.segments // 1000, S),
            'value': (total_fitted * args.steps) / dt, 'unit': 'EM iterations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': wl, 'segments': args.segments, 'states': S, 'restarts_total': total_fitted, 'restarts_this_rank': R,
                       'datasets': ND, 'restart_groups': nbatches, 'mstep': not args.no_mstep,
                       'world_size_observed': observed_world, 'backend': backend},
            'seg_state_cells_per_s': float(N1) * S * total_fitted * args.update_iters * args.steps / dt,
            'roofline': roof,
            'roofline_other': others,
            'variational_sweep': {'device_ms_per_sweep_all_restarts': sweep_ms,
                                  'hbm_frac_88B_per_cell_survey_model': hbm_frac(SURVEY_BYTES_PER_CELL),
                                  'hbm_frac_168B_per_cell_this_design': hbm_frac(168.0)},
            'device_ms_total': total_ms,
            'kernels': dict((k, {'ms': round(v[0], 3), 'n': v[1]}) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])),
            'elbo_best': float(np.nanmax(elbo)),
        }
        if gathered:
            line['final_gather'] = gathered
        if strong64 is not None:
            line['configs3_strong_64'] = strong64
        single = world == 1 and on_gpu and ND == 1 and not strong and not args.no_mstep
        if single and not args.no_fit_from_init:
            try:
                line['fit_from_init'] = fit_from_init(args, rs, device)
            except Exception as err:
                line['fit_from_init'] = _error_object(err)
        if single and not args.no_extra_states and args.groups != 1:
            # the forward-backward kernel at its other launch shape: all 16 restarts of the GPU in one launch (one restart group)
            try:
                line['roofline_one_group'] = one_group_roofline(args, rs, device)
            except Exception as err:
                line['roofline_one_group'] = _error_object(err)
        if single and not args.no_extra_states and args.max_cn == 8:
            # "also report S = 355 at max_cn = 12" (the reference's default max_copy_number):
            # same segments / restarts / step definition, reported next to the headline configuration
            try:
                line['states_355'] = extra_states(args, rs, device)
            except Exception as err:      # never let the extra measurement hide the headline number
                line['states_355'] = _error_object(err)
        if single and not args.no_extra_states and args.max_cn == 8 and args.restarts == 16:
            try:
                line['strong_scaling_proxy'] = strong_scaling_proxy(args, rs, device)
            except Exception as err:
                line['strong_scaling_proxy'] = _error_object(err)
        if single and not args.no_extra_states and args.max_cn == 8:
            try:
                line['unequal_chains'] = unequal_chains(args, rs, device, line['value'])
            except Exception as err:
                line['unequal_chains'] = _error_object(err)
        if cpu is not None:
            cpu355 = cpu.pop('states_355', None) if isinstance(cpu, dict) else None
            line['cpu_baseline'] = cpu
            if cpu355 is not None and isinstance(line.get('states_355'), dict):
                line['states_355']['cpu_baseline'] = cpu355
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline_object(dom, cells_per_launch, S, args, restarts_per_launch, traffic_file='traffic_r05.json'):
    if dom[0] is None:
        return None
    name, (ms, n) = dom
    avg_ms = ms / max(n, 1)
    alg = ALG_BYTES_PER_CELL.get(name)
    hbm_gbs = alg * cells_per_launch / (avg_ms * 1e-3) / 1e9
    traffic = None
    for tf in (traffic_file, 'traffic_r03.json', 'traffic_r02.json', 'traffic_r01.json'):
        try:   # PMC traffic of the same kernel on the same launch shape, measured offline (profiles/)
            tj = json.load(open(os.path.join(ROOT, 'profiles', tf)))
            w = tj['workload']
            if (w['segments'], w['states'], w['restarts']) == (args.segments, S, restarts_per_launch):
                traffic = tj['kernels'][name]['hbm_bytes_per_launch']
                break
        except Exception:
            continue
    fl = alg_flops_per_cell(name, S)
    if fl is not None:
        # the forward-backward recursion: S^2 FP64 FMAs per segment, direction and restart
        achieved = fl * cells_per_launch / (avg_ms * 1e-3) / 1e12
        return {'bound': 'fp64', 'kernel': name, 'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': 'profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel at this launch shape (tools/collect_profiles.sh), not collected in this run' % traffic_file,
                'avg_launch_ms': avg_ms, 'launches': n,
                'alg_flops_per_launch': fl * cells_per_launch, 'alg_bytes_per_launch': alg * cells_per_launch,
                'hbm_gbs_at_alg_bytes': hbm_gbs, 'hbm_frac_at_alg_bytes': hbm_gbs / HBM_PEAK_GBS,
                'note': 'FP64 FMA bound (the contract\'s "mfma" class: peak = dense FP64 rate, the same for vector and matrix FP64 on gfx950); '
                        'restart groups launch concurrently on separate streams, so avg_launch_ms includes time shared with the other group\'s kernels'}
    return {'bound': 'hbm', 'kernel': name, 'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': hbm_gbs / HBM_PEAK_GBS, 'traffic': traffic, 'avg_launch_ms': avg_ms, 'launches': n,
            'alg_bytes_per_launch': alg * cells_per_launch}


def _release(rs_main):
    import gc
    if rs_main is None:
        return
    for s_ in rs_main.sets:          # release the batches' device memory, streams and events NOW (the next measurement's streams are to find the hardware queues free)
        if hasattr(s_, 'close'):
            s_.close()
        s_.batch = None
        for m in s_.models:
            m.model = None
    gc.collect()


def fit_from_init(args, rs_main, device):
    """A real fit, wall clock: construct the models (remap, uploads), 5 EM iterations FROM THE INITIALISATION
    (reference defaults.py:154), Viterbi decode and result records, for the GPU's 16 restarts."""
    import torch
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    _release(rs_main)
    R = args.restarts
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=23, seed=0)
    params = synthetic.make_init_params(e, R, args.max_cn, num_clones=args.clones)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_kw = dict((item.split('=')[0], int(item.split('=')[1])) for item in args.host_option)
    rs = RestartGroups(e, params, args.max_cn, groups=args.groups, num_clones=args.clones, device=device, quiet=True, seeds=[1000 + i for i in range(R)], **host_kw)
    rs.synchronize()
    t1 = time.perf_counter()
    elbo = rs.fit(5, args.update_iters)
    rs.synchronize()
    t2 = time.perf_counter()
    res = rs.results()
    t3 = time.perf_counter()
    out = {'restarts': R, 'em_iterations': 5, 'construct_s': t1 - t0, 'em_s': t2 - t1, 'decode_and_results_s': t3 - t2, 'total_s': t3 - t0,
           'em_iterations_per_s_from_init': R * 5 / (t2 - t1), 'em_iterations_per_s_whole_fit': R * 5 / (t3 - t0),
           'elbo_best': float(np.nanmax(elbo)), 'failed_restarts': int(sum(1 for r in res if r['stats'].get('error_message')))}
    _release(rs)
    return out


def one_group_roofline(args, rs_main, device, traffic_file='traffic_r05_16.json', nsteps=6):
    """The headline workload with ALL restarts of the GPU in one restart group: what the forward-backward kernel reaches when a
    launch carries 16 restarts instead of 8 (its duration is the latency of the chain of steps, not a function of the restart
    count), and what the step then costs (the M-steps are no longer hidden behind another group's sweeps)."""
    import torch
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    _release(rs_main)
    R = args.restarts
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=23, seed=0)
    params = synthetic.make_init_params(e, R, args.max_cn, num_clones=args.clones)
    rs = RestartGroups(e, params, args.max_cn, groups=1, num_clones=args.clones, device=device, quiet=True, seeds=[1000 + i for i in range(R)])
    b = rs.batches[0]
    S, N1 = b.num_cn_states, b.num_segments
    for m, v in zip(rs.models, rs.calculate_elbo()):
        m.prev_elbo = float(v)
    rs.run(2, 0, args.update_iters)
    rs.synchronize(); torch.cuda.synchronize()
    b.profile_reset(); b.profile_enable(2)
    t0 = time.perf_counter()
    rs.run(nsteps, 2, args.update_iters)
    rs.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b.profile_enable(0)
    prof = rs.profile()
    out = roofline_object(('k_fb', prof['k_fb']), float(N1) * S * R, S, args, R, traffic_file=traffic_file)
    out['restart_groups'] = 1
    out['restarts_per_launch'] = R
    out['em_iterations_per_s_with_one_group'] = R * nsteps / dt
    out['note'] = 'one restart group: every forward-backward launch carries all %d restarts of the GPU' % R
    _release(rs)
    return out


def _timed_run(args, device, restarts, groups, max_cn, nsteps, warm, seeds_base=1000, chain_fractions=None):
    """Build `restarts` restarts of the headline experiment (at max_cn) in `groups` restart groups, run `warm` untimed and
    `nsteps` timed EM iterations with HIP-event kernel times: (rs, S, N1, seconds, elbo, profile)."""
    import torch
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=max_cn, num_chains=23, seed=0, chain_fractions=chain_fractions)
    params = synthetic.make_init_params(e, restarts, max_cn, num_clones=args.clones)
    rs = RestartGroups(e, params, max_cn, groups=groups, num_clones=args.clones, device=device, quiet=True, seeds=[seeds_base + i for i in range(restarts)])
    b = rs.batches[0]
    S, N1 = b.num_cn_states, b.num_segments
    for m, v in zip(rs.models, rs.calculate_elbo()):
        m.prev_elbo = float(v)
    rs.run(warm, 0, args.update_iters)
    rs.synchronize(); torch.cuda.synchronize()
    for b_ in rs.batches:
        b_.profile_reset(); b_.profile_enable(2)
    t0 = time.perf_counter()
    elbo = rs.run(nsteps, warm, args.update_iters)
    rs.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for b_ in rs.batches:
        b_.profile_enable(0)
    return rs, S, N1, dt, elbo, rs.profile()


class _RunInfo(object):
    """What the additional measurements read off the restart groups of a timed run, when the run was made in a child process."""
    class _B(object):
        def __init__(self, info):
            self._info = info

        def info(self, k):
            return self._info[str(k)]

    def __init__(self, paced, info):
        self.paced, self.batches, self.sets = paced, [self._B(info)], []


def _timed_run_isolated(args, device, restarts, groups, max_cn, nsteps, warm, unequal=False):
    """_timed_run in a process of its own (started here as a child; this process has released its batches and idles meanwhile).
    The restart groups of a process's FIRST measurement find the hardware queues unused and every stream gets its own; groups
    built later in the same process shared queues now and then even with their predecessors destroyed (states_355 110 instead of 150 EM
    it/s in one run of five).  A fit is a process of its own in production (the reference's `fit` task): that is what is measured.
    A child that fails raises SubRunFailed: the caller reports {'error', 'child_rc', 'child_stderr_tail'} in the measurement's object."""
    cmd = [sys.executable, os.path.abspath(__file__), '--sub-run', '%d,%d,%d,%d,%d,%d' % (restarts, groups, max_cn, nsteps, warm, 1 if unequal else 0),
           '--segments', str(args.segments), '--clones', str(args.clones), '--update-iters', str(args.update_iters)]
    for item in args.option:
        cmd += ['--option', item]
    for item in args.host_option:
        cmd += ['--host-option', item]
    if args.lib:
        cmd += ['--lib', args.lib]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired as err:
        # (subprocess.run has killed the child in the middle of its GPU work: reported, never measured again in this process -- a hang is a finding)
        raise SubRunFailed('additional measurement: child process killed after %d s' % 600, None, _tail(err.stderr))
    line = [l for l in res.stdout.splitlines() if l.startswith('SUB_RUN ')]
    if res.returncode != 0 or not line:
        raise SubRunFailed('additional measurement: child process failed (rc %s)' % res.returncode, res.returncode, _tail(res.stderr or res.stdout))
    j = json.loads(line[-1][len('SUB_RUN '):])
    prof = dict((k, (v[0], v[1])) for k, v in j['prof'].items())
    return _RunInfo(j['paced'], j['info']), j['S'], j['N1'], j['dt'], np.array([j['elbo_best']]), prof


class SubRunFailed(RuntimeError):
    """A --sub-run child that exited non-zero, printed no result or ran into the time limit.  There is NO in-process fallback:
    the in-process regime differs and a GPU fault or hang in an additional measurement must be visible in the bench line."""

    def __init__(self, message, rc, stderr_tail):
        RuntimeError.__init__(self, message)
        self.rc, self.stderr_tail = rc, stderr_tail


def _tail(text, n=600):
    if text is None:
        return ''
    if isinstance(text, bytes):
        text = text.decode('utf-8', 'replace')
    return text[-n:]


def _error_object(err):
    """What an additional measurement's object holds when it failed."""
    out = {'error': str(err)}
    if isinstance(err, SubRunFailed):
        out['measured_in'] = 'child process'
        out['child_rc'] = err.rc
        out['child_stderr_tail'] = err.stderr_tail
    return out


def sub_run(args):
    """The child of _timed_run_isolated: one timed run, its numbers as one line."""
    import torch
    restarts, groups, max_cn, nsteps, warm, unequal = [int(v) for v in args.sub_run.split(',')]
    torch.cuda.set_device(0)
    if args.lib:
        from remixt_amd import _lib as _libmod
        _libmod.LIB_PATH = os.path.abspath(args.lib)
    from remixt_amd import synthetic
    if args.option:
        from remixt_amd import bpmodel
        for item in args.option:
            name, value = item.split('=')
            bpmodel.set_default_option(name, int(value))
    rs, S, N1, dt, elbo, prof = _timed_run(args, 0, restarts, groups, max_cn, nsteps, warm, chain_fractions=synthetic.HUMAN_CHROMOSOME_MB if unequal else None)
    b = rs.batches[0]
    out = {'S': S, 'N1': N1, 'dt': dt, 'elbo_best': float(np.nanmax(elbo)), 'paced': bool(getattr(rs, 'paced', False)),
           'prof': dict((k, [v[0], v[1]]) for k, v in prof.items()), 'info': dict((str(k), b.info(k)) for k in (12, 13, 14, 15))}
    _release(rs)
    print('SUB_RUN ' + json.dumps(out), flush=True)
    return 0


def extra_states(args, rs_main, device):
    """EM iterations/s at max_cn = 12 (355 states: the reference's default max_copy_number, defaults.py:117, and the "~400 states"
    BASELINE.json's metric string is quoted on), everything else as the headline workload; 20 timed steps like the headline."""
    _release(rs_main)
    max_cn, R, nsteps = 12, args.restarts, 20
    # two restart groups, paced (RestartGroups paced='auto' above 200 states): 144 EM iterations/s against 134 for one group of 16 and 118
    # for two free-running groups (tools/s355_groups.sh)
    G355 = 2
    rs, S, N1, dt, elbo, prof = _timed_run_isolated(args, device, R, G355, max_cn, nsteps, 2)
    hot = [(k, prof[k]) for k in ALG_BYTES_PER_CELL if k in prof]
    dom = max(hot, key=lambda kv: kv[1][0]) if hot else (None, (0., 0))
    a355 = argparse.Namespace(**vars(args)); a355.max_cn = max_cn
    out = {'metric': 'EM iterations/sec (%dk seg x %d states)' % (args.segments // 1000, S), 'states': S, 'max_cn': max_cn, 'restart_groups': G355, 'paced_groups': bool(getattr(rs, 'paced', False)),
           'value': R * nsteps / dt, 'unit': 'EM iterations/s', 'ms_per_step': dt / nsteps * 1e3, 'steps': nsteps, 'warmup': 2,
           'seg_state_cells_per_s': float(N1) * S * R * args.update_iters * nsteps / dt, 'elbo_best': float(np.nanmax(elbo)),
           'forward_backward_kernel': {1: 'k_fbm', 2: 'k_fbv', 3: 'k_fbk', 4: 'k_fbq', 0: 'k_fb<0>'}.get(rs.batches[0].info(12)),
           'roofline': roofline_object(dom, float(N1) * S * R / G355, S, a355, R // G355, traffic_file='traffic_r05_s355_8.json'),
           'kernels': dict((k, {'ms': round(v[0], 3), 'n': v[1]}) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])),
           'measured_in': 'child process (rc 0)'}
    _release(rs)
    # the forward-backward kernel at its other launch shape here too: all 16 restarts in one launch (one restart group)
    try:
        out['roofline_one_group'] = one_group_roofline(a355, None, device, traffic_file='traffic_r05_s355.json', nsteps=3)
    except Exception as err:
        out['roofline_one_group'] = _error_object(err)
    return out


def unequal_chains(args, rs_main, device, headline_value):
    """The headline workload on chains in the proportions of a genome's chromosomes (GRCh37 lengths of 1 .. 22, X: 5 : 1) instead of 23 equal
    chains (the reference cuts chains at gaps, remixt/analysis/experiment.py:124-143): same segments, states, restarts,
    restart groups and step definition.  A forward-backward launch lasts as long as its slowest chain; k_fbm gives a long chain fewer
    restarts per workgroup (shorter steps) than a short one (rmx_api.hip fb_items_for)."""
    from remixt_amd import synthetic
    _release(rs_main)
    nsteps = 10
    rs, S, N1, dt, elbo, prof = _timed_run_isolated(args, device, args.restarts, args.groups, args.max_cn, nsteps, 2, unequal=True)
    fb = prof.get('k_fb', (0., 1))
    b = rs.batches[0]
    lens = sorted(int(round(f / float(sum(synthetic.HUMAN_CHROMOSOME_MB)) * args.segments)) for f in synthetic.HUMAN_CHROMOSOME_MB)
    out = {'workload': '%d segments in 23 chains proportional to the human chromosomes (longest %d, shortest %d segments; the headline has 23 x %d), %d states, %d restarts, %d restart groups'
                       % (args.segments, lens[-1], lens[0], args.segments // 23, S, args.restarts, args.groups),
           'value': args.restarts * nsteps / dt, 'unit': 'EM iterations/s', 'ms_per_step': dt / nsteps * 1e3, 'steps': nsteps, 'warmup': 2,
           'fb_avg_launch_ms': fb[0] / max(fb[1], 1), 'fb_restarts_per_workgroup': [b.info(13), b.info(15)],
           'ratio_to_headline': (args.restarts * nsteps / dt) / headline_value if headline_value else None,
           'elbo_best': float(np.nanmax(elbo)), 'measured_in': 'child process (rc 0)'}
    _release(rs)
    return out


def strong_64_over_ranks(args, rs_main, device, rank, world, dist, torch, on_gpu, kernel_module, host_kw):
    """BASELINE configs[3] on this process group: 64 restarts in total, restart i on rank i mod world, 8 timed EM iterations
    between barriers, MAX over ranks."""
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    _release(rs_main)
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=23, seed=0)
    T = args.strong_total
    params = synthetic.make_init_params(e, T, args.max_cn, num_clones=args.clones)
    ids = list(range(rank, T, world))
    groups = args.groups if len(ids) <= 16 else max(args.groups, (len(ids) + 15) // 16)
    rs = RestartGroups(e, [params[i] for i in ids], args.max_cn, groups=groups, num_clones=args.clones, device=device, quiet=True,
                       seeds=[1000 + i for i in ids], kernel_module=kernel_module, **host_kw)
    for m, v in zip(rs.models, rs.calculate_elbo()):
        m.prev_elbo = float(v)
    nsteps = 8 if kernel_module is None else 1
    warm = 2 if kernel_module is None else 0
    if warm:
        rs.run(warm, 0, args.update_iters)

    def fence():
        rs.synchronize()
        if on_gpu:
            torch.cuda.synchronize()
        dist.barrier()
    fence()
    t0 = time.perf_counter()
    rs.run(nsteps, warm, args.update_iters)
    fence()
    dt = time.perf_counter() - t0
    cdev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    _release(rs)
    return {'workload': 'BASELINE configs[3]: %d restarts in total over %d GPUs (%d on rank 0, %d restart groups each)' % (T, world, len(ids), groups),
            'scaling': 'strong', 'value': T * nsteps / dt, 'unit': 'EM iterations/s', 'ms_per_step': dt / nsteps * 1e3, 'steps': nsteps, 'warmup': warm,
            'restarts_total': T, 'restarts_this_rank': len(ids)}


def strong_scaling_proxy(args, rs_main, device):
    """BASELINE configs[3] without an 8-GPU node: the 64-restart job on THIS GPU, and one rank's share of it
    (8 restarts) on this GPU.  Restarts are independent and nothing is exchanged during EM, so 8 GPUs with 8 restarts each run at
    8 x the one-share rate: predicted speed-up of 8 GPUs over 1 on the fixed job = 8 x it/s(8) / it/s(64)."""
    _release(rs_main)
    nsteps, nsteps8 = 8, 20      # (the share's step is 27 ms: with 8 timed steps its rate scattered by 10 % and the prediction with it)
    rs, S, N1, dt64, _, prof64 = _timed_run_isolated(args, device, 64, 4, args.max_cn, nsteps, 2)
    fb64 = prof64.get('k_fb', (0., 1))
    _release(rs)
    rs, S, N1, dt8, _, prof8 = _timed_run_isolated(args, device, 8, args.groups, args.max_cn, nsteps8, 3)
    fb8 = prof8.get('k_fb', (0., 1))
    _release(rs)
    its64, its8 = 64 * nsteps / dt64, 8 * nsteps8 / dt8
    return {'workload': 'BASELINE configs[3]: 64 restarts, %d segments x %d states' % (args.segments, S),
            'one_gpu_64_restarts': {'value': its64, 'unit': 'EM iterations/s', 'ms_per_step': dt64 / nsteps * 1e3, 'restart_groups': 4,
                                    'fb_avg_launch_ms': fb64[0] / max(fb64[1], 1), 'fb_restarts_per_launch': 16},
            'one_rank_share_8_restarts': {'value': its8, 'unit': 'EM iterations/s', 'ms_per_step': dt8 / nsteps8 * 1e3, 'restart_groups': args.groups,
                                          'fb_avg_launch_ms': fb8[0] / max(fb8[1], 1), 'fb_restarts_per_launch': 8 // max(1, args.groups)},
            'predicted_speedup_8_gpus_over_1': 8. * its8 / its64, 'target': 6.0, 'measured_in': 'two child processes (rc 0, 0)',
            'note': 'a forward-backward launch is a chain of 2 173 dependent steps per chromosome: with 16 restarts per launch a workgroup carries four restarts on the '
                    'matrix cores (2.8 ms), a rank\'s 4-restart launches carry one per workgroup on the vector ALU (1.5 ms) -- shorter, but not four times shorter: one '
                    'GPU amortises the chain over 64 restarts, a rank holding 8 cannot'}


if __name__ == '__main__':
    main()
