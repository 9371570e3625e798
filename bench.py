#!/usr/bin/env python3
"""Benchmark of the ReMixT variational-EM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = ONE EM iteration (reference remixt/cn_model.py:409-418: 5
variational sweeps + the h M-step + the likelihood-parameter M-steps + the ELBO)
for EVERY restart resident on the GPU.  Workload at N=1 = BASELINE.json
configs[2]: 50k segments, 3 clones (normal + 2 tumour), max_cn = 8 (165 states),
16 (h, divergence-weight) restarts.  With N > 1 every rank fits its own 16
restarts (weak scaling; restarts are independent, reference
remixt/workflow.py:329-340) and the per-restart results are all-gathered once at
the end over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(dominant kernel, HIP-event timed on the batch stream) and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # dense FP64 rate of MI355X (vector = matrix on gfx950): 256 CUs x 4 SIMDs x 16 FMA lanes x 2.4 GHz
# algorithmic HBM bytes per (segment,state) cell per variational update, split by kernel
ALG_BYTES_PER_CELL = {
    'k_framelogprob': 64.0,      # read the 6 cached likelihood components, write f and exp(f - rowmax)
    'k_fb': 32.0,                # read exp(f - rowmax) (fwd) + write alpha + read it again (bwd) + write beta
    'k_marginals<true>': 72.0,   # read alpha, beta and the 6 cached components, write the posterior
}


def alg_flops_per_cell(name, S):
    """FP64 flops per (segment, state) cell: the forward and the backward recursion are S x S
    matrix-vector products per segment, i.e. S FMAs per cell and direction."""
    return 4.0 * S if name == 'k_fb' else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--segments', type=int, default=50000)
    ap.add_argument('--clones', type=int, default=3)
    ap.add_argument('--max-cn', type=int, default=8)
    ap.add_argument('--restarts', type=int, default=16, help='restarts per GPU')
    ap.add_argument('--update-iters', type=int, default=5)
    ap.add_argument('--groups', type=int, default=2, help='restart groups per GPU (own stream + host thread each; results do not depend on it)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-segments', type=int, default=800)
    ap.add_argument('--profile-all', action='store_true', help='HIP-event every kernel (default: only the variational-sweep kernels; the M-step objective kernels are ~3000 tiny launches per step)')
    ap.add_argument('--no-extra-states', action='store_true', help='skip the additional 355-state (max_cn = 12) measurement at N = 1')
    ap.add_argument('--no-mstep', action='store_true', help='diagnostic only: variational sweeps without M-steps (NOT the reported metric)')
    return ap.parse_args()


def cpu_baseline(args, cores_note=1):
    """Reference CPU path timed on this box's host cores on a bounded sample.

    One EM iteration of one restart on `cpu_sample_segments` segments with the same
    state grid; every loop of the reference is linear in N, so the per-EM-iteration
    time is scaled by N_full / N_sample.  kind = "reference" when the compiled
    reference kernel (oracle/_ref, built from /root/reference/remixt/bpmodel.pyx)
    travelled with the repo, else "port" (oracle/remixt_oracle.c)."""
    from remixt_amd import synthetic
    from remixt_amd.cn_model import BreakpointModel
    from oracle import refload
    kind = 'port'
    kern = None
    if refload.have_ref_binary():
        try:
            kern = refload.load_ref_bpmodel(); kind = 'reference'
        except Exception:
            kern = None
    if kern is None:
        from oracle import oracle as kern
        kern.build()
    ns = args.cpu_sample_segments
    e = synthetic.make_experiment(ns, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=4, seed=123)
    p = synthetic.make_init_params(e, 1, args.max_cn, num_clones=args.clones)[0]
    m = BreakpointModel(e.x, e.l, e.adjacencies, e.breakpoints, max_copy_number=args.max_cn,
                        divergence_weight=p['divergence_weight'], max_depth=p['max_depth'], kernel_module=kern, quiet=True)
    m.num_update_iter = args.update_iters
    m._attach_model(m._build_model(synthetic.h_init_from_params(p, args.clones)))
    m.prev_elbo = m.model.calculate_elbo()
    np.random.seed(0)
    t0 = time.time()
    m.em_iteration(0)
    dt = time.time() - t0
    scale = float(args.segments) / float(ns)
    return {
        'value': 1.0 / (dt * scale), 'unit': 'EM iterations/s', 'cores': 1, 'kind': kind,
        'sample': 'one EM iteration of one restart on %d segments x %d states (%.1f s), scaled linearly to %d segments'
                  % (ns, m.model.num_cn_states, dt, args.segments),
    }


def extra_states(args, rs_main, device):
    """EM iterations/s at max_cn = 12 (355 states), everything else as the headline workload."""
    import gc
    import torch
    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups
    for s_ in rs_main.sets:          # release the headline batches' device memory
        s_.batch = None
        for m in s_.models:
            m.model = None
    gc.collect()
    max_cn, R = 12, args.restarts
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=max_cn, num_chains=23, seed=0)
    params = synthetic.make_init_params(e, R, max_cn, num_clones=args.clones)
    # one group: at 355 states the forward-backward launch dominates and 4 restarts per workgroup beat the overlap of two groups
    rs = RestartGroups(e, params, max_cn, groups=1, num_clones=args.clones, device=device, quiet=True, seeds=[1000 + i for i in range(R)])
    S = rs.batches[0].num_cn_states
    for m, v in zip(rs.models, rs.calculate_elbo()):
        m.prev_elbo = float(v)
    rs.run(1, 0, args.update_iters)
    rs.synchronize(); torch.cuda.synchronize()
    nsteps = 2
    t0 = time.perf_counter()
    elbo = rs.run(nsteps, 1, args.update_iters)
    rs.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'states': S, 'max_cn': max_cn, 'restart_groups': 1, 'value': R * nsteps / dt, 'unit': 'EM iterations/s', 'ms_per_step': dt / nsteps * 1e3, 'steps': nsteps,
            'seg_state_cells_per_s': float(rs.batches[0].num_segments) * S * R * args.update_iters * nsteps / dt, 'elbo_best': float(np.max(elbo))}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # RCCL over xGMI; BENCH_DIST_BACKEND=gloo rehearses the multi-rank control flow where ranks share a GPU
        backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
        if backend != 'nccl':
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    device = local_rank if world > 1 else 0
    torch.cuda.set_device(device)

    from remixt_amd import synthetic
    from remixt_amd.restarts import RestartGroups, _pack

    R = args.restarts
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=23, seed=0)
    all_params = synthetic.make_init_params(e, R * world, args.max_cn, num_clones=args.clones)
    mine = all_params[rank::world]
    rs = RestartGroups(e, mine, args.max_cn, groups=args.groups, num_clones=args.clones, device=device, quiet=True,
                       seeds=[1000 + rank + world * i for i in range(R)])
    b = rs.batches[0]
    N1, S = b.num_segments, b.num_cn_states
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
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for b_ in rs.batches:
        b_.profile_reset(); b_.profile_enable(1 if args.profile_all else 2)
    fence()
    t0 = time.perf_counter()
    elbo = steps(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    for b_ in rs.batches:
        b_.profile_enable(0)
    prof = rs.profile()
    if world > 1:
        cdev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # final gather of the per-restart results (outside the timed region: it happens once per fit)
        res = rs.results()
        names = list(rs.models[0].likelihood_params)
        ids = list(e.breakpoints.keys())
        packs = [_pack(r_, len(e.x), args.clones, len(ids), len(names), ids, names) for r_ in res]
        ft = torch.from_numpy(np.stack([p_[0] for p_ in packs])).to(cdev)
        out = [torch.empty_like(ft) for _ in range(world)]
        dist.all_gather(out, ft)

    if rank == 0:
        total_ms = sum(v[0] for v in prof.values())
        # dominant kernel of the hot path = the data-parallel (segment x state) kernel with the largest
        # device time (the M-step's sampled-objective kernels are ~100 us host round trips over
        # <= 200 segments: latency, not a roofline subject; they are listed under "kernels")
        hot = [(k, prof[k]) for k in ALG_BYTES_PER_CELL if k in prof]
        dom = max(hot, key=lambda kv: kv[1][0]) if hot else (None, (0., 0))
        cells_total = float(N1) * S * R
        cells_per_launch = cells_total / len(rs.sets)     # every group launches over its own restarts
        roof = None
        if dom[0] is not None:
            name, (ms, n) = dom
            avg_ms = ms / max(n, 1)
            alg = ALG_BYTES_PER_CELL.get(name)
            hbm_gbs = alg * cells_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            try:   # PMC traffic of the same kernel on the same launch shape, measured offline (profiles/)
                tj = json.load(open(os.path.join(ROOT, 'profiles', 'traffic_r01.json')))
                w = tj['workload']
                if (w['segments'], w['states'], w['restarts']) == (args.segments, S, R // len(rs.sets)):
                    traffic = tj['kernels'][name]['hbm_bytes_per_launch']
            except Exception:
                traffic = None
            fl = alg_flops_per_cell(name, S)
            if fl is not None:
                # the forward-backward recursion: S^2 FP64 FMAs per segment, direction and restart on vector
                # v_fmac_f64 (no MFMA: matrix x vector with a sequential dependency between segments)
                achieved = fl * cells_per_launch / (avg_ms * 1e-3) / 1e12
                roof = {'bound': 'mfma', 'kernel': name, 'achieved': achieved, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': achieved / FP64_PEAK_TFLOPS, 'traffic': traffic, 'avg_launch_ms': avg_ms, 'launches': n,
                        'alg_flops_per_launch': fl * cells_per_launch, 'alg_bytes_per_launch': alg * cells_per_launch,
                        'hbm_gbs_at_alg_bytes': hbm_gbs,
                        'note': 'FP64 vector FMA bound (v_fmac_f64 with DPP row broadcast), peak = dense FP64 rate; '
                                'restart groups launch concurrently on separate streams, so avg_launch_ms includes time shared with the other group\'s kernels'}
            else:
                roof = {'bound': 'hbm', 'kernel': name, 'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': hbm_gbs / HBM_PEAK_GBS, 'traffic': traffic, 'avg_launch_ms': avg_ms, 'launches': n,
                        'alg_bytes_per_launch': alg * cells_per_launch}
        # the HBM-streaming passes of the sweep, same construction (algorithmic bytes / HIP-event launch time)
        others = []
        for name in ('k_framelogprob', 'k_marginals<true>'):
            if name in prof and prof[name][1]:
                avg = prof[name][0] / prof[name][1]
                gbs = ALG_BYTES_PER_CELL[name] * cells_per_launch / (avg * 1e-3) / 1e9
                others.append({'kernel': name, 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                               'avg_launch_ms': avg, 'launches': prof[name][1]})
        # whole variational update (all kernels of one sweep) against the 88 B/cell model
        upd = sum(prof.get(k, (0., 0))[0] for k in ('k_framelogprob', 'k_fb', 'k_marginals<true>', 'k_pairwise', 'k_brk_update',
                                                       'k_brk_lut', 'k_update_outlier_total', 'k_update_outlier_allele', 'k_update_allele_swap'))
        nupd = prof.get('k_fb', (0., 1))[1]
        sweep_ms = upd / max(nupd, 1)
        line = {
            'metric': 'EM iterations/sec (%dk seg x %d states x %d restarts/GPU)' % (args.segments // 1000, S, R),
            'value': (R * world * args.steps) / dt, 'unit': 'EM iterations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[2]: %d segments (%d after breakend remap), %d clones, max_cn=%d (%d states), %d restarts/GPU, %d variational sweeps + M-steps per EM iteration'
                                   % (args.segments, N1, args.clones, args.max_cn, S, R, args.update_iters),
                       'segments': args.segments, 'states': S, 'restarts_per_gpu': R, 'restart_groups': len(rs.sets), 'mstep': not args.no_mstep},
            'seg_state_cells_per_s': cells_total * world * args.update_iters * args.steps / dt,
            'roofline': roof,
            'roofline_other': others,
            'variational_sweep': {'device_ms_per_sweep_all_restarts': sweep_ms,
                                  'hbm_frac_168B_per_cell': (168.0 * cells_per_launch / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sweep_ms else None},
            'device_ms_total': total_ms,
            'kernels': dict((k, {'ms': round(v[0], 3), 'n': v[1]}) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])),
            'elbo_best': float(np.max(elbo)),
        }
        if world == 1 and not args.no_extra_states and args.max_cn == 8 and not args.no_mstep:
            # "also report S = 355 at max_cn = 12" (the reference's default max_copy_number):
            # same segments / restarts / step definition, reported next to the headline configuration
            try:
                line['states_355'] = extra_states(args, rs, device)
            except Exception as err:      # never let the extra measurement hide the headline number
                line['states_355'] = {'error': str(err)}
        if not args.no_cpu_baseline and world == 1:      # reported baseline, rank 0 at N = 1 only
            line['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
