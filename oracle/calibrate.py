"""Port-vs-reference calibration of the CPU baseline (build container only).

TEST / MEASUREMENT INFRASTRUCTURE: times ONE EM iteration of one restart on the same seeded sample
(a) on the compiled reference kernel (oracle/_ref, built from /root/reference/remixt/bpmodel.pyx by
oracle/build_ref.py) and (b) on the C restatement oracle/remixt_oracle.c, both driven by this
project's BreakpointModel host class (bit-identical to the reference's host class over the same
kernel, tests/test_oracle_vs_ref.py), single thread.  The ratio goes to profiles/cpu_calibration.json;
bench.py's cpu_baseline leg (which times the port on the GPU box, where the reference cannot be)
carries it in the bench line.

    python oracle/calibrate.py [--segments 400] [--max-cn 8]
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_em_iteration(kern, e, p, clones, max_cn, update_iters, seed):
    from remixt_amd import synthetic
    from remixt_amd.cn_model import BreakpointModel
    m = BreakpointModel(e.x, e.l, e.adjacencies, e.breakpoints, max_copy_number=max_cn, divergence_weight=p['divergence_weight'],
                        max_depth=p['max_depth'], kernel_module=kern, quiet=True, rng=np.random.RandomState(seed))
    m.num_update_iter = update_iters
    m._attach_model(m._build_model(synthetic.h_init_from_params(p, clones)))
    m.prev_elbo = m.model.calculate_elbo()
    t0 = time.perf_counter()
    m.em_iteration(0)
    return time.perf_counter() - t0, m.prev_elbo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--segments', type=int, default=400)
    ap.add_argument('--clones', type=int, default=3)
    ap.add_argument('--max-cn', type=int, default=8)
    ap.add_argument('--update-iters', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=2)
    args = ap.parse_args()
    from oracle import oracle, refload
    from remixt_amd import synthetic
    oracle.build()
    ref = refload.load_ref_bpmodel()
    e = synthetic.make_experiment(args.segments, num_clones=args.clones, max_copy_number=args.max_cn, num_chains=4, seed=123)
    p = synthetic.make_init_params(e, 1, args.max_cn, num_clones=args.clones)[0]
    t_ref, t_port, elbo = [], [], []
    for _ in range(args.repeats):
        a, ea = one_em_iteration(ref, e, p, args.clones, args.max_cn, args.update_iters, 1000)
        b, eb = one_em_iteration(oracle, e, p, args.clones, args.max_cn, args.update_iters, 1000)
        t_ref.append(a); t_port.append(b); elbo.append((ea, eb))
    out = {'reference_s': min(t_ref), 'port_s': min(t_port), 'port_over_reference': min(t_port) / min(t_ref),
           'elbo_reference': elbo[0][0], 'elbo_port': elbo[0][1],
           'sample': 'one EM iteration of one restart, %d segments x 165 states, single thread, best of %d' % (args.segments, args.repeats),
           'where': 'build container: %s, %d vCPU' % (platform.processor() or platform.machine(), os.cpu_count())}
    try:
        with open('/proc/cpuinfo') as f:
            names = [l.split(':', 1)[1].strip() for l in f if l.startswith('model name')]
        if names:
            out['where'] = 'build container: %s, %d vCPU' % (names[0], os.cpu_count())
    except Exception:
        pass
    path = os.path.join(ROOT, 'profiles', 'cpu_calibration.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
