"""Randomised differential check of the HIP path against the CPU oracle on SMALL problems of irregular shape (tests-only: imports the
oracle).  `python -m tests.fuzz_small --seeds 0:200` on a GPU box; tests/test_hip_fuzz.py runs a fixed handful of the same cases.

A case draws: segments (4 .. 90), chromosome chains (1 .. 6, down to two segments long), clones (2, 3), max copy number (2 .. 6),
breakpoints (1 .. N/3, some sharing a segment boundary), restarts per batch (1 .. 7: ragged quads of the matrix-core kernel) and the
forward-backward workgroup shape option; then posteriors, indicators, p_breakpoint, log Z and ELBO are compared after EVERY coordinate
update of two sweeps (1e-7 relative; the requirement is 1e-6) and the decoded paths bit for bit -- or, where the two lattices' inputs
differ in their last bits and two paths tie, to equal log-probability (1e-11 relative, under either side's arrays)."""
import argparse
import sys
import traceback

import numpy as np


def draw_case(seed):
    rng = np.random.RandomState(100003 * seed + 17)
    N = int(rng.choice([4, 5, 7, 12, 20, 33, 64, 90]))
    chains = int(rng.randint(1, min(6, N // 2) + 1))
    M = int(rng.choice([2, 3, 3]))
    max_cn = int(rng.choice([2, 3, 4, 5, 6]))
    nbrk = int(rng.choice([1, 2, max(1, N // 6), max(1, N // 3)]))
    R = int(rng.randint(1, 8))
    nv = int(rng.choice([0, 0, 1, 2, 4]))
    shared = bool(rng.randint(0, 2))
    return dict(seed=seed, N=N, chains=chains, M=M, max_cn=max_cn, nbrk=nbrk, R=R, fb_nv=nv, shared=shared)


def run_case(case, oracle_mod):
    from remixt_amd import synthetic
    from tests import helpers as H
    from tests.test_hip_bench_shapes import _two_sets, _compare_after_every_update
    e = synthetic.make_experiment(case['N'], num_clones=case['M'], max_copy_number=case['max_cn'], num_chains=case['chains'],
                                  seed=case['seed'], num_breakpoints=case['nbrk'])
    if case['shared']:
        try:
            e.breakpoints = H.add_shared_boundary_breakpoints(e)
        except RuntimeError:        # every boundary of a very small problem already carries a breakpoint
            pass
    ps = synthetic.make_init_params(e, case['R'], case['max_cn'], num_clones=case['M'])
    dev, ora = _two_sets(oracle_mod, e, ps, case['max_cn'], case['M'], options={'fb_nv': case['fb_nv']})
    _compare_after_every_update(dev, ora, rtol=1e-7, elbo_rtol=1e-7, ties_ok=True)
    b = dev.batch
    return b.num_cn_states, b.info(12), b.info(13)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', default='0:40')
    args = ap.parse_args(argv)
    lo, hi = [int(v) for v in args.seeds.split(':')]
    from oracle import oracle
    oracle.build()
    bad = 0
    for seed in range(lo, hi):
        case = draw_case(seed)
        try:
            S, kern, nv = run_case(case, oracle)
            print('ok  ', case, 'states', S, 'fb kernel', kern, 'nv', nv, flush=True)
        except Exception as err:        # report every failing case, then fail
            bad += 1
            print('FAIL', case, type(err).__name__, str(err).splitlines()[0][:300], flush=True)
            traceback.print_exc(limit=3)
    print('%d cases, %d failed' % (hi - lo, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
