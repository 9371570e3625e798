"""Where a round of the lock-step h M-step goes: time inside the batched objective call vs the optimisers' Python steps
(one restart group of 8 at the benchmark configuration, nothing else on the GPU unless OTHER=1 keeps a second group sweeping)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from remixt_amd import synthetic
from remixt_amd.restarts import RestartSet
e = synthetic.make_experiment(50000, num_clones=3, max_copy_number=8, num_chains=23, seed=0)
ps = synthetic.make_init_params(e, 8, 8)
rs = RestartSet(e, ps, 8, num_clones=3, quiet=True, seeds=[1000 + i for i in range(8)])
for m, v in zip(rs.models, rs.calculate_elbo()):
    m.prev_elbo = float(v)
for i in range(2):
    rs.em_iteration(i, 5)
b = rs.batch
make = b.h_batch_evaluator
acc = {'in': 0., 'n': 0, 't_first': None, 't_last': None}
def factory(restarts):
    ev = make(restarts)
    def timed(ids, xs):
        t0 = time.perf_counter()
        out = ev(ids, xs)
        t1 = time.perf_counter()
        acc['in'] += t1 - t0; acc['n'] += 1
        if acc['t_first'] is None:
            acc['t_first'] = t0
        acc['t_last'] = t1
        return out
    return timed
b.h_batch_evaluator = factory
tot = 0.
NIT = 6
c0 = [b.info(i) for i in (60, 61, 62, 63)]
for i in range(NIT):
    acc['t_first'] = None
    rs.em_iteration(2 + i, 5)
    tot += acc['t_last'] - acc['t_first']
print('%d rounds per M-step; per round: %.1f us in the objective call, %.1f us outside (optimiser steps); %.2f ms per M-step' % (
    acc['n'] / NIT, acc['in'] / acc['n'] * 1e6, (tot - acc['in']) / acc['n'] * 1e6, tot / NIT * 1e3))

c1 = [b.info(i) for i in (60, 61, 62, 63)]
n = max(c1[3] - c0[3], 1)
print('inside the C call, per batched sampled-objective round (%d rounds, h rounds and others): %.1f us preparing + launching, %.1f us waiting for the device, %.1f us after the wait'
      % (n, (c1[0] - c0[0]) / n * 1e-3, (c1[1] - c0[1]) / n * 1e-3, (c1[2] - c0[2]) / n * 1e-3))
