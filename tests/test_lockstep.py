"""CPU: the generator restatement of scipy's 1-D Nelder-Mead is scipy's, evaluation by evaluation."""
import numpy as np
import scipy.optimize

from remixt_amd import lockstep


def _functions():
    rng = np.random.RandomState(0)
    fs = []
    for i in range(40):
        a, b, c = rng.uniform(1, 2000), rng.uniform(0.1, 5), rng.uniform(0, 0.05)
        lo, hi = 10., 2000.

        def f(v, a=a, b=b, c=c, lo=lo, hi=hi):
            x = float(v[0])
            if x < lo or x > hi:
                return np.inf
            return b * (np.log(x) - np.log(a)) ** 2 + c * np.sin(x / 37.) + 1e3
        fs.append((f, rng.uniform(10, 2000)))
    # flat (immediate convergence), monotone (runs into the bound), zero start, max-iteration case
    fs.append((lambda v: 5.0, 100.))
    fs.append((lambda v: float(v[0]) if 10 <= v[0] <= 2000 else np.inf, 500.))
    fs.append((lambda v: (float(v[0]) - 3.) ** 2, 0.))
    fs.append((lambda v: -abs(float(v[0])) ** 1.5, 1.0))
    return fs


def test_generator_equals_scipy_fmin():
    for f, x0 in _functions():
        calls = []

        def rec(v):
            assert v.shape == (1,)
            calls.append(float(v[0]))
            return f(v)
        ref = scipy.optimize.fmin(rec, x0, full_output=1, disp=False)
        ref_calls = list(calls); calls.clear()
        g = lockstep.fmin_1d(x0)
        try:
            x = next(g)
            while True:
                x = g.send(rec(x))
        except StopIteration as stop:
            xopt, fopt, it, nf, warn = stop.value
        assert calls == ref_calls
        assert xopt[0] == ref[0][0] and (fopt == ref[1] or (np.isnan(fopt) and np.isnan(ref[1])))
        assert (it, nf, warn) == (ref[2], ref[3], ref[4])


def test_run_lockstep_batches_rounds():
    fs = _functions()[:12]
    rounds = []

    def evaluate(ids, xs):
        rounds.append(len(ids))
        return [fs[i][0](x) for i, x in zip(ids, xs)]
    res = lockstep.run_lockstep([lockstep.fmin_1d(x0) for _, x0 in fs], evaluate)
    for (f, x0), r in zip(fs, res):
        ref = scipy.optimize.fmin(f, x0, full_output=1, disp=False)
        assert r[0][0] == ref[0][0] and r[3] == ref[3]
    assert rounds[0] == 12 and max(rounds) == 12 and len(rounds) == max(r[3] for r in res)
