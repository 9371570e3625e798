"""CPU: the generator restatement of scipy's 1-D Nelder-Mead is scipy's, evaluation by evaluation."""
import numpy as np
import pytest
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


def test_lbfgsb_gen_reproduces_scipy_minimize():
    """The reverse-communication L-BFGS-B generator visits the points of scipy.optimize.minimize(method=
    'L-BFGS-B', jac=..., bounds=...) in the same order and returns the same result, bit for bit (the h
    M-step of BreakpointModel.update_h, cn_model.py:500-508, for all restarts in lock step)."""
    import scipy.optimize
    from remixt_amd import lockstep
    if not lockstep.lbfgsb_available():
        pytest.skip('scipy layout other than 1.15: RestartSet falls back to scipy.optimize.minimize per restart')
    rng = np.random.RandomState(3)
    for trial in range(6):
        n = 3
        A = rng.randn(n, n); A = A @ A.T + np.eye(n)
        c = rng.randn(n) * (trial + 1)
        fun = lambda x: float(0.5 * x @ A @ x - c @ x + np.sum(np.log1p(x * x)))
        grad = lambda x: A @ x - c + 2 * x / (1 + x * x)
        seen = []

        def f_rec(x):
            seen.append(np.array(x)); return fun(x)
        x0 = rng.rand(n) * 12.0          # partly outside the upper bound: scipy clips
        bounds = [(1e-8, 10.)] * n
        ref = scipy.optimize.minimize(f_rec, x0, method='L-BFGS-B', jac=grad, bounds=bounds)
        gen = lockstep.lbfgsb_gen(x0, bounds)
        pts = []
        try:
            x = next(gen)
            while True:
                pts.append(np.array(x))
                x = gen.send((fun(x), grad(x)))
        except StopIteration as stop:
            res = stop.value
        assert len(pts) == len(seen) and all(np.array_equal(a, b) for a, b in zip(pts, seen))
        assert np.array_equal(res.x, ref.x) and res.fun == ref.fun and res.nfev == ref.nfev and res.nit == ref.nit
        assert res.success == ref.success and res.message == ref.message


def test_lbfgsb_lockstep_reproduces_scipy_minimize_for_every_run():
    """The flat lock-step driver (what RestartSet's h M-step runs): several runs at once, each visiting the points of its own
    scipy.optimize.minimize(method='L-BFGS-B') in the same order and ending on the same result bit for bit -- also when the runs
    finish after different numbers of evaluations and when a start lies outside the bounds."""
    import scipy.optimize
    from remixt_amd import lockstep
    if not lockstep.lbfgsb_available():
        pytest.skip('scipy layout other than 1.15')
    rng = np.random.RandomState(11)
    n, R = 3, 7
    probs = []
    for r in range(R):
        A = rng.randn(n, n); A = A @ A.T + np.eye(n)
        c = rng.randn(n) * (r + 1)
        probs.append((A, c))
    fun = lambda r, x: float(0.5 * x @ probs[r][0] @ x - probs[r][1] @ x + np.sum(np.log1p(x * x)))
    grad = lambda r, x: probs[r][0] @ x - probs[r][1] + 2 * x / (1 + x * x)
    x0s = [rng.rand(n) * 12.0 for _ in range(R)]
    bounds = [(1e-8, 10.)] * n
    refs, seen_ref = [], []
    for r in range(R):
        seen = []
        refs.append(scipy.optimize.minimize(lambda x, r=r, seen=seen: (seen.append(np.array(x)), fun(r, x))[1], x0s[r], method='L-BFGS-B',
                                            jac=lambda x, r=r: grad(r, x), bounds=bounds))
        seen_ref.append(seen)
    seen = [[] for _ in range(R)]
    rounds = []

    def evaluate(ids, X):
        rounds.append(list(ids))
        F = np.zeros(len(ids)); G = np.zeros((len(ids), n))
        for j, r in enumerate(ids):
            seen[r].append(np.array(X[j])); F[j] = fun(r, X[j]); G[j] = grad(r, X[j])
        return F, G
    res = lockstep.lbfgsb_lockstep(x0s, bounds, evaluate)
    assert rounds[0] == list(range(R)) and len(rounds) == max(len(s_) for s_ in seen)          # shared rounds
    assert len(set(len(s_) for s_ in seen)) > 1                                                     # ... of runs of different lengths
    for r in range(R):
        assert len(seen[r]) == len(seen_ref[r]) and all(np.array_equal(a, b) for a, b in zip(seen[r], seen_ref[r]))
        assert np.array_equal(res[r].x, refs[r].x) and res[r].fun == refs[r].fun and res[r].nfev == refs[r].nfev and res[r].nit == refs[r].nit
        assert res[r].success == refs[r].success and res[r].message == refs[r].message
        assert np.array_equal(res[r].jac, refs[r].jac)


def test_lockstep_runs_generators_together():
    from remixt_amd import lockstep
    calls = []

    def evaluate(ids, xs):
        calls.append(list(ids))
        return [float((x[0] - 3.0 - i) ** 2) for i, x in zip(ids, xs)]
    out = lockstep.run_lockstep([lockstep.fmin_1d(1.0), lockstep.fmin_1d(2.0), lockstep.fmin_1d(10.0)], evaluate)
    assert [abs(o[0][0] - (3.0 + i)) < 1e-3 for i, o in enumerate(out)] == [True] * 3
    assert calls[0] == [0, 1, 2] and len(calls) < sum(o[3] for o in out)      # rounds are shared


def test_sample_without_replacement_matches_numpy_distribution():
    """Seeded restarts draw their M-step samples with one cumulative sum; the draws must follow the
    distribution of numpy's choice(replace=False, p=...) (successive sampling)."""
    from collections import Counter
    from remixt_amd.cn_model import _sample_without_replacement as draw
    p = np.array([.4, .3, .15, .1, .04, .01])
    r1, r2 = np.random.RandomState(1), np.random.RandomState(2)
    c1, c2 = Counter(), Counter()
    for _ in range(20000):
        c1[tuple(draw(r1, 6, 2, p))] += 1
        c2[tuple(r2.choice(6, 2, replace=False, p=p))] += 1
    for k in c2:
        if c2[k] > 400:
            assert abs(c1[k] - c2[k]) < 6 * np.sqrt(c2[k]), (k, c1[k], c2[k])
    s = draw(np.random.RandomState(0), 5000, 200)
    assert len(set(s.tolist())) == 200 and s.min() >= 0 and s.max() < 5000
    with pytest.raises(ValueError):
        draw(np.random.RandomState(0), 6, 3, np.array([.5, .5, 0, 0, 0, 0]))


@pytest.mark.parametrize('n,size,zero_frac', [(50000, 200, 0.0), (3000, 200, 0.5), (400, 40, 0.8), (250, 25, 0.0)])
def test_weight_column_sampling_equals_the_dense_weight_path(n, size, zero_frac):
    """The restart driver draws its weighted M-step samples from a COLUMN of the outlier indicator array (WeightColumn ->
    rmx_weighted_sample_round, one native call per round of draws, no normalised copy, no mask): same indices in the same
    order and the same RNG stream afterwards as the dense path -- including rounds that have to be repeated because of
    duplicate draws (few positive weights) and the "fewer non-zero entries" error."""
    from remixt_amd import cn_model as cm
    if cm._native_sample_round() is None:
        pytest.skip('libremixt_hip.so not built')
    rng = np.random.RandomState(n)
    for col in (0, 1):
        q = rng.rand(n, 2)
        q[rng.rand(n) < zero_frac, col] = 0.
        w = q[:, col]
        norm = w.sum()
        r1, r2 = np.random.RandomState(5), np.random.RandomState(5)
        a = cm._sample_without_replacement(r1, n, size, cm.WeightColumn(q, col, norm))
        b = cm._sample_without_replacement(r2, n, size, w / norm)
        assert np.array_equal(a, b) and len(set(a.tolist())) == size
        assert r1.rand() == r2.rand()                                  # the two consumed the same number of draws
    q = np.zeros((n, 2)); q[:size - 1, 0] = 1.
    with pytest.raises(ValueError):
        cm._sample_without_replacement(np.random.RandomState(1), n, size, cm.WeightColumn(q, 0, q[:, 0].sum()))
