"""Lock-step 1-D optimisers for the likelihood-parameter M-steps of several restarts.

The reference updates each likelihood parameter with
`scipy.optimize.brute(nll, ranges=[bounds], full_output=True)` (remixt/cn_model.py:553-558): a
20-point grid followed by a `scipy.optimize.fmin` (Nelder-Mead) polish -- ~45 sequential objective
evaluations, each of which is a ~100 us device round trip here.  Restarts are independent, so their
optimisers can advance together and every round of evaluations be ONE batched launch.

`fmin_1d` restates scipy's Nelder-Mead (scipy/optimize/_optimize.py `_minimize_neldermead`, the
N = 1 case with fmin's defaults xatol = fatol = 1e-4, maxiter = maxfun = 200, no bounds) as a
generator that yields the point it wants evaluated and receives the value: same floating-point
operations in the same order, hence the same evaluation sequence and result as scipy
(tests/test_lockstep.py checks this bit for bit).  `run_lockstep` drives many such generators with
a batch evaluator.
"""
import numpy as np


def fmin_1d(x0, xatol=1e-4, fatol=1e-4, maxiter=200, maxfun=200):
    """Generator form of scipy.optimize.fmin(func, x0, full_output=1, disp=False) for one variable.

    Protocol: `x = next(gen)` / `x = gen.send(f)` give the next point (ndarray of shape (1,)) to
    evaluate; StopIteration.value = (xopt (1,), fopt, iterations, funcalls, warnflag).
    """
    rho, chi, psi, sigma = 1, 2, 0.5, 0.5
    nonzdelt, zdelt = 0.05, 0.00025
    x0 = np.atleast_1d(x0).flatten()
    x0 = np.asarray(x0, dtype=np.float64)
    N = 1
    sim = np.empty((N + 1, N), dtype=x0.dtype)
    sim[0] = x0
    y = np.array(x0, copy=True)
    if y[0] != 0:
        y[0] = (1 + nonzdelt) * y[0]
    else:
        y[0] = zdelt
    sim[1] = y
    fsim = np.full((N + 1,), np.inf, dtype=float)
    fcalls = 0

    class _MaxFun(Exception):
        pass

    # every evaluation goes through this inline pattern:
    #   if fcalls >= maxfun: raise _MaxFun ; fcalls += 1 ; f = yield copy(x)
    try:
        for k in range(N + 1):
            if fcalls >= maxfun:
                raise _MaxFun()
            fcalls += 1
            fsim[k] = yield np.copy(sim[k])
    except _MaxFun:
        pass
    finally:
        ind = np.argsort(fsim)
        sim = np.take(sim, ind, 0)
        fsim = np.take(fsim, ind, 0)
    ind = np.argsort(fsim)
    fsim = np.take(fsim, ind, 0)
    sim = np.take(sim, ind, 0)

    iterations = 1
    while fcalls < maxfun and iterations < maxiter:
        try:
            if (np.max(np.ravel(np.abs(sim[1:] - sim[0]))) <= xatol and
                    np.max(np.abs(fsim[0] - fsim[1:])) <= fatol):
                break
            xbar = np.add.reduce(sim[:-1], 0) / N
            xr = (1 + rho) * xbar - rho * sim[-1]
            if fcalls >= maxfun:
                raise _MaxFun()
            fcalls += 1
            fxr = yield np.copy(xr)
            doshrink = 0
            if fxr < fsim[0]:
                xe = (1 + rho * chi) * xbar - rho * chi * sim[-1]
                if fcalls >= maxfun:
                    raise _MaxFun()
                fcalls += 1
                fxe = yield np.copy(xe)
                if fxe < fxr:
                    sim[-1] = xe
                    fsim[-1] = fxe
                else:
                    sim[-1] = xr
                    fsim[-1] = fxr
            else:
                if fxr < fsim[-2]:
                    sim[-1] = xr
                    fsim[-1] = fxr
                else:
                    if fxr < fsim[-1]:
                        xc = (1 + psi * rho) * xbar - psi * rho * sim[-1]
                        if fcalls >= maxfun:
                            raise _MaxFun()
                        fcalls += 1
                        fxc = yield np.copy(xc)
                        if fxc <= fxr:
                            sim[-1] = xc
                            fsim[-1] = fxc
                        else:
                            doshrink = 1
                    else:
                        xcc = (1 - psi) * xbar + psi * sim[-1]
                        if fcalls >= maxfun:
                            raise _MaxFun()
                        fcalls += 1
                        fxcc = yield np.copy(xcc)
                        if fxcc < fsim[-1]:
                            sim[-1] = xcc
                            fsim[-1] = fxcc
                        else:
                            doshrink = 1
                    if doshrink:
                        for j in range(1, N + 1):
                            sim[j] = sim[0] + sigma * (sim[j] - sim[0])
                            if fcalls >= maxfun:
                                raise _MaxFun()
                            fcalls += 1
                            fsim[j] = yield np.copy(sim[j])
            iterations += 1
        except _MaxFun:
            pass
        finally:
            ind = np.argsort(fsim)
            sim = np.take(sim, ind, 0)
            fsim = np.take(fsim, ind, 0)
    x = sim[0]
    fval = np.min(fsim)
    warnflag = 0
    if fcalls >= maxfun:
        warnflag = 1
    elif iterations >= maxiter:
        warnflag = 2
    return x, fval, iterations, fcalls, warnflag


def run_lockstep(generators, evaluate):
    """Advance all generators together.  `evaluate(ids, xs)` gets the indices of the generators that
    are waiting and their points (list of (1,) arrays) and returns the function values in order.
    Returns the list of the generators' return values."""
    n = len(generators)
    results = [None] * n
    pending = {}
    for i, g in enumerate(generators):
        try:
            pending[i] = next(g)
        except StopIteration as stop:
            results[i] = stop.value
    while pending:
        ids = sorted(pending)
        fs = evaluate(ids, [pending[i] for i in ids])
        nxt = {}
        for i, f in zip(ids, fs):
            try:
                nxt[i] = generators[i].send(f)
            except StopIteration as stop:
                results[i] = stop.value
        pending = nxt
    return results
