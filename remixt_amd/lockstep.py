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


class LbfgsbResult(object):
    """The fields of scipy's OptimizeResult that BreakpointModel.update_h looks at."""

    def __init__(self, x, fun, jac, nfev, nit, status, message, success):
        self.x, self.fun, self.jac, self.nfev, self.njev, self.nit = x, fun, jac, nfev, nfev, nit
        self.status, self.message, self.success = status, message, success

    def __str__(self):
        return ' message: {}\n success: {}\n  status: {}\n     fun: {}\n       x: {}\n     nit: {}\n     jac: {}\n    nfev: {}'.format(
            self.message, self.success, self.status, self.fun, self.x, self.nit, self.jac, self.nfev)


def lbfgsb_available():
    """The reverse-communication L-BFGS-B driver below calls scipy's private `_lbfgsb.setulb` with the
    argument list of scipy 1.15; other layouts make callers fall back to scipy.optimize.minimize."""
    try:
        import scipy
        from scipy.optimize import _lbfgsb, _lbfgsb_py   # noqa: F401
        major, minor = [int(v) for v in scipy.__version__.split('.')[:2]]
        return (major, minor) == (1, 15) and hasattr(_lbfgsb_py, 'status_messages') and hasattr(_lbfgsb_py, 'task_messages')
    except Exception:
        return False


def lbfgsb_gen(x0, bounds, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxiter=15000, maxls=20):
    """Generator form of scipy.optimize.minimize(fun, x0, method='L-BFGS-B', jac=grad, bounds=bounds)
    (scipy 1.15 `_minimize_lbfgsb`): yields the point at which it wants (f, g), receives the pair.

    Same sequence as scipy: ScalarFunction evaluates (f, g) at the clipped x0 when it is built and
    serves the first request from that cache; afterwards one (f, g) evaluation per `task == FG`.
    StopIteration.value is an LbfgsbResult."""
    from scipy.optimize import _lbfgsb
    from scipy.optimize._lbfgsb_py import status_messages, task_messages
    m = maxcor
    factr = ftol / np.finfo(float).eps
    x0 = np.asarray(x0, dtype=np.float64).ravel()
    n = x0.shape[0]
    lo = np.array([b[0] for b in bounds], dtype=np.float64)
    hi = np.array([b[1] for b in bounds], dtype=np.float64)
    x0 = np.clip(x0, lo, hi)
    nbd = np.full(n, 2, dtype=np.int32)       # finite lower and upper bounds on every variable
    low_bnd = lo.copy(); upper_bnd = hi.copy()

    x = np.array(x0, dtype=np.float64)
    f = np.array(0.0, dtype=np.int32)
    g = np.zeros((n,), dtype=np.int32)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, dtype=np.int32)
    task = np.zeros(2, dtype=np.int32)
    ln_task = np.zeros(2, dtype=np.int32)
    lsave = np.zeros(4, dtype=np.int32)
    isave = np.zeros(44, dtype=np.int32)
    dsave = np.zeros(29, dtype=np.float64)

    # ScalarFunction.__init__: f and g at x0
    cx = np.copy(x0)
    cf, cg = yield np.copy(cx)
    cg = np.atleast_1d(np.asarray(cg, dtype=np.float64))
    nfev = 1
    n_iterations = 0
    while True:
        g = g.astype(np.float64)
        _lbfgsb.setulb(m, x, low_bnd, upper_bnd, nbd, f, g, factr, gtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
        if task[0] == 3:
            if not np.array_equal(x, cx):
                cx = np.copy(x)
                cf, cg = yield np.copy(cx)
                cg = np.atleast_1d(np.asarray(cg, dtype=np.float64))
                nfev += 1
            f, g = cf, cg
        elif task[0] == 1:
            n_iterations += 1
            if n_iterations >= maxiter:
                task[0] = 5; task[1] = 504
            elif nfev > maxfun:
                task[0] = 5; task[1] = 502
        else:
            break
    if task[0] == 4:
        warnflag = 0
    elif nfev > maxfun or n_iterations >= maxiter:
        warnflag = 1
    else:
        warnflag = 2
    msg = status_messages[task[0]] + ": " + task_messages[task[1]]
    return LbfgsbResult(x, f, g, nfev, n_iterations, warnflag, msg, warnflag == 0)


class _LbfgsbState(object):
    """Everything scipy's `_minimize_lbfgsb` keeps for one run (scipy 1.15), preallocated: the lean form of lbfgsb_gen for
    lbfgsb_lockstep (no generator, no per-iteration array allocation).  Same calls of `_lbfgsb.setulb` with the same values."""
    __slots__ = ('n', 'm', 'x', 'lo', 'hi', 'nbd', 'g', 'wa', 'iwa', 'task', 'ln_task', 'lsave', 'isave', 'dsave', 'cx', 'cf', 'cg',
                 'f', 'nfev', 'nit', 'done', 'factr', 'gtol', 'maxls', 'maxfun', 'maxiter')

    def __init__(self, x0, lo, hi, maxcor, ftol, gtol, maxfun, maxiter, maxls):
        n = len(x0)
        self.n, self.m = n, maxcor
        self.lo, self.hi = lo, hi
        self.nbd = np.full(n, 2, dtype=np.int32)
        self.x = np.clip(np.asarray(x0, dtype=np.float64).ravel(), lo, hi)
        self.g = np.zeros(n, dtype=np.float64)
        m = maxcor
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32); self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32); self.isave = np.zeros(44, dtype=np.int32); self.dsave = np.zeros(29, dtype=np.float64)
        self.cx = self.x.copy(); self.cf = 0.0; self.cg = np.zeros(n, dtype=np.float64)
        self.f = 0.0
        self.nfev = 0; self.nit = 0; self.done = False
        self.factr = ftol / np.finfo(float).eps; self.gtol = gtol; self.maxls = maxls; self.maxfun = maxfun; self.maxiter = maxiter


def lbfgsb_lockstep(x0s, bounds, evaluate, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxiter=15000, maxls=20):
    """scipy.optimize.minimize(method='L-BFGS-B', jac=True-style (f, g), bounds=bounds) for several starting points at once, the
    runs advancing in lock step: `evaluate(ids, X)` gets the indices of the runs that wait for an evaluation and their points as
    the rows of X (a (k, n) float64 array it may keep) and returns (F (k,), G (k, n)).  Each run makes exactly the calls of
    `_lbfgsb.setulb` that lbfgsb_gen -- i.e. scipy's own driver -- makes, with the same values (tests/test_lockstep.py compares the
    two and scipy.optimize.minimize bit for bit); only the Python around them is flat: no generators, arrays made once.
    Returns the list of LbfgsbResult."""
    from scipy.optimize import _lbfgsb
    from scipy.optimize._lbfgsb_py import status_messages, task_messages
    setulb = _lbfgsb.setulb
    lo = np.array([b[0] for b in bounds], dtype=np.float64)
    hi = np.array([b[1] for b in bounds], dtype=np.float64)
    runs = [_LbfgsbState(x0, lo, hi, maxcor, ftol, gtol, maxfun, maxiter, maxls) for x0 in x0s]
    n = len(lo)
    X = np.zeros((len(runs), n), dtype=np.float64)

    def advance(s, resumed):
        """Run `s` until it wants (f, g) at a new point (True) or ends (False).  resumed: the evaluation an FG task asked for has
        arrived -- finish that branch (f, g = cf, cg) first."""
        if resumed:
            s.f = s.cf; s.g[:] = s.cg
        while True:
            setulb(s.m, s.x, s.lo, s.hi, s.nbd, s.f, s.g, s.factr, s.gtol, s.wa, s.iwa, s.task, s.lsave, s.isave, s.dsave, s.maxls, s.ln_task)
            t0 = s.task[0]
            if t0 == 3:
                if not np.array_equal(s.x, s.cx):
                    s.cx[:] = s.x
                    return True
                s.f = s.cf; s.g[:] = s.cg
            elif t0 == 1:
                s.nit += 1
                if s.nit >= s.maxiter:
                    s.task[0] = 5; s.task[1] = 504
                elif s.nfev > s.maxfun:
                    s.task[0] = 5; s.task[1] = 502
            else:
                s.done = True
                return False

    # ScalarFunction.__init__: (f, g) at the clipped x0 -- the first FG request is served from this evaluation
    waiting = list(range(len(runs)))
    first = True
    while waiting:
        k = len(waiting)
        for j, i in enumerate(waiting):
            X[j] = runs[i].cx
        F, G = evaluate(waiting, X[:k])
        nxt = []
        for j, i in enumerate(waiting):
            s = runs[i]
            s.cf = float(F[j]); s.cg[:] = G[j]; s.nfev += 1
            if advance(s, not first):
                nxt.append(i)
        first = False
        waiting = nxt
    out = []
    for s in runs:
        t0 = int(s.task[0])
        if t0 == 4:
            warnflag = 0
        elif s.nfev > s.maxfun or s.nit >= s.maxiter:
            warnflag = 1
        else:
            warnflag = 2
        msg = status_messages[t0] + ": " + task_messages[int(s.task[1])]
        out.append(LbfgsbResult(s.x, s.f, s.cg.copy(), s.nfev, s.nit, warnflag, msg, warnflag == 0))
    return out
