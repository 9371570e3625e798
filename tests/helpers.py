"""Shared helpers: small seeded problems and attribute-by-attribute comparison."""
import numpy as np

from remixt_amd import synthetic
from remixt_amd.cn_model import BreakpointModel

STATE_ATTRS = ['framelogprob', 'posterior_marginals', 'p_breakpoint', 'p_outlier_total', 'p_outlier_allele', 'p_allele_swap']
DENSE_ATTRS = ['log_transmat', 'cached_log_transmat', 'joint_posterior_marginals']

RTOL = 1e-6   # north_star tolerance for posteriors / log-likelihood
ATOL = 1e-9   # probabilities below 1e-9 are compared absolutely


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    return np.allclose(a, b, rtol=rtol, atol=atol)


def maxerr(a, b):
    a = np.asarray(a, dtype=float); b = np.asarray(b, dtype=float)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / (ATOL / RTOL + np.maximum(np.abs(a), np.abs(b)))))


def add_shared_boundary_breakpoints(e):
    """e.breakpoints plus two breakpoints with one breakend each on the SAME boundary -- (a, side 1) and (a + 1, side 0) --
    so that the segment remap has to insert a zero-length segment there (cn_model.py:86-161)."""
    N = len(e.l)
    used = set(be for bp in e.breakpoints.values() for be in bp)
    brk = dict(e.breakpoints)
    far = [(n, sd) for n in range(N - 2, 0, -1) for sd in (0, 1) if (n, sd) not in used]
    for a in range(4, N - 6):
        ends = [(a, 1), (a + 1, 0)]
        if (a, a + 1) in e.adjacencies and not any(x in used for x in ends):
            f = [x for x in far if abs(x[0] - a) > 2][:2]
            brk['shared_a'] = frozenset([(a, 1), f[0]])
            brk['shared_b'] = frozenset([(a + 1, 0), f[1]])
            return brk
    raise RuntimeError('no free boundary')


def make_model(kernel, N=120, M=3, max_cn=3, chains=4, seed=0, restart=0, normal_contamination=True, experiment=None, **kw):
    e = experiment if experiment is not None else synthetic.make_experiment(N, num_clones=M, max_copy_number=max_cn, num_chains=chains, seed=seed)
    ps = synthetic.make_init_params(e, restart + 1, max_cn, num_clones=M)[restart]
    m = BreakpointModel(e.x, e.l, e.adjacencies, e.breakpoints, max_copy_number=max_cn,
                        divergence_weight=ps['divergence_weight'], max_depth=ps['max_depth'],
                        normal_contamination=normal_contamination, kernel_module=kernel, quiet=True, **kw)
    h = synthetic.h_init_from_params(ps, M)
    return m, h, e


def attach(m, h):
    m._attach_model(m._build_model(np.asarray(h, dtype=float)))
    return m.model


def compare_models(a, b, attrs=STATE_ATTRS, dense=False, tag=''):
    worst = {}
    for name in list(attrs) + (DENSE_ATTRS if dense else []):
        va, vb = np.asarray(getattr(a, name)), np.asarray(getattr(b, name))
        assert va.shape == vb.shape, (tag, name, va.shape, vb.shape)
        worst[name] = maxerr(va, vb)
        assert close(va, vb), '%s %s: max rel err %.3e' % (tag, name, worst[name])
    return worst
