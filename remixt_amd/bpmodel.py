"""MI355X-native `RemixtModel`: the object protocol of the reference's Cython
class `remixt.bpmodel.RemixtModel` (reference remixt/bpmodel.pyx:397-1210) on
top of the HIP C ABI (include/remixt_amd.h).

Two objects:

* `RemixtBatch` -- one dataset (segments, state tables, topology) resident in
  HBM with R independent restarts; coordinate updates and ELBOs run for any
  restart range in single launches.
* `RemixtModel` -- the drop-in per-model view.  Constructed with the
  reference's positional signature it owns a batch of one; `batch.model(r)`
  gives the same protocol on restart r of a shared batch.

State lives on the device.  Array attributes are copied to the host on read and
to the device on assignment (the reference hands out memoryviews of host
buffers; code that mutates a returned array in place must assign it back).
`log_transmat`, `cached_log_transmat` and `joint_posterior_marginals`
((N-1) x S x S) are never stored: they are materialised only when read.
"""
import ctypes as C

import numpy as np

from . import _lib

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

RMX_EVALUE, RMX_EASSERT, RMX_EDEVICE, RMX_EUNSUPPORTED, RMX_EARG = 1, 2, 3, 4, 5

MAX_CLONES = 4      # RMX_MAX_CLONES (include/remixt_amd.h)
PARAM_IDS = {
    'negbin_r_0': 0, 'negbin_r_1': 1, 'negbin_hdel_mu': 2, 'negbin_hdel_r_0': 3, 'negbin_hdel_r_1': 4,
    'betabin_M_0': 5, 'betabin_M_1': 6, 'betabin_loh_p': 7, 'betabin_loh_M_0': 8, 'betabin_loh_M_1': 9,
    'prior_outlier_total': 10, 'prior_outlier_allele': 11, 'divergence_weight': 12, 'hmm_log_norm_const': 13,
}
ARRAY_IDS = {
    'h': 0, 'p_breakpoint': 1, 'p_allele_swap': 2, 'p_outlier_total': 3, 'p_outlier_allele': 4,
    'posterior_marginals': 5, 'framelogprob': 6, 'total_likelihood_mask': 7, 'allele_likelihood_mask': 8,
    'log_transmat': 9, 'cached_log_transmat': 10, 'joint_posterior_marginals': 11, 'state_sequence': 12,
}
_WRITABLE = {'h', 'p_breakpoint', 'p_allele_swap', 'p_outlier_total', 'p_outlier_allele', 'posterior_marginals',
             'total_likelihood_mask', 'allele_likelihood_mask'}
_STATE_TABLES = {'cn_states_total': 0, 'num_alleles_subclonal': 1, 'is_hdel': 2, 'is_loh': 3}


# enum rmx_option_id (include/remixt_amd.h)
OPTION_IDS = dict((n, i) for i, n in enumerate((
    'fb_kernel', 'fb_nv', 'fb_breakend_codes', 'fuse_sweeps', 'two_streams', 'viterbi_plain', 'search_mode', 'ell_dense', 'strip',
    'cell_cache', 'sparse_trial', 'fb_debug', 'pairwise_kernel', 'pace_sweeps', 'fb_wg_budget', 'trial_kernel', 'grad_kernel', 'stream_pool', 'viterbi_cluster', 'traceback', 'cu_partition')))


def set_default_option(name, value):
    """Process-wide default of a tuning option for batches created afterwards (tests and A/B measurements)."""
    lib = _lib.load()
    rc = lib.rmx_set_default_option(OPTION_IDS[name], int(value))
    if rc:
        _raise(lib, rc)


def last_error_restarts():
    """Restarts flagged by the calling thread's last failing call (rmx_last_error_restarts)."""
    lib = _lib.load()
    n = lib.rmx_last_error_restarts(None, 0)      # the count first: the list is as long as the batch has restarts
    if n <= 0:
        return []
    buf = (C.c_int32 * n)()
    n = min(n, lib.rmx_last_error_restarts(buf, n))
    return [int(buf[i]) for i in range(n)]


def _raise(lib, rc):
    msg = lib.rmx_last_error().decode()
    if rc == RMX_EVALUE:
        err = ValueError(msg)
        err.restarts = last_error_restarts()
        raise err
    if rc == RMX_EASSERT:
        err = AssertionError(msg)
        err.restarts = last_error_restarts()
        raise err
    if rc == RMX_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == RMX_EARG:
        raise ValueError('bad argument: ' + msg)
    raise RuntimeError('HIP backend failure: ' + msg)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def compress_cn_states(cn_states, max_classes=64):
    """(N,S,M,2) int64 -> (classes (C,S,M,2), seg_class (N,) int32)."""
    lib = _lib.load()
    cn_states = _i64(cn_states)
    N, S, M, A = cn_states.shape
    if A != 2:
        raise ValueError('cn_states must have shape (num_segments, num_cn_states, num_clones, num_alleles)')
    seg_class = np.zeros(N, dtype=np.int32)
    classes = np.zeros((max_classes, S, M, 2), dtype=np.int64)
    nc = C.c_int32(0)
    rc = lib.rmx_compress_cn_states(cn_states.ctypes.data_as(_ip), N, S, M, max_classes,
                                    seg_class.ctypes.data_as(_i32p), classes.ctypes.data_as(_ip), C.byref(nc))
    if rc:
        _raise(lib, rc)
    return classes[:nc.value].copy(), seg_class


class RemixtBatch(object):
    """R restarts of one dataset on one GPU."""

    def __init__(self, num_clones, num_segments, num_breakpoints, normal_contamination,
                 cn_classes, seg_class, brk_states, h_init, l, x, y,
                 is_telomere, breakpoint_idx, breakpoint_orient, transition_penalty,
                 divergence_weight, device=0):
        self._lib = lib = _lib.load()
        self._handle = None
        cn_classes = _i64(cn_classes)
        if cn_classes.ndim != 4:
            raise ValueError('cn_classes must have shape (num_classes, num_cn_states, num_clones, 2)')
        brk_states = _i64(brk_states)
        seg_class = np.ascontiguousarray(seg_class, dtype=np.int32)
        h_init = np.atleast_2d(_f64(h_init))
        R = h_init.shape[0]
        divergence_weight = _f64(np.broadcast_to(np.asarray(divergence_weight, dtype=np.float64), (R,)))
        C_, S, M, A = cn_classes.shape
        # shape validation of bpmodel.pyx:509-529
        if M != num_clones or A != 2 or seg_class.shape[0] != num_segments:
            raise ValueError('cn_states must have shape (num_segments, num_cn_states, num_clones, num_alleles)')
        if brk_states.ndim != 2 or brk_states.shape[1] != num_clones:
            raise ValueError('cn_states must have shape (num_brk_states, num_clones)')
        if h_init.shape[1] != num_clones:
            raise ValueError('h must have length equal to num_clones')
        is_telomere = _i64(is_telomere); breakpoint_idx = _i64(breakpoint_idx); breakpoint_orient = _i64(breakpoint_orient)
        if is_telomere.shape[0] != num_segments:
            raise ValueError('is_telomere must have length equal to num_segments')
        if breakpoint_idx.shape[0] != num_segments:
            raise ValueError('breakpoint_idx must have length equal to num_segments')
        if breakpoint_orient.shape[0] != num_segments:
            raise ValueError('breakpoint_orient must have length equal to num_segments')
        l = _f64(l); x = _f64(x); y = _f64(y)
        if l.shape != (num_segments,) or x.shape != (num_segments,) or y.shape != (num_segments, 2):
            raise ValueError('l, x, y must have shapes (N,), (N,), (N, 2)')
        pr = _lib.RmxProblem()
        pr.num_clones = M; pr.num_segments = num_segments; pr.num_breakpoints = num_breakpoints
        pr.num_cn_states = S; pr.num_brk_states = brk_states.shape[0]; pr.num_classes = C_
        pr.normal_contamination = int(bool(normal_contamination))
        pr.cn_classes = cn_classes.ctypes.data_as(_ip); pr.seg_class = seg_class.ctypes.data_as(_i32p)
        pr.brk_states = brk_states.ctypes.data_as(_ip)
        pr.l = l.ctypes.data_as(_dp); pr.x = x.ctypes.data_as(_dp); pr.y = y.ctypes.data_as(_dp)
        pr.is_telomere = is_telomere.ctypes.data_as(_ip); pr.breakpoint_idx = breakpoint_idx.ctypes.data_as(_ip)
        pr.breakpoint_orient = breakpoint_orient.ctypes.data_as(_ip)
        pr.transition_penalty = float(transition_penalty)
        handle = C.c_void_p()
        rc = lib.rmx_batch_create(C.byref(pr), R, h_init.ctypes.data_as(_dp), divergence_weight.ctypes.data_as(_dp),
                                  int(device), C.byref(handle))
        if rc:
            _raise(lib, rc)
        self._handle = handle
        self.device = int(device)
        self.num_restarts = R
        self.num_clones = M
        self.num_segments = int(num_segments)
        self.num_breakpoints = int(num_breakpoints)
        self.num_cn_states = S
        self.num_brk_states = int(brk_states.shape[0])
        self.num_alleles = 2
        self.normal_contamination = bool(normal_contamination)
        self.transition_penalty = abs(float(transition_penalty))
        self.cn_classes = cn_classes
        self.seg_class = seg_class
        self.brk_states = brk_states
        self.is_telomere = is_telomere
        self.breakpoint_idx = breakpoint_idx
        self.breakpoint_orient = breakpoint_orient
        self.l = l; self.x = x; self.y = y
        self.cn_max = self.info(0)
        self._transition_model = 0
        self._samples = {}

    # -- lifetime ------------------------------------------------------------
    def close(self):
        if self._handle is not None:
            self._lib.rmx_batch_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            _raise(self._lib, rc)

    def info(self, what):
        out = C.c_int64(0)
        self._ck(self._lib.rmx_info(self._handle, what, C.byref(out)))
        return int(out.value)

    def synchronize(self):
        self._ck(self._lib.rmx_synchronize(self._handle))

    def set_option(self, name, value):
        """Tuning option of this batch (include/remixt_amd.h rmx_option_id): which equivalent kernel / launch shape runs."""
        self._ck(self._lib.rmx_set_option(self._handle, OPTION_IDS[name], int(value)))

    def get_option(self, name):
        out = C.c_int32(0)
        self._ck(self._lib.rmx_get_option(self._handle, OPTION_IDS[name], C.byref(out)))
        return int(out.value)

    def model(self, r):
        return RemixtModel._from_batch(self, r)

    # -- attributes ----------------------------------------------------------
    def array_shape(self, name):
        N, S, M, K, B = self.num_segments, self.num_cn_states, self.num_clones, self.num_breakpoints, self.num_brk_states
        return {
            'h': (M,), 'p_breakpoint': (K, B), 'p_allele_swap': (N, 2), 'p_outlier_total': (N, 2),
            'p_outlier_allele': (N, 2), 'posterior_marginals': (N, S), 'framelogprob': (N, S),
            'total_likelihood_mask': (N,), 'allele_likelihood_mask': (N,), 'log_transmat': (N - 1, S, S),
            'cached_log_transmat': (N - 1, S, S), 'joint_posterior_marginals': (N - 1, S, S), 'state_sequence': (N,),
        }[name]

    def get_array(self, r, name):
        shape = self.array_shape(name)
        dt = np.int64 if name in ('total_likelihood_mask', 'allele_likelihood_mask', 'state_sequence') else np.float64
        out = np.zeros(shape, dtype=dt)
        if out.size:
            self._ck(self._lib.rmx_get_array(self._handle, r, ARRAY_IDS[name], out.ctypes.data_as(C.c_void_p)))
        return out

    def fetch_indicators(self, r0=None, r1=None):
        """(p_outlier_total, p_outlier_allele) of restarts [r0, r1) as [r1 - r0][N][2] float64 arrays: views of pinned memory
        the batch owns, overwritten by the next call (rmx_fetch_indicators: two copies queued in order on the batch stream -- behind everything already queued there -- and waited for)."""
        r0, r1 = self._range(r0, r1)
        pt, pa = _dp(), _dp()
        self._ck(self._lib.rmx_fetch_indicators(self._handle, r0, r1, C.byref(pt), C.byref(pa)))
        shape = (r1 - r0, self.num_segments, 2)
        return np.ctypeslib.as_array(pt, shape=shape), np.ctypeslib.as_array(pa, shape=shape)

    def set_array(self, r, name, value):
        if name not in _WRITABLE:
            raise AttributeError('%s is read-only' % name)
        shape = self.array_shape(name)
        dt = np.int64 if name in ('total_likelihood_mask', 'allele_likelihood_mask') else np.float64
        v = np.ascontiguousarray(value, dtype=dt)
        if v.shape != shape:
            raise ValueError('%s must have shape %s' % (name, shape))
        if v.size:
            self._ck(self._lib.rmx_set_array(self._handle, r, ARRAY_IDS[name], v.ctypes.data_as(C.c_void_p)))

    def get_param(self, r, name):
        out = C.c_double(0.)
        self._ck(self._lib.rmx_get_param(self._handle, r, PARAM_IDS[name], C.byref(out)))
        return float(out.value)

    def set_param(self, r, name, value):
        self._ck(self._lib.rmx_set_param(self._handle, r, PARAM_IDS[name], float(value)))

    @property
    def transition_model(self):
        return self._transition_model

    @transition_model.setter
    def transition_model(self, value):
        self._ck(self._lib.rmx_set_transition_model(self._handle, int(value)))
        self._transition_model = int(value)

    def state_table(self, name):
        N, S, M = self.num_segments, self.num_cn_states, self.num_clones
        shape = (N, S, M) if name == 'cn_states_total' else (N, S)
        out = np.zeros(shape, dtype=np.int64)
        self._ck(self._lib.rmx_get_state_table(self._handle, _STATE_TABLES[name], out.ctypes.data_as(_ip)))
        return out

    # -- batched operations on restarts [r0, r1) -------------------------------
    def _range(self, r0, r1):
        if r0 is None:
            r0 = 0
        if r1 is None:
            r1 = self.num_restarts
        return int(r0), int(r1)

    def update_framelogprob(self, r0=None, r1=None):
        self._ck(self._lib.rmx_update_framelogprob(self._handle, *self._range(r0, r1)))

    def update_p_cn(self, r0=None, r1=None):
        self._ck(self._lib.rmx_update_p_cn(self._handle, *self._range(r0, r1)))

    def update_p_breakpoint(self, r0=None, r1=None):
        self._ck(self._lib.rmx_update_p_breakpoint(self._handle, *self._range(r0, r1)))

    def update_p_outlier_total(self, r0=None, r1=None):
        self._ck(self._lib.rmx_update_p_outlier_total(self._handle, *self._range(r0, r1)))

    def update_p_outlier_allele(self, r0=None, r1=None):
        self._ck(self._lib.rmx_update_p_outlier_allele(self._handle, *self._range(r0, r1)))

    def update_p_allele_swap(self, r0=None, r1=None):
        self._ck(self._lib.rmx_update_p_allele_swap(self._handle, *self._range(r0, r1)))

    def variational_update(self, iters=1, r0=None, r1=None):
        r0, r1 = self._range(r0, r1)
        self._ck(self._lib.rmx_variational_update(self._handle, r0, r1, int(iters)))

    def _scalar_range(self, fn, r0, r1):
        r0, r1 = self._range(r0, r1)
        out = np.zeros(r1 - r0, dtype=np.float64)
        self._ck(fn(self._handle, r0, r1, out.ctypes.data_as(_dp)))
        return out

    def calculate_elbo(self, r0=None, r1=None):
        return self._scalar_range(self._lib.rmx_calculate_elbo, r0, r1)

    def calculate_elbo_begin(self, r0=None, r1=None):
        """Queue the ELBO of restarts [r0, r1) without waiting (rmx_calculate_elbo_begin); calculate_elbo_end() returns the values."""
        r0, r1 = self._range(r0, r1)
        self._ck(self._lib.rmx_calculate_elbo_begin(self._handle, r0, r1))
        self._elbo_pending = r1 - r0

    def calculate_elbo_end(self):
        n = getattr(self, '_elbo_pending', 0)
        out = np.zeros(max(n, 1), dtype=np.float64)
        self._elbo_pending = 0
        self._ck(self._lib.rmx_calculate_elbo_end(self._handle, out.ctypes.data_as(_dp)))
        return out[:n]

    def calculate_variational_energy(self, r0=None, r1=None):
        return self._scalar_range(self._lib.rmx_calculate_variational_energy, r0, r1)

    def calculate_variational_entropy(self, r0=None, r1=None):
        return self._scalar_range(self._lib.rmx_calculate_variational_entropy, r0, r1)

    def _use_sample(self, r, sample):
        """Upload the sample mask unless it is the very array object last uploaded for this restart
        (the M-steps evaluate hundreds of times on one mask they never modify).  A caller that
        mutates a mask in place must pass a new array (or call invalidate_sample)."""
        if self._samples.get(r) is sample:
            return
        s64 = _i64(sample)
        if s64.shape != (self.num_segments,):
            raise ValueError('sample must have length num_segments')
        self._ck(self._lib.rmx_set_sample(self._handle, r, s64.ctypes.data_as(_ip)))
        self._samples[r] = sample

    def invalidate_sample(self, r=None):
        if r is None:
            self._samples.clear()
        else:
            self._samples.pop(r, None)

    def expected_log_likelihood(self, r, sample, want_grad=False):
        self._use_sample(r, sample)
        ell = C.c_double(0.)
        grad = np.zeros(self.num_clones, dtype=np.float64) if want_grad else None
        self._ck(self._lib.rmx_expected_log_likelihood(
            self._handle, r, None, C.byref(ell), grad.ctypes.data_as(_dp) if want_grad else None))
        return float(ell.value), grad

    def expected_log_likelihood_param_grid(self, r, name, values, sample):
        self._use_sample(r, sample)
        v = _f64(values).ravel()
        out = np.zeros(len(v), dtype=np.float64)
        self._ck(self._lib.rmx_expected_ll_param_grid(self._handle, r, PARAM_IDS[name], v.ctypes.data_as(_dp), len(v),
                                                      out.ctypes.data_as(_dp)))
        return out

    def expected_log_likelihood_batch(self, restarts, name, values):
        """E[ll] on each listed restart's current sample with likelihood parameter `name` set to the
        matching entry of `values` (and left there): the evaluation round of lock-step optimisers."""
        rl = np.ascontiguousarray(restarts, dtype=np.int32)
        v = _f64(values).ravel()
        if rl.shape != v.shape:
            raise ValueError('one value per listed restart')
        out = np.zeros(len(v), dtype=np.float64)
        self._ck(self._lib.rmx_expected_ll_batch(self._handle, len(v), rl.ctypes.data_as(_i32p), PARAM_IDS[name],
                                                 v.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
        return out

    def param_search(self, restarts, name, lo, hi, grid):
        """scipy.optimize.brute over `grid` + Nelder-Mead polish of likelihood parameter `name` for all
        listed restarts in lock step (rmx_param_search); returns the optimum per restart."""
        rl = np.ascontiguousarray(restarts, dtype=np.int32)
        g = _f64(grid).ravel()
        out = np.zeros(len(rl), dtype=np.float64)
        self._ck(self._lib.rmx_param_search(self._handle, len(rl), rl.ctypes.data_as(_i32p), PARAM_IDS[name], float(lo), float(hi),
                                            g.ctypes.data_as(_dp), len(g), out.ctypes.data_as(_dp)))
        return out

    def set_sample_slot(self, r, slot, sample):
        """The M-step sample of restart r for parameter slot `slot` of param_search_multi."""
        s64 = _i64(sample)
        if s64.shape != (self.num_segments,):
            raise ValueError('sample must have length num_segments')
        self._ck(self._lib.rmx_set_sample_slot(self._handle, int(r), int(slot), s64.ctypes.data_as(_ip)))

    def set_sample_lists(self, entries):
        """M-step samples of several restarts in one call (rmx_set_sample_lists).  `entries`: (restart, slot, mask, indices) with
        slot -1 = the restart's current sample (what _use_sample uploads) or 0..3 = its parameter slot of param_search_multi;
        `indices` = np.flatnonzero(mask), `mask` the dense array the per-restart calls take (kept for _use_sample's identity test)."""
        if not entries:
            return
        rl = np.array([e[0] for e in entries], dtype=np.int32)
        sl = np.array([e[1] for e in entries], dtype=np.int32)
        idx = [np.ascontiguousarray(e[3], dtype=np.int32).ravel() for e in entries]
        off = np.zeros(len(entries) + 1, dtype=np.int32)
        off[1:] = np.cumsum([len(i) for i in idx])
        flat = np.concatenate(idx) if len(idx) else np.zeros(0, dtype=np.int32)
        flat = np.ascontiguousarray(flat, dtype=np.int32)
        self._ck(self._lib.rmx_set_sample_lists(self._handle, len(entries), rl.ctypes.data_as(_i32p), sl.ctypes.data_as(_i32p),
                                                off.ctypes.data_as(_i32p), flat.ctypes.data_as(_i32p)))
        for r, slot, mask, _ in entries:
            if slot < 0:
                self._samples[int(r)] = mask

    def param_search_multi(self, restarts, names, los, his, grids):
        """The searches of param_search for several of the four standard likelihood parameters at once
        (rmx_param_search_multi; samples from set_sample_slot, slot = position in `names`).  Returns
        (xopt, lastval), each [len(names)][len(restarts)]; the model is not modified.  Raises
        NotImplementedError when the request does not qualify -- fall back to param_search."""
        rl = np.ascontiguousarray(restarts, dtype=np.int32)
        ids = np.ascontiguousarray([PARAM_IDS[n] for n in names], dtype=np.int32)
        lo = _f64(los).ravel(); hi = _f64(his).ravel()
        g = _f64(grids)
        if g.ndim != 2 or g.shape[0] != len(ids) or len(lo) != len(ids) or len(hi) != len(ids):
            raise ValueError('one grid, lower and upper bound per parameter')
        xopt = np.zeros((len(ids), len(rl))); last = np.zeros((len(ids), len(rl)))
        self._ck(self._lib.rmx_param_search_multi(self._handle, len(rl), rl.ctypes.data_as(_i32p), len(ids), ids.ctypes.data_as(_i32p),
                                                  lo.ctypes.data_as(_dp), hi.ctypes.data_as(_dp), g.ctypes.data_as(_dp), g.shape[1],
                                                  xopt.ctypes.data_as(_dp), last.ctypes.data_as(_dp)))
        return xopt, last

    def expected_log_likelihood_h_batch(self, restarts, hs):
        """(E[ll], dE[ll]/dh) on each listed restart's current sample with h set to the matching row
        of `hs` (and left there): the evaluation round of the lock-step h M-step."""
        rl = np.ascontiguousarray(restarts, dtype=np.int32)
        h = _f64(hs).reshape(len(rl), self.num_clones)
        out = np.zeros((len(rl), 1 + MAX_CLONES), dtype=np.float64)
        self._ck(self._lib.rmx_expected_ll_h_batch(self._handle, len(rl), rl.ctypes.data_as(_i32p),
                                                   h.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
        return out[:, 0].copy(), out[:, 1:1 + self.num_clones].copy()

    def h_batch_evaluator(self, restarts):
        """`evaluate(ids, xs)` for lockstep.run_lockstep over the listed restarts: (-E[ll], -dE[ll]/dh) of restart restarts[ids[k]] at
        h = xs[k], through expected_log_likelihood_h_batch's entry point with the argument buffers and their ctypes views made once
        (a round of the lock-step h M-step is a few hundred microseconds; the generic wrapper's conversions were a quarter of it)."""
        live = [int(r) for r in restarts]
        n, M = len(live), self.num_clones
        rl = np.zeros(max(n, 1), dtype=np.int32)
        h = np.zeros((max(n, 1), M), dtype=np.float64)
        out = np.zeros((max(n, 1), 1 + MAX_CLONES), dtype=np.float64)
        p_rl, p_h, p_out = rl.ctypes.data_as(_i32p), h.ctypes.data_as(_dp), out.ctypes.data_as(_dp)
        fn, handle, lib = self._lib.rmx_expected_ll_h_batch, self._handle, self._lib

        neg = np.zeros((max(n, 1), 1 + MAX_CLONES), dtype=np.float64)

        def evaluate(ids, xs):
            """(-E[ll] (k,), -dE[ll]/dh (k, M)) at the rows of xs (a (k, M) array or a list of k vectors); the arrays are views
            of buffers the next call overwrites."""
            k = len(ids)
            for j in range(k):
                rl[j] = live[ids[j]]
            h[:k] = xs
            rc = fn(handle, k, p_rl, p_h, p_out)
            if rc:
                _raise(lib, rc)
            np.negative(out[:k], out=neg[:k])
            return neg[:k, 0], neg[:k, 1:1 + M]
        return evaluate

    def expected_log_likelihood_full(self, r0=None, r1=None):
        """E[ll] over all segments for restarts [r0, r1)."""
        return self._scalar_range(self._lib.rmx_expected_ll_full, r0, r1)

    def expected_log_likelihood_full_trial(self, r0=None, r1=None):
        """E[ll] over all segments at h / parameter values that are on trial (rmx_expected_ll_full_trial):
        nothing of the restarts' committed state changes; follow with set_param / set h (accept) or
        rollback_param / rollback_h (reject)."""
        return self._scalar_range(self._lib.rmx_expected_ll_full_trial, r0, r1)

    # component of E[ll] each of the four standard likelihood parameters moves (column of expected_log_likelihood_components)
    PARAM_COMPONENT = {'negbin_r_0': 0, 'negbin_r_1': 1, 'betabin_M_0': 2, 'betabin_M_1': 3}

    def expected_log_likelihood_components(self, r0=None, r1=None, trial=False):
        """[r1-r0][4]: E[ll] over all segments split into the components negbin_r_0, negbin_r_1, betabin_M_0, betabin_M_1 move,
        at the committed values or (trial=True) with changed parameters on trial (rmx_expected_ll_components); trial=2: from the
        scratch expectations the last trial pass left behind, without another pass."""
        r0, r1 = self._range(r0, r1)
        out = np.zeros((r1 - r0, 4), dtype=np.float64)
        self._ck(self._lib.rmx_expected_ll_components(self._handle, r0, r1, int(trial), out.ctypes.data_as(_dp)))
        return out

    def rollback_param(self, r, name, value):
        v = _f64([value])
        self._ck(self._lib.rmx_trial_rollback(self._handle, int(r), PARAM_IDS[name], v.ctypes.data_as(_dp)))

    def rollback_h(self, r, h):
        v = _f64(h).ravel()
        if len(v) != self.num_clones:
            raise ValueError('h must have one entry per clone')
        self._ck(self._lib.rmx_trial_rollback(self._handle, int(r), -1, v.ctypes.data_as(_dp)))

    def infer_cn(self, r):
        cn = np.zeros((self.num_segments, self.num_clones, 2), dtype=np.int64)
        lp = C.c_double(0.)
        self._ck(self._lib.rmx_infer_cn(self._handle, r, cn.ctypes.data_as(_ip), C.byref(lp)))
        return cn, float(lp.value)

    def infer_cn_batch(self, r0, nr):
        """Viterbi decode of restarts r0 .. r0+nr-1 in one call: (cn [nr][N][M][2], logprob [nr])."""
        cn = np.zeros((nr, self.num_segments, self.num_clones, 2), dtype=np.int64)
        lp = np.zeros(nr)
        self._ck(self._lib.rmx_infer_cn_batch(self._handle, int(r0), int(nr), cn.ctypes.data_as(_ip), lp.ctypes.data_as(_dp)))
        return cn, lp

    # -- measurement ------------------------------------------------------------
    def timer_start(self):
        self._ck(self._lib.rmx_timer_start(self._handle))

    def timer_stop(self):
        ms = C.c_double(0.)
        self._ck(self._lib.rmx_timer_stop(self._handle, C.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on=True):
        self._ck(self._lib.rmx_profile_enable(self._handle, int(on)))    # 0 off, 1 all kernels, 2 sweep kernels only

    def profile_reset(self):
        self._ck(self._lib.rmx_profile_reset(self._handle))

    def profile(self):
        """{kernel name: (total_ms, launches)} since the last reset."""
        out = {}
        for i in range(self._lib.rmx_num_kernels()):
            ms = C.c_double(0.); n = C.c_int64(0)
            self._ck(self._lib.rmx_profile_get(self._handle, i, C.byref(ms), C.byref(n)))
            if n.value:
                out[self._lib.rmx_kernel_name(i).decode()] = (float(ms.value), int(n.value))
        return out


class RemixtModel(object):
    """Drop-in for remixt.bpmodel.RemixtModel (bpmodel.pyx:397)."""

    _own = ('_batch', '_r', '_owns_batch')

    def __init__(self, num_clones, num_segments, num_breakpoints, normal_contamination,
                 cn_states, brk_states, h_init, l, x, y, is_telomere, breakpoint_idx,
                 breakpoint_orient, transition_penalty, divergence_weight, device=0):
        cn_states = np.asarray(cn_states)
        if cn_states.ndim != 4 or cn_states.shape[0] != num_segments or cn_states.shape[2] != num_clones or cn_states.shape[3] != 2:
            raise ValueError('cn_states must have shape (num_segments, num_cn_states, num_clones, num_alleles)')
        classes, seg_class = compress_cn_states(cn_states)
        batch = RemixtBatch(num_clones, num_segments, num_breakpoints, normal_contamination,
                            classes, seg_class, brk_states, np.asarray(h_init, dtype=np.float64)[None, :], l, x, y,
                            is_telomere, breakpoint_idx, breakpoint_orient, transition_penalty,
                            [float(divergence_weight)], device=device)
        object.__setattr__(self, '_batch', batch)
        object.__setattr__(self, '_r', 0)
        object.__setattr__(self, '_owns_batch', True)

    @classmethod
    def from_classes(cls, num_clones, num_segments, num_breakpoints, normal_contamination,
                     cn_classes, seg_class, brk_states, h_init, l, x, y, is_telomere, breakpoint_idx,
                     breakpoint_orient, transition_penalty, divergence_weight, device=0):
        """Same as the constructor with the state array given class-compressed
        (cn_classes (C,S,M,2), seg_class (N,)): avoids building the (N,S,M,2) array."""
        batch = RemixtBatch(num_clones, num_segments, num_breakpoints, normal_contamination,
                            cn_classes, seg_class, brk_states, np.asarray(h_init, dtype=np.float64)[None, :], l, x, y,
                            is_telomere, breakpoint_idx, breakpoint_orient, transition_penalty,
                            [float(divergence_weight)], device=device)
        self = cls.__new__(cls)
        object.__setattr__(self, '_batch', batch)
        object.__setattr__(self, '_r', 0)
        object.__setattr__(self, '_owns_batch', True)
        return self

    @classmethod
    def _from_batch(cls, batch, r):
        self = cls.__new__(cls)
        object.__setattr__(self, '_batch', batch)
        object.__setattr__(self, '_r', int(r))
        object.__setattr__(self, '_owns_batch', False)
        return self

    # -- attribute protocol ---------------------------------------------------------
    def __getattr__(self, name):
        b = object.__getattribute__(self, '_batch')
        r = object.__getattribute__(self, '_r')
        if name in ARRAY_IDS:
            return b.get_array(r, name)
        if name in PARAM_IDS:
            return b.get_param(r, name)
        if name in _STATE_TABLES:
            return b.state_table(name)
        if name == 'cn_states':
            return b.cn_classes[b.seg_class]
        if name in ('num_clones', 'num_segments', 'num_breakpoints', 'num_alleles', 'num_cn_states', 'num_brk_states',
                    'cn_max', 'normal_contamination', 'transition_penalty', 'transition_model', 'brk_states',
                    'is_telomere', 'breakpoint_idx', 'breakpoint_orient', 'l', 'x', 'y'):
            return getattr(b, name)
        if name == 'breakpoint_side':
            side = np.zeros(b.num_segments, dtype=np.int64)
            seen = np.zeros(max(b.num_breakpoints, 1), dtype=np.int64)
            for n in range(b.num_segments):
                k = b.breakpoint_idx[n]
                if k < 0:
                    continue
                side[n] = seen[k]
                seen[k] += 1
            return side
        raise AttributeError(name)

    def __setattr__(self, name, value):
        b = object.__getattribute__(self, '_batch')
        r = object.__getattribute__(self, '_r')
        if name in ARRAY_IDS:
            b.set_array(r, name, value)
        elif name in PARAM_IDS:
            if name == 'hmm_log_norm_const':
                raise AttributeError('hmm_log_norm_const is read-only')
            b.set_param(r, name, value)
        elif name == 'transition_model':
            b.transition_model = value
        else:
            raise AttributeError('cannot set attribute %r' % name)

    def __dir__(self):
        return sorted(set(list(ARRAY_IDS) + list(PARAM_IDS) + list(_STATE_TABLES) + [
            'cn_states', 'num_clones', 'num_segments', 'num_breakpoints', 'num_alleles', 'num_cn_states',
            'num_brk_states', 'cn_max', 'normal_contamination', 'transition_penalty', 'transition_model',
            'brk_states', 'is_telomere', 'breakpoint_idx', 'breakpoint_orient', 'breakpoint_side', 'l', 'x', 'y']))

    # -- methods (cpdef surface of bpmodel.pyx) ----------------------------------------
    def update_framelogprob(self):
        self._batch.update_framelogprob(self._r, self._r + 1)

    def update_p_cn(self):
        self._batch.update_p_cn(self._r, self._r + 1)

    def update_p_breakpoint(self):
        self._batch.update_p_breakpoint(self._r, self._r + 1)

    def update_p_outlier_total(self):
        self._batch.update_p_outlier_total(self._r, self._r + 1)

    def update_p_outlier_allele(self):
        self._batch.update_p_outlier_allele(self._r, self._r + 1)

    def update_p_allele_swap(self):
        self._batch.update_p_allele_swap(self._r, self._r + 1)

    def calculate_elbo(self):
        return float(self._batch.calculate_elbo(self._r, self._r + 1)[0])

    def calculate_variational_energy(self):
        return float(self._batch.calculate_variational_energy(self._r, self._r + 1)[0])

    def calculate_variational_entropy(self):
        return float(self._batch.calculate_variational_entropy(self._r, self._r + 1)[0])

    def calculate_expected_log_likelihood(self, sample):
        return self._batch.expected_log_likelihood(self._r, sample)[0]

    def calculate_expected_log_likelihood_param_grid(self, name, values, sample):
        """E[ll] for each value of the likelihood parameter `name` (left at values[-1]), one round trip."""
        return self._batch.expected_log_likelihood_param_grid(self._r, name, values, sample)

    def calculate_expected_log_likelihood_partial_h(self, sample, partial_h):
        _, g = self._batch.expected_log_likelihood(self._r, sample, want_grad=True)
        partial_h[:] = g

    def calculate_log_transmat(self, log_transmat):
        """bpmodel.pyx:639-684 with the *current* p_breakpoint, into the caller's (N-1, S, S) array."""
        b = self._batch
        out = np.ascontiguousarray(log_transmat, dtype=np.float64)
        if out.shape != (b.num_segments - 1, b.num_cn_states, b.num_cn_states):
            raise ValueError('log_transmat must have shape (num_segments - 1, num_cn_states, num_cn_states)')
        b._ck(b._lib.rmx_calculate_log_transmat(b._handle, self._r, out.ctypes.data_as(_dp)))
        if out is not log_transmat:
            np.asarray(log_transmat)[...] = out

    def calculate_log_likelihood_total(self, n, s, u):
        out = C.c_double(0.)
        b = self._batch
        b._ck(b._lib.rmx_log_likelihood_total(b._handle, self._r, int(n), int(s), int(u), C.byref(out)))
        return float(out.value)

    def calculate_log_likelihood_allele(self, n, s, v, w):
        out = C.c_double(0.)
        b = self._batch
        b._ck(b._lib.rmx_log_likelihood_allele(b._handle, self._r, int(n), int(s), int(v), int(w), C.byref(out)))
        return float(out.value)

    # the remaining per-cell cpdef methods (bpmodel.pyx:686-749, 778-807, 855-896); partial_h is the caller's (M,) array
    def _cell(self, which, n, s, u=0, v=0, w=0):
        b = self._batch
        if not (0 <= int(n) < b.num_segments and 0 <= int(s) < b.num_cn_states):
            raise IndexError('segment / state index out of range')       # the reference's bounds-checked memoryviews
        out = np.zeros(MAX_CLONES)
        b._ck(b._lib.rmx_cell_quantity(b._handle, self._r, int(n), int(s), which, int(u), int(v), int(w), out.ctypes.data_as(_dp)))
        return out

    def calculate_expected_total_reads(self, n, s):
        return float(self._cell(0, n, s)[0])

    def calculate_expected_total_reads_partial_h(self, n, s, partial_h):
        partial_h[:] = self._cell(1, n, s)[:self._batch.num_clones]

    def calculate_expected_allele_ratio(self, n, s):
        return float(self._cell(2, n, s)[0])

    def calculate_expected_allele_ratio_partial_h(self, n, s, partial_h):
        partial_h[:] = self._cell(3, n, s)[:self._batch.num_clones]

    def calculate_log_prior_cn(self, n, s):
        return float(self._cell(4, n, s)[0])

    def calculate_log_likelihood_total_partial_h(self, n, s, u, partial_h):
        partial_h[:] = self._cell(5, n, s, u=u)[:self._batch.num_clones]

    def calculate_log_likelihood_allele_partial_h(self, n, s, v, w, partial_h):
        partial_h[:] = self._cell(6, n, s, v=v, w=w)[:self._batch.num_clones]

    def infer_cn(self, cn):
        out, _ = self._batch.infer_cn(self._r)
        cn[...] = out


def sum_product(framelogprob, log_transmat, alphas, betas, device=0):
    """bpmodel.pyx:1213-1246 on caller-supplied dense inputs (HIP)."""
    lib = _lib.load()
    f = _f64(framelogprob); T = _f64(log_transmat)
    N, S = f.shape
    if T.shape != (N - 1, S, S):
        raise ValueError('log_transmat must have shape (N-1, S, S)')
    a = np.zeros_like(f); b = np.zeros_like(f)
    Tp = T if T.size else np.zeros(1)
    rc = lib.rmx_sum_product(f.ctypes.data_as(_dp), Tp.ctypes.data_as(_dp), a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), N, S, device)
    if rc:
        _raise(lib, rc)
    alphas[...] = a
    betas[...] = b


def max_product(framelogprob, log_transmat, state_sequence, device=0):
    """bpmodel.pyx:1296-1333 on caller-supplied dense inputs (HIP); returns the log probability."""
    lib = _lib.load()
    f = _f64(framelogprob); T = _f64(log_transmat)
    N, S = f.shape
    if T.shape != (N - 1, S, S):
        raise ValueError('log_transmat must have shape (N-1, S, S)')
    ss = np.zeros(N, dtype=np.int64)
    lp = C.c_double(0.)
    Tp = T if T.size else np.zeros(1)
    rc = lib.rmx_max_product(f.ctypes.data_as(_dp), Tp.ctypes.data_as(_dp), ss.ctypes.data_as(_ip), C.byref(lp), N, S, device)
    if rc:
        _raise(lib, rc)
    state_sequence[...] = ss
    return float(lp.value)
