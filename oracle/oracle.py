"""ctypes front-end of oracle/remixt_oracle.c (the CPU restatement).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package remixt_amd/.

`RemixtModel` here mirrors the attribute / method protocol of the reference's
Cython class (remixt/bpmodel.pyx:397-1210) closely enough that the same driver
code (tests, BreakpointModel host) can run on the reference binary, on this
oracle and on the HIP backend and diff the results.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libremixt_oracle.so")
# tests/test_sanitizers_cpu.py: an ASan / UBSan build of the same source, made by the test, loaded instead
PREBUILT = os.environ.get("RMX_ORACLE_LIB")
if PREBUILT:
    LIB_PATH = PREBUILT


def build(force=False):
    if PREBUILT:
        return LIB_PATH
    src = os.path.join(HERE, "remixt_oracle.c")
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= os.path.getmtime(src)):
        return LIB_PATH
    # -ffp-contract=off: keep the reference's (x86-64 gcc, no FMA) rounding sequence
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off",
                           "-o", LIB_PATH, src, "-lm"])
    return LIB_PATH


_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    L.rmxo_create.restype = C.c_void_p
    L.rmxo_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int, _ip, C.c_int,
                              _dp, _dp, _dp, _dp, _ip, _ip, _ip, C.c_double, C.c_double]
    L.rmxo_destroy.argtypes = [C.c_void_p]
    for name in ("rmxo_update_framelogprob", "rmxo_update_p_cn", "rmxo_update_p_breakpoint",
                 "rmxo_update_p_outlier_total", "rmxo_update_p_outlier_allele", "rmxo_update_p_allele_swap",
                 "rmxo_err"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("rmxo_variational_entropy", "rmxo_variational_energy", "rmxo_calculate_elbo"):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [C.c_void_p]
    L.rmxo_calculate_log_transmat.argtypes = [C.c_void_p, _dp]
    L.rmxo_calculate_log_transmat.restype = None
    L.rmxo_expected_log_likelihood.restype = C.c_double
    L.rmxo_expected_log_likelihood.argtypes = [C.c_void_p, _ip]
    L.rmxo_expected_log_likelihood_partial_h.restype = C.c_int
    L.rmxo_expected_log_likelihood_partial_h.argtypes = [C.c_void_p, _ip, _dp]
    L.rmxo_infer_cn.restype = C.c_int
    L.rmxo_infer_cn.argtypes = [C.c_void_p, _ip, _ip]
    L.rmxo_log_likelihood_total.restype = C.c_double
    L.rmxo_log_likelihood_total.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.rmxo_log_likelihood_allele.restype = C.c_double
    L.rmxo_log_likelihood_allele.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.rmxo_cell_quantity.restype = C.c_int
    L.rmxo_cell_quantity.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    L.rmxo_sum_product.restype = None
    L.rmxo_sum_product.argtypes = [_dp, _dp, _dp, _dp, C.c_int, C.c_int]
    L.rmxo_max_product.restype = C.c_double
    L.rmxo_max_product.argtypes = [_dp, _dp, _ip, C.c_int, C.c_int]
    L.rmxo_errmsg.restype = C.c_char_p
    L.rmxo_errmsg.argtypes = [C.c_void_p]
    L.rmxo_clear_err.argtypes = [C.c_void_p]
    L.rmxo_dim.restype = C.c_int
    L.rmxo_dim.argtypes = [C.c_void_p, C.c_int]
    L.rmxo_array.restype = C.c_void_p
    L.rmxo_array.argtypes = [C.c_void_p, C.c_int]
    L.rmxo_scalar.restype = _dp
    L.rmxo_scalar.argtypes = [C.c_void_p, C.c_int]
    L.rmxo_transition_model.restype = C.POINTER(C.c_int)
    L.rmxo_transition_model.argtypes = [C.c_void_p]
    for name, nargs in (("rmxo_digamma", 1), ("rmxo_negbin_ll", 3), ("rmxo_negbin_ll_partial_mu", 3),
                        ("rmxo_betabin_ll", 4), ("rmxo_betabin_ll_partial_p", 4)):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [C.c_double] * nargs
    _lib = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _pd(a):
    return a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


_ARRAYS = {  # name -> (id, dtype, shape lambda)
    "h": (0, np.float64, lambda m: (m.num_clones,)),
    "p_breakpoint": (1, np.float64, lambda m: (m.num_breakpoints, m.num_brk_states)),
    "framelogprob": (2, np.float64, lambda m: (m.num_segments, m.num_cn_states)),
    "log_transmat": (3, np.float64, lambda m: (m.num_segments - 1, m.num_cn_states, m.num_cn_states)),
    "cached_log_transmat": (4, np.float64, lambda m: (m.num_segments - 1, m.num_cn_states, m.num_cn_states)),
    "posterior_marginals": (5, np.float64, lambda m: (m.num_segments, m.num_cn_states)),
    "joint_posterior_marginals": (6, np.float64, lambda m: (m.num_segments - 1, m.num_cn_states, m.num_cn_states)),
    "p_allele_swap": (7, np.float64, lambda m: (m.num_segments, 2)),
    "p_outlier_total": (8, np.float64, lambda m: (m.num_segments, 2)),
    "p_outlier_allele": (9, np.float64, lambda m: (m.num_segments, 2)),
    "total_likelihood_mask": (10, np.int64, lambda m: (m.num_segments,)),
    "allele_likelihood_mask": (11, np.int64, lambda m: (m.num_segments,)),
    "cn_states_total": (12, np.int64, lambda m: (m.num_segments, m.num_cn_states, m.num_clones)),
    "num_alleles_subclonal": (13, np.int64, lambda m: (m.num_segments, m.num_cn_states)),
    "is_hdel": (14, np.int64, lambda m: (m.num_segments, m.num_cn_states)),
    "is_loh": (15, np.int64, lambda m: (m.num_segments, m.num_cn_states)),
    "breakpoint_side": (16, np.int64, lambda m: (m.num_segments,)),
    "cn_states": (17, np.int64, lambda m: (m.num_segments, m.num_cn_states, m.num_clones, 2)),
    "brk_states": (18, np.int64, lambda m: (m.num_brk_states, m.num_clones)),
    "is_telomere": (19, np.int64, lambda m: (m.num_segments,)),
    "breakpoint_idx": (20, np.int64, lambda m: (m.num_segments,)),
    "breakpoint_orient": (21, np.int64, lambda m: (m.num_segments,)),
    "l": (22, np.float64, lambda m: (m.num_segments,)),
    "x": (23, np.float64, lambda m: (m.num_segments,)),
    "y": (24, np.float64, lambda m: (m.num_segments, 2)),
}
_SCALARS = {
    "negbin_r_0": 0, "negbin_r_1": 1, "negbin_hdel_mu": 2, "negbin_hdel_r_0": 3, "negbin_hdel_r_1": 4,
    "betabin_M_0": 5, "betabin_M_1": 6, "betabin_loh_p": 7, "betabin_loh_M_0": 8, "betabin_loh_M_1": 9,
    "prior_outlier_total": 10, "prior_outlier_allele": 11, "hmm_log_norm_const": 12,
    "transition_penalty": 13, "divergence_weight": 14,
}


class RemixtModel(object):
    """Oracle twin of remixt.bpmodel.RemixtModel (bpmodel.pyx:397)."""

    def __init__(self, num_clones, num_segments, num_breakpoints, normal_contamination,
                 cn_states, brk_states, h_init, l, x, y, is_telomere, breakpoint_idx,
                 breakpoint_orient, transition_penalty, divergence_weight):
        L = lib()
        cn_states = _i64(cn_states)
        brk_states = _i64(brk_states)
        if cn_states.ndim != 4 or brk_states.ndim != 2:
            raise ValueError("bad state table rank")
        S = cn_states.shape[1]
        B = brk_states.shape[0]
        # validation of bpmodel.pyx:509-529
        if (cn_states.shape[0] != num_segments or cn_states.shape[2] != num_clones or cn_states.shape[3] != 2):
            raise ValueError('cn_states must have shape (num_segments, num_cn_states, num_clones, num_alleles)')
        if brk_states.shape[1] != num_clones:
            raise ValueError('cn_states must have shape (num_brk_states, num_clones)')
        h_init = _f64(h_init)
        if h_init.shape[0] != num_clones:
            raise ValueError('h must have length equal to num_clones')
        is_telomere = _i64(is_telomere)
        breakpoint_idx = _i64(breakpoint_idx)
        breakpoint_orient = _i64(breakpoint_orient)
        if is_telomere.shape[0] != num_segments:
            raise ValueError('is_telomere must have length equal to num_segments')
        if breakpoint_idx.shape[0] != num_segments:
            raise ValueError('breakpoint_idx must have length equal to num_segments')
        if breakpoint_orient.shape[0] != num_segments:
            raise ValueError('breakpoint_orient must have length equal to num_segments')
        if breakpoint_idx.max() + 1 != num_breakpoints:
            raise ValueError('breakpoint_idx must have maximum of num_breakpoints positive indices')
        l = _f64(l); x = _f64(x); y = _f64(y)
        object.__setattr__(self, "_h", None)
        self._L = L
        self._p = L.rmxo_create(int(num_clones), int(num_segments), int(num_breakpoints), int(bool(normal_contamination)),
                                _pi(cn_states), S, _pi(brk_states), B, _pd(h_init), _pd(l), _pd(x), _pd(y),
                                _pi(is_telomere), _pi(breakpoint_idx), _pi(breakpoint_orient),
                                float(transition_penalty), float(divergence_weight))
        if not self._p:
            raise ValueError("oracle create failed")
        self.num_clones = L.rmxo_dim(self._p, 0)
        self.num_segments = L.rmxo_dim(self._p, 1)
        self.num_breakpoints = L.rmxo_dim(self._p, 2)
        self.num_cn_states = L.rmxo_dim(self._p, 3)
        self.num_brk_states = L.rmxo_dim(self._p, 4)
        self.cn_max = L.rmxo_dim(self._p, 5)
        self.num_alleles = 2
        self.normal_contamination = bool(normal_contamination)

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                self._L.rmxo_destroy(self._p)
                self._p = None
        except Exception:
            pass

    # -- attribute protocol --------------------------------------------------
    def _view(self, name):
        aid, dt, shp = _ARRAYS[name]
        shape = shp(self)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dt)
        ptr = self._L.rmxo_array(self._p, aid)
        ct = C.c_double if dt == np.float64 else C.c_int64
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).reshape(shape)

    def __getattr__(self, name):
        if name in _ARRAYS:
            return self._view(name)
        if name in _SCALARS:
            return float(self._L.rmxo_scalar(self._p, _SCALARS[name])[0])
        if name == "transition_model":
            return int(self._L.rmxo_transition_model(self._p)[0])
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _ARRAYS:
            self._view(name)[...] = np.asarray(value)
        elif name in _SCALARS:
            self._L.rmxo_scalar(self._p, _SCALARS[name])[0] = float(value)
        elif name == "transition_model":
            self._L.rmxo_transition_model(self._p)[0] = int(value)
        else:
            object.__setattr__(self, name, value)

    def _check(self):
        e = self._L.rmxo_err(self._p)
        if e:
            msg = self._L.rmxo_errmsg(self._p).decode()
            self._L.rmxo_clear_err(self._p)
            if e == 2:
                raise AssertionError(msg)
            raise ValueError(msg)

    # -- methods (bpmodel.pyx cpdef surface) ----------------------------------
    def update_framelogprob(self):
        self._L.rmxo_update_framelogprob(self._p); self._check()

    def calculate_log_transmat(self, out):
        assert out.dtype == np.float64 and out.flags.c_contiguous
        self._L.rmxo_calculate_log_transmat(self._p, _pd(out)); self._check()

    def update_p_cn(self):
        self._L.rmxo_update_p_cn(self._p); self._check()

    def update_p_breakpoint(self):
        self._L.rmxo_update_p_breakpoint(self._p); self._check()

    def update_p_outlier_total(self):
        self._L.rmxo_update_p_outlier_total(self._p); self._check()

    def update_p_outlier_allele(self):
        self._L.rmxo_update_p_outlier_allele(self._p); self._check()

    def update_p_allele_swap(self):
        self._L.rmxo_update_p_allele_swap(self._p); self._check()

    def calculate_variational_entropy(self):
        v = self._L.rmxo_variational_entropy(self._p); self._check(); return v

    def calculate_variational_energy(self):
        v = self._L.rmxo_variational_energy(self._p); self._check(); return v

    def calculate_elbo(self):
        v = self._L.rmxo_calculate_elbo(self._p); self._check(); return v

    def calculate_expected_log_likelihood(self, sample):
        s = _i64(sample)
        v = self._L.rmxo_expected_log_likelihood(self._p, _pi(s)); self._check(); return v

    def calculate_expected_log_likelihood_partial_h(self, sample, partial_h):
        s = _i64(sample)
        out = np.zeros(self.num_clones)
        self._L.rmxo_expected_log_likelihood_partial_h(self._p, _pi(s), _pd(out)); self._check()
        partial_h[:] = out

    def calculate_log_likelihood_total(self, n, s, u):
        v = self._L.rmxo_log_likelihood_total(self._p, n, s, u); self._check(); return v

    def calculate_log_likelihood_allele(self, n, s, v, w):
        r = self._L.rmxo_log_likelihood_allele(self._p, n, s, v, w); self._check(); return r

    # remaining per-cell cpdef methods (bpmodel.pyx:686-749, 778-807, 855-896)
    def _cell(self, which, n, s, u=0, v=0, w=0):
        out = np.zeros(8)
        self._L.rmxo_cell_quantity(self._p, int(n), int(s), which, int(u), int(v), int(w), _pd(out)); self._check()
        return out

    def calculate_expected_total_reads(self, n, s):
        return float(self._cell(0, n, s)[0])

    def calculate_expected_total_reads_partial_h(self, n, s, partial_h):
        partial_h[:] = self._cell(1, n, s)[:self.num_clones]

    def calculate_expected_allele_ratio(self, n, s):
        return float(self._cell(2, n, s)[0])

    def calculate_expected_allele_ratio_partial_h(self, n, s, partial_h):
        partial_h[:] = self._cell(3, n, s)[:self.num_clones]

    def calculate_log_prior_cn(self, n, s):
        return float(self._cell(4, n, s)[0])

    def calculate_log_likelihood_total_partial_h(self, n, s, u, partial_h):
        partial_h[:] = self._cell(5, n, s, u=u)[:self.num_clones]

    def calculate_log_likelihood_allele_partial_h(self, n, s, v, w, partial_h):
        partial_h[:] = self._cell(6, n, s, v=v, w=w)[:self.num_clones]

    def infer_cn(self, cn):
        out = np.zeros((self.num_segments, self.num_clones, 2), dtype=np.int64)
        out[...] = cn
        ss = np.zeros(self.num_segments, dtype=np.int64)
        self._L.rmxo_infer_cn(self._p, _pi(out), _pi(ss)); self._check()
        cn[...] = out
        return ss


def sum_product(framelogprob, log_transmat, alphas, betas):
    f = _f64(framelogprob); T = _f64(log_transmat)
    a = np.zeros_like(f); b = np.zeros_like(f)
    lib().rmxo_sum_product(_pd(f), _pd(T), _pd(a), _pd(b), f.shape[0], f.shape[1])
    alphas[...] = a; betas[...] = b


def max_product(framelogprob, log_transmat, state_sequence):
    f = _f64(framelogprob); T = _f64(log_transmat)
    ss = np.zeros(f.shape[0], dtype=np.int64)
    lp = lib().rmxo_max_product(_pd(f), _pd(T), _pi(ss), f.shape[0], f.shape[1])
    state_sequence[...] = ss
    return lp


def digamma(x):
    return lib().rmxo_digamma(float(x))


def negbin_ll(x, mu, r):
    return lib().rmxo_negbin_ll(float(x), float(mu), float(r))


def betabin_ll(k, n, p, M):
    return lib().rmxo_betabin_ll(float(k), float(n), float(p), float(M))
