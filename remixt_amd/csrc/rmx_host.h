// rmx_host.h -- the pieces of the C ABI that never touch the device, as plain C++ (no HIP types), so that the CPU test
// suite can compile them with gcc -fsanitize=address,undefined (tests/csrc/host_sanitize.cpp, tests/test_sanitizers_cpu.py):
//   compress_cn_states   rmx_compress_cn_states: dense (N,S,M,2) int64 state array -> class tables + class id per segment
//   weighted_search      rmx_weighted_search: numpy's cumsum / searchsorted(side='right') for the weighted M-step samples
//                        (reference remixt/cn_model.py:475-480)
//   weighted_sample_round  rmx_weighted_sample_round: a whole round of that sampling from a strided weight column
//   Nm1                  scipy.optimize.fmin (Nelder-Mead, one variable) as a resumable state machine: the polish stage
//                        of scipy.optimize.brute in BreakpointModel.update_param (reference remixt/cn_model.py:553-561)
// Status codes: 0 ok, 4 = RMX_EUNSUPPORTED, 5 = RMX_EARG (include/remixt_amd.h).
#ifndef RMX_HOST_H
#define RMX_HOST_H
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

// (Nm1 also runs inside a kernel: k_param_search_nm)
#if defined(__HIPCC__)
#define RMX_HD __host__ __device__
#else
#define RMX_HD
#endif

namespace rmxh {

inline int compress_cn_states(const int64_t *cn_states, int32_t N, int32_t S, int32_t M, int32_t max_classes,
                              int32_t *seg_class_out, int64_t *classes_out, int32_t *num_classes) {
    if (!cn_states || !seg_class_out || !classes_out || !num_classes) return 5;
    const size_t tsz = (size_t)S * M * 2;
    int C = 0;
    for (int n = 0; n < N; n++) {
        const int64_t *t = cn_states + (size_t)n * tsz;
        int found = -1;
        if (n > 0 && memcmp(t, classes_out + (size_t)seg_class_out[n - 1] * tsz, tsz * 8) == 0) found = seg_class_out[n - 1];
        for (int c = 0; c < C && found < 0; c++) if (memcmp(t, classes_out + (size_t)c * tsz, tsz * 8) == 0) found = c;
        if (found < 0) {
            if (C >= max_classes) return 4;
            memcpy(classes_out + (size_t)C * tsz, t, tsz * 8);
            found = C++;
        }
        seg_class_out[n] = found;
    }
    *num_classes = C;
    return 0;
}

inline int weighted_search(const double *p, int64_t n, const double *u, int32_t k, int64_t *out, int64_t *positive) {
    if (!p || n < 1 || (k > 0 && (!u || !out))) return 5;
    std::vector<double> cdf((size_t)n);
    double acc = 0.;
    int64_t pos = 0;
    for (int64_t i = 0; i < n; i++) { acc += p[i]; cdf[(size_t)i] = acc; pos += p[i] > 0.; }
    const double last = cdf[(size_t)n - 1];
    for (int64_t i = 0; i < n; i++) cdf[(size_t)i] /= last;
    for (int32_t j = 0; j < k; j++) {
        const int64_t idx = (int64_t)(std::upper_bound(cdf.begin(), cdf.end(), u[j]) - cdf.begin());
        out[j] = std::min<int64_t>(idx, n - 1);
    }
    if (positive) *positive = pos;
    return 0;
}

// One round of the weighted M-step sampling (remixt_amd/cn_model.py _sample_without_replacement: `size` distinct indices, successively
// with probability proportional to the weights = numpy's choice(replace=False, p=...), reference remixt/cn_model.py:475-480): the
// weights are w[i * stride] / norm (a column of an (N, 2) indicator array, normalised like numpy's `weights / weights.sum()`), the
// k uniform draws u select indices from their cumulative sum exactly as weighted_search does, and the indices not seen before are
// appended to found[0 .. *nfound) in the order of the draws (numpy: unique(concatenate(found, new)) by first occurrence), up to cap.
inline int weighted_sample_round(const double *w, int64_t n, int64_t stride, double norm, const double *u, int32_t k,
                                 int64_t *found, int32_t *nfound, int32_t cap, int64_t *positive) {
    if (!w || n < 1 || stride < 1 || !(norm > 0.) || (k > 0 && !u) || !found || !nfound || *nfound < 0 || *nfound > cap) return 5;
    std::vector<double> cdf((size_t)n);
    double acc = 0.;
    int64_t pos = 0;
    for (int64_t i = 0; i < n; i++) { const double p = w[(size_t)(i * stride)] / norm; acc += p; cdf[(size_t)i] = acc; pos += p > 0.; }
    const double last = cdf[(size_t)n - 1];
    for (int64_t i = 0; i < n; i++) cdf[(size_t)i] /= last;
    if (positive) *positive = pos;
    std::vector<int64_t> seen(found, found + *nfound);
    std::sort(seen.begin(), seen.end());
    int32_t nf = *nfound;
    for (int32_t j = 0; j < k && nf < cap; j++) {
        int64_t idx = (int64_t)(std::upper_bound(cdf.begin(), cdf.end(), u[j]) - cdf.begin());
        idx = std::min<int64_t>(idx, n - 1);
        auto it = std::lower_bound(seen.begin(), seen.end(), idx);
        if (it != seen.end() && *it == idx) continue;
        seen.insert(it, idx);
        found[nf++] = idx;
    }
    *nfound = nf;
    return 0;
}

// scipy.optimize.fmin (Nelder-Mead, one variable, xatol = fatol = 1e-4, maxiter = maxfun = 200) as a
// resumable state machine: same floating-point operations in the same order as scipy's
// _minimize_neldermead (python twin: remixt_amd/lockstep.py fmin_1d; tests compare the two).
struct Nm1 {
    enum { START, W_INIT0, W_INIT1, W_XR, W_XE, W_XC, W_XCC, W_SHRINK, DONE };
    double s0 = 0, s1 = 0, f0 = INFINITY, f1 = INFINITY, xbar = 0, xr = 0, fxr = 0, xe = 0, xc = 0, xcc = 0, req = 0, last = 0;
    int fcalls = 0, iters = 0, state = START;
    static constexpr int maxfun = 200, maxiter = 200;
    static constexpr double xatol = 1e-4, fatol = 1e-4;
    RMX_HD bool request(double x, int next) { if (fcalls >= maxfun) return false; fcalls++; req = x; last = x; state = next; return true; }
    RMX_HD void sort() { if (f1 < f0) { double t_ = f0; f0 = f1; f1 = t_; t_ = s0; s0 = s1; s1 = t_; } }
    // feed the value of the last request (ignored on the first call); true = `req` holds the next point
    RMX_HD bool advance(double x0, double f) {
#pragma clang fp contract(off)
        switch (state) {
        case START:
            s0 = x0; s1 = x0 != 0. ? (1 + 0.05) * x0 : 0.00025;
            if (request(s0, W_INIT0)) return true;
            goto init_done;
        case W_INIT0:
            f0 = f;
            if (request(s1, W_INIT1)) return true;
            goto init_done;
        case W_INIT1:
            f1 = f;
            goto init_done;
        case W_XR:
            fxr = f;
            if (fxr < f0) {
                xe = 3. * xbar - 2. * s1;
                if (request(xe, W_XE)) return true;
                goto maxfun_exit;
            }
            // N = 1: fsim[-2] is fsim[0], so "fxr < fsim[-2]" cannot hold here
            if (fxr < f1) {
                xc = 1.5 * xbar - 0.5 * s1;
                if (request(xc, W_XC)) return true;
                goto maxfun_exit;
            }
            xcc = 0.5 * xbar + 0.5 * s1;
            if (request(xcc, W_XCC)) return true;
            goto maxfun_exit;
        case W_XE:
            if (f < fxr) { s1 = xe; f1 = f; } else { s1 = xr; f1 = fxr; }
            goto iter_done;
        case W_XC:
            if (f <= fxr) { s1 = xc; f1 = f; goto iter_done; }
            goto shrink;
        case W_XCC:
            if (f < f1) { s1 = xcc; f1 = f; goto iter_done; }
            goto shrink;
        case W_SHRINK:
            f1 = f;
            goto iter_done;
        default:
            return false;
        }
    shrink:
        s1 = s0 + 0.5 * (s1 - s0);
        if (request(s1, W_SHRINK)) return true;
        goto maxfun_exit;
    init_done:
        sort();
        iters = 1;
        goto loop_top;
    iter_done:
        iters++;
    maxfun_exit:
        sort();
    loop_top:
        if (fcalls < maxfun && iters < maxiter) {
            if (!(fabs(s1 - s0) <= xatol && fabs(f0 - f1) <= fatol)) {
                xbar = s0 / 1;
                xr = 2. * xbar - 1. * s1;
                if (request(xr, W_XR)) return true;
                goto maxfun_exit;      // cannot happen (fcalls < maxfun was just checked); mirrors the python flow
            }
        }
        state = DONE;
        return false;
    }
    RMX_HD double xopt() const { return s0; }
    // the points the NEXT call of advance() can request, whatever value the pending request gets (same
    // expressions as above): after the first initial point the second one; after a reflection the
    // expansion, the outside and the inside contraction
    int lookahead(double out[3]) const {
#pragma clang fp contract(off)
        if (state == W_INIT0) { out[0] = s1; return 1; }
        if (state == W_XR) { out[0] = 3. * xbar - 2. * s1; out[1] = 1.5 * xbar - 0.5 * s1; out[2] = 0.5 * xbar + 0.5 * s1; return 3; }
        return 0;
    }
};

}  // namespace rmxh
#endif
