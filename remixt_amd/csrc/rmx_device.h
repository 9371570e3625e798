// rmx_device.h -- device-side data model, math helpers and the per-cell
// read-count likelihoods of the ReMixT variational HMM, for gfx950.
//
// Reference behaviour (paths relative to the reference root):
//   negbin / betabin formulas        remixt/bpmodel.pyx:238-394
//   digamma (AS 103)                 remixt/bpmodel.pyx:162-235
//   per-cell likelihood branches     remixt/bpmodel.pyx:751-896
// Nothing here is derived from the reference's code layout: the reference
// evaluates every (segment,state,u,v,w) cell from scratch with 12 lgamma calls;
// here everything that depends only on the segment (8 lgamma-differences) or
// only on the state (depth, allele ratio, 4 lgamma) is hoisted into small
// per-restart tables, leaving 8 lgamma + 4 log per cell.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/remixt_amd.h"

// device error bits (per restart), translated to the reference's exceptions on the host
#define RMX_ERR_NAN_LL 1u        // ValueError 'll is nan' (bpmodel.pyx:269, :344)
#define RMX_ERR_BAD_P 2u         // ValueError 'p <= 0 or (1 - p) <= 0' (:335, :383)
#define RMX_ERR_TOTAL_DEPTH 4u   // ValueError 'total_depth <= 0' (:721, :738)
#define RMX_ERR_LOH_P 8u         // ValueError 'expected p for loh state' (:829)
#define RMX_ERR_NAN_F 16u        // AssertionError nan framelogprob (:936)
#define RMX_ERR_NAN_AB 32u       // AssertionError nan alphas / betas (:943-944)
#define RMX_ERR_NAN_POST 64u     // AssertionError nan posterior marginals (:952, :962)
#define RMX_ERR_DIGAMMA 128u     // ValueError 'x <= 0.0' in digamma (:207)
#define RMX_ERR_NAN_GRAD 256u    // ValueError 'partial_* is nan' (:298, :391)
#define RMX_ERR_WAIT 512u        // (no reference counterpart) a block of a one-launch search (search_mode 7) waited for a partner's partial sum until its watchdog ran out

// state-table flag bits
#define ST_HDEL_NB 1u      // use the hdel NB branch (:760)
#define ST_LOH_M 2u        // use loh dispersion / constants (:823-834)
#define ST_E_TD 4u         // evaluating the allele ll raises total_depth <= 0
#define ST_E_LOH 8u        // evaluating the allele ll raises 'expected p'
#define ST_E_BADP 16u      // betabin raises p <= 0 or 1-p <= 0
#define ST_GZ_ALLELE 32u   // allele gradient is zero (:867)

struct RestartParams {
    double h[RMX_MAX_CLONES];
    double p[RMX_P_COUNT];
    double logr[2];   // log(negbin_r_0), log(negbin_r_1) (filled by the host when uploading)
};

struct Dev {
    int N, S, SP, M, K, B, C, D, cn_max, NC, NBE, TC, nc, tmodel, R, pad0;
    double pen;
    // ---- shared, read-only --------------------------------------------------
    const double *l, *x, *y, *logl;      // [N], [N], [N][2], log(l) [N]
    const uint8_t *mask_t, *mask_a;      // [N]
    const int32_t *seg_class;            // [N]
    const int32_t *tclass;               // [N] transition class of (n,n+1); -1: telomere / last
    const int32_t *brk_slot;             // [N] breakend slot of (n,n+1) or -1
    const int32_t *brk_idx, *brk_orient; // [N]
    const int32_t *be_n;                 // [NBE] slot -> n (ascending)
    const int32_t *chain_be;             // [NC][2] slots lo, hi: the breakend adjacencies inside chain c are slots [lo, hi)
    const int32_t *be_cls;               // [NBE][2] state-table class of segments n, n+1
    const int32_t *chain_tc;             // [NC] transition class of a chain whose segments share one state-table class, or -1
    const int32_t *chain_cls;            // [NC] that state-table class
    const int32_t *chain_list_fast, *chain_list_generic, *chain_list_all;  // chains by FB kernel
    const int32_t *chain_start, *chain_end; // [NC]
    const uint8_t *chain_end_flag;       // [N]
    const int8_t *cn;                    // [C][S][M][2]
    const int8_t *tot;                   // [C][S][M]
    const uint8_t *sflags;               // [C][S] bit0 hdel, bit1 loh, bits2-3 #subclonal alleles
    const int32_t *brk_states;           // [B][M]
    const int32_t *bk_ptr, *bk_slots;    // CSR breakpoint -> slots (ascending n)
    const double *Tval, *Wf, *Wb;        // [TC][S][S]: log T (reference order), exp(T) (q=i,o=j), exp(T)^T
    const int8_t *af, *ab;               // [TC][S][S] allele-flip term (q=i,o=j) and its transpose
    const uint16_t *pcode;               // [TC][S][SPC] pair codes for k_pairwise_be2, columns in jord order: (index into pe2_lt) | allele distance << 10; or null
    const int32_t *jord, *jmeta;         // [C][S] columns sorted by the tumour clones' totals; totals t1 | t2 << 8 | run-end flags << 16
    // ---- per restart ----------------------------------------------------------
    RestartParams *rp;                   // [R]
    double *stD, *stP, *stM, *stLg;      // [R][C][SP]; stM [R][C][2][SP]; stLg [R][C][4][SP]
    double *stLogD;                      // [R][C][SP] log of the state's expected depth
    uint32_t *stFlags;                   // [R][C][SP]
    uint32_t *stFlagsAgg;                // [R][C] what the states of a table raise together: 1 some ST_E_TD, 2 some ST_E_LOH, 4 some ST_E_BADP without either
    double *segc;                        // [R][8][N]
    double *qt, *qa, *qs;                // [R][N][2]
    double *pbrk;                        // [R][K][B]
    double *f, *fe, *fa, *fb, *post;     // [R][N][SP]; fe = exp(f - rowmax)
    double *fe_alt;                      // [R][N][SP] second fe buffer: a fused marginal pass writes the NEXT sweep's scaled
                                         // emissions here while this sweep's breakend reductions still read fe (the host swaps)
    uint16_t *sig_idx; uint8_t *sig_cnt; // [R][N][RMX_SIGK], [R][N]: states with posterior mass >= RMX_POST_EPS per segment (count 255: more than RMX_SIGK, use all states), or null
    double *lc;                          // [R][6][N][SP] cached cell likelihoods (LT0, LT1, LA00, LA01, LA10, LA11) or null
    double *fmax, *mrow;                 // [R][N]
    double *A, *Bv;                      // [R][N][2], [R][N][4]
    double *rowPF, *rowPP, *rowZ;        // [R][N]
    double *pd_lt, *pd_cached;           // [R][NBE][M][D]
    double *pe_lt;                       // [R][NBE][MDP] exp(-pen * pd_lt), rows padded to 16 bytes
    double *pe2_lt;                      // [R][NBE][PE2P] product of pe_lt over the clones by tumour-clone differences (M <= 3), or null
    double *pe2x_lt;                     // [ceil(R/4)][NBE][PE2P][4] the same, the four restarts of a k_fbm workgroup interleaved, or null
    double *hist;                        // [R][NBE][M][D]
    double *be_jt, *be_ja;               // [R][NBE]
    uint32_t *err;                       // [R]
};

__device__ __forceinline__ size_t rs_off(const Dev &d, int r, int n) { return ((size_t)r * d.N + n) * d.SP; }

// ---- bpmodel.pyx:606-616 -------------------------------------------------------
__device__ __forceinline__ double g_transition(int tmodel, int cn_diff) {
    if (tmodel == 0) return (double)(cn_diff < 0 ? -cn_diff : cn_diff);
    return cn_diff == 0 ? 0.0 : 1.0;
}

// ---- bpmodel.pyx:162-235 (AS 103) ------------------------------------------------
__device__ inline double digamma_as103(double x, unsigned &err) {
    if (x <= 0.0) { err |= RMX_ERR_DIGAMMA; return __builtin_nan(""); }
    if (x <= 0.000001) return -0.57721566490153286060 - 1.0 / x + 1.6449340668482264365 * x;
    double value = 0.0, x2 = x;
    while (x2 < 8.5) { value = value - 1.0 / x2; x2 = x2 + 1.0; }
    double r = 1.0 / x2;
    value = value + log(x2) - 0.5 * r;
    r = r * r;
    value = (value - r * (1.0 / 12.0 - r * (1.0 / 120.0 - r * (1.0 / 252.0 - r * (1.0 / 240.0 - r * (1.0 / 132.0))))));
    return value;
}

// reciprocal for scale factors (v_rcp_f64 + one Newton step)
__device__ __forceinline__ double fast_rcp(double m) {
    double r = __builtin_amdgcn_rcp(m);
    const double e = fma(-m, r, 1.0);
    return fma(r, e, r);
}

// ---- natural logarithm for positive finite normal arguments -----------------------------------
// fdlibm's e_log.c scheme (argument reduction to [sqrt(1/2), sqrt(2)), s = f/(2+f), degree-14
// polynomial in s) with the hardware frexp instructions and a Newton-refined reciprocal: ~30
// instructions against ~60 for the library routine, < 2 ulp.  Callers guarantee x > 0, finite,
// not subnormal (counts plus positive dispersions / depths).
__device__ __forceinline__ double fast_log_pos(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);      // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s_ = f * fast_rcp(2.0 + f);
    const double z = s_ * s_, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s_ * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// ---- exponential for the indicator updates and the scaled emissions -------------------------------
// exp(x) for x <= ~700: k = rint(x / ln 2), r = x - k ln 2 (two-part constant), Taylor polynomial of degree 13 on |r| <= 0.35,
// v_ldexp_f64 (which also rounds into the subnormals); below -746 the result is 0 (covers -inf), NaN stays NaN.  21 instructions
// against 48 for the library routine; 0.88 ulp worst case over 2e7 arguments in [-700, 0] (tools/micro/exp_fast_check.cpp; libm: 0.51).
// The fused marginal pass is bound by VALU issue (one instruction = 4 cycles of a SIMD whatever it does): seven of these and two
// logarithms per segment were 500 of its 1 350 instructions.
__device__ __forceinline__ double exp_fast(double x) {
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;              // 1/13! ... 1/2!
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = ldexp(p, (int)k);
    return x < -746. ? 0. : v;
}
// log of a prior probability: the lean routine inside (0, 1), the library's at the ends (log(0) = -inf as numpy gives it)
__device__ __forceinline__ double log_prior(double p) { return (p > 1e-300 && p < 1.) ? fast_log_pos(p) : log(p); }

// log(1 + t) for t > 0 with the rounding of 1 + t compensated
__device__ __forceinline__ double fast_log1p_pos(double t) {
    const double u = 1.0 + t;
    const double dlt = t - (u - 1.0);
    return fast_log_pos(u) + dlt * fast_rcp(u);
}

// ---- fast log-gamma for positive arguments -------------------------------------------
// Stirling series with 6 correction terms for z >= 16 (truncation error < 1e-17
// relative), upward recurrence below.  Replaces libm lgamma in the per-cell hot
// loops (8 evaluations per (segment,state) cell); accuracy is pinned against the
// oracle by the parity tests (well inside the 1e-6 budget).  z <= 0 -> NaN.
__device__ __forceinline__ double lgamma_pos(double z) {
    if (!(z > 0.)) return __builtin_nan("");
    double shift = 1.;
    bool shifted = false;
    while (z < 16.) { shift *= z; z += 1.; shifted = true; }
    const double r = fast_rcp(z), r2 = r * r;
    double c = 691.0 / 360360.0;
    c = fma(-c, r2, 1.0 / 1188.0);
    c = fma(-c, r2, 1.0 / 1680.0);
    c = fma(-c, r2, 1.0 / 1260.0);
    c = fma(-c, r2, 1.0 / 360.0);
    c = fma(-c, r2, 1.0 / 12.0);
    double v = fma(z - 0.5, fast_log_pos(z), -z) + 0.91893853320467274178 + c * r;
    if (shifted) v -= log(shift);
    return v;
}

// ---- per-segment context ----------------------------------------------------------
struct SegCtx {
    double x, l, y0, y1, ys, logl;
    int mt, ma;
    double cnb[4];   // [u*2+var]  lgamma(x+r) - lgamma(x+1) - lgamma(r); var 1 = hdel dispersion
    double cbb[4];   // [v*2+var]  lgamma(n+1)-lgamma(k+1)-lgamma(n-k+1)-lgamma(n+M)+lgamma(M); var 1 = loh dispersion
};

__device__ __forceinline__ void load_seg(const Dev &d, int r, int n, SegCtx &c) {
    c.x = d.x[n]; c.l = d.l[n]; c.logl = d.logl[n]; c.y0 = d.y[2 * (size_t)n]; c.y1 = d.y[2 * (size_t)n + 1]; c.ys = c.y0 + c.y1;
    c.mt = d.mask_t[n]; c.ma = d.mask_a[n];
    const double *sc = d.segc + (size_t)r * 8 * d.N + n;
#pragma unroll
    for (int i = 0; i < 4; i++) { c.cnb[i] = sc[(size_t)i * d.N]; c.cbb[i] = sc[(size_t)(4 + i) * d.N]; }
}

__device__ __forceinline__ double nb_state_part(double x, double mu, double r) {
    // x*log(p) + r*log(1-p) with p = mu/(r+mu), p outside [0,1] -> 0.5 (bpmodel.pyx:261-267)
    double p = mu / (r + mu);
    if (p < 0. || p > 1.) p = 0.5;
    return x * log(p) + r * log(1 - p);
}

// ---- register-resident variant ---------------------------------------------------------------
// The strip kernels keep the table entries of the states a lane owns in registers across many
// segments, and take the per-segment values as wave-uniform scalars.
struct StateRegs { double D, logD, p, M0, M1, lgA0, lgB0, lgA1, lgB1, nsub; unsigned fl; };

__device__ __forceinline__ void load_state_regs(const Dev &d, int r, int cls, int s, StateRegs &st) {
    const size_t base = (size_t)r * d.C + cls, cs = d.SP;
    const size_t si = base * cs + s;
    st.D = d.stD[si]; st.logD = d.stLogD[si]; st.p = d.stP[si];
    st.M0 = d.stM[(base * 2 + 0) * cs + s]; st.M1 = d.stM[(base * 2 + 1) * cs + s];
    st.lgA0 = d.stLg[(base * 4 + 0) * cs + s]; st.lgB0 = d.stLg[(base * 4 + 1) * cs + s];
    st.lgA1 = d.stLg[(base * 4 + 2) * cs + s]; st.lgB1 = d.stLg[(base * 4 + 3) * cs + s];
    st.fl = d.stFlags[si];
    st.nsub = (double)((d.sflags[(size_t)cls * d.S + s] >> 2) & 3);
}

// Posterior mass below which a state's term post * ll is dropped from an expectation when ALL 64 states
// of a wave are below it: |ll| < 1e7, so such a group adds less than 64 * 1e-23 to sums of magnitude
// >= 1 -- far under half an ulp, the rounded sum is the same.  (Posteriors concentrate on a few
// neighbouring states; whole 64-state groups are skipped on most segments.)
#define RMX_POST_EPS 1e-30
#define RMX_SIGK 64        // capacity of the per-segment list of states with posterior mass (32 until round 4: the weighted M-step samples pick the segments with the longest lists, and an overflowed list walks all states)

// component mask of cell_ll_regs: which of the six values the caller needs
#define CM_LT0 1
#define CM_LT1 2
#define CM_LA0 4   // LA[0], LA[1]  (v = 0)
#define CM_LA1 8   // LA[2], LA[3]  (v = 1)
#define CM_ALL 15

template <int MASK>
__device__ __forceinline__ void cell_ll_regs(const RestartParams &rp, const SegCtx &sc, const StateRegs &st,
                                             double LT[2], double LA[4], unsigned &err) {
    const unsigned fl = st.fl;
    LT[0] = LT[1] = 0.;
    if ((MASK & (CM_LT0 | CM_LT1)) && sc.mt) {
        if (fl & ST_HDEL_NB) {
            const double mu = rp.p[RMX_P_NEGBIN_HDEL_MU];
            if (MASK & CM_LT0) LT[0] = sc.cnb[1] + nb_state_part(sc.x, mu, rp.p[RMX_P_NEGBIN_HDEL_R_0]);
            if (MASK & CM_LT1) LT[1] = sc.cnb[3] + nb_state_part(sc.x, mu, rp.p[RMX_P_NEGBIN_HDEL_R_1]);
        } else {
            const double mu = st.D * sc.l;
            if (mu > 0.) {
                const double lmu = sc.logl + st.logD;
                const double rmu = fast_rcp(mu);
                if (MASK & CM_LT0) { const double r0_ = rp.p[RMX_P_NEGBIN_R_0]; const double L0 = fast_log1p_pos(r0_ * rmu); LT[0] = sc.cnb[0] + (r0_ * ((rp.logr[0] - lmu) - L0) - sc.x * L0); }
                if (MASK & CM_LT1) { const double r1_ = rp.p[RMX_P_NEGBIN_R_1]; const double L1 = fast_log1p_pos(r1_ * rmu); LT[1] = sc.cnb[2] + (r1_ * ((rp.logr[1] - lmu) - L1) - sc.x * L1); }
            } else {
                if (MASK & CM_LT0) LT[0] = sc.cnb[0] + nb_state_part(sc.x, mu, rp.p[RMX_P_NEGBIN_R_0]);
                if (MASK & CM_LT1) LT[1] = sc.cnb[2] + nb_state_part(sc.x, mu, rp.p[RMX_P_NEGBIN_R_1]);
            }
        }
        if (LT[0] != LT[0] || LT[1] != LT[1]) err |= RMX_ERR_NAN_LL;
    }
    LA[0] = LA[1] = LA[2] = LA[3] = 0.;
    if ((MASK & (CM_LA0 | CM_LA1)) && sc.ma) {
        if (fl & ST_E_TD) err |= RMX_ERR_TOTAL_DEPTH;
        if (fl & ST_E_LOH) err |= RMX_ERR_LOH_P;
        if (sc.ys == 0. || (fl & (ST_E_TD | ST_E_LOH))) {
        } else if (fl & ST_E_BADP) { err |= RMX_ERR_BAD_P; }
        else {
            const double p = st.p;
            const int var = (fl & ST_LOH_M) ? 1 : 0;
            if (MASK & CM_LA0) {
                const double a = st.M0 * p, b = st.M0 * (1 - p);
                const double base = (var ? sc.cbb[1] : sc.cbb[0]) - st.lgA0 - st.lgB0;   // (no runtime-indexed array: that would live in scratch)
                LA[0] = base + lgamma_pos(sc.y0 + a) + lgamma_pos(sc.ys - sc.y0 + b);
                LA[1] = base + lgamma_pos(sc.y1 + a) + lgamma_pos(sc.ys - sc.y1 + b);
            }
            if (MASK & CM_LA1) {
                const double a = st.M1 * p, b = st.M1 * (1 - p);
                const double base = (var ? sc.cbb[3] : sc.cbb[2]) - st.lgA1 - st.lgB1;
                LA[2] = base + lgamma_pos(sc.y0 + a) + lgamma_pos(sc.ys - sc.y0 + b);
                LA[3] = base + lgamma_pos(sc.y1 + a) + lgamma_pos(sc.ys - sc.y1 + b);
            }
            if (LA[0] != LA[0] || LA[1] != LA[1] || LA[2] != LA[2] || LA[3] != LA[3]) err |= RMX_ERR_NAN_LL;
        }
    }
}

// The part of cell_ll_regs' error reporting that does not need the cell evaluated: the state-table
// flags of invalid states raise for every segment that looks at the allele likelihood, whatever the
// state's posterior mass -- kernels that skip states without posterior mass call this for them, so
// that the reference's ValueErrors (bpmodel.pyx:717-733, 823-853) are raised in the same situations.
template <int MASK>
__device__ __forceinline__ void cell_static_errors(const SegCtx &sc, unsigned fl, unsigned &err) {
    if ((MASK & (CM_LA0 | CM_LA1)) && sc.ma) {
        if (fl & ST_E_TD) err |= RMX_ERR_TOTAL_DEPTH;
        if (fl & ST_E_LOH) err |= RMX_ERR_LOH_P;
        if (!(sc.ys == 0. || (fl & (ST_E_TD | ST_E_LOH))) && (fl & ST_E_BADP)) err |= RMX_ERR_BAD_P;
    }
}

// ... for ALL states of a (restart, class) table at once, from the aggregate k_state_tables leaves in d.stFlagsAgg: what the loop of
// cell_static_errors over the S states of a segment's table reports (the sparse M-step kernels walked 660 bytes of flags per sampled segment for it)
template <int MASK>
__device__ __forceinline__ void table_static_errors(const SegCtx &sc, unsigned agg, unsigned &err) {
    if ((MASK & (CM_LA0 | CM_LA1)) && sc.ma) {
        if (agg & 1u) err |= RMX_ERR_TOTAL_DEPTH;
        if (agg & 2u) err |= RMX_ERR_LOH_P;
        if (sc.ys != 0. && (agg & 4u)) err |= RMX_ERR_BAD_P;
    }
}

// The six likelihood values of one (segment,state) cell, table-driven entry point:
//   LT[u]      = calculate_log_likelihood_total(n,s,u)      (bpmodel.pyx:751-776)
//   LA[v*2+w]  = calculate_log_likelihood_allele(n,s,v,w)   (bpmodel.pyx:809-853)
__device__ inline void cell_ll(const Dev &d, const RestartParams &rp, const SegCtx &sc, int r, int cls, int s,
                               double LT[2], double LA[4], unsigned &err) {
    StateRegs st;
    load_state_regs(d, r, cls, s, st);
    cell_ll_regs<CM_ALL>(rp, sc, st, LT, LA, err);
}

// prior of bpmodel.pyx:746-749: -1.0 * num_alleles_subclonal * l * divergence_weight
__device__ __forceinline__ double cell_prior(const Dev &d, const RestartParams &rp, const SegCtx &sc, int cls, int s) {
    const double nsub = (double)((d.sflags[(size_t)cls * d.S + s] >> 2) & 3);
    return -1.0 * nsub * sc.l * rp.p[RMX_P_DIVERGENCE_WEIGHT];
}

// two-element _exp_normalize (bpmodel.pyx:120-128), same operation order
__device__ __forceinline__ void exp_normalize2(double lp0, double lp1, double &y0, double &y1) {
    const double vmax = lp0 > lp1 ? lp0 : lp1;   // _max: strict > from -inf
    double ps = 0.; ps += exp_fast(lp0 - vmax); ps += exp_fast(lp1 - vmax);
    const double norm = fast_log_pos(ps) + vmax;      // (ps in [1, 2], or NaN)
    y0 = exp_fast(lp0 - norm); y1 = exp_fast(lp1 - norm);
    const double s = y0 + y1;
    y0 /= s; y1 /= s;
}

// The logits of update_p_outlier_allele (bpmodel.pyx:1005-1023) and update_p_allele_swap (:1025-1042) from the four allele expectations:
// sums of two products, which the compiler is free to contract into fused multiply-adds either way round -- it did so differently in
// different kernels, visible where one product is 1e-60 of the other (one indicator in 4 200 off by an ulp).  One helper for the
// stand-alone kernels and the fused passes, with the fused multiply-adds written out.
__device__ __forceinline__ void outlier_allele_logits(double prior0, double prior1, double qs0, double qs1, double b0, double b1, double b2, double b3,
                                                      double &lp0, double &lp1) {
#pragma clang fp contract(off)
    lp0 = fma(qs0, b0, prior0); lp0 = fma(qs1, b1, lp0);
    lp1 = fma(qs0, b2, prior1); lp1 = fma(qs1, b3, lp1);
}
__device__ __forceinline__ void allele_swap_logits(double qa0, double qa1, double b0, double b1, double b2, double b3, double &lp0, double &lp1) {
#pragma clang fp contract(off)
    lp0 = qa0 * b0; lp1 = qa0 * b1;
    lp0 = fma(qa1, b2, lp0); lp1 = fma(qa1, b3, lp1);
}

__device__ __forceinline__ double xlogx(double v) { return v > 0. ? v * log(v) : 0.; }

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int off) { return __shfl_xor(v, off, 64); }

// ---- DPP cross-lane helpers (no LDS round trip, unlike __shfl_xor's ds_bpermute) ----------
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the aligned group of PP lanes (PP a power of two <= 64); every lane of the group gets the sum.
// xor-1 / xor-2 are quad permutes; once quads (half rows) hold uniform values, the half-row (row)
// mirror pairs each lane with one of the other quad (half row), which is all a butterfly step needs.
__device__ __forceinline__ double quad_group_sum(double v, int PP) {
    if (PP >= 2) v += dpp_mov_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    if (PP >= 4) v += dpp_mov_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    if (PP >= 8) v += dpp_mov_f64<0x141>(v);    // row_half_mirror
    if (PP >= 16) v += dpp_mov_f64<0x140>(v);   // row_mirror
    if (PP >= 32) v += __shfl_xor(v, 16, 64);
    if (PP >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
// maximum over the 64 lanes of a wave of non-negative values; result valid in every lane.
__device__ __forceinline__ double wave_max_nonneg(double v) {
    v = fmax(v, dpp_mov_f64<0x121>(v));   // row_ror:1
    v = fmax(v, dpp_mov_f64<0x122>(v));   // row_ror:2
    v = fmax(v, dpp_mov_f64<0x124>(v));   // row_ror:4
    v = fmax(v, dpp_mov_f64<0x128>(v));   // row_ror:8  -> every lane holds its row's (16 lanes) maximum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r = v;
#pragma unroll
    for (int row = 0; row < 4; row++)
        r = fmax(r, __hiloint2double(__builtin_amdgcn_readlane(hi, row * 16), __builtin_amdgcn_readlane(lo, row * 16)));
    return r;
}
// Row scaling of the forward / backward vectors.  Any positive per-row factor is legal (posteriors are
// self-normalised and hmm_log_norm_const adds log(factor) per row, see rowZ); the factor used is the
// power of two 2^e <= max < 2^(e+1): multiplying by 2^-e is exact, needs no reciprocal, and the
// reduction only has to find the largest HIGH DWORD of the (non-negative) values -- a 32-bit integer
// maximum, single-instruction DPP steps instead of a 64-bit floating-point butterfly.
__device__ __forceinline__ void pow2_scale(unsigned hi, double &scale, double &inv) {
    // (selects, not branches: this sits in the tail of every forward-backward step)
    const unsigned ef = (hi >> 20) & 0x7ffu;
    const bool ok = ef - 1u < 0x7feu;                        // a normal number; else inf / nan (a nan's sign bit only raises the key) or a vanished row (zero / denormal)
    const int hs = ok ? (int)(ef << 20) : (ef == 0x7ffu ? 0x7ff00000 : 0);
    const int hv = ok ? (int)((2046u - ef) << 20) : 0;
    scale = __hiloint2double(hs, 0); inv = __hiloint2double(hv, 0);
}
// maximum over the 64 lanes of a wave of unsigned keys (wave-uniform result)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ void lds_max_u32(void *lds_ptr, unsigned v) {
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_ptr;
    asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// LDS 64-bit unsigned atomic max, one instruction (atomicMax() makes hipcc emit a per-lane scalar loop)
__device__ __forceinline__ void lds_max_u64(void *lds_ptr, unsigned long long v) {
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds_ptr;
    asm volatile("ds_max_u64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// sum / max over aligned groups of G lanes (G in {8,16,32,64}); every lane gets its group's result.
// DPP butterflies inside a row of 16 lanes, v_readlane across rows: fixed evaluation order
// (deterministic) and no LDS round trips (__shfl_xor lowers to ds_bpermute).
__device__ __forceinline__ double group_sum(double v, int G) {
    v += dpp_mov_f64<0xB1>(v);                 // xor 1
    v += dpp_mov_f64<0x4E>(v);                 // xor 2
    v += dpp_mov_f64<0x141>(v);                // xor 4 (row_half_mirror on quad-uniform values)
    if (G >= 16) v += dpp_mov_f64<0x140>(v);   // xor 8 (row_mirror on half-row-uniform values)
    if (G >= 32) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
        const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        if (G >= 64) v = ((r0 + r1) + r2) + r3;
        else v = (threadIdx.x & 32) ? (r2 + r3) : (r0 + r1);
    }
    return v;
}
__device__ __forceinline__ double group_max(double v, int G) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    if (G >= 16) v = fmax(v, dpp_mov_f64<0x140>(v));
    if (G >= 32) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
        const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        if (G >= 64) v = fmax(fmax(r0, r1), fmax(r2, r3));
        else v = (threadIdx.x & 32) ? fmax(r2, r3) : fmax(r0, r1);
    }
    return v;
}

// the same with the number of waves taken from the launch (blocks of 64 * k threads): the per-wave sums are
// added in wave order, so a block of fewer waves whose dropped lanes would have contributed zeros gives the
// same bits
__device__ inline double block_sum_rt(double v, double *scratch /* >= blockDim.x/64 */) {
    v = group_sum(v, 64);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double t = 0.;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += scratch[i];
    return t;
}
// block-wide deterministic sum (fixed tree): all threads must call; result valid in thread 0
template <int NT> __device__ inline double block_sum(double v, double *scratch /* >= NT/64 */) {
    v = group_sum(v, 64);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double t = 0.;
    if (threadIdx.x == 0) for (int i = 0; i < NT / 64; i++) t += scratch[i];
    return t;
}
