/*
 * oracle/remixt_oracle.c -- CPU restatement of ReMixT's variational-HMM kernel.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the "port"
 * CPU baseline.  It is never linked into, imported by, or called from the
 * product path (remixt_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * It restates, in plain C99 / float64, the algorithm of the reference's
 * remixt/bpmodel.pyx (class RemixtModel + sum_product + max_product) with the
 * same dense (N-1) x S x S arrays, the same loop nests and the same
 * accumulation order, so that results agree with the compiled reference to the
 * last bit wherever libm agrees.  Each function cites the reference lines it
 * follows (paths relative to /root/reference).
 *
 * Parity pin: validated against the reference itself built by
 * oracle/build_ref.py (tests/test_oracle_vs_ref.py, build container only) and
 * against the committed golden vectors in tests/golden/ (generated from the
 * reference by oracle/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RMXO_OK 0
#define RMXO_EVALUE 1   /* reference raises ValueError */
#define RMXO_EASSERT 2  /* reference raises AssertionError */

typedef struct rmxo_model {
    int M, N, K, A, cn_max, normal_contamination, S, B;
    int64_t *cn_states;        /* [N][S][M][2] */
    int64_t *cn_states_total;  /* [N][S][M] */
    int64_t *brk_states;       /* [B][M] */
    int64_t *num_alleles_subclonal; /* [N][S] */
    int64_t *is_hdel, *is_loh; /* [N][S] */
    int64_t *is_telomere, *breakpoint_idx, *breakpoint_orient, *breakpoint_side; /* [N] */
    double transition_penalty, divergence_weight;
    double *p_breakpoint;      /* [K][B] */
    double hmm_log_norm_const;
    double *framelogprob;      /* [N][S] */
    double *log_transmat, *cached_log_transmat, *joint_posterior_marginals; /* [N-1][S][S] */
    double *posterior_marginals; /* [N][S] */
    double *p_allele_swap, *p_outlier_total, *p_outlier_allele; /* [N][2] */
    double prior_outlier_total, prior_outlier_allele;
    double *l, *x, *y;         /* [N], [N], [N][2] */
    int64_t *total_likelihood_mask, *allele_likelihood_mask; /* [N] */
    double *h;                 /* [M] */
    double negbin_r_0, negbin_r_1, negbin_hdel_mu, negbin_hdel_r_0, negbin_hdel_r_1;
    double betabin_M_0, betabin_M_1, betabin_loh_p, betabin_loh_M_0, betabin_loh_M_1;
    int transition_model;
    double *p_d;               /* [(cn_max+1)*2] */
    int p_d_len;
    int err;
    char errmsg[256];
} rmxo_model;

#define CN(m_, n, s, c, a) ((m_)->cn_states[((((size_t)(n)) * (m_)->S + (s)) * (m_)->M + (c)) * 2 + (a)])
#define TOT(m_, n, s, c) ((m_)->cn_states_total[(((size_t)(n)) * (m_)->S + (s)) * (m_)->M + (c)])
#define T3(arr, m_, n, i, j) ((arr)[(((size_t)(n)) * (m_)->S + (i)) * (m_)->S + (j)])

/* ---- bpmodel.pyx:21-159 reductions ------------------------------------ */
static double v_max(const double *v, size_t n) {            /* :21-45 */
    double vmax = -INFINITY;
    for (size_t i = 0; i < n; i++) if (v[i] > vmax) vmax = v[i];
    return vmax;
}
static int v_argmax(const double *v, size_t n) {            /* :48-55, strict > */
    double vmax = -INFINITY; int imax = 0;
    for (size_t i = 0; i < n; i++) if (v[i] > vmax) { vmax = v[i]; imax = (int)i; }
    return imax;
}
static double v_logsum(const double *v, size_t n) {         /* :77-107 */
    double vmax = v_max(v, n), power_sum = 0;
    for (size_t i = 0; i < n; i++) power_sum += exp(v[i] - vmax);
    return log(power_sum) + vmax;
}
static double v_entropy(const double *v, size_t n) {        /* :110-117 */
    double e = 0;
    for (size_t i = 0; i < n; i++) if (v[i] > 0.) e += v[i] * log(v[i]);
    return e;
}
static void v_exp_normalize(double *Y, const double *X, size_t n) { /* :120-159 */
    double normalize = v_logsum(X, n);
    for (size_t i = 0; i < n; i++) Y[i] = exp(X[i] - normalize);
    normalize = 0.;
    for (size_t i = 0; i < n; i++) normalize += Y[i];
    for (size_t i = 0; i < n; i++) Y[i] /= normalize;
}

/* ---- bpmodel.pyx:162-235 digamma (AS 103) ------------------------------ */
static double digamma_as103(rmxo_model *m, double x) {
    const double c = 8.5, euler_mascheroni = 0.57721566490153286060;
    double r, value, x2;
    if (x <= 0.0) {
        if (m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "x <= 0.0 for x: %g", x); }
        return NAN;
    }
    if (x <= 0.000001) return -euler_mascheroni - 1.0 / x + 1.6449340668482264365 * x;
    value = 0.0; x2 = x;
    while (x2 < c) { value = value - 1.0 / x2; x2 = x2 + 1.0; }
    r = 1.0 / x2;
    value = value + log(x2) - 0.5 * r;
    r = r * r;
    value = (value - r * (1.0 / 12.0 - r * (1.0 / 120.0 - r * (1.0 / 252.0 - r * (1.0 / 240.0 - r * (1.0 / 132.0))))));
    return value;
}

/* ---- bpmodel.pyx:238-301 negative binomial ----------------------------- */
static double negbin_ll(rmxo_model *m, double x, double mu, double r) {
    double nb_p = mu / (r + mu);
    if (nb_p < 0. || nb_p > 1.) nb_p = 0.5;
    double ll = (lgamma(x + r) - lgamma(x + 1) - lgamma(r) + x * log(nb_p) + r * log(1 - nb_p));
    if (isnan(ll) && m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "ll is nan for x: %g, mu: %g, r: %g", x, mu, r); }
    return ll;
}
static double negbin_ll_partial_mu(rmxo_model *m, double x, double mu, double r) {
    double p = x / mu - (r + x) / (r + mu);
    if (isnan(p) && m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "partial_mu is nan for x: %g, mu: %g, r: %g", x, mu, r); }
    return p;
}
/* ---- bpmodel.pyx:304-394 beta binomial --------------------------------- */
static double betabin_ll(rmxo_model *m, double k, double n, double p, double M) {
    if (p <= 0. || (1 - p) <= 0.) {
        if (m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "p <= 0 or (1 - p) <= 0. for p: %g", p); }
        return NAN;
    }
    double ll = (lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1)
        + lgamma(k + M * p) + lgamma(n - k + M * (1 - p))
        - lgamma(n + M)
        - lgamma(M * p) - lgamma(M * (1 - p))
        + lgamma(M));
    if (isnan(ll) && m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "ll is nan for k: %g, n: %g, p: %g, M: %g", k, n, p, M); }
    return ll;
}
static double betabin_ll_partial_p(rmxo_model *m, double k, double n, double p, double M) {
    if (p <= 0. || (1 - p) <= 0.) {
        if (m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "p <= 0 or (1 - p) <= 0. for p: %g", p); }
        return NAN;
    }
    double pp = (M * digamma_as103(m, k + M * p)
        + (-M) * digamma_as103(m, n - k + M * (1 - p))
        - M * digamma_as103(m, M * p)
        - (-M) * digamma_as103(m, M * (1 - p)));
    if (isnan(pp) && m && !m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "partial_p is nan"); }
    return pp;
}

/* exported scalar helpers so tests can pin a2-a4 directly */
double rmxo_digamma(double x) { return digamma_as103(NULL, x); }
double rmxo_negbin_ll(double x, double mu, double r) { return negbin_ll(NULL, x, mu, r); }
double rmxo_negbin_ll_partial_mu(double x, double mu, double r) { return negbin_ll_partial_mu(NULL, x, mu, r); }
double rmxo_betabin_ll(double k, double n, double p, double M) { return betabin_ll(NULL, k, n, p, M); }
double rmxo_betabin_ll_partial_p(double k, double n, double p, double M) { return betabin_ll_partial_p(NULL, k, n, p, M); }

/* ---- bpmodel.pyx:606-616 ------------------------------------------------ */
static inline double calc_transition(const rmxo_model *m, double cn_diff) {
    if (m->transition_model == 0) return fabs(cn_diff);
    else if (m->transition_model == 1) return (cn_diff == 0) ? 0. : 1.;
    return 0.; /* Cython: falls off the end -> 0.0 */
}
static inline int pd_index(const rmxo_model *m, int d) { return d < 0 ? d + m->p_d_len : d; } /* wraparound(True) */

/* ---- bpmodel.pyx:639-684 calculate_log_transmat ------------------------- */
void rmxo_calculate_log_transmat(rmxo_model *m, double *lt) {
    const int S = m->S, M = m->M;
    if (m->N > 1) memset(lt, 0, sizeof(double) * (size_t)(m->N - 1) * S * S);
    for (int n = 0; n < m->N - 1; n++) {
        if (m->is_telomere[n] > 0) continue;
        else if (m->breakpoint_idx[n] < 0) {
            for (int c = 0; c < M; c++)
                for (int s1 = 0; s1 < S; s1++)
                    for (int s2 = 0; s2 < S; s2++)
                        T3(lt, m, n, s1, s2) += -m->transition_penalty * calc_transition(m, (double)(TOT(m, n, s1, c) - TOT(m, n + 1, s2, c)));
        } else {
            for (int c = 0; c < M; c++) {
                for (int i = 0; i < m->p_d_len; i++) m->p_d[i] = 0.;
                for (int d = -m->cn_max - 1; d < m->cn_max + 2; d++)
                    for (int sb = 0; sb < m->B; sb++)
                        m->p_d[pd_index(m, d)] += m->p_breakpoint[(size_t)m->breakpoint_idx[n] * m->B + sb]
                            * calc_transition(m, (double)(d - m->breakpoint_orient[n] * m->brk_states[(size_t)sb * M + c]));
                for (int s1 = 0; s1 < S; s1++)
                    for (int s2 = 0; s2 < S; s2++)
                        T3(lt, m, n, s1, s2) += -m->transition_penalty * m->p_d[pd_index(m, (int)(TOT(m, n, s1, c) - TOT(m, n + 1, s2, c)))];
            }
        }
        double ach[2];
        for (int s1 = 0; s1 < S; s1++)
            for (int s2 = 0; s2 < S; s2++) {
                for (int flip = 0; flip < 2; flip++) {
                    ach[flip] = 0.;
                    for (int c = 0; c < M; c++) {
                        for (int a = 0; a < 2; a++) {
                            int oa = flip == 1 ? 1 - a : a;
                            ach[flip] += calc_transition(m, (double)(CN(m, n, s1, c, a) - CN(m, n + 1, s2, c, oa)));
                        }
                        ach[flip] -= calc_transition(m, (double)(TOT(m, n, s1, c) - TOT(m, n + 1, s2, c)));
                    }
                }
                T3(lt, m, n, s1, s2) += -m->transition_penalty * (ach[0] < ach[1] ? ach[0] : ach[1]);
            }
    }
}

/* ---- bpmodel.pyx:686-749 ------------------------------------------------ */
static double expected_total_reads(const rmxo_model *m, int n, int s) {
    double mu = 0.;
    for (int c = 0; c < m->M; c++) mu += m->h[c] * (double)TOT(m, n, s, c);
    mu *= m->l[n];
    return mu;
}
static double expected_allele_ratio(rmxo_model *m, int n, int s, double *minor_out, double *total_out) {
    double minor_depth = 0., total_depth = 0.;
    for (int c = 0; c < m->M; c++) {
        minor_depth += m->h[c] * (double)CN(m, n, s, c, 0);
        total_depth += m->h[c] * (double)TOT(m, n, s, c);
    }
    if (total_depth <= 0) {
        if (!m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "total_depth <= 0 for s: %d", s); }
        return NAN;
    }
    if (minor_out) *minor_out = minor_depth;
    if (total_out) *total_out = total_depth;
    return minor_depth / total_depth;
}
static double log_prior_cn(const rmxo_model *m, int n, int s) {   /* :746-749 */
    return -1.0 * (double)m->num_alleles_subclonal[(size_t)n * m->S + s] * m->l[n] * m->divergence_weight;
}

/* ---- bpmodel.pyx:751-776 ------------------------------------------------ */
double rmxo_log_likelihood_total(rmxo_model *m, int n, int s, int u) {
    double mu, r;
    if (m->total_likelihood_mask[n] == 0) return 0.;
    if (!m->normal_contamination && m->is_hdel[(size_t)n * m->S + s] == 1) {
        mu = m->negbin_hdel_mu;
        r = (u == 0) ? m->negbin_hdel_r_0 : m->negbin_hdel_r_1;
    } else {
        mu = expected_total_reads(m, n, s);
        r = (u == 0) ? m->negbin_r_0 : m->negbin_r_1;
    }
    return negbin_ll(m, m->x[n], mu, r);
}
/* ---- bpmodel.pyx:778-807 ------------------------------------------------ */
static void log_likelihood_total_partial_h(rmxo_model *m, int n, int s, int u, double *ph) {
    if (m->total_likelihood_mask[n] == 0) { for (int c = 0; c < m->M; c++) ph[c] = 0.; return; }
    if (!m->normal_contamination && m->is_hdel[(size_t)n * m->S + s] == 1) { for (int c = 0; c < m->M; c++) ph[c] = 0.; return; }
    double mu = expected_total_reads(m, n, s);
    double r = (u == 0) ? m->negbin_r_0 : m->negbin_r_1;
    double pmu = negbin_ll_partial_mu(m, m->x[n], mu, r);
    for (int c = 0; c < m->M; c++) ph[c] = m->l[n] * (double)TOT(m, n, s, c);
    for (int c = 0; c < m->M; c++) ph[c] *= pmu;
}
/* ---- bpmodel.pyx:809-853 ------------------------------------------------ */
double rmxo_log_likelihood_allele(rmxo_model *m, int n, int s, int v, int w) {
    double p, Mv;
    if (m->allele_likelihood_mask[n] == 0) return 0.;
    if (m->is_hdel[(size_t)n * m->S + s] == 1) p = 0.;
    else p = expected_allele_ratio(m, n, s, NULL, NULL);
    if (!m->normal_contamination && m->is_loh[(size_t)n * m->S + s] == 1) {
        if (p == 0.) p = m->betabin_loh_p;
        else if (p == 1.) p = 1. - m->betabin_loh_p;
        else {
            if (!m->err) { m->err = RMXO_EVALUE; snprintf(m->errmsg, sizeof m->errmsg, "expected p %g for loh state %d", p, s); }
            return NAN;
        }
        Mv = (v == 0) ? m->betabin_loh_M_0 : m->betabin_loh_M_1;
    } else {
        Mv = (v == 0) ? m->betabin_M_0 : m->betabin_M_1;
    }
    double allelic = m->y[(size_t)n * 2 + 0] + m->y[(size_t)n * 2 + 1];
    if (allelic == 0) return 0.;
    double minor = (w == 0) ? m->y[(size_t)n * 2 + 0] : m->y[(size_t)n * 2 + 1];
    return betabin_ll(m, minor, allelic, p, Mv);
}
/* ---- bpmodel.pyx:855-896 ------------------------------------------------ */
static void log_likelihood_allele_partial_h(rmxo_model *m, int n, int s, int v, int w, double *ph) {
    const int M = m->M;
    if (m->allele_likelihood_mask[n] == 0) { for (int c = 0; c < M; c++) ph[c] = 0.; return; }
    if (!m->normal_contamination && m->is_loh[(size_t)n * m->S + s] == 1) { for (int c = 0; c < M; c++) ph[c] = 0.; return; }
    double minor_depth, total_depth;
    double p = expected_allele_ratio(m, n, s, &minor_depth, &total_depth);
    if (m->err) { for (int c = 0; c < M; c++) ph[c] = NAN; return; }
    double Mv = (v == 0) ? m->betabin_M_0 : m->betabin_M_1;
    double allelic = m->y[(size_t)n * 2 + 0] + m->y[(size_t)n * 2 + 1];
    if (allelic == 0) { for (int c = 0; c < M; c++) ph[c] = 0.; return; }
    double minor = (w == 0) ? m->y[(size_t)n * 2 + 0] : m->y[(size_t)n * 2 + 1];
    double pp = betabin_ll_partial_p(m, minor, allelic, p, Mv);
    /* :726-744 (recomputes the depths; same values) */
    for (int c = 0; c < M; c++)
        ph[c] = (((double)CN(m, n, s, c, 0) * total_depth - minor_depth * (double)TOT(m, n, s, c)) / (total_depth * total_depth));
    for (int c = 0; c < M; c++) ph[c] *= pp;
}

/* the per-cell cpdef methods as one exported entry (same numbering as rmx_cell_quantity in include/remixt_amd.h):
 * 0 expected_total_reads, 1 its partial_h, 2 expected_allele_ratio, 3 its partial_h (:727-745), 4 log_prior_cn,
 * 5 log_likelihood_total_partial_h (u), 6 log_likelihood_allele_partial_h (v, w) */
int rmxo_cell_quantity(rmxo_model *m, int n, int s, int which, int u, int v, int w, double *out) {
    const int M = m->M;
    double minor_depth = 0., total_depth = 0.;
    switch (which) {
    case 0: out[0] = expected_total_reads(m, n, s); break;
    case 1: for (int c = 0; c < M; c++) out[c] = m->l[n] * (double)TOT(m, n, s, c); break;
    case 2: out[0] = expected_allele_ratio(m, n, s, NULL, NULL); break;
    case 3:
        expected_allele_ratio(m, n, s, &minor_depth, &total_depth);
        if (!m->err)
            for (int c = 0; c < M; c++)
                out[c] = (((double)CN(m, n, s, c, 0) * total_depth - minor_depth * (double)TOT(m, n, s, c)) / (total_depth * total_depth));
        break;
    case 4: out[0] = log_prior_cn(m, n, s); break;
    case 5: log_likelihood_total_partial_h(m, n, s, u, out); break;
    case 6: log_likelihood_allele_partial_h(m, n, s, v, w, out); break;
    default: return -1;
    }
    return m->err;
}

/* ---- bpmodel.pyx:898-919 update_framelogprob ---------------------------- */
int rmxo_update_framelogprob(rmxo_model *m) {
    const int S = m->S;
    for (int n = 0; n < m->N; n++)
        for (int s = 0; s < S; s++) {
            double f = 0.;
            for (int u = 0; u < 2; u++)
                f += (m->p_outlier_total[(size_t)n * 2 + u] * rmxo_log_likelihood_total(m, n, s, u));
            for (int v = 0; v < 2; v++)
                for (int w = 0; w < 2; w++)
                    f += (m->p_outlier_allele[(size_t)n * 2 + v] * m->p_allele_swap[(size_t)n * 2 + w] * rmxo_log_likelihood_allele(m, n, s, v, w));
            f += log_prior_cn(m, n, s);
            m->framelogprob[(size_t)n * S + s] = f;
            if (m->err) return m->err;
        }
    return m->err;
}

/* ---- bpmodel.pyx:1213-1246 sum_product ---------------------------------- */
void rmxo_sum_product(const double *f, const double *lt, double *alphas, double *betas, int N, int S) {
    double *wb = (double *)malloc(sizeof(double) * S);
    for (int i = 0; i < S; i++) alphas[i] = f[i];
    for (int n = 1; n < N; n++)
        for (int j = 0; j < S; j++) {
            for (int i = 0; i < S; i++) wb[i] = alphas[(size_t)(n - 1) * S + i] + lt[((size_t)(n - 1) * S + i) * S + j];
            alphas[(size_t)n * S + j] = v_logsum(wb, S) + f[(size_t)n * S + j];
        }
    for (int i = 0; i < S; i++) betas[(size_t)(N - 1) * S + i] = 0.0;
    for (int n = N - 2; n >= 0; n--)
        for (int i = 0; i < S; i++) {
            for (int j = 0; j < S; j++)
                wb[j] = (lt[((size_t)n * S + i) * S + j] + f[(size_t)(n + 1) * S + j] + betas[(size_t)(n + 1) * S + j]);
            betas[(size_t)n * S + i] = v_logsum(wb, S);
        }
    free(wb);
}

/* ---- bpmodel.pyx:1296-1333 max_product ---------------------------------- */
double rmxo_max_product(const double *f, const double *lt, int64_t *state_sequence, int N, int S) {
    double *wb = (double *)malloc(sizeof(double) * S);
    double *lat = (double *)calloc((size_t)N * S, sizeof(double));
    for (int i = 0; i < S; i++) lat[i] = f[i];
    for (int n = 1; n < N; n++)
        for (int j = 0; j < S; j++) {
            for (int i = 0; i < S; i++) wb[i] = lat[(size_t)(n - 1) * S + i] + lt[((size_t)(n - 1) * S + i) * S + j];
            lat[(size_t)n * S + j] = v_max(wb, S) + f[(size_t)n * S + j];
        }
    /* np.argmax: first maximum; NaN-free inputs assumed */
    int max_pos = 0; { double vm = lat[(size_t)(N - 1) * S]; for (int i = 1; i < S; i++) if (lat[(size_t)(N - 1) * S + i] > vm) { vm = lat[(size_t)(N - 1) * S + i]; max_pos = i; } }
    state_sequence[N - 1] = max_pos;
    double logprob = lat[(size_t)(N - 1) * S + max_pos];
    for (int n = N - 2; n >= 0; n--) {
        for (int i = 0; i < S; i++) wb[i] = lat[(size_t)n * S + i] + lt[((size_t)n * S + i) * S + state_sequence[n + 1]];
        state_sequence[n] = v_argmax(wb, S);
    }
    free(wb); free(lat);
    return logprob;
}

static int any_nan(const double *v, size_t n) { for (size_t i = 0; i < n; i++) if (isnan(v[i])) return 1; return 0; }

/* ---- bpmodel.pyx:921-962 update_p_cn ------------------------------------ */
int rmxo_update_p_cn(rmxo_model *m) {
    const int N = m->N, S = m->S;
    double *alphas = (double *)malloc(sizeof(double) * (size_t)N * S);
    double *betas = (double *)malloc(sizeof(double) * (size_t)N * S);
    double *lpm = (double *)malloc(sizeof(double) * S);
    double *ljpm = (double *)malloc(sizeof(double) * (size_t)S * S);
    int rc = rmxo_update_framelogprob(m);
    if (rc) goto done;
    if (any_nan(m->framelogprob, (size_t)N * S)) { m->err = RMXO_EASSERT; snprintf(m->errmsg, sizeof m->errmsg, "nan in framelogprob"); goto done; }
    rmxo_calculate_log_transmat(m, m->log_transmat);
    rmxo_sum_product(m->framelogprob, m->log_transmat, alphas, betas, N, S);
    if (any_nan(alphas, (size_t)N * S) || any_nan(betas, (size_t)N * S)) { m->err = RMXO_EASSERT; snprintf(m->errmsg, sizeof m->errmsg, "nan in alphas/betas"); goto done; }
    m->hmm_log_norm_const = v_logsum(alphas + (size_t)(N - 1) * S, S);
    for (int n = 0; n < N; n++) {
        for (int s = 0; s < S; s++) lpm[s] = alphas[(size_t)n * S + s] + betas[(size_t)n * S + s];
        v_exp_normalize(m->posterior_marginals + (size_t)n * S, lpm, S);
    }
    if (any_nan(m->posterior_marginals, (size_t)N * S)) { m->err = RMXO_EASSERT; snprintf(m->errmsg, sizeof m->errmsg, "nan in posterior_marginals"); goto done; }
    for (int n = 0; n < N - 1; n++) {
        for (int s = 0; s < S; s++)
            for (int s_ = 0; s_ < S; s_++)
                ljpm[(size_t)s * S + s_] = (alphas[(size_t)n * S + s] + T3(m->log_transmat, m, n, s, s_)
                    + m->framelogprob[(size_t)(n + 1) * S + s_] + betas[(size_t)(n + 1) * S + s_]);
        v_exp_normalize(m->joint_posterior_marginals + (size_t)n * S * S, ljpm, (size_t)S * S);
    }
    if (N > 1 && any_nan(m->joint_posterior_marginals, (size_t)(N - 1) * S * S)) { m->err = RMXO_EASSERT; snprintf(m->errmsg, sizeof m->errmsg, "nan in joint_posterior_marginals"); }
done:
    free(alphas); free(betas); free(lpm); free(ljpm);
    return m->err;
}

/* ---- bpmodel.pyx:618-637 + 964-985 update_p_breakpoint ------------------- */
int rmxo_update_p_breakpoint(rmxo_model *m) {
    const int S = m->S, B = m->B;
    double *logp = (double *)calloc((size_t)(m->K > 0 ? m->K : 1) * B, sizeof(double));
    for (int n = 0; n < m->N - 1; n++) {
        if (m->breakpoint_idx[n] < 0) continue;
        double *lp = logp + (size_t)m->breakpoint_idx[n] * B;
        const double *pcn = m->joint_posterior_marginals + (size_t)n * S * S;
        const double mult = -m->transition_penalty;
        for (int c = 0; c < m->M; c++) {
            for (int i = 0; i < m->p_d_len; i++) m->p_d[i] = 0.;
            for (int s1 = 0; s1 < S; s1++)
                for (int s2 = 0; s2 < S; s2++) {
                    int d = (int)(TOT(m, n, s1, c) - TOT(m, n + 1, s2, c));
                    m->p_d[pd_index(m, d)] += pcn[(size_t)s1 * S + s2];
                }
            for (int sb = 0; sb < B; sb++)
                for (int d = -m->cn_max - 1; d < m->cn_max + 2; d++)
                    lp[sb] += mult * m->p_d[pd_index(m, d)] * calc_transition(m, (double)(d - m->breakpoint_orient[n] * m->brk_states[(size_t)sb * m->M + c]));
        }
    }
    for (int k = 0; k < m->K; k++) v_exp_normalize(m->p_breakpoint + (size_t)k * B, logp + (size_t)k * B, B);
    free(logp);
    rmxo_calculate_log_transmat(m, m->cached_log_transmat);
    return m->err;
}

/* ---- bpmodel.pyx:987-1042 indicator updates ------------------------------ */
int rmxo_update_p_outlier_total(rmxo_model *m) {
    double lp[2];
    for (int n = 0; n < m->N; n++) {
        lp[0] = log(1. - m->prior_outlier_total);
        lp[1] = log(m->prior_outlier_total);
        for (int s = 0; s < m->S; s++)
            for (int u = 0; u < 2; u++)
                lp[u] += (m->posterior_marginals[(size_t)n * m->S + s] * rmxo_log_likelihood_total(m, n, s, u));
        if (m->err) return m->err;
        v_exp_normalize(m->p_outlier_total + (size_t)n * 2, lp, 2);
    }
    return m->err;
}
int rmxo_update_p_outlier_allele(rmxo_model *m) {
    double lp[2];
    for (int n = 0; n < m->N; n++) {
        lp[0] = log(1. - m->prior_outlier_allele);
        lp[1] = log(m->prior_outlier_allele);
        for (int s = 0; s < m->S; s++)
            for (int v = 0; v < 2; v++)
                for (int w = 0; w < 2; w++)
                    lp[v] += (m->p_allele_swap[(size_t)n * 2 + w] * m->posterior_marginals[(size_t)n * m->S + s] * rmxo_log_likelihood_allele(m, n, s, v, w));
        if (m->err) return m->err;
        v_exp_normalize(m->p_outlier_allele + (size_t)n * 2, lp, 2);
    }
    return m->err;
}
int rmxo_update_p_allele_swap(rmxo_model *m) {
    double lp[2];
    for (int n = 0; n < m->N; n++) {
        lp[0] = 0.; lp[1] = 0.;
        for (int s = 0; s < m->S; s++)
            for (int v = 0; v < 2; v++)
                for (int w = 0; w < 2; w++)
                    lp[w] += (m->p_outlier_allele[(size_t)n * 2 + v] * m->posterior_marginals[(size_t)n * m->S + s] * rmxo_log_likelihood_allele(m, n, s, v, w));
        if (m->err) return m->err;
        v_exp_normalize(m->p_allele_swap + (size_t)n * 2, lp, 2);
    }
    return m->err;
}

/* ---- bpmodel.pyx:1044-1123 ELBO ------------------------------------------ */
double rmxo_variational_entropy(rmxo_model *m) {
    const size_t NS = (size_t)m->N * m->S, NSS = (size_t)(m->N - 1) * m->S * m->S;
    double entropy = 0., acc;
    entropy += -m->hmm_log_norm_const;
    /* np.sum of elementwise products: numpy pairwise summation; the oracle uses a
       plain left-to-right sum (differences are O(1e-16) relative, see tests) */
    acc = 0.; for (size_t i = 0; i < NS; i++) acc += m->posterior_marginals[i] * m->framelogprob[i];
    entropy += acc;
    acc = 0.; for (size_t i = 0; i < NSS; i++) acc += m->joint_posterior_marginals[i] * m->log_transmat[i];
    entropy += acc;
    entropy += v_entropy(m->p_breakpoint, (size_t)m->K * m->B);
    entropy += v_entropy(m->p_outlier_total, (size_t)m->N * 2);
    entropy += v_entropy(m->p_outlier_allele, (size_t)m->N * 2);
    entropy += v_entropy(m->p_allele_swap, (size_t)m->N * 2);
    return entropy;
}
double rmxo_variational_energy(rmxo_model *m) {
    const int N = m->N, S = m->S;
    double energy = 0.;
    for (int n = 0; n < N; n++)
        for (int s = 0; s < S; s++)
            energy += (m->posterior_marginals[(size_t)n * S + s] * log_prior_cn(m, n, s));
    for (int n = 0; n < N; n++) {
        for (int s = 0; s < S; s++)
            for (int u = 0; u < 2; u++)
                energy += (m->posterior_marginals[(size_t)n * S + s] * m->p_outlier_total[(size_t)n * 2 + u] * rmxo_log_likelihood_total(m, n, s, u));
        energy += (m->p_outlier_total[(size_t)n * 2 + 0] * log(1. - m->prior_outlier_total));
        energy += (m->p_outlier_total[(size_t)n * 2 + 1] * log(m->prior_outlier_total));
    }
    for (int n = 0; n < N; n++) {
        for (int s = 0; s < S; s++)
            for (int v = 0; v < 2; v++)
                for (int w = 0; w < 2; w++)
                    energy += (m->posterior_marginals[(size_t)n * S + s] * m->p_outlier_allele[(size_t)n * 2 + v] * m->p_allele_swap[(size_t)n * 2 + w] * rmxo_log_likelihood_allele(m, n, s, v, w));
        energy += (m->p_outlier_allele[(size_t)n * 2 + 0] * log(1. - m->prior_outlier_allele));
        energy += (m->p_outlier_allele[(size_t)n * 2 + 1] * log(m->prior_outlier_allele));
    }
    for (int n = 0; n < N - 1; n++)
        for (int s = 0; s < S; s++)
            for (int s_ = 0; s_ < S; s_++)
                energy += T3(m->joint_posterior_marginals, m, n, s, s_) * T3(m->cached_log_transmat, m, n, s, s_);
    return energy;
}
double rmxo_calculate_elbo(rmxo_model *m) { return rmxo_variational_energy(m) - rmxo_variational_entropy(m); }

/* ---- bpmodel.pyx:1125-1195 M-step objectives ------------------------------ */
double rmxo_expected_log_likelihood(rmxo_model *m, const int64_t *sample) {
    const int N = m->N, S = m->S;
    double energy = 0.;
    for (int n = 0; n < N; n++) {
        if (sample[n] == 0) continue;
        for (int s = 0; s < S; s++)
            for (int u = 0; u < 2; u++)
                energy += (m->posterior_marginals[(size_t)n * S + s] * m->p_outlier_total[(size_t)n * 2 + u] * rmxo_log_likelihood_total(m, n, s, u));
    }
    for (int n = 0; n < N; n++) {
        if (sample[n] == 0) continue;
        for (int s = 0; s < S; s++)
            for (int v = 0; v < 2; v++)
                for (int w = 0; w < 2; w++)
                    energy += (m->posterior_marginals[(size_t)n * S + s] * m->p_outlier_allele[(size_t)n * 2 + v] * m->p_allele_swap[(size_t)n * 2 + w] * rmxo_log_likelihood_allele(m, n, s, v, w));
    }
    return energy;
}
int rmxo_expected_log_likelihood_partial_h(rmxo_model *m, const int64_t *sample, double *partial_h) {
    const int N = m->N, S = m->S, M = m->M;
    double seg[16];
    for (int c = 0; c < M; c++) partial_h[c] = 0.;
    for (int n = 0; n < N; n++) {
        if (sample[n] == 0) continue;
        for (int s = 0; s < S; s++)
            for (int u = 0; u < 2; u++) {
                log_likelihood_total_partial_h(m, n, s, u, seg);
                for (int c = 0; c < M; c++)
                    partial_h[c] += (m->posterior_marginals[(size_t)n * S + s] * m->p_outlier_total[(size_t)n * 2 + u] * seg[c]);
            }
    }
    for (int n = 0; n < N; n++) {
        if (sample[n] == 0) continue;
        for (int s = 0; s < S; s++)
            for (int v = 0; v < 2; v++)
                for (int w = 0; w < 2; w++) {
                    log_likelihood_allele_partial_h(m, n, s, v, w, seg);
                    for (int c = 0; c < M; c++)
                        partial_h[c] += (m->posterior_marginals[(size_t)n * S + s] * m->p_outlier_allele[(size_t)n * 2 + v] * m->p_allele_swap[(size_t)n * 2 + w] * seg[c]);
                }
    }
    return m->err;
}

/* ---- bpmodel.pyx:1197-1210 infer_cn --------------------------------------- */
int rmxo_infer_cn(rmxo_model *m, int64_t *cn /* [N][M][2] */, int64_t *state_sequence /* [N] or NULL */) {
    int64_t *ss = state_sequence ? state_sequence : (int64_t *)calloc(m->N, sizeof(int64_t));
    rmxo_max_product(m->framelogprob, m->log_transmat, ss, m->N, m->S);
    for (int n = 0; n < m->N; n++)
        for (int c = 0; c < m->M; c++)
            for (int ell0 = 0; ell0 < 2; ell0++) {
                int ell = ell0;
                if (m->p_allele_swap[(size_t)n * 2 + 1] > m->p_allele_swap[(size_t)n * 2 + 0]) ell = 1 - ell;
                /* the flipped index is used on BOTH sides (reference quirk: the swap is a no-op) */
                cn[((size_t)n * m->M + c) * 2 + ell] = CN(m, n, ss[n], c, ell);
            }
    if (!state_sequence) free(ss);
    return m->err;
}

/* ---- bpmodel.pyx:461-604 __cinit__ ---------------------------------------- */
static void *dupmem(const void *src, size_t bytes) { void *p = malloc(bytes ? bytes : 1); if (src && bytes) memcpy(p, src, bytes); return p; }

rmxo_model *rmxo_create(int num_clones, int num_segments, int num_breakpoints, int normal_contamination,
                        const int64_t *cn_states, int S, const int64_t *brk_states, int B,
                        const double *h_init, const double *l, const double *x, const double *y,
                        const int64_t *is_telomere, const int64_t *breakpoint_idx, const int64_t *breakpoint_orient,
                        double transition_penalty, double divergence_weight) {
    rmxo_model *m = (rmxo_model *)calloc(1, sizeof(rmxo_model));
    const int N = num_segments, M = num_clones, K = num_breakpoints;
    m->M = M; m->N = N; m->K = K; m->A = 2; m->S = S; m->B = B; m->normal_contamination = normal_contamination;
    if (M > 16) { free(m); return NULL; }
    m->cn_states = (int64_t *)dupmem(cn_states, sizeof(int64_t) * (size_t)N * S * M * 2);
    m->brk_states = (int64_t *)dupmem(brk_states, sizeof(int64_t) * (size_t)B * M);
    m->h = (double *)dupmem(h_init, sizeof(double) * M);
    m->l = (double *)dupmem(l, sizeof(double) * N);
    m->x = (double *)dupmem(x, sizeof(double) * N);
    m->y = (double *)dupmem(y, sizeof(double) * N * 2);
    int64_t mx = 0;
    for (size_t i = 0; i < (size_t)N * S * M * 2; i++) if (cn_states[i] > mx) mx = cn_states[i];
    for (size_t i = 0; i < (size_t)B * M; i++) if (brk_states[i] > mx) mx = brk_states[i];
    m->cn_max = (int)mx;
    m->total_likelihood_mask = (int64_t *)malloc(sizeof(int64_t) * N);
    m->allele_likelihood_mask = (int64_t *)malloc(sizeof(int64_t) * N);
    for (int n = 0; n < N; n++) { m->total_likelihood_mask[n] = 1; m->allele_likelihood_mask[n] = 1; }
    m->cn_states_total = (int64_t *)calloc((size_t)N * S * M, sizeof(int64_t));
    m->num_alleles_subclonal = (int64_t *)calloc((size_t)N * S, sizeof(int64_t));
    m->is_hdel = (int64_t *)calloc((size_t)N * S, sizeof(int64_t));
    m->is_loh = (int64_t *)calloc((size_t)N * S, sizeof(int64_t));
    for (int n = 0; n < N; n++)
        for (int s = 0; s < S; s++) {
            for (int c = 0; c < M; c++)
                for (int a = 0; a < 2; a++) TOT(m, n, s, c) += CN(m, n, s, c, a);
            /* :505  sum over alleles of (max over tumour clones != min over tumour clones) */
            int nsub = 0;
            for (int a = 0; a < 2; a++) {
                if (M > 1) {
                    int64_t lo = CN(m, n, s, 1, a), hi = lo;
                    for (int c = 2; c < M; c++) { int64_t v = CN(m, n, s, c, a); if (v < lo) lo = v; if (v > hi) hi = v; }
                    if (hi != lo) nsub++;
                }
            }
            m->num_alleles_subclonal[(size_t)n * S + s] = nsub;
            /* :506 all entries zero */
            int hd = 1;
            for (int c = 0; c < M; c++) for (int a = 0; a < 2; a++) if (CN(m, n, s, c, a) != 0) hd = 0;
            m->is_hdel[(size_t)n * S + s] = hd;
            /* :507 any allele with zero copies summed over clones */
            int loh = 0;
            for (int a = 0; a < 2; a++) { int64_t t = 0; for (int c = 0; c < M; c++) t += CN(m, n, s, c, a); if (t == 0) loh = 1; }
            m->is_loh[(size_t)n * S + s] = loh;
        }
    m->is_telomere = (int64_t *)dupmem(is_telomere, sizeof(int64_t) * N);
    m->breakpoint_idx = (int64_t *)dupmem(breakpoint_idx, sizeof(int64_t) * N);
    m->breakpoint_orient = (int64_t *)dupmem(breakpoint_orient, sizeof(int64_t) * N);
    m->transition_penalty = fabs(transition_penalty);
    m->divergence_weight = fabs(divergence_weight);
    m->breakpoint_side = (int64_t *)calloc(N, sizeof(int64_t));
    {
        int64_t *sides = (int64_t *)calloc(K > 0 ? K : 1, sizeof(int64_t));
        for (int n = 0; n < N; n++) {
            if (m->breakpoint_idx[n] < 0) continue;
            m->breakpoint_side[n] = sides[m->breakpoint_idx[n]];
            sides[m->breakpoint_idx[n]] += 1;
        }
        free(sides);
    }
    /* :547-554 favour single copy change */
    m->p_breakpoint = (double *)calloc((size_t)(K > 0 ? K : 1) * B, sizeof(double));
    for (int k = 0; k < K; k++) {
        double tot = 0.;
        for (int sb = 0; sb < B; sb++) {
            int64_t bm = brk_states[(size_t)sb * M];
            for (int c = 1; c < M; c++) if (brk_states[(size_t)sb * M + c] > bm) bm = brk_states[(size_t)sb * M + c];
            if (bm > 1) continue;
            m->p_breakpoint[(size_t)k * B + sb] = 1.; tot += 1.;
        }
        for (int sb = 0; sb < B; sb++) m->p_breakpoint[(size_t)k * B + sb] /= tot;
    }
    m->hmm_log_norm_const = 0.;
    size_t NS = (size_t)N * S, NSS = (size_t)(N > 1 ? N - 1 : 0) * S * S;
    m->framelogprob = (double *)malloc(sizeof(double) * NS);
    for (size_t i = 0; i < NS; i++) m->framelogprob[i] = 1.0;
    m->log_transmat = (double *)calloc(NSS ? NSS : 1, sizeof(double));
    m->cached_log_transmat = (double *)calloc(NSS ? NSS : 1, sizeof(double));
    m->posterior_marginals = (double *)malloc(sizeof(double) * NS);
    /* :564-567  ones, then divided by the row sum (float division, not 1/S literal) */
    for (size_t i = 0; i < NS; i++) m->posterior_marginals[i] = 1.0 / (double)S;
    m->joint_posterior_marginals = (double *)malloc(sizeof(double) * (NSS ? NSS : 1));
    for (size_t i = 0; i < NSS; i++) m->joint_posterior_marginals[i] = 1.0 / (double)((size_t)S * S);
    m->p_allele_swap = (double *)malloc(sizeof(double) * N * 2);
    m->p_outlier_total = (double *)malloc(sizeof(double) * N * 2);
    m->p_outlier_allele = (double *)malloc(sizeof(double) * N * 2);
    m->prior_outlier_total = 0.01; m->prior_outlier_allele = 0.01;
    for (int n = 0; n < N; n++) {
        m->p_allele_swap[2 * n] = 0.5; m->p_allele_swap[2 * n + 1] = 0.5;
        m->p_outlier_total[2 * n] = 1. - m->prior_outlier_total; m->p_outlier_total[2 * n + 1] = m->prior_outlier_total;
        m->p_outlier_allele[2 * n] = 1. - m->prior_outlier_allele; m->p_outlier_allele[2 * n + 1] = m->prior_outlier_allele;
    }
    m->negbin_r_0 = 500.; m->negbin_r_1 = 10.; m->negbin_hdel_mu = 1e-5; m->negbin_hdel_r_0 = 10.; m->negbin_hdel_r_1 = 1.;
    m->betabin_M_0 = 500.; m->betabin_M_1 = 10.; m->betabin_loh_p = 1e-3; m->betabin_loh_M_0 = 10.; m->betabin_loh_M_1 = 1.;
    m->transition_model = 0;
    m->p_d_len = (m->cn_max + 1) * 2;
    m->p_d = (double *)calloc(m->p_d_len, sizeof(double));
    rmxo_calculate_log_transmat(m, m->cached_log_transmat);
    return m;
}

void rmxo_destroy(rmxo_model *m) {
    if (!m) return;
    free(m->cn_states); free(m->cn_states_total); free(m->brk_states); free(m->num_alleles_subclonal);
    free(m->is_hdel); free(m->is_loh); free(m->is_telomere); free(m->breakpoint_idx); free(m->breakpoint_orient);
    free(m->breakpoint_side); free(m->p_breakpoint); free(m->framelogprob); free(m->log_transmat);
    free(m->cached_log_transmat); free(m->joint_posterior_marginals); free(m->posterior_marginals);
    free(m->p_allele_swap); free(m->p_outlier_total); free(m->p_outlier_allele); free(m->l); free(m->x); free(m->y);
    free(m->total_likelihood_mask); free(m->allele_likelihood_mask); free(m->h); free(m->p_d);
    free(m);
}

/* ---- accessors for the ctypes wrapper ------------------------------------- */
int rmxo_err(rmxo_model *m) { return m->err; }
const char *rmxo_errmsg(rmxo_model *m) { return m->errmsg; }
void rmxo_clear_err(rmxo_model *m) { m->err = 0; m->errmsg[0] = 0; }
int rmxo_dim(rmxo_model *m, int which) {
    switch (which) { case 0: return m->M; case 1: return m->N; case 2: return m->K; case 3: return m->S; case 4: return m->B; case 5: return m->cn_max; }
    return -1;
}
void *rmxo_array(rmxo_model *m, int id) {
    switch (id) {
    case 0: return m->h; case 1: return m->p_breakpoint; case 2: return m->framelogprob; case 3: return m->log_transmat;
    case 4: return m->cached_log_transmat; case 5: return m->posterior_marginals; case 6: return m->joint_posterior_marginals;
    case 7: return m->p_allele_swap; case 8: return m->p_outlier_total; case 9: return m->p_outlier_allele;
    case 10: return m->total_likelihood_mask; case 11: return m->allele_likelihood_mask; case 12: return m->cn_states_total;
    case 13: return m->num_alleles_subclonal; case 14: return m->is_hdel; case 15: return m->is_loh; case 16: return m->breakpoint_side;
    case 17: return m->cn_states; case 18: return m->brk_states; case 19: return m->is_telomere; case 20: return m->breakpoint_idx;
    case 21: return m->breakpoint_orient; case 22: return m->l; case 23: return m->x; case 24: return m->y;
    }
    return NULL;
}
double *rmxo_scalar(rmxo_model *m, int id) {
    switch (id) {
    case 0: return &m->negbin_r_0; case 1: return &m->negbin_r_1; case 2: return &m->negbin_hdel_mu; case 3: return &m->negbin_hdel_r_0;
    case 4: return &m->negbin_hdel_r_1; case 5: return &m->betabin_M_0; case 6: return &m->betabin_M_1; case 7: return &m->betabin_loh_p;
    case 8: return &m->betabin_loh_M_0; case 9: return &m->betabin_loh_M_1; case 10: return &m->prior_outlier_total;
    case 11: return &m->prior_outlier_allele; case 12: return &m->hmm_log_norm_const; case 13: return &m->transition_penalty;
    case 14: return &m->divergence_weight;
    }
    return NULL;
}
int *rmxo_transition_model(rmxo_model *m) { return &m->transition_model; }
