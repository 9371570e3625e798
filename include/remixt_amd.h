/*
 * remixt_amd.h -- C ABI of the MI355X-native ReMixT variational-HMM kernel.
 *
 * This is the drop-in boundary below the reference's Python object protocol.
 * The reference has no FFI on this path: `remixt/cn_model.py` talks to the
 * Cython class `remixt.bpmodel.RemixtModel` (reference remixt/bpmodel.pyx:397).
 * Every entry point below replaces one method / attribute of that class (cited
 * per function, paths relative to the reference root).  remixt_amd/bpmodel.py is
 * the ctypes binding that re-creates the `RemixtModel` protocol on top of it;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  All host arrays are
 *     C-contiguous float64 / int64 exactly as the reference passes them.
 *   - a `rmx_batch` holds ONE dataset (segments, state tables, topology) and R
 *     independent "restarts" (the reference's one-process-per-init_id fan-out,
 *     remixt/workflow.py:329-340).  The reference's RemixtModel == batch of 1.
 *   - every function returns an int status: 0 ok; RMX_E* otherwise, with a
 *     message retrievable through rmx_last_error().  RMX_EVALUE / RMX_EASSERT
 *     mirror the reference's ValueError / AssertionError sites.
 *   - restart ranges are [r0, r1).  All work is queued on the batch's HIP
 *     stream; functions that return host values synchronise that stream.
 *   - nothing is computed on the CPU: if no HIP device is usable the create
 *     call fails with RMX_EDEVICE (there is no fallback path).
 */
#ifndef REMIXT_AMD_H
#define REMIXT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMX_OK 0
#define RMX_EVALUE 1     /* reference: ValueError (bad shapes, nan ll, invalid p, x<=0 in digamma) */
#define RMX_EASSERT 2    /* reference: AssertionError (nan in alphas/betas/marginals, bpmodel.pyx:936-962) */
#define RMX_EDEVICE 3    /* HIP runtime failure / no device */
#define RMX_EUNSUPPORTED 4 /* outside the supported envelope (documented in DESIGN.md) */
#define RMX_EARG 5       /* bad argument to this C API */

#define RMX_MAX_CLONES 4

typedef struct rmx_batch rmx_batch;

/* Read-only problem description == positional arguments of
 * RemixtModel.__cinit__ (bpmodel.pyx:461-476), with the (N,S,M,2) state array
 * given in class-compressed form: segment n uses table cn_classes[seg_class[n]].
 * (cn_model.py:359-364 builds at most a handful of distinct tables: they differ
 * only in the normal-clone row.)  rmx_compress_cn_states() converts the dense
 * reference array. */
typedef struct rmx_problem {
    int32_t num_clones;            /* M (<= RMX_MAX_CLONES) */
    int32_t num_segments;          /* N */
    int32_t num_breakpoints;       /* K */
    int32_t num_cn_states;         /* S */
    int32_t num_brk_states;        /* B */
    int32_t num_classes;           /* C */
    int32_t normal_contamination;  /* bool */
    int32_t reserved;
    const int64_t *cn_classes;     /* [C][S][M][2] */
    const int32_t *seg_class;      /* [N] */
    const int64_t *brk_states;     /* [B][M] */
    const double *l;               /* [N] segment lengths */
    const double *x;               /* [N] total read counts */
    const double *y;               /* [N][2] allele read counts */
    const int64_t *is_telomere;    /* [N] */
    const int64_t *breakpoint_idx; /* [N] (-1 = none) */
    const int64_t *breakpoint_orient; /* [N] */
    double transition_penalty;     /* fabs() taken, bpmodel.pyx:534 */
} rmx_problem;

/* ids for rmx_set_param / rmx_get_param: the `cdef public` float attributes of
 * RemixtModel (bpmodel.pyx:433-454, 417-418, 422) */
enum rmx_param_id {
    RMX_P_NEGBIN_R_0 = 0, RMX_P_NEGBIN_R_1, RMX_P_NEGBIN_HDEL_MU, RMX_P_NEGBIN_HDEL_R_0, RMX_P_NEGBIN_HDEL_R_1,
    RMX_P_BETABIN_M_0, RMX_P_BETABIN_M_1, RMX_P_BETABIN_LOH_P, RMX_P_BETABIN_LOH_M_0, RMX_P_BETABIN_LOH_M_1,
    RMX_P_PRIOR_OUTLIER_TOTAL, RMX_P_PRIOR_OUTLIER_ALLELE, RMX_P_DIVERGENCE_WEIGHT,
    RMX_P_HMM_LOG_NORM_CONST /* read-only */, RMX_P_COUNT
};

/* ids for rmx_get_array / rmx_set_array: array attributes of RemixtModel
 * (bpmodel.pyx:405-442).  Shapes in the reference's layout. */
enum rmx_array_id {
    RMX_A_H = 0,                   /* f64 [M]            rw */
    RMX_A_P_BREAKPOINT,            /* f64 [K][B]         rw */
    RMX_A_P_ALLELE_SWAP,           /* f64 [N][2]         rw */
    RMX_A_P_OUTLIER_TOTAL,         /* f64 [N][2]         rw */
    RMX_A_P_OUTLIER_ALLELE,        /* f64 [N][2]         rw */
    RMX_A_POSTERIOR_MARGINALS,     /* f64 [N][S]         rw */
    RMX_A_FRAMELOGPROB,            /* f64 [N][S]         r  */
    RMX_A_TOTAL_LIKELIHOOD_MASK,   /* i64 [N]            rw (shared by the batch) */
    RMX_A_ALLELE_LIKELIHOOD_MASK,  /* i64 [N]            rw (shared by the batch) */
    RMX_A_LOG_TRANSMAT,            /* f64 [N-1][S][S]    r  materialised on request only */
    RMX_A_CACHED_LOG_TRANSMAT,     /* f64 [N-1][S][S]    r  materialised on request only */
    RMX_A_JOINT_POSTERIOR_MARGINALS, /* f64 [N-1][S][S]  r  materialised on request only */
    RMX_A_STATE_SEQUENCE,          /* i64 [N]            r  last Viterbi path */
    RMX_A_COUNT
};

/* -- lifetime ------------------------------------------------------------- */
/* RemixtModel.__cinit__ (bpmodel.pyx:461-604) for R restarts at once.
 * h_init: [R][M]; divergence_weight: [R] (fabs taken, :535). */
int rmx_batch_create(const rmx_problem *problem, int32_t num_restarts, const double *h_init,
                     const double *divergence_weight, int32_t device, rmx_batch **out);
int rmx_batch_destroy(rmx_batch *b);
/* Dense (N,S,M,2) int64 -> class-compressed form.  seg_class_out: [N];
 * classes_out: caller buffer for up to max_classes tables of S*M*2 int64.
 * Returns the number of classes through *num_classes (RMX_EUNSUPPORTED when
 * more than max_classes distinct tables occur). */
int rmx_compress_cn_states(const int64_t *cn_states, int32_t N, int32_t S, int32_t M,
                           int32_t max_classes, int32_t *seg_class_out, int64_t *classes_out,
                           int32_t *num_classes);
const char *rmx_last_error(void);
/* The restarts whose device-side checks (the reference's ValueError / AssertionError sites) fired in the calling thread's last
 * failing call: up to `cap` indices into out (out may be NULL with cap 0), returns their number.  Every flagged restart's
 * error state is cleared when the call reports, and rmx_last_error() describes the lowest one; a batched caller fails exactly
 * the listed restarts.  A failing call that flagged no restart (bad argument, device failure, unsupported shape) leaves the
 * list empty: it never describes an earlier call. */
int rmx_last_error_restarts(int32_t *out, int32_t cap);
/* use an externally owned HIP stream (e.g. torch's current stream); NULL = own */
int rmx_set_stream(rmx_batch *b, void *hip_stream);
int rmx_synchronize(rmx_batch *b);
/* derived sizes: 0 cn_max (bpmodel.pyx:489), 1 num chains, 2 num transition classes,
 * 3 num breakend segments, 4 padded row stride of [N][S] device arrays; 10 / 11 chains on the register-resident
 * forward-backward kernels / on the general one; 12 the forward-backward kernel the last update_p_cn launched for the
 * former (1 k_fbm: FP64 matrix cores at 4 restarts per workgroup, vector FMA at 2 / 1; 2 k_fbv: two-phase vector FMA, 3 k_fbk: weights from packed copy numbers, 4 k_fbq: matrix cores with
 * weights from 8-bit codes; 0: general kernel k_fb<0> only), 13 restarts per workgroup of that launch, 14 the lattice kernel of
 * the last decode (1 k_viterbi_reg, 2 k_viterbi_code, 3 k_viterbi, 4 k_viterbi_max, 5 k_viterbi_code_max, 6 k_viterbi_sad_max: above ~380 states); 18 workgroups per restart of that lattice, 19 the trace-back (1 parallel, 0 the sequential walk), 54 decodes repeated after a lattice cluster's watchdog ran out; 15 the largest number of restarts per workgroup of that launch (13 is the
 * smallest: k_fbm gives long chains fewer restarts per workgroup than short ones) */
int rmx_info(rmx_batch *b, int32_t what, int64_t *out);
/* -- tuning options ------------------------------------------------------- */
/* Not part of the reference protocol: which of this library's equivalent kernels / launch shapes run.  Results do not
 * depend on them beyond rounding (tests/test_hip_*.py compare the alternatives); they exist so that tests can put a
 * small problem on the launch shape a large one gets, and for A/B measurements.  rmx_set_default_option applies to
 * batches created afterwards (process-wide); rmx_set_option to one batch (creation-time options: RMX_EARG). */
enum rmx_option_id {
    RMX_OPT_FB_KERNEL = 0,      /* forward-backward: 0 auto, 1 general single-vector kernel for every chain, 2 tabulated weights
                                   instead of on-the-fly weights for grids beyond the register-resident kernel (S > 176), 3 the two-phase
                                   vector kernels (k_fbv up to 176 states, k_fbk above) instead of k_fbm / k_fbq */
    RMX_OPT_FB_NV,              /* restarts advanced by one forward-backward workgroup: 0 auto (k_fbm: 4 on the matrix cores, or 2 / 1 on the
                                   vector ALU while that many workgroups fit the chip's 256 CUs at once), 1, 2, 4 (the shapes that exist) */
    RMX_OPT_FB_BREAKEND_CODES,  /* 1 (default): breakend steps from pair codes + clone-product tables; 0: per-clone distance tables */
    RMX_OPT_FUSE_SWEEPS,        /* 1 (default): marginals + indicator updates + next frame pass as one kernel between sweeps */
    RMX_OPT_TWO_STREAMS,        /* 1 (default): breakend branch of a sweep on a second stream next to the marginal pass */
    RMX_OPT_VITERBI_PLAIN,      /* decode: 0 (default) the reference's own formulation -- maxima forward, the lattice rows kept, arg-maxima recomputed in the
                                   trace-back (k_viterbi_max / k_viterbi_code_max + k_backtrace_max, k_viterbi_sad_max + k_backtrace_sad above ~380 states; round 5); 1 the table-reading lattice kernel with
                                   back-pointers (k_viterbi); 2 round 4's register / code-table lattices with back-pointers */
    RMX_OPT_SEARCH_MODE,        /* parameter searches: 5 (default since round 5) the four standard searches together in rounds the device drives: optimiser
                                   state on the device, a kernel pair per round, queued back to back (half the latency of the host-driven rounds);
                                   0 the same shared rounds driven from the host (the default until round 4);
                                   1 one parameter at a time; 2 with table rebuilds per candidate; 3 with look-ahead evaluations;
                                   4 on the full objective; 6 = 0 with the final sums of a Nelder-Mead round folded into the objective kernel (last-block ticket: same bits, one launch
                                   fewer per round, but a release fence per block -- measured 5 % slower on the headline, not the default);
                                   7 = 5 as ONE launch: a request's blocks stay resident, publish their partial sums (sc1 stores, no fence) and each advances its own
                                   copy of the request's optimiser on all of them (same bits as 5) */
    RMX_OPT_ELL_DENSE,          /* 1: sampled objectives over all states instead of the lists of states with posterior mass */
    RMX_OPT_STRIP,              /* 1 (default): strip kernels for the (segment x state) passes when 32 < S <= 384 */
    RMX_OPT_CELL_CACHE,         /* creation time, 1 (default): cache the six likelihood values of every cell */
    RMX_OPT_SPARSE_TRIAL,       /* creation time, 1 (default): keep per-segment lists of states with posterior mass */
    RMX_OPT_FB_DEBUG,           /* creation time, 1: cycle counters of the forward-backward kernel through rmx_info(20..) */
    RMX_OPT_PAIRWISE_KERNEL,    /* breakend pairwise reductions: 0 auto (above 200 states k_pairwise_sp: the state pairs above the posterior threshold; else k_pairwise_be2), 1 general kernel
                                   (k_pairwise), 2 the dense pair-code kernel (k_pairwise_be2), 3 k_pairwise_sp, 4 k_pairwise_sp with a block of one wave per adjacency */
    RMX_OPT_PACE_SWEEPS,        /* 1: rmx_variational_update reaches each sweep's forward-backward point only after the previous sweep's
                                   forward-backward launch has finished on the device, instead of queueing all its sweeps at once; for restart groups that share a GPU -- at 355 states two paced groups of 8
                                   make 144 EM iterations/s, free-running ones 118 */
    RMX_OPT_FB_WG_BUDGET,       /* workgroups a forward-backward launch may have side by side when the shapes of its chains are chosen (0: 256, one per CU
                                   of an MI355X): tests set a small number to put a small problem on the mixed shapes a genome gets */
    RMX_OPT_TRIAL_KERNEL,       /* M-step trial passes over the lists of states with posterior mass: 0 (default) cells laid out flat over the threads
                                   (k_trial_flat: a segmented sum in list order), 1 a quarter wave per segment (k_trial_sparse) */
    RMX_OPT_GRAD_KERNEL,        /* h M-step rounds (objective + gradient on the samples): 0 (default) the lane chains laid out flat over the threads
                                   (k_gradflat_round; the same per-segment sums to the bit), 1 half a wave per sampled segment with the final sums
                                   folded in (round 4's form), 2 half a wave per segment and the final sums as a kernel of their own */
    RMX_OPT_STREAM_POOL,        /* 1 (default): a batch's two streams come from a process-wide pool per device and return to it when the batch is destroyed
                                   (streams are never destroyed: which hardware queue a role gets is decided once per process); 0: created and destroyed per batch */
    RMX_OPT_VITERBI_CLUSTER,    /* lattice above 176 states (k_viterbi_sad_max, default transition model): 0 (default) 4 workgroups per restart up to 300 states, 8
                                   above, each a share of the target states, the rows exchanged through memory step by step (halved while restarts x
                                   workgroups > 64; one where the device already holds 192 such workgroups); 1 one workgroup per restart (up to ~380
                                   states that is round 5's code-table lattice k_viterbi_code_max); 2 / 4 / 8 that many; 102 / 104 / 108: a test of the clusters' watchdog (one member never
                                   publishes its first row: the waits run out and the decode is repeated with one workgroup per restart) */
    RMX_OPT_TRACEBACK,          /* trace-back of the kept lattice rows (default transition model): 0 (default) in parallel -- the first arg-maximum of every target
                                   state of every row on the whole chip (k_bp_all), then the walk as a composition of maps (k_chase_compose / _ends / _fill);
                                   1 the sequential walk on one wave per restart (k_backtrace_max / k_backtrace_sad) */
    RMX_OPT_CU_PARTITION,       /* creation time: 0 (default) the batch's streams use the whole device; parts * 16 + index (parts 2 / 4 / 8): they are created with a CU mask
                                   -- range `index` of `parts` equal ranges of the device's CUs (hipExtStreamCreateWithCUMask): restart groups that do not share CUs */
    RMX_OPT_COUNT
};
int rmx_set_default_option(int32_t option_id, int32_t value);
int rmx_set_option(rmx_batch *b, int32_t option_id, int32_t value);
int rmx_get_option(rmx_batch *b, int32_t option_id, int32_t *value);

/* -- attributes ----------------------------------------------------------- */
int rmx_set_param(rmx_batch *b, int32_t r, int32_t param_id, double value);
int rmx_get_param(rmx_batch *b, int32_t r, int32_t param_id, double *value);
int rmx_set_transition_model(rmx_batch *b, int32_t model);  /* bpmodel.pyx:456, 606-616 */
int rmx_set_array(rmx_batch *b, int32_t r, int32_t array_id, const void *host_src);
int rmx_get_array(rmx_batch *b, int32_t r, int32_t array_id, void *host_dst);
/* The attributes p_outlier_total / p_outlier_allele (bpmodel.pyx:419-421) of restarts [r0, r1) in one transfer into registered
 * host memory the batch owns: *total / *allele point to [r1 - r0][N][2] float64, valid until the next call.  The two copies are
 * queued IN ORDER on the batch's stream (behind everything already queued there) and the call returns when they have landed.  What BreakpointModel.get_param_sample_weight (cn_model.py:323-352) reads at the start of every M-step. */
int rmx_fetch_indicators(rmx_batch *b, int32_t r0, int32_t r1, const double **total, const double **allele);
/* read-only derived state tables, (N,S[,M]) int64 like the reference attributes
 * cn_states_total / num_alleles_subclonal / is_hdel / is_loh (bpmodel.pyx:497-507):
 * which = 0..3 in that order */
/* calculate_log_transmat(out) (bpmodel.pyx:639-684): dense (N-1) x S x S log transition array for the CURRENT
 * p_breakpoint of restart r into a host array of 8 (N-1) S^2 bytes; no model state changes. */
int rmx_calculate_log_transmat(rmx_batch *b, int32_t r, double *dst);
/* Host-only helper of the weighted M-step sampling (cn_model.py:475-480 with p = weights): the index every
 * uniform draw u[j] selects from cumsum(p) / sum -- numpy's cumsum / searchsorted(side='right') with the same
 * accumulation order; *positive = count_nonzero(p > 0).  Runs without the GIL, no device involved. */
int rmx_weighted_search(const double *p, int64_t n, const double *u, int32_t k, int64_t *out, int64_t *positive);
/* One round of that sampling without a normalised copy of the weights (host only, no GIL, no device): weights w[i * stride] / norm
 * (a column of the (N, 2) outlier indicators the sample of negbin_r_* / betabin_M_* is weighted with, cn_model.py:323-352); the k
 * uniform draws u pick indices from the cumulative sum as rmx_weighted_search does; indices not yet in found[0 .. *nfound) are
 * appended in the order of the draws, up to cap (numpy's unique-by-first-occurrence of the concatenation).  *positive = number of
 * positive weights (fewer than the sample size: numpy's "Fewer non-zero entries in p than size"). */
int rmx_weighted_sample_round(const double *w, int64_t n, int64_t stride, double norm, const double *u, int32_t k,
                              int64_t *found, int32_t *nfound, int32_t cap, int64_t *positive);
int rmx_get_state_table(rmx_batch *b, int32_t which, int64_t *host_dst);

/* -- coordinate updates (bpmodel.pyx cpdef methods), restarts [r0,r1) ------ */
int rmx_update_framelogprob(rmx_batch *b, int32_t r0, int32_t r1);      /* :898-919 */
int rmx_update_p_cn(rmx_batch *b, int32_t r0, int32_t r1);              /* :921-962 */
int rmx_update_p_breakpoint(rmx_batch *b, int32_t r0, int32_t r1);      /* :964-985 */
int rmx_update_p_outlier_total(rmx_batch *b, int32_t r0, int32_t r1);   /* :987-1003 */
int rmx_update_p_outlier_allele(rmx_batch *b, int32_t r0, int32_t r1);  /* :1005-1023 */
int rmx_update_p_allele_swap(rmx_batch *b, int32_t r0, int32_t r1);     /* :1025-1042 */
/* the five updates in BreakpointModel.variational_update order (cn_model.py:444-460),
 * `iters` times, without host round trips */
int rmx_variational_update(rmx_batch *b, int32_t r0, int32_t r1, int32_t iters);

/* -- objectives ----------------------------------------------------------- */
/* out: [r1-r0] */
int rmx_calculate_elbo(rmx_batch *b, int32_t r0, int32_t r1, double *elbo_out);       /* :1119-1123 */
/* calculate_elbo in two halves (round 5): _begin queues the ELBO of restarts [r0, r1) on the batch's stream -- kernels and the copy of the
 * results -- and returns without waiting; _end waits for it and returns the values (and raises what rmx_calculate_elbo would have).  The
 * batched EM driver queues the next iteration's sweeps in between: the value is only recorded (cn_model.py:420-428), nothing waits for it. */
int rmx_calculate_elbo_begin(rmx_batch *b, int32_t r0, int32_t r1);
int rmx_calculate_elbo_end(rmx_batch *b, double *elbo_out);
int rmx_calculate_variational_energy(rmx_batch *b, int32_t r0, int32_t r1, double *out);  /* :1060-1117 */
int rmx_calculate_variational_entropy(rmx_batch *b, int32_t r0, int32_t r1, double *out); /* :1044-1058 */
/* sample: int64 [N] 0/1 mask as in the reference (:1125, :1159); partial_h_out may be
 * NULL; when given it receives [M] (:1159-1195). */
int rmx_expected_log_likelihood(rmx_batch *b, int32_t r, const int64_t *sample, double *ell_out,
                                double *partial_h_out);
/* The M-steps of BreakpointModel (cn_model.py:482-569) evaluate the objective hundreds of times on
 * one fixed sample: rmx_set_sample uploads the mask once; rmx_expected_log_likelihood with
 * sample == NULL then re-uses it. */
int rmx_set_sample(rmx_batch *b, int32_t r, const int64_t *sample);
/* E[ll] on the current sample for G values of one likelihood parameter (the grid stage of
 * scipy.optimize.brute, cn_model.py:553-558) with one host round trip; G <= 64.  Leaves the
 * parameter at values[G-1], as G sequential evaluations would. */
int rmx_expected_ll_param_grid(rmx_batch *b, int32_t r, int32_t param_id, const double *values, int32_t G, double *out);
/* Lock-step M-steps over several restarts (remixt_amd/restarts.py): one candidate value of one
 * likelihood parameter per listed restart (distinct restarts), each evaluated on its own current
 * sample; out[i] belongs to restarts[i].  One host round trip for the whole list. */
int rmx_expected_ll_batch(rmx_batch *b, int32_t nreq, const int32_t *restarts, int32_t param_id, const double *values, double *out);
/* The whole scipy.optimize.brute search of BreakpointModel.update_param (cn_model.py:553-561) for one
 * likelihood parameter of every listed restart, in lock step: G grid values (np.mgrid[lo:hi:G*1j]),
 * first arg-min, then scipy.optimize.fmin (Nelder-Mead) restated as a host state machine; every
 * round of objective evaluations is one rmx_expected_ll_batch.  nll is +inf outside [lo, hi]
 * (cn_model.py:542-543).  xopt[i] = the optimiser's result for restarts[i]; the parameter is left at
 * the value of the restart's last evaluation, as the sequential scipy run leaves it. */
int rmx_param_search(rmx_batch *b, int32_t nreq, const int32_t *restarts, int32_t param_id, double lo, double hi,
                     const double *grid, int32_t G, double *xopt);
/* The four standard searches of every listed restart in the same evaluation rounds (they move disjoint
 * likelihood components and do not write to the model while searching, cn_model.py:533-561 x 4):
 * param_ids [nparams] a subset of negbin_r_0/1, betabin_M_0/1; slot j = position in param_ids, its
 * samples given by rmx_set_sample_slot; lo / hi [nparams]; grids [nparams][G]; xopt / lastval
 * [nparams][nreq] = optimiser result and last evaluated point (the state the acceptance test of
 * cn_model.py:563-569 looks at).  The model is not modified.  RMX_EUNSUPPORTED: use rmx_param_search. */
int rmx_set_sample_slot(rmx_batch *b, int32_t r, int32_t slot, const int64_t *sample);
/* The M-step samples (cn_model.py:475-480) of several restarts in one call and one device transfer: list i
 * is the ascending segment indices indices[offsets[i] .. offsets[i+1]) of restart restarts[i]; slots[i] = -1:
 * the restart's current sample (as rmx_set_sample), 0..3: its parameter slot (as rmx_set_sample_slot). */
int rmx_set_sample_lists(rmx_batch *b, int32_t nlists, const int32_t *restarts, const int32_t *slots,
                         const int32_t *offsets, const int32_t *indices);
int rmx_param_search_multi(rmx_batch *b, int32_t nreq, const int32_t *restarts, int32_t nparams,
                           const int32_t *param_ids, const double *lo, const double *hi,
                           const double *grids, int32_t G, double *xopt, double *lastval);
/* Lock-step h M-step (BreakpointModel.update_h, cn_model.py:482-531): one candidate haploid-depth
 * vector h[i][0..M) per listed restart; out[i][0] = E[ll] (calculate_expected_log_likelihood,
 * bpmodel.pyx:1125-1157) and out[i][1..M] = dE[ll]/dh (calculate_expected_log_likelihood_partial_h,
 * bpmodel.pyx:1159-1195) on restart i's current sample; out is [nreq][1 + RMX_MAX_CLONES].  Leaves
 * h[i] set on the restart, as `model.h = h` followed by the two calls would. */
int rmx_expected_ll_h_batch(rmx_batch *b, int32_t nreq, const int32_t *restarts, const double *h, double *out);
/* E[ll] over ALL segments (the reference passes a mask of ones, cn_model.py:497, :524, :549, :563)
 * for restarts [r0, r1); out: [r1-r0]. */
int rmx_expected_ll_full(rmx_batch *b, int32_t r0, int32_t r1, double *out);
/* The M-step's accept test (`ell_after < ell_before`, cn_model.py:497-505, 563-569) without committing the
 * tried values: full-data E[ll] at the current (changed) h / parameters, evaluated into scratch
 * per-segment expectations -- same numbers as rmx_expected_ll_full -- leaving the restart's own
 * expectations, cell cache and staleness flags untouched.  rmx_trial_rollback then puts the previous
 * value back (param_id >= 0: likelihood parameter, values[0]; param_id < 0: h, values[0..M)) and declares
 * the untouched expectations current again, so a rejected update costs no second pass over the cells.
 * Valid only in the sequence rmx_expected_ll_full -> sampled evaluations -> rmx_expected_ll_full_trial ->
 * accept (set the new value) | rmx_trial_rollback. */
int rmx_expected_ll_full_trial(rmx_batch *b, int32_t r0, int32_t r1, double *out);
int rmx_trial_rollback(rmx_batch *b, int32_t r, int32_t param_id, const double *values);
/* The full-data E[ll] split into the four likelihood components that negbin_r_0, negbin_r_1, betabin_M_0 and
 * betabin_M_1 move (out [r1-r0][4]; their sum is rmx_expected_ll_full up to rounding): trial = 0 at the committed values,
 * trial = 1 with the changed parameters on trial (as rmx_expected_ll_full_trial).  The accept tests of the four
 * parameters (cn_model.py:563-569, one update_param after the other) then take ONE pass over the cells: E[ll] with
 * parameter j on trial and the earlier ones decided is a sum of component values of the two calls.  Afterwards, per
 * parameter: rmx_set_param (accept) or rmx_trial_rollback (reject; only that parameter's components become current again).
 * trial = 2: the component sums of the expectations the LAST trial pass over this range left in scratch (rmx_expected_ll_full_trial or
 * trial = 1), without another pass: right after the h M-step's accept test they are E[ll] at (accepted h, committed parameters),
 * i.e. the "before" of the parameter accept tests, for every restart whose h was accepted (restarts whose h was rolled back: trial = 0,
 * which costs no pass either).  A restart's own expectations then stay stale until the next ELBO / sweep: ONE full refresh per EM
 * iteration instead of one after the h M-step and one after the parameter M-steps.
 * trial = 3: the same per restart, for a range the last rmx_expected_ll_full_trial covered: the scratch sums for the restarts whose own
 * expectations are stale (h kept), the sums of their own expectations for the others (rolled back, or untouched) -- the mixed outcome
 * of a batch's h accept tests without a refresh pass; RMX_EUNSUPPORTED if that trial pass is not the last one over the range. */
int rmx_expected_ll_components(rmx_batch *b, int32_t r0, int32_t r1, int32_t trial, double *out);
/* per-cell values, for tests (:751-776, :809-853): u/v/w in {0,1} */
int rmx_log_likelihood_total(rmx_batch *b, int32_t r, int32_t n, int32_t s, int32_t u, double *out);
int rmx_log_likelihood_allele(rmx_batch *b, int32_t r, int32_t n, int32_t s, int32_t v, int32_t w, double *out);

/* The other per-cell cpdef methods of RemixtModel for segment n, state s (no caller in cn_model.py; part of the protocol):
 * which = 0 calculate_expected_total_reads (bpmodel.pyx:686-698) -> out[0]
 *         1 calculate_expected_total_reads_partial_h (:700-708) -> out[0..M)
 *         2 calculate_expected_allele_ratio (:710-725) -> out[0]     (RMX_EVALUE: total_depth <= 0)
 *         3 calculate_expected_allele_ratio_partial_h (:727-745) -> out[0..M)
 *         4 calculate_log_prior_cn (:747-750) -> out[0]
 *         5 calculate_log_likelihood_total_partial_h (:778-807), outlier state u -> out[0..M)
 *         6 calculate_log_likelihood_allele_partial_h (:855-896), outlier state v, allele w -> out[0..M)
 * out must hold RMX_MAX_CLONES doubles. */
int rmx_cell_quantity(rmx_batch *b, int32_t r, int32_t n, int32_t s, int32_t which, int32_t u, int32_t v, int32_t w, double *out);

/* -- decoding ------------------------------------------------------------- */
/* infer_cn (:1197-1210): Viterbi over the framelogprob / log_transmat of the last
 * update_p_cn; cn_out int64 [N][M][2]; logprob_out may be NULL */
int rmx_infer_cn(rmx_batch *b, int32_t r, int64_t *cn_out, double *logprob_out);
/* the same for restarts r0 .. r0+nr-1 with their lattices running side by side: cn_out int64
 * [nr][N][M][2], logprob_out [nr] or NULL (the per-restart decode of analysis/pipeline.py:196-206) */
int rmx_infer_cn_batch(rmx_batch *b, int32_t r0, int32_t nr, int64_t *cn_out, double *logprob_out);

/* -- module-level functions on caller-supplied dense inputs ----------------- */
/* sum_product (:1213-1246): f [N][S], T [N-1][S][S] -> alphas, betas [N][S] */
int rmx_sum_product(const double *f, const double *T, double *alphas, double *betas,
                    int32_t N, int32_t S, int32_t device);
/* max_product (:1296-1333): returns path int64 [N] and the log probability */
int rmx_max_product(const double *f, const double *T, int64_t *state_sequence, double *logprob,
                    int32_t N, int32_t S, int32_t device);

/* -- measurement (bench.py): HIP events on the batch stream ----------------- */
int rmx_timer_start(rmx_batch *b);
int rmx_timer_stop(rmx_batch *b, double *elapsed_ms);
/* per-kernel accumulated device time since the last reset (HIP events around
 * each launch).  rmx_profile_enable: 0 = off, 1 = every kernel, 2 = only the kernels of the
 * variational sweep (the M-step objective kernels are launched thousands of times per EM
 * iteration; two event records per launch would perturb the host loop).  kernel ids: see
 * rmx_kernel_name(). */
int rmx_profile_enable(rmx_batch *b, int32_t on);
int rmx_profile_get(rmx_batch *b, int32_t kernel_id, double *total_ms, int64_t *launches);
int rmx_profile_reset(rmx_batch *b);
const char *rmx_kernel_name(int32_t kernel_id);
int rmx_num_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* REMIXT_AMD_H */
