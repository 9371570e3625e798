// rmx_api.hip -- host side of the C ABI declared in include/remixt_amd.h.
// Owns device memory, the restart batch bookkeeping (what is stale after which
// attribute write) and the launch sequences that replace the methods of
// remixt.bpmodel.RemixtModel (reference remixt/bpmodel.pyx:397-1210).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <limits>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <memory>

#include "rmx_kernels.h"
#include "rmx_host.h"

// ---------------------------------------------------------------------------
static thread_local std::string g_err;
// the restarts a failing batched call flagged (this thread's last failure): rmx_last_error_restarts.  A failure that
// flags no restart (bad argument, device error, unsupported shape) leaves the list EMPTY: fail() clears it, the
// device-check reporters fill it and report through fail_flagged()
static thread_local std::vector<int32_t> g_err_restarts;
static int fail(int code, const std::string &msg) { g_err = msg; g_err_restarts.clear(); return code; }
static int fail_flagged(int code, const std::string &msg) { g_err = msg; return code; }
static inline long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return fail(RMX_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

enum KernelId {
    KID_STATE_TABLES = 0, KID_SEG_CONST, KID_FRAMELOGPROB, KID_FB, KID_MARGINALS, KID_MARGINALS_AB, KID_OUTLIER_TOTAL,
    KID_OUTLIER_ALLELE, KID_ALLELE_SWAP, KID_BRK_LUT, KID_PAIRWISE, KID_BRK_UPDATE, KID_ELBO_SEG, KID_ELBO_FINAL,
    KID_ELL_LIST, KID_ELL_FINAL, KID_ELL_FULL, KID_VITERBI, KID_BACKTRACE, KID_OTHER, KID_COUNT
};
static const char *kKernelNames[KID_COUNT] = {
    "k_state_tables", "k_seg_const", "k_framelogprob", "k_fb", "k_marginals<true>", "k_marginals<false>", "k_update_outlier_total",
    "k_update_outlier_allele", "k_update_allele_swap", "k_brk_lut", "k_pairwise", "k_brk_update", "k_elbo_seg", "k_elbo_final",
    "k_ell_list", "k_ell_final", "k_ell_full", "k_viterbi", "k_backtrace", "other"};

struct ProfRec { int id; hipEvent_t a, b; };

// Process-wide pool of HIP streams per device and ROLE (0: a batch's main stream -- sweeps and M-step; 1: its second stream -- the breakend branch
// of a sweep).  The runtime maps streams onto a handful of hardware queues when they are created (the least referenced queue), and streams that
// share a queue run their kernels one after the other: with a stream pair created and destroyed per batch, WHICH streams of a later pair of
// restart groups shared a queue depended on what earlier batches a Python program had dropped but not yet destroyed (two groups' forward-backward
// launches serialised: 106-111 instead of 148 EM iterations/s at 355 states, profiles/r04_hw_queues.txt).  Pooled streams are created on demand,
// handed back when their batch is destroyed and never destroyed themselves: a role's queue is decided once per process, and a batch built later
// gets a stream with the placement the first ones got (profiles/r05_stream_pool.txt).
// (pool key: role 0 / 1 and the CU partition of option cu_partition -- 0: the whole device; parts * 16 + index: partition `index` of `parts` equal ranges of the CU mask)
struct StreamPool { std::mutex mu; std::map<int, std::vector<hipStream_t>> idle; int created[2] = {0, 0}; };
static StreamPool g_stream_pool[16];
static hipError_t create_stream_for(int dev, int part, hipStream_t *out) {
    if (part <= 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    const int parts = part >> 4, idx = part & 15;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
    const int ncu = prop.multiProcessorCount, per = ncu / parts;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c = idx * per; c < (idx + 1) * per; c++) mask[c >> 5] |= 1u << (c & 31);
    return hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
}
static int pool_acquire(int dev, int role, hipStream_t *out, int part = 0) {
    StreamPool &p = g_stream_pool[dev & 15];
    std::lock_guard<std::mutex> lk(p.mu);
    std::vector<hipStream_t> &idle = p.idle[role + 2 * part];
    if (!idle.empty()) { *out = idle.front(); idle.erase(idle.begin()); return RMX_OK; }      // (the oldest first)
    if (create_stream_for(dev, part, out) != hipSuccess) return RMX_EDEVICE;
    p.created[role]++;
    return RMX_OK;
}
static void pool_release(int dev, int role, hipStream_t s, int part = 0) {
    hipStreamSynchronize(s);
    StreamPool &p = g_stream_pool[dev & 15];
    std::lock_guard<std::mutex> lk(p.mu);
    p.idle[role + 2 * part].push_back(s);
}

// tuning options (include/remixt_amd.h rmx_option_id): process-wide defaults, copied into a batch at creation
static int g_opt_default[RMX_OPT_COUNT] = {0, 0, 1, 1, 1, 0, 5, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0};      // (search_mode 5 since round 5)
static std::mutex g_opt_mu;

struct rmx_batch {
    hipEvent_t ev_pace = nullptr;      // pace_sweeps: the forward-backward of the previous sweep of this call has been queued up to here
    int opt[RMX_OPT_COUNT];
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // breakend branch of a sweep (pairwise reductions, p_breakpoint) next to the marginal pass
    hipEvent_t ev_fb = nullptr, ev_brk = nullptr;
    bool own_stream = false, pooled_stream = false, pooled_stream2 = false;
    Dev d{};
    int R = 0;
    // host copies of the problem (for table rebuilds / decoding)
    std::vector<int64_t> cn_classes;   // [C][S][M][2]
    std::vector<int32_t> seg_class, brk_idx, brk_orient, tclass, brk_slot, be_n;
    std::vector<int64_t> brk_states, is_telomere;
    std::vector<std::pair<int, int>> tc_pairs;
    std::vector<double> Tsum;          // [TC] sum of plain log-transition entries (current model)
    // per-restart host state
    std::vector<RestartParams> rp;
    std::vector<char> tables_dirty, segc_dirty, ab_dirty;
    std::vector<double> segc_built;      // [R][8] the dispersion parameters the per-segment constants on the device were built from
    std::vector<char> sig_valid;       // per restart: the lists of states with posterior mass belong to the current posterior
    std::vector<int> cache_stale;      // components of the cell cache that are not current (CM_* bits); 15 = nothing cached
    bool use_cache = false;
    std::vector<int> comp_dirty;       // which components of (A, B, PF/PP) are stale: CM_* bits, 16 = PF/PP
    std::vector<int> comp_base;        // ... of them, those stale because h changed (a parameter's rollback does not make these current again)
    std::vector<int> lt_valid;
    std::vector<int> lt_model, cached_model;   // per restart: the transition model of the log_transmat / cached_log_transmat snapshot
    double *Tval_m[2] = {nullptr, nullptr}; int8_t *af_m[2] = {nullptr, nullptr};   // plain tables of either transition model once it has been current
    std::vector<double> plain_T_init;  // [R]
    std::vector<double> logZ;          // last hmm_log_norm_const
    std::vector<char> logz_dirty;
    int *d_lt_valid = nullptr;
    // scratch
    double *d_partial = nullptr;       // ELBO partials [R][ELBO_BLOCKS][3]; E[ll] component partials [R][4][ELBO_BLOCKS]
    double *d_be_e = nullptr;          // ELBO: energy term of every breakend slot [R][NBE]
    double *d_out4 = nullptr;          // [R][4]
    double *d_ell_partial = nullptr;   // [max(N,ELBO_BLOCKS)][1+MAXC]
    double *d_ell_out = nullptr;       // [1+MAXC]
    double *h_pinned = nullptr;        // pinned staging [max(4R, 16)]
    hipEvent_t ev_copy = nullptr;
    double *h_ind = nullptr;           // pinned [2][R][N][2]: the outlier indicators of all restarts for the M-step's weighted samples (rmx_fetch_indicators)
    int32_t *h_lists = nullptr;        // pinned staging of rmx_set_sample_lists (read by the device in place)
    size_t h_lists_cap = 0;
    hipEvent_t ev_lists = nullptr;     // the scatter kernel that last read h_lists
    uint32_t *h_err = nullptr;         // pinned [4R+64]: landing area of check_errors / per-request error words
    int32_t *d_sample = nullptr;       // [R][N] index lists of the current M-step samples
    NmState *d_nm_state = nullptr;      // [64]: optimiser state of the device-driven search rounds (k_search_round / k_search_advance)
    NmLayout nm_lay = {nullptr, nullptr, nullptr}; double *d_nm_partial = nullptr; size_t nm_partial_cap = 0;      // their cell layout and per-block sums
    int32_t *d_msample = nullptr, *d_mcounts = nullptr;   // [4][R][N], [4][R]: per parameter slot, for rmx_param_search_multi
    std::vector<int> msample_count;    // [4][R], -1 = not set
    double *d_mpartial = nullptr; size_t mpartial_cap = 0;
    std::vector<std::vector<int64_t>> sample_cache; std::vector<int> sample_count;
    double *d_grid_out = nullptr;      // [R][64][1+MAXC]
    GradFlatLayout gf_lay = {nullptr, nullptr, nullptr, nullptr};      // layout of the flat h-round kernel (made once per M-step: gf_sig)
    std::vector<double> gf_sig;        // what the layout was made for: request list, sample / list epochs, likelihood parameters
    std::vector<long long> sample_epoch, sig_epoch;
    int32_t *gf_nblk_host = nullptr;   // [16] host-visible: blocks per request
    bool gf_first = false;
    unsigned *d_done = nullptr;        // [max(R, 64)] per-request completion tickets of the fused objective kernels (zero between launches)
    int32_t *d_rlist = nullptr, *d_counts = nullptr; RestartParams *d_rp_stage = nullptr; double *d_batch_out = nullptr;   // [R] each
    void *h_batch = nullptr;           // pinned staging for the batched objective
    std::vector<int32_t> plain_list; int32_t *d_plain_list = nullptr; double *d_plain_jt = nullptr;
    // viterbi
    int last_search_blocks = 0, last_search_persist = 0;
    unsigned *d_vflag = nullptr; int cluster_timeouts = 0;      // "a cluster member gave up waiting" of the last lattice launch; how often that has happened
    int last_viterbi_wgs = 1;      // workgroups per restart of the last lattice (k_viterbi_sad_max<., true>: clusters)
    int32_t *d_vit_special = nullptr; int n_vit_special = -1;      // adjacencies that are not plain class-0 ones, ascending (k_viterbi_max)
    double *h_elbo = nullptr; hipEvent_t ev_elbo = nullptr; bool elbo_pending = false, elbo_sync = false; int elbo_r0 = 0, elbo_r1 = 0;      // rmx_calculate_elbo_begin / _end
    double *d_vrow = nullptr; size_t vrow_cap = 0;      // [nr][N][SR] lattice rows of k_viterbi_max / k_viterbi_code_max (pads 0)
    uint16_t *d_bp = nullptr; double *d_final = nullptr; int64_t *d_path = nullptr; double *d_logprob = nullptr;
    std::vector<int64_t> last_path; int vit_cap = 0; size_t bp_cap = 0;
    uint16_t *d_comp = nullptr; int32_t *d_ends = nullptr; size_t comp_cap = 0, ends_cap = 0; int last_traceback = 0;      // parallel trace-back: composed maps [nr][NBLK][S], block end states [nr][NBLK]
    uint8_t *d_vit_code = nullptr; double *d_vit_val = nullptr; bool vit_code_ok = false, vit_mul_ok = false;   // 8-bit codes of T(i, o) of class 0 + their values (k_viterbi_code)
    // FB launch configuration
    FbLaunch fbG{}; size_t fbG_lds = 0;   // generic kernel configuration
    int n_fast = 0, n_generic = 0;
    int num_cus = 256;                                   // compute units of the batch's device
    std::vector<int32_t> h_list_fast, h_chain_len;       // chains on the register-resident kernels; segments of every chain
    struct FbItems { int n = 0; int4 *dev = nullptr; int nv_min = 4, nv_max = 1; };
    std::map<std::tuple<int, int, int>, FbItems> fb_items;   // work-item tables of k_fbm launches by (r0, r1, pinned restarts per workgroup)
    // which kernels the last update_p_cn / decode launched (rmx_info 12..14; tests assert the shape they mean to cover)
    int last_fb_kernel = 0, last_fb_nv = 0, last_fb_nv_max = 0, last_viterbi = 0;
    // host-side stage clocks of the batched sampled-objective rounds (rmx_info 60..63): ns spent preparing + launching, ns waiting for the
    // device, ns after the wait, rounds
    long long t_launch_ns = 0, t_wait_ns = 0, t_post_ns = 0, n_rounds = 0;
    unsigned long long *d_dbg = nullptr;
    int fbv_rpt = 0;   // rows per slice of the multi-vector kernel (0 = not applicable)
    int G = 64;
    // all device allocations (freed on destroy)
    std::vector<void *> allocs;
    // profiling
    int fbk_kmax = 0;      // largest allele distance of a uniform class (k_fbq keeps 64 table entries, k_fbk FBK_WKN)
    bool fbk_ok = false;   // k_fbk usable: log-weights of every uniform class equal -pen * min(SAD, SAD swapped)
    std::vector<double> trial_comp; int trial_comp_r0 = -1, trial_comp_r1 = -1;   // component sums of the last rmx_expected_ll_full_trial's scratch expectations
    int scratch_r0 = -1, scratch_r1 = -1;   // restarts whose scratch expectations (d_A2, d_Bv2) the last trial pass wrote (-1: none)
    double *d_A2 = nullptr, *d_Bv2 = nullptr;   // [R][N][2], [R][N][4]: (A, B) of a trial parameter value (rmx_expected_ll_full_trial)
    uint32_t *d_cnpack = nullptr, *d_totpack = nullptr, *d_cnpack2 = nullptr; double *d_wk = nullptr;   // [C][S], [C][S], [C][S] (third tumour clone), [TC][FBK_WKN]
    int pe2p = 0;     // padded row length of pe2_lt (0: no product table)
    int spc = 0;      // row stride of the pair-code table
    bool pcode_ok = false;
    int max_adist = 0; // largest allele distance in af / ab
    int be_cap = 4;   // LDS ints for a chain's breakend adjacency list (largest chain + pad)
    int prof = 0;     // 0 off, 1 all kernels, 2 variational-sweep kernels only
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[KID_COUNT] = {0};
    long long prof_n[KID_COUNT] = {0};
    hipEvent_t tm_a = nullptr, tm_b = nullptr;
    std::vector<hipEvent_t> done_ev;   // [R] completion marker of a restart's last queued read-back
    std::mutex mu;                     // guards the profiling lists and launch sequences of concurrent callers
};

template <typename T> static int dalloc(rmx_batch *b, T **p, size_t count) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(RMX_EDEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    b->allocs.push_back(q);
    *p = (T *)q;
    return RMX_OK;
}
static void dfree(rmx_batch *b, void *q) {
    if (!q) return;
    auto it = std::find(b->allocs.begin(), b->allocs.end(), q);
    if (it != b->allocs.end()) b->allocs.erase(it);
    hipFree(q);
}
template <typename T> static int dupload(rmx_batch *b, const T **p, const std::vector<T> &v) {
    T *q = nullptr;
    int rc = dalloc(b, &q, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHK(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return RMX_OK;
}

// ---- profiling wrappers ---------------------------------------------------
static hipEvent_t get_event(rmx_batch *b) {
    if (!b->ev_pool.empty()) { hipEvent_t e = b->ev_pool.back(); b->ev_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
static void prof_collect(rmx_batch *b) {
    for (auto &pr : b->prof_pending) {
        hipEventSynchronize(pr.b);
        float ms = 0; hipEventElapsedTime(&ms, pr.a, pr.b);
        b->prof_ms[pr.id] += ms; b->prof_n[pr.id] += 1;
        b->ev_pool.push_back(pr.a); b->ev_pool.push_back(pr.b);
    }
    b->prof_pending.clear();
}
struct ProfScope {
    rmx_batch *b; int id; hipEvent_t a{}, e{};
    // callers hold b->mu (see LOCK()) whenever profiling may be on
    bool on;
    // level 1: every kernel; level 2: only the kernels of the variational sweep (a few dozen launches
    // per EM iteration -- the M-step objective kernels are launched thousands of times and two event
    // records per launch would slow the host loop being measured)
    static bool sweep_kernel(int id) {
        return id == KID_FRAMELOGPROB || id == KID_FB || id == KID_MARGINALS || id == KID_PAIRWISE || id == KID_BRK_LUT ||
               id == KID_BRK_UPDATE || id == KID_OUTLIER_TOTAL || id == KID_OUTLIER_ALLELE || id == KID_ALLELE_SWAP;
    }
    ProfScope(rmx_batch *b_, int id_) : b(b_), id(id_) {
        on = b->prof == 1 || (b->prof == 2 && sweep_kernel(id));
        if (on) { a = get_event(b); e = get_event(b); hipEventRecord(a, b->stream); }
    }
    // start again from here on the stream, under another kernel id
    void restart(int id_) { id = id_; if (on) hipEventRecord(a, b->stream); }
    void cancel() { if (on) { b->ev_pool.push_back(a); b->ev_pool.push_back(e); on = false; } }
    ~ProfScope() {
        if (on) { hipEventRecord(e, b->stream); b->prof_pending.push_back({id, a, e}); if (b->prof_pending.size() > 8192) prof_collect(b); }
    }
};

// ---- error translation -------------------------------------------------------
static int translate_error(rmx_batch *b, int r, uint32_t v);
// Every flagged error word of a call is cleared on the device before the call reports -- a word left set would
// surface in a later, unrelated call of that restart -- and the lowest flagged restart is the one reported;
// the full list stays readable through rmx_last_error_restarts so that a batched caller can fail only those.
static int check_errors(rmx_batch *b, int r0, int r1) {
    uint32_t *e = b->h_err;      // pinned: the copy is queued behind the caller's own result copy, one wait for both
    HIPCHK(hipMemcpyAsync(e, b->d.err, sizeof(uint32_t) * b->R, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    int first = -1;
    for (int r = r0; r < r1; r++) if (e[r]) { if (first < 0) { first = r; g_err_restarts.clear(); } g_err_restarts.push_back(r); }
    if (first < 0) return RMX_OK;
    HIPCHK(hipMemsetAsync(b->d.err + r0, 0, sizeof(uint32_t) * (size_t)(r1 - r0), b->stream));
    return translate_error(b, first, e[first]);
}
// the same for a request list whose error words came back with the results (words[i] belongs to restart_of(i))
template <typename F> static int report_request_errors(rmx_batch *b, int n, const uint32_t *words, F restart_of) {
    int first = -1;
    for (int i = 0; i < n; i++) {
        if (!words[i]) continue;
        const int r = restart_of(i);
        if (first < 0) { first = i; g_err_restarts.clear(); }
        if (std::find(g_err_restarts.begin(), g_err_restarts.end(), r) == g_err_restarts.end()) {
            g_err_restarts.push_back(r);
            HIPCHK(hipMemsetAsync(b->d.err + r, 0, sizeof(uint32_t), b->stream));
        }
    }
    if (first < 0) return RMX_OK;
    return translate_error(b, restart_of(first), words[first]);
}
static int translate_error(rmx_batch *b, int r, uint32_t v) {
    char buf[160];
    if (v & RMX_ERR_NAN_LL) { snprintf(buf, sizeof buf, "ll is nan (restart %d)", r); return fail_flagged(RMX_EVALUE, buf); }
    if (v & RMX_ERR_TOTAL_DEPTH) { snprintf(buf, sizeof buf, "total_depth <= 0 (restart %d)", r); return fail_flagged(RMX_EVALUE, buf); }
    if (v & RMX_ERR_LOH_P) { snprintf(buf, sizeof buf, "expected p 0 or 1 for loh state (restart %d)", r); return fail_flagged(RMX_EVALUE, buf); }
    if (v & RMX_ERR_BAD_P) { snprintf(buf, sizeof buf, "p <= 0 or (1 - p) <= 0. (restart %d)", r); return fail_flagged(RMX_EVALUE, buf); }
    if (v & RMX_ERR_DIGAMMA) { snprintf(buf, sizeof buf, "x <= 0.0 in digamma (restart %d)", r); return fail_flagged(RMX_EVALUE, buf); }
    if (v & RMX_ERR_NAN_GRAD) { snprintf(buf, sizeof buf, "partial derivative is nan (restart %d)", r); return fail_flagged(RMX_EVALUE, buf); }
    if (v & RMX_ERR_WAIT) { snprintf(buf, sizeof buf, "a one-launch parameter search (search_mode 7) gave up waiting for a block that was not resident (restart %d)", r); return fail_flagged(RMX_EDEVICE, buf); }
    if (v & RMX_ERR_NAN_F) { snprintf(buf, sizeof buf, "nan in framelogprob (restart %d)", r); return fail_flagged(RMX_EASSERT, buf); }
    if (v & RMX_ERR_NAN_AB) { snprintf(buf, sizeof buf, "nan in alphas/betas (restart %d)", r); return fail_flagged(RMX_EASSERT, buf); }
    if (v & RMX_ERR_NAN_POST) { snprintf(buf, sizeof buf, "nan in posterior marginals (restart %d)", r); return fail_flagged(RMX_EASSERT, buf); }
    snprintf(buf, sizeof buf, "device error bits 0x%x (restart %d)", v, r); return fail_flagged(RMX_EVALUE, buf);
}

// Completion + error check for ONE restart that does not wait for other restarts' queued work:
// the error word travels with the results and the wait is on the restart's own event.
static int finish_restart(rmx_batch *b, int r, uint32_t *err_host) {
    HIPCHK(hipMemcpyAsync(err_host, b->d.err + r, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipEventRecord(b->done_ev[r], b->stream));
    return RMX_OK;
}
// ---- transition tables ---------------------------------------------------------
static inline double g_host(int model, int64_t dd) { return model == 0 ? (double)(dd < 0 ? -dd : dd) : (dd == 0 ? 0. : 1.); }

static int build_transitions(rmx_batch *b) {
    const int S = b->d.S, M = b->d.M, TC = (int)b->tc_pairs.size(), model = b->d.tmodel;
    const double pen = b->d.pen;
    const size_t SS = (size_t)S * S;
    std::vector<double> Tval(SS * TC), Wf(SS * TC), Wb(SS * TC);
    std::vector<int8_t> af(SS * TC), ab(SS * TC);
    b->Tsum.assign(TC, 0.);
    for (int tc = 0; tc < TC; tc++) {
        const int64_t *A = b->cn_classes.data() + (size_t)b->tc_pairs[tc].first * S * M * 2;
        const int64_t *Bc = b->cn_classes.data() + (size_t)b->tc_pairs[tc].second * S * M * 2;
        double tsum = 0.;
        for (int i = 0; i < S; i++)
            for (int j = 0; j < S; j++) {
                // bpmodel.pyx:652-656 then :670-684, same accumulation order
                double T = 0.;
                for (int m = 0; m < M; m++) {
                    int64_t ti = A[(i * M + m) * 2] + A[(i * M + m) * 2 + 1], tj = Bc[(j * M + m) * 2] + Bc[(j * M + m) * 2 + 1];
                    T += -pen * g_host(model, ti - tj);
                }
                double ach[2];
                for (int flip = 0; flip < 2; flip++) {
                    ach[flip] = 0.;
                    for (int m = 0; m < M; m++) {
                        for (int a = 0; a < 2; a++) {
                            int oa = flip ? 1 - a : a;
                            ach[flip] += g_host(model, A[(i * M + m) * 2 + a] - Bc[(j * M + m) * 2 + oa]);
                        }
                        int64_t ti = A[(i * M + m) * 2] + A[(i * M + m) * 2 + 1], tj = Bc[(j * M + m) * 2] + Bc[(j * M + m) * 2 + 1];
                        ach[flip] -= g_host(model, ti - tj);
                    }
                }
                const double amin = std::min(ach[0], ach[1]);
                T += -pen * amin;
                Tval[tc * SS + (size_t)i * S + j] = T;
                Wf[tc * SS + (size_t)i * S + j] = std::exp(T);
                Wb[tc * SS + (size_t)j * S + i] = std::exp(T);
                af[tc * SS + (size_t)i * S + j] = (int8_t)amin;
                b->max_adist = std::max(b->max_adist, (int)amin);
                ab[tc * SS + (size_t)j * S + i] = (int8_t)amin;
                tsum += T;
            }
        b->Tsum[tc] = tsum;
    }
    // pair codes + column order for k_pairwise_be2 (M in {2, 3}; index < 1024, allele distance < 64)
    b->pcode_ok = false;
    if (TC > 0 && b->d.pcode && (M == 2 || M == 3) && b->max_adist < 64 && (M == 2 ? b->d.D : b->d.D * b->d.D) <= 1024) {
        const int C = b->d.C, D = b->d.D, off = b->d.cn_max + 1, SPC = b->spc;
        auto tot_of = [&](int cls, int s_, int c) { const int64_t *t_ = b->cn_classes.data() + ((size_t)cls * S + s_) * M * 2 + (size_t)c * 2; return (int)(t_[0] + t_[1]); };
        std::vector<int32_t> jord((size_t)C * S), jmeta((size_t)C * S);
        for (int cls = 0; cls < C; cls++) {
            std::vector<int> ord(S);
            for (int j = 0; j < S; j++) ord[j] = j;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
                const int x1 = tot_of(cls, x, 1), y1 = tot_of(cls, y, 1);
                if (x1 != y1) return x1 < y1;
                return M == 3 && tot_of(cls, x, 2) < tot_of(cls, y, 2);
            });
            for (int jj = 0; jj < S; jj++) {
                const int j = ord[jj], t1 = tot_of(cls, j, 1), t2 = M == 3 ? tot_of(cls, j, 2) : 0;
                const bool last = jj + 1 == S;
                const int n1 = last ? -1 : tot_of(cls, ord[jj + 1], 1), n2 = last || M < 3 ? -1 : tot_of(cls, ord[jj + 1], 2);
                int fl = 0;
                if (last || n1 != t1 || (M == 3 && n2 != t2)) fl |= 1;
                if (last || n1 != t1) fl |= 2;
                jord[(size_t)cls * S + jj] = j;
                jmeta[(size_t)cls * S + jj] = t1 | (t2 << 8) | (fl << 16);
            }
        }
        const int S8 = (S + 7) & ~7;
        std::vector<uint16_t> pcode((size_t)TC * S8 * SPC, 0);
        for (int tc = 0; tc < TC; tc++) {
            const int ca = b->tc_pairs[tc].first, cb = b->tc_pairs[tc].second;
            for (int jj = 0; jj < S; jj++) {
                const int j = jord[(size_t)cb * S + jj];
                for (int i = 0; i < S; i++) {
                    int idx = 0;
                    for (int c = 1; c < M; c++) idx = idx * D + (tot_of(ca, i, c) - tot_of(cb, j, c) + off);
                    pcode[((size_t)tc * S8 + jj) * SPC + i] = (uint16_t)(idx | ((int)af[tc * SS + (size_t)i * S + j] << 10));
                }
            }
        }
        HIPCHK(hipMemcpy((void *)b->d.pcode, pcode.data(), pcode.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((void *)b->d.jord, jord.data(), jord.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((void *)b->d.jmeta, jmeta.data(), jmeta.size() * 4, hipMemcpyHostToDevice));
        b->pcode_ok = true;
    }
    // k_fbk (state grids too large for register-resident weights): byte-packed allele copies / totals of the
    // tumour clones per class, exp(-pen*k) per transition class, and the check that the closed form
    // -pen * min(SAD(cn_q, cn_o), SAD(cn_q, swap(cn_o))) reproduces the tabulated log-weights exactly
    b->fbk_ok = false;
    b->fbk_kmax = 0;
    if (TC > 0 && M >= 2 && M <= 4 && model == 0 && b->d.cn_max <= 30 && b->d_cnpack) {
        const int C = b->d.C;
        std::vector<uint32_t> cnp((size_t)C * S), ttp((size_t)C * S), cnp2((size_t)C * S, 0u);      // (cnp2: the third tumour clone's alleles, four clones only)
        for (int cls = 0; cls < C; cls++)
            for (int s_ = 0; s_ < S; s_++) {
                const int64_t *t_ = b->cn_classes.data() + ((size_t)cls * S + s_) * M * 2;
                uint32_t cp = 0, tp = 0;
                uint32_t cp2 = 0;
                for (int c = 1; c < M; c++) {
                    if (c <= 2) { cp |= ((uint32_t)t_[c * 2] & 0xff) << (16 * (c - 1)); cp |= ((uint32_t)t_[c * 2 + 1] & 0xff) << (16 * (c - 1) + 8); }
                    else { cp2 |= ((uint32_t)t_[c * 2] & 0xff); cp2 |= ((uint32_t)t_[c * 2 + 1] & 0xff) << 8; }
                    tp |= ((uint32_t)(t_[c * 2] + t_[c * 2 + 1]) & 0xff) << (8 * (c - 1));
                }
                cnp[(size_t)cls * S + s_] = cp; ttp[(size_t)cls * S + s_] = tp; cnp2[(size_t)cls * S + s_] = cp2;
            }
        auto sad = [](uint32_t x, uint32_t y) { int r_ = 0; for (int i = 0; i < 4; i++) r_ += std::abs((int)((x >> (8 * i)) & 0xff) - (int)((y >> (8 * i)) & 0xff)); return r_; };
        auto swp = [](uint32_t x) { return ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu); };
        bool ok = true;
        std::vector<double> wk((size_t)TC * FBK_WKN, 0.);
        int kmax = 0;
        for (int tc = 0; tc < TC && ok; tc++) {
            for (int k = 0; k < FBK_WKN; k++) wk[(size_t)tc * FBK_WKN + k] = std::exp(-pen * (double)k);
            const int ca = b->tc_pairs[tc].first, cb = b->tc_pairs[tc].second;
            if (ca != cb) continue;      // only chains of one class use the kernel
            for (int i = 0; i < S && ok; i++)
                for (int j = 0; j < S; j++) {
                    const int k = std::min(sad(cnp[(size_t)ca * S + i], cnp[(size_t)cb * S + j]) + sad(cnp2[(size_t)ca * S + i], cnp2[(size_t)cb * S + j]),
                                           sad(cnp[(size_t)ca * S + i], swp(cnp[(size_t)cb * S + j])) + sad(cnp2[(size_t)ca * S + i], swp(cnp2[(size_t)cb * S + j])));
                    const int kt = sad(ttp[(size_t)ca * S + i], ttp[(size_t)cb * S + j]);
                    if (k >= FBK_WKN || Tval[tc * SS + (size_t)i * S + j] != -pen * (double)k || (int)af[tc * SS + (size_t)i * S + j] != k - kt ||
                        Wf[tc * SS + (size_t)i * S + j] != wk[(size_t)tc * FBK_WKN + k]) { ok = false; break; }
                    kmax = std::max(kmax, k);
                }
        }
        if (ok) {
            HIPCHK(hipMemcpy(b->d_cnpack, cnp.data(), cnp.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(b->d_cnpack2, cnp2.data(), cnp2.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(b->d_totpack, ttp.data(), ttp.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(b->d_wk, wk.data(), wk.size() * 8, hipMemcpyHostToDevice));
            b->fbk_ok = true; b->fbk_kmax = kmax;
        }
    }
    // Viterbi lattice for grids beyond the register-resident kernel: codes of the distinct values of class 0
    b->vit_code_ok = false;
    if (TC > 0 && S <= 384) {      // (all grids since round 5: the trace-back of the maxima lattice looks its transition values up in the same table)
        std::map<double, int> ids;
        std::vector<uint8_t> code(SS);
        std::vector<double> vals(256, 0.);
        bool ok = true;
        // the multiple form first: every value is (-pen) * k for an integer k <= 254, bit for bit (integer multiples of the penalty: true for
        // both transition models whenever the reference's term-by-term accumulation is exact, e.g. the default penalty 10) -- then the code IS k,
        // and a consumer may form the value with one multiplication instead of a table lookup (k_backtrace_max)
        bool mul = pen > 0.;
        for (int i = 0; i < S && mul; i++)
            for (int j = 0; j < S; j++) {
                const double T = Tval[(size_t)i * S + j];
                const double kk = std::nearbyint(-T / pen);
                if (!(kk >= 0. && kk <= 254.) || kk * (-pen) != T) { mul = false; break; }
            }
        b->vit_mul_ok = mul;
        for (int i = 0; i < S && ok; i++)
            for (int j = 0; j < S; j++) {
                const double T = Tval[(size_t)i * S + j];
                int id;
                if (mul) { id = (int)std::nearbyint(-T / pen); vals[id] = T; }
                else {
                    auto it = ids.find(T);
                    if (it == ids.end()) { id = (int)ids.size(); if (id >= 255) { ok = false; break; } ids[T] = id; vals[id] = T; }
                    else id = it->second;
                }
                code[(size_t)j * S + i] = (uint8_t)id;       // transposed: row = target state o, column = source state i
            }
        if (ok) {
            vals[255] = -INFINITY;
            int rc2;
            if (!b->d_vit_code && ((rc2 = dalloc(b, &b->d_vit_code, SS)) || (rc2 = dalloc(b, &b->d_vit_val, 256)))) return rc2;
            HIPCHK(hipMemcpy(b->d_vit_code, code.data(), SS, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(b->d_vit_val, vals.data(), 256 * 8, hipMemcpyHostToDevice));
            b->vit_code_ok = true;
        }
    }
    if (TC > 0) {
        HIPCHK(hipMemcpy((void *)b->d.Tval, Tval.data(), Tval.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((void *)b->d.Wf, Wf.data(), Wf.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((void *)b->d.Wb, Wb.data(), Wb.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((void *)b->d.af, af.data(), af.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy((void *)b->d.ab, ab.data(), ab.size(), hipMemcpyHostToDevice));
        // the plain tables of this model, kept for the snapshots (log_transmat / cached_log_transmat) that outlive a model change
        int rc2;
        if (!b->Tval_m[model] && ((rc2 = dalloc(b, &b->Tval_m[model], Tval.size())) || (rc2 = dalloc(b, &b->af_m[model], af.size())))) return rc2;
        HIPCHK(hipMemcpy(b->Tval_m[model], Tval.data(), Tval.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->af_m[model], af.data(), af.size(), hipMemcpyHostToDevice));
    }
    return RMX_OK;
}
// the batch's device view with the plain tables of transition model `model` (snapshot consumers)
static Dev dev_for_model(const rmx_batch *b, int model) {
    Dev d2 = b->d;
    if (model != b->d.tmodel && b->Tval_m[model]) { d2.Tval = b->Tval_m[model]; d2.af = b->af_m[model]; d2.tmodel = model; }
    return d2;
}

static double plain_T_mean_sum(rmx_batch *b) {
    // sum over plain (non-telomere, non-breakend) adjacencies of mean(T): the transition
    // factor of the energy under the uniform initial joint (bpmodel.pyx:566-567, :1112-1115)
    double acc = 0.;
    const double ss = (double)((size_t)b->d.S * b->d.S);
    for (int n = 0; n + 1 < b->d.N; n++)
        if (b->tclass[n] >= 0 && b->brk_slot[n] < 0) acc += b->Tsum[b->tclass[n]] / ss;
    return acc;
}

// ---- FB configuration -----------------------------------------------------------
typedef void (*fb_kernel_t)(FbArgs);
static fb_kernel_t fb_kernel_for(int) { return k_fb<1024>; }
typedef void (*fbv_kernel_t)(FbvArgs);
static fbv_kernel_t fbv_kernel_for(int rpt, int nv, int blk) {
    (void)blk;
#define FBV_CASE(R_) \
    if (rpt == R_) { if (nv == 1) return k_fbv<R_, 1, 768>; if (nv == 2) return k_fbv<R_, 2, 768>; return k_fbv<R_, 4, 768>; }
    FBV_CASE(2) FBV_CASE(6) FBV_CASE(14) FBV_CASE(22)
#undef FBV_CASE
    return nullptr;
}
static const size_t kLdsBudget = 150 * 1024;   // of the CU's 160 KiB
static void fb_layout(rmx_batch *b, int rpt, int P, FbLaunch &L, size_t &lds, int *amat_lds) {
    const int S = b->d.S;
    L.P = P;
    L.NT = ((S * P + 63) / 64) * 64;
    int span = std::max(S, P * (rpt > 0 ? rpt : (S + P - 1) / P));
    L.SPAD = ((span + 7) / 8) * 8;
    const int mdp = (b->d.M * b->d.D + 1) & ~1;
    size_t fixed = (size_t)(2 * L.SPAD + (size_t)P * b->d.SP + 4 + mdp + 128) * 8 + (size_t)FB_NBUF * 2 * 64 * 4 +
                   (((size_t)b->d.C * S * b->d.M + 15) & ~(size_t)15) + 64;
    if (amat_lds) {
        *amat_lds = (rpt > 0 && fixed + (size_t)S * S + (size_t)FB_NBUF * 2 * b->d.SP * 8 <= kLdsBudget) ? 1 : 0;
        if (*amat_lds) fixed += (size_t)S * S;
    }
    int blk = 8;
    while (blk > 1 && fixed + (size_t)FB_NBUF * blk * b->d.SP * 8 > kLdsBudget) blk--;
    L.BLK = std::max(1, blk);
    lds = fixed + (size_t)FB_NBUF * L.BLK * b->d.SP * 8;
}
static void configure_fb(rmx_batch *b) {
    const int S = b->d.S;
    // general single-vector kernel: as many slices of the reduction index as fit 1024 threads
    int PG = 1;
    while (S * (PG + 1) <= 1024 && (S + PG) / (PG + 1) >= 1 && PG < 16) PG++;
    // multi-vector register kernel (chains of one state-table class): rows per slice for 8 slices
    b->fbv_rpt = 0;
    if (b->opt[RMX_OPT_FB_KERNEL] != 1) { for (int v : {2, 6, 14, 22}) if (8 * v >= S) { b->fbv_rpt = v; break; } }
    fb_layout(b, 0, PG, b->fbG, b->fbG_lds, nullptr);
    // more than 64 KiB of dynamic LDS needs an explicit opt-in per kernel
    hipFuncSetAttribute((const void *)fb_kernel_for(0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->fbG_lds);
}

// ---- staleness handling ----------------------------------------------------------
static void fill_logr(RestartParams &rp) { rp.logr[0] = std::log(rp.p[RMX_P_NEGBIN_R_0]); rp.logr[1] = std::log(rp.p[RMX_P_NEGBIN_R_1]); }

// State tables (cheap: C x S entries) are rebuilt whenever h / a likelihood parameter changed; the
// per-segment constant table ([8][N] lgamma differences) only when a full pass over all segments
// needs it (need_segc) -- the sampled M-step objective evaluates its own segments' constants.
static int ensure_tables(rmx_batch *b, int r0, int r1, bool need_segc = true) {
    {
        // parameters travel by value in the kernel arguments, up to 16 restarts per launch
        StageArgs sa;
        int n_ = 0;
        auto flush = [&]() {
            if (!n_) return;
            ProfScope ps(b, KID_STATE_TABLES);
            if (n_ == 1) hipLaunchKernelGGL(k_state_tables_one, dim3(b->d.C), dim3(256), 0, b->stream, b->d, (int)sa.rlist[0], sa.rp[0]);
            else hipLaunchKernelGGL(k_state_tables_many, dim3(b->d.C, n_), dim3(256), 0, b->stream, b->d, sa);
            n_ = 0;
        };
        for (int r = r0; r < r1; r++) {
            if (!b->tables_dirty[r]) continue;
            fill_logr(b->rp[r]);
            sa.rlist[n_] = r; sa.rp[n_] = b->rp[r];
            if (++n_ == 16) flush();
            b->tables_dirty[r] = 0; b->segc_dirty[r] = 1; b->ab_dirty[r] = 1;
        }
        flush();
    }
    if (need_segc) {
        int r = r0;
        while (r < r1) {
            // the per-segment constants are functions of the eight dispersion parameters only (seg_const_value): a table rebuild for a new h, the h
            // rounds and a rolled-back h leave them valid -- the flag says "maybe", the parameters they were last built from decide
            auto stale = [&](int q) {
                if (!b->segc_dirty[q]) return false;
                static const int ids[8] = {RMX_P_NEGBIN_R_0, RMX_P_NEGBIN_HDEL_R_0, RMX_P_NEGBIN_R_1, RMX_P_NEGBIN_HDEL_R_1, RMX_P_BETABIN_M_0, RMX_P_BETABIN_LOH_M_0, RMX_P_BETABIN_M_1, RMX_P_BETABIN_LOH_M_1};
                bool same = b->segc_built.size() == (size_t)b->R * 8;
                for (int k = 0; same && k < 8; k++) same = memcmp(&b->segc_built[(size_t)q * 8 + k], &b->rp[q].p[ids[k]], 8) == 0;
                if (same) { b->segc_dirty[q] = 0; return false; }
                if (b->segc_built.size() != (size_t)b->R * 8) b->segc_built.assign((size_t)b->R * 8, std::numeric_limits<double>::quiet_NaN());
                for (int k = 0; k < 8; k++) b->segc_built[(size_t)q * 8 + k] = b->rp[q].p[ids[k]];
                return true;
            };
            if (!stale(r)) { r++; continue; }
            int e = r + 1;
            while (e < r1 && stale(e)) e++;
            { ProfScope ps(b, KID_SEG_CONST); hipLaunchKernelGGL(k_seg_const, dim3((b->d.N + 255) / 256, e - r), dim3(256), 0, b->stream, b->d, r); }
            for (int i = r; i < e; i++) b->segc_dirty[i] = 0;
            r = e;
        }
    }
    HIPCHK(hipGetLastError());
    return RMX_OK;
}
// ---- strip kernels (S > 32, at most 4 states per lane) --------------------------------------------
typedef void (*cells_kernel_t)(Dev, int);
// cache: 0 none, 1 evaluate + store, 2 read
template <int NS> static cells_kernel_t cells_kernel_ns(int mode, int mask, int cache) {
    if (mode == 0) return cache == 2 ? k_cells<NS, 0, CM_ALL, 2> : (cache == 1 ? k_cells<NS, 0, CM_ALL, 1> : k_cells<NS, 0, CM_ALL, 0>);
    if (mode == 1) return cache == 2 ? k_cells<NS, 1, CM_ALL, 2> : k_cells<NS, 1, CM_ALL, 0>;
    if (mode == 3) return k_cells<NS, 3, CM_ALL, 2>;
    if (cache == 2) return k_cells<NS, 2, 31, 2>;
    if (cache == 1) {
        switch (mask) {
        case 1: return k_cells<NS, 2, 1, 1>; case 2: return k_cells<NS, 2, 2, 1>;
        case 3: return k_cells<NS, 2, 3, 1>; case 4: return k_cells<NS, 2, 4, 1>; case 8: return k_cells<NS, 2, 8, 1>;
        case 12: return k_cells<NS, 2, 12, 1>; case 15: return k_cells<NS, 2, 15, 1>; default: return k_cells<NS, 2, 31, 1>;
        }
    }
    switch (mask) {
    case 1: return k_cells<NS, 2, 1, 0>; case 2: return k_cells<NS, 2, 2, 0>;
    case 3: return k_cells<NS, 2, 3, 0>; case 4: return k_cells<NS, 2, 4, 0>; case 8: return k_cells<NS, 2, 8, 0>;
    case 12: return k_cells<NS, 2, 12, 0>; case 15: return k_cells<NS, 2, 15, 0>; default: return k_cells<NS, 2, 31, 0>;
    }
}
static bool use_strip(rmx_batch *b) { return b->d.S > 32 && b->d.S <= 384 && b->opt[RMX_OPT_STRIP]; }
static cells_kernel_t cells_kernel(rmx_batch *b, int mode, int mask, int cache) {
    const int ns = (b->d.S + 63) / 64;
    switch (ns) { case 1: return cells_kernel_ns<1>(mode, mask, cache); case 2: return cells_kernel_ns<2>(mode, mask, cache);
                  case 3: return cells_kernel_ns<3>(mode, mask, cache); case 4: return cells_kernel_ns<4>(mode, mask, cache);
                  case 5: return cells_kernel_ns<5>(mode, mask, cache); default: return cells_kernel_ns<6>(mode, mask, cache); }
}
// block size of the sampled-objective kernels: one lane per state, whole waves, at most 256
static dim3 ell_block(rmx_batch *b) { return dim3(std::min(256, ((b->d.S + 63) / 64) * 64)); }
// single-component sampled objectives can run from the lists of states with posterior mass when every
// listed restart's lists belong to its current posterior (both evaluation paths of rmx_param_search ask
// this same question, so they sum the same terms in the same order)
static bool ell_sparse_ok(rmx_batch *b, int n, const int32_t *restarts) {
    if (!b->d.sig_cnt || b->opt[RMX_OPT_ELL_DENSE]) return false;
    for (int i = 0; i < n; i++) if (!b->sig_valid[restarts[i]]) return false;
    return true;
}
static dim3 strip_grid(rmx_batch *b, int nr) { return dim3((b->d.N + 4 * STRIP_RPW - 1) / (4 * STRIP_RPW), nr); }
// smallest instantiated component mask covering `m`
static int cover_mask(int m) {
    if (m & 16) return 31;
    if ((m & 3) && (m & 12)) return 15;
    if ((m & 3) == 3) return 3;
    if (m & 1) return 1;
    if (m & 2) return 2;
    if ((m & 12) == 12) return 12;
    if (m & 4) return 4;
    if (m & 8) return 8;
    return 0;
}

static dim3 row_grid(rmx_batch *b, int nr) { int rows = 256 / b->G; return dim3((b->d.N + rows - 1) / rows, nr); }

static int ensure_ab(rmx_batch *b, int r0, int r1) {
    int rc = ensure_tables(b, r0, r1);
    if (rc) return rc;
    int r = r0;
    while (r < r1) {
        if (!b->ab_dirty[r]) { r++; continue; }
        // contiguous run of restarts whose stale components are covered by the same kernel; -1 = every
        // cell value is current in the cell cache, only the (A, B) reductions are redone from it
        auto variant = [&](int i) {
            if (!use_strip(b)) return 31;
            if (b->use_cache && b->cache_stale[i] == 0 && b->comp_dirty[i]) return -1;
            return cover_mask(b->comp_dirty[i]);
        };
        const int mask = variant(r);
        int e = r;
        while (e < r1 && b->ab_dirty[e] && variant(e) == mask) e++;
        if (use_strip(b)) {
            if (mask == -1) {
                ProfScope ps(b, KID_MARGINALS_AB);
                hipLaunchKernelGGL(cells_kernel(b, 2, 31, 2), strip_grid(b, e - r), dim3(256), 0, b->stream, b->d, r);
            } else if (mask) {
                ProfScope ps(b, KID_MARGINALS_AB);
                hipLaunchKernelGGL(cells_kernel(b, 2, mask, b->use_cache ? 1 : 0), strip_grid(b, e - r), dim3(256), 0, b->stream, b->d, r);
                if (b->use_cache) for (int i = r; i < e; i++) b->cache_stale[i] &= ~(mask & 15);
            }
        } else {
            ProfScope ps(b, KID_MARGINALS_AB); hipLaunchKernelGGL(k_marginals<false>, row_grid(b, e - r), dim3(256), 0, b->stream, b->d, r, b->G);
        }
        for (int i = r; i < e; i++) { b->ab_dirty[i] = 0; b->comp_dirty[i] = 0; b->comp_base[i] = 0; }
        r = e;
    }
    HIPCHK(hipGetLastError());
    return RMX_OK;
}

static int launch_pairwise_breakends(rmx_batch *b, int r0, int r1, int mode) {
    if (b->d.NBE == 0) return RMX_OK;
    ProfScope ps(b, KID_PAIRWISE);
    const Dev &d = b->d;
    const int nt = ((d.S + 63) / 64) * 64;
    const size_t lds2 = ((size_t)((d.S + 7) & ~7) + b->pe2p + 128 + (size_t)nt * (d.M - 1) * (d.cn_max + 2) + nt + 3 * d.M * d.D) * 8 + (size_t)((d.S + 7) & ~7) * 4 + (size_t)d.S * 4 + 64;
    const int S8p = (d.S + 7) & ~7;
    const int sp_rows = PSP_ROWS;
    const size_t lds_sp = (size_t)(4 * S8p + sp_rows * 2 * (d.cn_max + 2) + 2 * sp_rows + (d.M * d.D + 4) + 64) * 8 + (size_t)3 * S8p * 4 + (size_t)sp_rows * 4 + 64;
    // the sparse kernel (state pairs above the posterior threshold only) wherever the pair codes exist: auto, or option 3; 2 = the dense kernel
    // (auto: above ~200 states, where the dense kernel's S^2 pairs per adjacency outweigh the sparse kernel's per-block latency chain; at 165 states the
    // two are within 2 % of each other in the benchmark, the dense one ahead)
    const bool want_sp = b->opt[RMX_OPT_PAIRWISE_KERNEL] >= 3 || (b->opt[RMX_OPT_PAIRWISE_KERNEL] == 0 && d.S > 200);
    const int sp_threads = b->opt[RMX_OPT_PAIRWISE_KERNEL] >= 4 ? 64 : 256;
    if (mode == 0 && b->pcode_ok && want_sp && lds_sp + 512 <= 64 * 1024) {      // (+ the kernel's static __shared__: no opt-in above 64 KiB, the dense kernels take over)
        hipLaunchKernelGGL(k_pairwise_sp, dim3(d.NBE, r1 - r0), dim3(sp_threads), lds_sp, b->stream, b->d, r0, b->pe2p, b->spc);
    } else if (mode == 0 && b->pcode_ok && lds2 <= 150 * 1024 && (b->opt[RMX_OPT_PAIRWISE_KERNEL] == 0 || b->opt[RMX_OPT_PAIRWISE_KERNEL] == 2)) {
        HIPCHK(hipFuncSetAttribute((const void *)k_pairwise_be2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_pairwise_be2, dim3(d.NBE, r1 - r0), dim3(nt), lds2, b->stream, b->d, r0, b->pe2p, b->spc);
    } else {
        hipLaunchKernelGGL(k_pairwise, dim3(d.NBE, r1 - r0), dim3(256), 0, b->stream, b->d, r0, mode, (const int32_t *)nullptr, (double *)nullptr, PairAux{});
    }
    HIPCHK(hipGetLastError());
    return RMX_OK;
}
static int launch_brk_lut(rmx_batch *b, int r0, int r1, double *dst, double *edst) {
    if (b->d.NBE == 0) return RMX_OK;
    ProfScope ps(b, KID_BRK_LUT);
    hipLaunchKernelGGL(k_brk_lut, dim3(b->d.NBE, r1 - r0), dim3(64), 0, b->stream, b->d, r0, dst, edst, edst ? b->d.pe2_lt : (double *)nullptr, b->pe2p);
    HIPCHK(hipGetLastError());
    return RMX_OK;
}

#define RANGE_CHECK() if (!b || r0 < 0 || r1 > b->R || r0 >= r1) return fail(RMX_EARG, "bad restart range")
// HIP's current device is per host thread and starts at 0: every entry point binds the calling thread to the
// batch's device first (restart groups, M-step helpers and result collection call from their own threads;
// allocations and events made there must belong to the batch's GPU, not to GPU 0)
static inline void bind_device(const rmx_batch *b);
#define BIND(b) do { if (b) bind_device(b); } while (0)

static inline void bind_device(const rmx_batch *b) {
    (void)hipSetDevice(b->device);      // thread-local and cheap; not cached: other libraries may change the thread's device
}
// ===========================================================================
extern "C" {

const char *rmx_last_error(void) { return g_err.c_str(); }
int rmx_last_error_restarts(int32_t *out, int32_t cap) {
    const int n = (int)g_err_restarts.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = g_err_restarts[i];
    return n;
}

static bool option_value_ok(int id, int v) {
    switch (id) {
    case RMX_OPT_FB_KERNEL: return v >= 0 && v <= 3;
    case RMX_OPT_FB_NV: return v == 0 || v == 1 || v == 2 || v == 4;      // the workgroup shapes that exist (k_fbm and k_fbv / k_fbk 1 / 2 / 4, k_fbq 4)
    case RMX_OPT_SEARCH_MODE: return v >= 0 && v <= 7;
    case RMX_OPT_PAIRWISE_KERNEL: return v >= 0 && v <= 4;
    case RMX_OPT_FB_WG_BUDGET: return v >= 0 && v <= 4096;
    case RMX_OPT_GRAD_KERNEL: return v >= 0 && v <= 2;
    case RMX_OPT_VITERBI_CLUSTER: return v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 102 || v == 104 || v == 108;
    case RMX_OPT_TRACEBACK: return v == 0 || v == 1;
    case RMX_OPT_CU_PARTITION: return v == 0 || (((v >> 4) == 2 || (v >> 4) == 4 || (v >> 4) == 8) && (v & 15) < (v >> 4));
    case RMX_OPT_VITERBI_PLAIN: return v >= 0 && v <= 2;
    default: return v == 0 || v == 1;
    }
}
int rmx_set_default_option(int32_t id, int32_t value) {
    if (id < 0 || id >= RMX_OPT_COUNT || !option_value_ok(id, value)) return fail(RMX_EARG, "bad option id / value");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    g_opt_default[id] = value;
    return RMX_OK;
}
static void configure_fb(rmx_batch *b);
int rmx_set_option(rmx_batch *b, int32_t id, int32_t value) {
    if (!b || id < 0 || id >= RMX_OPT_COUNT || !option_value_ok(id, value)) return fail(RMX_EARG, "bad option id / value");
    if (id == RMX_OPT_CELL_CACHE || id == RMX_OPT_SPARSE_TRIAL || id == RMX_OPT_FB_DEBUG || id == RMX_OPT_STREAM_POOL || id == RMX_OPT_CU_PARTITION) return fail(RMX_EARG, "creation-time option: use rmx_set_default_option before rmx_batch_create");
    BIND(b);      // configure_fb sets function attributes (the > 64 KiB LDS opt-in) on the calling thread's current device
    b->opt[id] = value;
    if (id == RMX_OPT_FB_KERNEL) configure_fb(b);
    return RMX_OK;
}
int rmx_get_option(rmx_batch *b, int32_t id, int32_t *value) {
    if (!b || !value || id < 0 || id >= RMX_OPT_COUNT) return fail(RMX_EARG, "bad option id");
    BIND(b);
    *value = b->opt[id];
    return RMX_OK;
}

int rmx_compress_cn_states(const int64_t *cn_states, int32_t N, int32_t S, int32_t M, int32_t max_classes,
                           int32_t *seg_class_out, int64_t *classes_out, int32_t *num_classes) {
    const int rc = rmxh::compress_cn_states(cn_states, N, S, M, max_classes, seg_class_out, classes_out, num_classes);
    if (rc == RMX_EARG) return fail(RMX_EARG, "null argument");
    if (rc == RMX_EUNSUPPORTED) return fail(RMX_EUNSUPPORTED, "too many distinct per-segment state tables");
    return rc;
}

int rmx_batch_create(const rmx_problem *pr, int32_t R, const double *h_init, const double *divw, int32_t device, rmx_batch **out) {
    if (!pr || !out || !h_init || !divw) return fail(RMX_EARG, "null argument");
    const int N = pr->num_segments, S = pr->num_cn_states, M = pr->num_clones, K = pr->num_breakpoints, B = pr->num_brk_states, C = pr->num_classes;
    if (M < 1 || M > RMX_MAX_CLONES) return fail(RMX_EUNSUPPORTED, "num_clones must be in [1, 4]");
    if (S < 1 || S > 1024) return fail(RMX_EUNSUPPORTED, "num_cn_states must be in [1, 1024]");
    if (N < 1 || R < 1 || C < 1 || B < 1) return fail(RMX_EARG, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RMX_EDEVICE, "no HIP device available (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RMX_EDEVICE, "bad device ordinal");
    HIPCHK(hipSetDevice(device));

    // reference validation (bpmodel.pyx:528)
    int64_t maxidx = -1;
    for (int n = 0; n < N; n++) maxidx = std::max(maxidx, pr->breakpoint_idx[n]);
    if (maxidx + 1 != K) return fail(RMX_EVALUE, "breakpoint_idx must have maximum of num_breakpoints positive indices");
    for (int n = 0; n < N; n++) if (pr->seg_class[n] < 0 || pr->seg_class[n] >= C) return fail(RMX_EARG, "seg_class out of range");

    rmx_batch *b = new rmx_batch();
    { std::lock_guard<std::mutex> lk(g_opt_mu); memcpy(b->opt, g_opt_default, sizeof b->opt); }
    b->device = device; b->R = R;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) b->num_cus = cus; }
    if (b->opt[RMX_OPT_STREAM_POOL]) { if (pool_acquire(device, 0, &b->stream, b->opt[RMX_OPT_CU_PARTITION]) != RMX_OK) { delete b; return fail(RMX_EDEVICE, "hipStreamCreate failed"); } b->pooled_stream = true; }
    else HIPCHK(create_stream_for(device, b->opt[RMX_OPT_CU_PARTITION], &b->stream));
    b->own_stream = true;
    Dev &d = b->d;
    d.N = N; d.S = S; d.SP = ((S + 7) / 8) * 8; d.M = M; d.K = K; d.B = B; d.C = C; d.nc = pr->normal_contamination ? 1 : 0;
    d.tmodel = 0; d.R = R; d.pen = std::fabs(pr->transition_penalty);
    b->cn_classes.assign(pr->cn_classes, pr->cn_classes + (size_t)C * S * M * 2);
    b->seg_class.assign(pr->seg_class, pr->seg_class + N);
    b->brk_states.assign(pr->brk_states, pr->brk_states + (size_t)B * M);
    b->is_telomere.assign(pr->is_telomere, pr->is_telomere + N);
    int64_t cnmax = 0;
    for (auto v : b->cn_classes) { if (v < 0 || v > 100) { delete b; return fail(RMX_EUNSUPPORTED, "copy number out of range [0,100]"); } cnmax = std::max(cnmax, v); }
    for (auto v : b->brk_states) { if (v < 0 || v > 100) { delete b; return fail(RMX_EUNSUPPORTED, "breakpoint copy number out of range"); } cnmax = std::max(cnmax, v); }
    d.cn_max = (int)cnmax; d.D = 2 * d.cn_max + 3;
    if (d.D > 64) { delete b; return fail(RMX_EUNSUPPORTED, "cn_max > 30"); }

    // derived state tables
    std::vector<int8_t> cn8((size_t)C * S * M * 2), tot8((size_t)C * S * M);
    std::vector<uint8_t> sflags((size_t)C * S);
    for (int c = 0; c < C; c++)
        for (int s = 0; s < S; s++) {
            const int64_t *t = b->cn_classes.data() + ((size_t)c * S + s) * M * 2;
            bool hdel = true; int nsub = 0; bool loh = false;
            for (int m = 0; m < M; m++) {
                cn8[(((size_t)c * S + s) * M + m) * 2] = (int8_t)t[m * 2];
                cn8[(((size_t)c * S + s) * M + m) * 2 + 1] = (int8_t)t[m * 2 + 1];
                int64_t tt = t[m * 2] + t[m * 2 + 1];
                if (tt > 127) { delete b; return fail(RMX_EUNSUPPORTED, "total copy number > 127"); }
                tot8[((size_t)c * S + s) * M + m] = (int8_t)tt;
                if (t[m * 2] != 0 || t[m * 2 + 1] != 0) hdel = false;
            }
            for (int a = 0; a < 2; a++) {
                int64_t sum = 0;
                for (int m = 0; m < M; m++) sum += t[m * 2 + a];
                if (sum == 0) loh = true;
                if (M > 1) {
                    int64_t lo = t[2 + a], hi = lo;
                    for (int m = 2; m < M; m++) { lo = std::min(lo, t[m * 2 + a]); hi = std::max(hi, t[m * 2 + a]); }
                    if (hi != lo) nsub++;
                }
            }
            sflags[(size_t)c * S + s] = (uint8_t)((hdel ? 1 : 0) | (loh ? 2 : 0) | (nsub << 2));
            // guard: |tot_i - tot_j| must index the distance tables
            for (int m = 0; m < M; m++) if (tot8[((size_t)c * S + s) * M + m] > d.cn_max + 1) { delete b; return fail(RMX_EUNSUPPORTED, "total copy number exceeds cn_max + 1"); }
        }

    // topology: chains, transition classes, breakend slots
    b->brk_idx.resize(N); b->brk_orient.resize(N); b->tclass.assign(N, -1); b->brk_slot.assign(N, -1);
    std::vector<int32_t> cstart, cend; std::vector<uint8_t> cendflag(N, 0);
    std::map<std::pair<int, int>, int> tcmap;
    int start = 0;
    for (int n = 0; n < N; n++) {
        b->brk_idx[n] = (int32_t)pr->breakpoint_idx[n]; b->brk_orient[n] = (int32_t)pr->breakpoint_orient[n];
        const bool last = (n == N - 1), tel = pr->is_telomere[n] > 0;
        if (!last && !tel) {
            auto key = std::make_pair(b->seg_class[n], b->seg_class[n + 1]);
            auto it = tcmap.find(key);
            if (it == tcmap.end()) { int id = (int)b->tc_pairs.size(); tcmap[key] = id; b->tc_pairs.push_back(key); b->tclass[n] = id; }
            else b->tclass[n] = it->second;
        }
        if (!last && b->brk_idx[n] >= 0) { b->brk_slot[n] = (int32_t)b->be_n.size(); b->be_n.push_back(n); }
        if (last || tel) { cstart.push_back(start); cend.push_back(n); cendflag[n] = 1; start = n + 1; }
    }
    d.NC = (int)cstart.size(); d.NBE = (int)b->be_n.size(); d.TC = (int)b->tc_pairs.size();
    std::vector<int32_t> chain_tc(d.NC, 0), list_fast, list_gen, list_all(d.NC), be_cls(2 * (size_t)d.NBE);
    std::vector<int32_t> chain_cls(d.NC, 0);
    for (int c = 0; c < d.NC; c++) {
        list_all[c] = c;
        bool uniform = true;
        for (int n = cstart[c]; n <= cend[c]; n++) if (b->seg_class[n] != b->seg_class[cstart[c]]) uniform = false;
        chain_cls[c] = b->seg_class[cstart[c]];
        // transition class of (cls, cls); a single-segment chain never looks at it
        int tc0 = -1;
        if (uniform) {
            if (cend[c] > cstart[c]) tc0 = b->tclass[cstart[c]];
            else tc0 = d.TC > 0 ? 0 : -1;
        }
        chain_tc[c] = tc0;
        (tc0 >= 0 ? list_fast : list_gen).push_back(c);
    }
    // be_n ascends with the slot, so the breakend adjacencies of a chain are a slot interval
    std::vector<int32_t> chain_be(2 * (size_t)d.NC);
    for (int c = 0; c < d.NC; c++) {
        chain_be[2 * c] = (int32_t)(std::lower_bound(b->be_n.begin(), b->be_n.end(), cstart[c]) - b->be_n.begin());
        chain_be[2 * c + 1] = (int32_t)(std::lower_bound(b->be_n.begin(), b->be_n.end(), cend[c]) - b->be_n.begin());
    }
    b->be_cap = 4;
    for (int c = 0; c < d.NC; c++) b->be_cap = std::max(b->be_cap, (int)(chain_be[2 * c + 1] - chain_be[2 * c]) + 4);
    for (int s_ = 0; s_ < d.NBE; s_++) { be_cls[2 * s_] = b->seg_class[b->be_n[s_]]; be_cls[2 * s_ + 1] = b->seg_class[b->be_n[s_] + 1]; }
    b->n_fast = (int)list_fast.size(); b->n_generic = (int)list_gen.size();
    b->h_list_fast.assign(list_fast.begin(), list_fast.end());
    b->h_chain_len.resize(d.NC);
    for (int c = 0; c < d.NC; c++) b->h_chain_len[c] = cend[c] - cstart[c] + 1;
    if (d.TC > 4096) { delete b; return fail(RMX_EUNSUPPORTED, "too many transition classes"); }
    std::vector<int32_t> bk_ptr(K + 1, 0), bk_slots(d.NBE);
    for (int s = 0; s < d.NBE; s++) bk_ptr[b->brk_idx[b->be_n[s]] + 1]++;
    for (int k = 0; k < K; k++) bk_ptr[k + 1] += bk_ptr[k];
    { std::vector<int32_t> fill(bk_ptr.begin(), bk_ptr.end() - 1);
      for (int s = 0; s < d.NBE; s++) bk_slots[fill[b->brk_idx[b->be_n[s]]]++] = s; }

    int rc;
#define UP(field, vec) if ((rc = dupload(b, &d.field, vec))) { rmx_batch_destroy(b); return rc; }
    std::vector<double> lv(pr->l, pr->l + N), xv(pr->x, pr->x + N), yv(pr->y, pr->y + 2 * (size_t)N);
    std::vector<uint8_t> ones(N, 1);
    std::vector<double> loglv(N);
    for (int n = 0; n < N; n++) loglv[n] = std::log(lv[n]);
    std::vector<int32_t> brkst((size_t)B * M);
    for (size_t i = 0; i < brkst.size(); i++) brkst[i] = (int32_t)b->brk_states[i];
    UP(l, lv) UP(logl, loglv) UP(x, xv) UP(y, yv) UP(mask_t, ones) UP(mask_a, ones) UP(seg_class, b->seg_class) UP(tclass, b->tclass)
    UP(brk_slot, b->brk_slot) UP(brk_idx, b->brk_idx) UP(brk_orient, b->brk_orient) UP(be_n, b->be_n) UP(chain_start, cstart)
    UP(chain_end, cend) UP(chain_end_flag, cendflag) UP(cn, cn8) UP(tot, tot8) UP(sflags, sflags) UP(brk_states, brkst)
    UP(bk_ptr, bk_ptr) UP(bk_slots, bk_slots) UP(chain_tc, chain_tc) UP(chain_cls, chain_cls) UP(chain_list_fast, list_fast) UP(chain_list_generic, list_gen)
    UP(chain_list_all, list_all) UP(be_cls, be_cls) UP(chain_be, chain_be)
#undef UP
    const size_t SS = (size_t)S * S;
#define DA(field, type, count) { type *p_ = nullptr; if ((rc = dalloc(b, &p_, (size_t)(count)))) { rmx_batch_destroy(b); return rc; } d.field = p_; }
    DA(Tval, double, SS * d.TC) DA(Wf, double, SS * d.TC) DA(Wb, double, SS * d.TC) DA(af, int8_t, SS * d.TC) DA(ab, int8_t, SS * d.TC)
    const size_t RN = (size_t)R * N, RNS = RN * d.SP, RCS = (size_t)R * C * d.SP;
    DA(rp, RestartParams, R) DA(stLogD, double, RCS) DA(stD, double, RCS) DA(stP, double, RCS) DA(stM, double, RCS * 2) DA(stLg, double, RCS * 4) DA(stFlags, uint32_t, RCS) DA(stFlagsAgg, uint32_t, (size_t)R * C)
    DA(segc, double, RN * 8) DA(qt, double, RN * 2) DA(qa, double, RN * 2) DA(qs, double, RN * 2) DA(pbrk, double, (size_t)R * K * B)
    DA(f, double, RNS) DA(fe, double, RNS) DA(fe_alt, double, RNS) DA(fa, double, RNS) DA(fb, double, RNS) DA(post, double, RNS) DA(fmax, double, RN) DA(mrow, double, RN)
    DA(A, double, RN * 2) DA(Bv, double, RN * 4) DA(rowPF, double, RN) DA(rowPP, double, RN) DA(rowZ, double, RN)
    const size_t BEW = (size_t)R * d.NBE * M * d.D;
    DA(pd_lt, double, BEW) DA(pe_lt, double, (size_t)R * d.NBE * ((M * d.D + 1) & ~1) + 2)
    d.pe2_lt = nullptr; d.pe2x_lt = nullptr; b->pe2p = 0;
    d.pcode = nullptr; d.jord = nullptr; d.jmeta = nullptr;
    if (M >= 2 && M <= 3 && d.D <= 63 && d.NBE > 0) {
        int n2 = M == 2 ? d.D : d.D * d.D;
        b->pe2p = (n2 + 1) & ~1;
        DA(pe2_lt, double, (size_t)R * d.NBE * b->pe2p + 2)
        if (d.D <= 31) {      // (the matrix-core kernels' quad-interleaved copy and product codes: grids up to max_cn 14; above, k_fbk reads pe2_lt alone -- round 5)
            DA(pe2x_lt, double, (size_t)((R + 3) / 4) * d.NBE * b->pe2p * 4 + 2)
            HIPCHK(hipMemset(d.pe2x_lt, 0, ((size_t)((R + 3) / 4) * d.NBE * b->pe2p * 4 + 2) * 8));
            b->spc = ((S + 63) / 64) * 64;
            DA(pcode, uint16_t, (size_t)std::max(d.TC, 1) * ((S + 7) & ~7) * b->spc) DA(jord, int32_t, (size_t)C * S) DA(jmeta, int32_t, (size_t)C * S)
        }
    } else {
        if (M == 4 && d.D <= 21 && d.NBE > 0) {      // four clones: the clone-product table of a breakend has D^3 entries (max_cn 8: 6 859); k_fbk is its only reader (round 4)
            b->pe2p = (d.D * d.D * d.D + 1) & ~1;
            DA(pe2_lt, double, (size_t)R * d.NBE * b->pe2p + 2)
        }
    }
    if ((rc = dalloc(b, &b->d_A2, RN * 2)) || (rc = dalloc(b, &b->d_Bv2, RN * 4))) { rmx_batch_destroy(b); return rc; }
    if ((rc = dalloc(b, &b->d_cnpack, (size_t)C * S)) || (rc = dalloc(b, &b->d_cnpack2, (size_t)C * S)) || (rc = dalloc(b, &b->d_totpack, (size_t)C * S)) || (rc = dalloc(b, &b->d_wk, (size_t)std::max(d.TC, 1) * FBK_WKN))) { rmx_batch_destroy(b); return rc; }
    DA(pd_cached, double, BEW) DA(hist, double, BEW) DA(be_jt, double, (size_t)R * d.NBE) DA(be_ja, double, (size_t)R * d.NBE)
    DA(err, uint32_t, R)
#undef DA
    d.lc = nullptr; d.sig_idx = nullptr; d.sig_cnt = nullptr;
    if (S > 32 && S <= 384 && b->opt[RMX_OPT_SPARSE_TRIAL]) {
        uint16_t *pi_ = nullptr; uint8_t *pc_ = nullptr;
        if (dalloc(b, &pi_, RN * RMX_SIGK) == RMX_OK && dalloc(b, &pc_, RN) == RMX_OK) { d.sig_idx = pi_; d.sig_cnt = pc_; }
    }
    {
        const size_t bytes = RNS * 6 * 8;
        const bool want = b->opt[RMX_OPT_CELL_CACHE] != 0;
        if (want && S > 32 && S <= 384 && bytes <= ((size_t)96 << 30)) { double *p_ = nullptr; if (dalloc(b, &p_, RNS * 6) == RMX_OK) { d.lc = p_; b->use_cache = true; } }
    }
    if ((rc = dalloc(b, &b->d_lt_valid, R)) || (rc = dalloc(b, &b->d_partial, (size_t)R * ELBO_BLOCKS * 5)) || (rc = dalloc(b, &b->d_be_e, (size_t)R * std::max(d.NBE, 1))) || (rc = dalloc(b, &b->d_out4, (size_t)R * 5)) ||
        (rc = dalloc(b, &b->d_ell_partial, (size_t)R * std::max(N, ELBO_BLOCKS) * (1 + RMX_MAX_CLONES))) || (rc = dalloc(b, &b->d_ell_out, (size_t)R * 8)) ||
        (rc = dalloc(b, &b->d_sample, (size_t)R * N)) || (rc = dalloc(b, &b->d_grid_out, (size_t)R * 64 * (1 + RMX_MAX_CLONES))) ||
        (rc = dalloc(b, &b->d_done, std::max(R, 64))) || (rc = dalloc(b, &b->d_rlist, R)) || (rc = dalloc(b, &b->d_counts, R)) || (rc = dalloc(b, &b->d_rp_stage, R)) || (rc = dalloc(b, &b->d_batch_out, (size_t)R * 8))) { rmx_batch_destroy(b); return rc; }
    HIPCHK(hipMemset(b->d_done, 0, sizeof(unsigned) * (size_t)std::max(R, 64)));
    HIPCHK(hipHostMalloc((void **)&b->h_pinned, sizeof(double) * (size_t)(R + 1) * 64 * (1 + RMX_MAX_CLONES)));
    HIPCHK(hipHostMalloc((void **)&b->h_err, sizeof(uint32_t) * ((size_t)R * 4 + 64)));      // one word per request of a round (<= 4R in rmx_param_search_multi)
    HIPCHK(hipHostMalloc(&b->h_batch, (size_t)R * (sizeof(RestartParams) + 64) + 4096));
    HIPCHK(hipEventCreate(&b->tm_a)); HIPCHK(hipEventCreate(&b->tm_b));
    b->done_ev.resize(R);
    for (int r = 0; r < R; r++) HIPCHK(hipEventCreateWithFlags(&b->done_ev[r], hipEventDisableTiming));

    if ((rc = build_transitions(b))) { rmx_batch_destroy(b); return rc; }
    // envelope of the scaled linear-domain recursion: exp(T) must stay a normal double
    // with head-room for one step of products ("numerical envelope")
    {
        double tmin = 0.;
        // (breakend adjacencies are not bounded a priori; a vanishing row is reported as RMX_EASSERT)
        std::vector<double> tv((size_t)SS * d.TC);
        if (d.TC) HIPCHK(hipMemcpy(tv.data(), d.Tval, tv.size() * 8, hipMemcpyDeviceToHost));
        for (double v : tv) tmin = std::min(tmin, v);
        if (tmin < -650.) { rmx_batch_destroy(b); return fail(RMX_EUNSUPPORTED, "transition_log_prob * max copy-number change exceeds 650 nats: outside the linear-domain envelope"); }
    }

    // per-restart initial state (bpmodel.pyx:546-597)
    b->sample_cache.assign(R, std::vector<int64_t>()); b->sample_count.assign(R, -1);
    b->rp.resize(R); b->tables_dirty.assign(R, 1); b->segc_dirty.assign(R, 1); b->ab_dirty.assign(R, 1); b->comp_dirty.assign(R, 31); b->comp_base.assign(R, 31); b->cache_stale.assign(R, 15); b->sig_valid.assign(R, 0); b->sample_epoch.assign(R, 0); b->sig_epoch.assign(R, 0); b->lt_valid.assign(R, 0); b->lt_model.assign(R, 0); b->cached_model.assign(R, 0); b->logZ.assign(R, 0.); b->logz_dirty.assign(R, 0);
    for (int r = 0; r < R; r++) {
        RestartParams &p = b->rp[r];
        memset(&p, 0, sizeof p);
        for (int m = 0; m < M; m++) p.h[m] = h_init[(size_t)r * M + m];
        p.p[RMX_P_NEGBIN_R_0] = 500.; p.p[RMX_P_NEGBIN_R_1] = 10.; p.p[RMX_P_NEGBIN_HDEL_MU] = 1e-5; p.p[RMX_P_NEGBIN_HDEL_R_0] = 10.;
        p.p[RMX_P_NEGBIN_HDEL_R_1] = 1.; p.p[RMX_P_BETABIN_M_0] = 500.; p.p[RMX_P_BETABIN_M_1] = 10.; p.p[RMX_P_BETABIN_LOH_P] = 1e-3;
        p.p[RMX_P_BETABIN_LOH_M_0] = 10.; p.p[RMX_P_BETABIN_LOH_M_1] = 1.; p.p[RMX_P_PRIOR_OUTLIER_TOTAL] = 0.01; p.p[RMX_P_PRIOR_OUTLIER_ALLELE] = 0.01;
        p.p[RMX_P_DIVERGENCE_WEIGHT] = std::fabs(divw[r]);
    }
    {
        std::vector<double> q2(RN * 2), v;
        for (size_t i = 0; i < RN; i++) { q2[2 * i] = 1. - 0.01; q2[2 * i + 1] = 0.01; }
        HIPCHK(hipMemcpy(d.qt, q2.data(), q2.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.qa, q2.data(), q2.size() * 8, hipMemcpyHostToDevice));
        for (size_t i = 0; i < RN * 2; i++) q2[i] = 0.5;
        HIPCHK(hipMemcpy(d.qs, q2.data(), q2.size() * 8, hipMemcpyHostToDevice));
        // p_breakpoint: uniform over states with max <= 1 (bpmodel.pyx:547-554)
        std::vector<double> pb((size_t)K * B, 0.);
        if (K > 0) {
            double cnt = 0.;
            std::vector<double> row(B, 0.);
            for (int sb = 0; sb < B; sb++) { int64_t mx = 0; for (int m = 0; m < M; m++) mx = std::max(mx, b->brk_states[(size_t)sb * M + m]); if (mx <= 1) { row[sb] = 1.; cnt += 1.; } }
            for (int sb = 0; sb < B; sb++) row[sb] /= cnt;
            for (int k = 0; k < K; k++) std::copy(row.begin(), row.end(), pb.begin() + (size_t)k * B);
            for (int r = 0; r < R; r++) HIPCHK(hipMemcpy(d.pbrk + (size_t)r * K * B, pb.data(), pb.size() * 8, hipMemcpyHostToDevice));
        }
        // framelogprob = 1, posterior = 1/S, logZ rows = 0 (bpmodel.pyx:556-567)
        // (filled on the device: as pageable host copies these two planes were 1 GB over PCIe, half of the constructor's time at 50 000 x 165 x 8)
        hipLaunchKernelGGL(k_fill_f64, dim3(2048), dim3(256), 0, b->stream, d.f, RNS, 1.0);
        hipLaunchKernelGGL(k_fill_f64, dim3(2048), dim3(256), 0, b->stream, d.post, RNS, 1.0 / (double)S);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(b->stream));
        HIPCHK(hipMemset(d.rowZ, 0, RN * 8)); HIPCHK(hipMemset(d.fmax, 0, RN * 8)); HIPCHK(hipMemset(d.mrow, 0, RN * 8));
        HIPCHK(hipMemset(d.fa, 0, RNS * 8)); HIPCHK(hipMemset(d.fb, 0, RNS * 8)); HIPCHK(hipMemset(d.fe, 0, RNS * 8)); HIPCHK(hipMemset(d.fe_alt, 0, RNS * 8));
        HIPCHK(hipMemset(d.err, 0, R * 4)); HIPCHK(hipMemset(b->d_lt_valid, 0, R * 4));
        HIPCHK(hipMemset(d.hist, 0, BEW * 8)); HIPCHK(hipMemset(d.be_jt, 0, (size_t)R * d.NBE * 8)); HIPCHK(hipMemset(d.be_ja, 0, (size_t)R * d.NBE * 8));
        HIPCHK(hipMemset(d.pd_lt, 0, BEW * 8));
    }
    if (b->opt[RMX_OPT_FB_DEBUG]) { if ((rc = dalloc(b, &b->d_dbg, 32))) { rmx_batch_destroy(b); return rc; } HIPCHK(hipMemset(b->d_dbg, 0, 256)); }
    b->G = S > 32 ? 64 : (S > 16 ? 32 : (S > 8 ? 16 : 8));
    configure_fb(b);
    if (b->fbG_lds > 160 * 1024) { rmx_batch_destroy(b); return fail(RMX_EUNSUPPORTED, "LDS budget exceeded"); }
    // cached_log_transmat of the constructor (:604) + pairwise reductions of the uniform joint
    if ((rc = launch_brk_lut(b, 0, R, d.pd_cached, nullptr)) || (rc = launch_pairwise_breakends(b, 0, R, 1))) { rmx_batch_destroy(b); return rc; }
    b->plain_T_init.assign(R, plain_T_mean_sum(b));
    // plain adjacency list (only used when exact energy / entropy parts are requested)
    for (int n = 0; n + 1 < N; n++) if (b->tclass[n] >= 0 && b->brk_slot[n] < 0) b->plain_list.push_back(n);
    HIPCHK(hipStreamSynchronize(b->stream));
    *out = b;
    return RMX_OK;
}

int rmx_batch_destroy(rmx_batch *b) { BIND(b);
    if (!b) return RMX_OK;
    hipSetDevice(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    prof_collect(b);
    for (void *p : b->allocs) hipFree(p);
    if (b->h_pinned) hipHostFree(b->h_pinned);
    if (b->h_lists) hipHostFree(b->h_lists);
    if (b->ev_lists) hipEventDestroy(b->ev_lists);
    if (b->h_err) hipHostFree(b->h_err);
    if (b->h_batch) hipHostFree(b->h_batch);
    if (b->gf_nblk_host) hipHostFree(b->gf_nblk_host);
    if (b->h_elbo) { hipHostFree(b->h_elbo); hipEventDestroy(b->ev_elbo); }
    for (auto e : b->ev_pool) hipEventDestroy(e);
    if (b->tm_a) hipEventDestroy(b->tm_a);
    if (b->tm_b) hipEventDestroy(b->tm_b);
    for (auto e : b->done_ev) hipEventDestroy(e);
    if (b->stream2) { hipStreamSynchronize(b->stream2); if (b->pooled_stream2) pool_release(b->device, 1, b->stream2, b->opt[RMX_OPT_CU_PARTITION]); else hipStreamDestroy(b->stream2); hipEventDestroy(b->ev_fb); hipEventDestroy(b->ev_brk); }
    if (b->ev_copy) hipEventDestroy(b->ev_copy);
    if (b->ev_pace) hipEventDestroy(b->ev_pace);
    if (b->h_ind) { hipHostUnregister(b->h_ind); free(b->h_ind); }
    if (b->own_stream && b->stream) { if (b->pooled_stream) pool_release(b->device, 0, b->stream, b->opt[RMX_OPT_CU_PARTITION]); else hipStreamDestroy(b->stream); }
    delete b;
    return RMX_OK;
}

int rmx_set_stream(rmx_batch *b, void *s) { BIND(b);
    if (!b) return fail(RMX_EARG, "null batch");
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->own_stream) { if (b->pooled_stream) pool_release(b->device, 0, b->stream, b->opt[RMX_OPT_CU_PARTITION]); else hipStreamDestroy(b->stream); b->own_stream = false; b->pooled_stream = false; }
    if (s) b->stream = (hipStream_t)s;
    else { HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)); b->own_stream = true; }
    return RMX_OK;
}
int rmx_synchronize(rmx_batch *b) { BIND(b); HIPCHK(hipStreamSynchronize(b->stream)); return RMX_OK; }

int rmx_info(rmx_batch *b, int32_t what, int64_t *out) { BIND(b);
    switch (what) {
    case 0: *out = b->d.cn_max; break; case 1: *out = b->d.NC; break; case 2: *out = b->d.TC; break; case 3: *out = b->d.NBE; break;
    case 4: *out = b->d.SP; break; case 5: *out = b->fbv_rpt; break; case 6: *out = b->fbG.P; break; case 7: *out = b->fbG.NT; break;
    case 8: *out = b->fbG.BLK; break; case 9: *out = (int64_t)b->fbG_lds; break; case 10: *out = b->n_fast; break; case 11: *out = b->n_generic; break;
    case 60: *out = b->t_launch_ns; break; case 61: *out = b->t_wait_ns; break; case 62: *out = b->t_post_ns; break; case 63: *out = b->n_rounds; break;
    case 12: *out = b->last_fb_kernel; break; case 13: *out = b->last_fb_nv; break; case 14: *out = b->last_viterbi; break; case 15: *out = b->last_fb_nv_max; break;
    case 18: *out = b->last_viterbi_wgs; break; case 19: *out = b->last_traceback; break;
    case 54: *out = b->cluster_timeouts; break;      // decodes repeated with one workgroup per restart after a lattice cluster's watchdog ran out
    case 52: *out = b->last_search_blocks; break; case 53: *out = b->last_search_persist; break;      // blocks of the last device-driven search; 1: one launch (k_search_persist)
    case 16: { StreamPool &p_ = g_stream_pool[b->device & 15]; std::lock_guard<std::mutex> lk(p_.mu); *out = p_.created[0] + p_.created[1]; break; }      // streams the device's pool has created so far
    case 17: { StreamPool &p_ = g_stream_pool[b->device & 15]; std::lock_guard<std::mutex> lk(p_.mu); { size_t n_ = 0; for (auto &kv : p_.idle) n_ += kv.second.size(); *out = (int64_t)n_; } break; }   // ... of them idle
    case 20: case 21: case 22: case 23: case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39:
    case 40: case 41: case 42: case 43: case 44: case 45: case 46: case 47: case 48: case 49: case 50: case 51:
        { if (!b->d_dbg) { *out = 0; break; } unsigned long long v[32]; HIPCHK(hipStreamSynchronize(b->stream)); HIPCHK(hipMemcpy(v, b->d_dbg, 256, hipMemcpyDeviceToHost)); *out = (int64_t)v[what - 20]; break; }
    default: return fail(RMX_EARG, "bad info id");
    }
    return RMX_OK;
}

// ---- attributes -----------------------------------------------------------------
// components of (A, B, PF/PP) a likelihood parameter moves (CM_* bits, 16 = PF/PP), by parameter id
static const int kParamComponents[RMX_P_HMM_LOG_NORM_CONST] = {CM_LT0, CM_LT1, CM_LT0 | CM_LT1, CM_LT0, CM_LT1, CM_LA0, CM_LA1, CM_LA0 | CM_LA1, CM_LA0, CM_LA1, 0, 0, 16};
int rmx_set_param(rmx_batch *b, int32_t r, int32_t id, double v) { BIND(b);
    if (r < 0 || r >= b->R || id < 0 || id >= RMX_P_HMM_LOG_NORM_CONST) return fail(RMX_EARG, "bad restart / param id");
    if (id == RMX_P_DIVERGENCE_WEIGHT) v = std::fabs(v);
    b->rp[r].p[id] = v; b->tables_dirty[r] = 1; b->ab_dirty[r] = 1;
    b->comp_dirty[r] |= kParamComponents[id];
    b->cache_stale[r] |= (kParamComponents[id] & 15);
    return RMX_OK;
}
int rmx_get_param(rmx_batch *b, int32_t r, int32_t id, double *v) { BIND(b);
    if (r < 0 || r >= b->R || id < 0 || id >= RMX_P_COUNT) return fail(RMX_EARG, "bad restart / param id");
    if (id == RMX_P_HMM_LOG_NORM_CONST && b->logz_dirty[r]) {
        double *dst = b->d_ell_out + (size_t)r * 8;
        hipLaunchKernelGGL(k_logz, dim3(1), dim3(256), 0, b->stream, b->d, r, dst);
        HIPCHK(hipMemcpyAsync(&b->logZ[r], dst, 8, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        b->logz_dirty[r] = 0;
    }
    *v = (id == RMX_P_HMM_LOG_NORM_CONST) ? b->logZ[r] : b->rp[r].p[id];
    return RMX_OK;
}
int rmx_set_transition_model(rmx_batch *b, int32_t model) { BIND(b);
    if (model != 0 && model != 1) return fail(RMX_EUNSUPPORTED, "transition_model must be 0 or 1");
    if (model == b->d.tmodel) return RMX_OK;
    HIPCHK(hipStreamSynchronize(b->stream));
    b->d.tmodel = model;
    return build_transitions(b);
}

static int array_shape(rmx_batch *b, int id, size_t *count, bool *is_int) {
    const Dev &d = b->d; *is_int = false;
    switch (id) {
    case RMX_A_H: *count = d.M; break;
    case RMX_A_P_BREAKPOINT: *count = (size_t)d.K * d.B; break;
    case RMX_A_P_ALLELE_SWAP: case RMX_A_P_OUTLIER_TOTAL: case RMX_A_P_OUTLIER_ALLELE: *count = (size_t)d.N * 2; break;
    case RMX_A_POSTERIOR_MARGINALS: case RMX_A_FRAMELOGPROB: *count = (size_t)d.N * d.S; break;
    case RMX_A_TOTAL_LIKELIHOOD_MASK: case RMX_A_ALLELE_LIKELIHOOD_MASK: case RMX_A_STATE_SEQUENCE: *count = d.N; *is_int = true; break;
    case RMX_A_LOG_TRANSMAT: case RMX_A_CACHED_LOG_TRANSMAT: case RMX_A_JOINT_POSTERIOR_MARGINALS: *count = (size_t)(d.N - 1) * d.S * d.S; break;
    default: return fail(RMX_EARG, "bad array id");
    }
    return RMX_OK;
}

int rmx_set_array(rmx_batch *b, int32_t r, int32_t id, const void *src) { BIND(b);
    if (r < 0 || r >= b->R || !src) return fail(RMX_EARG, "bad restart / null source");
    Dev &d = b->d;
    const size_t RN = (size_t)r * d.N;
    switch (id) {
    case RMX_A_H:
        for (int m = 0; m < d.M; m++) b->rp[r].h[m] = ((const double *)src)[m];
        b->tables_dirty[r] = 1; b->ab_dirty[r] = 1; b->comp_dirty[r] = 31; b->comp_base[r] = 31; b->cache_stale[r] = 15; return RMX_OK;
    case RMX_A_P_BREAKPOINT:
        if (d.K) HIPCHK(hipMemcpyAsync(d.pbrk + (size_t)r * d.K * d.B, src, (size_t)d.K * d.B * 8, hipMemcpyHostToDevice, b->stream));
        break;
    case RMX_A_P_ALLELE_SWAP: HIPCHK(hipMemcpyAsync(d.qs + RN * 2, src, (size_t)d.N * 16, hipMemcpyHostToDevice, b->stream)); break;
    case RMX_A_P_OUTLIER_TOTAL: HIPCHK(hipMemcpyAsync(d.qt + RN * 2, src, (size_t)d.N * 16, hipMemcpyHostToDevice, b->stream)); break;
    case RMX_A_P_OUTLIER_ALLELE: HIPCHK(hipMemcpyAsync(d.qa + RN * 2, src, (size_t)d.N * 16, hipMemcpyHostToDevice, b->stream)); break;
    case RMX_A_POSTERIOR_MARGINALS:
        HIPCHK(hipMemcpy2DAsync(d.post + RN * d.SP, (size_t)d.SP * 8, src, (size_t)d.S * 8, (size_t)d.S * 8, d.N, hipMemcpyHostToDevice, b->stream));
        b->ab_dirty[r] = 1; b->comp_dirty[r] = 31; b->comp_base[r] = 31; b->sig_valid[r] = 0; b->sig_epoch[r]++; break;
    case RMX_A_TOTAL_LIKELIHOOD_MASK: case RMX_A_ALLELE_LIKELIHOOD_MASK: {
        std::vector<uint8_t> m8(d.N);
        for (int n = 0; n < d.N; n++) m8[n] = ((const int64_t *)src)[n] != 0;
        HIPCHK(hipStreamSynchronize(b->stream));
        HIPCHK(hipMemcpy((void *)(id == RMX_A_TOTAL_LIKELIHOOD_MASK ? d.mask_t : d.mask_a), m8.data(), d.N, hipMemcpyHostToDevice));
        for (int i = 0; i < b->R; i++) { b->ab_dirty[i] = 1; b->comp_dirty[i] = 31; b->comp_base[i] = 31; b->cache_stale[i] = 15; }
        return RMX_OK; }
    default: return fail(RMX_EARG, "array is read-only or unknown");
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}

int rmx_get_array(rmx_batch *b, int32_t r, int32_t id, void *dst) { BIND(b);
    if (r < 0 || r >= b->R || !dst) return fail(RMX_EARG, "bad restart / null destination");
    Dev &d = b->d;
    const size_t RN = (size_t)r * d.N;
    switch (id) {
    case RMX_A_H: for (int m = 0; m < d.M; m++) ((double *)dst)[m] = b->rp[r].h[m]; return RMX_OK;
    case RMX_A_P_BREAKPOINT: if (d.K) HIPCHK(hipMemcpyAsync(dst, d.pbrk + (size_t)r * d.K * d.B, (size_t)d.K * d.B * 8, hipMemcpyDeviceToHost, b->stream)); break;
    case RMX_A_P_ALLELE_SWAP: HIPCHK(hipMemcpyAsync(dst, d.qs + RN * 2, (size_t)d.N * 16, hipMemcpyDeviceToHost, b->stream)); break;
    case RMX_A_P_OUTLIER_TOTAL: HIPCHK(hipMemcpyAsync(dst, d.qt + RN * 2, (size_t)d.N * 16, hipMemcpyDeviceToHost, b->stream)); break;
    case RMX_A_P_OUTLIER_ALLELE: HIPCHK(hipMemcpyAsync(dst, d.qa + RN * 2, (size_t)d.N * 16, hipMemcpyDeviceToHost, b->stream)); break;
    case RMX_A_POSTERIOR_MARGINALS:
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)d.S * 8, d.post + RN * d.SP, (size_t)d.SP * 8, (size_t)d.S * 8, d.N, hipMemcpyDeviceToHost, b->stream)); break;
    case RMX_A_FRAMELOGPROB:
        HIPCHK(hipMemcpy2DAsync(dst, (size_t)d.S * 8, d.f + RN * d.SP, (size_t)d.SP * 8, (size_t)d.S * 8, d.N, hipMemcpyDeviceToHost, b->stream)); break;
    case RMX_A_TOTAL_LIKELIHOOD_MASK: case RMX_A_ALLELE_LIKELIHOOD_MASK: {
        std::vector<uint8_t> m8(d.N);
        HIPCHK(hipStreamSynchronize(b->stream));
        HIPCHK(hipMemcpy(m8.data(), id == RMX_A_TOTAL_LIKELIHOOD_MASK ? d.mask_t : d.mask_a, d.N, hipMemcpyDeviceToHost));
        for (int n = 0; n < d.N; n++) ((int64_t *)dst)[n] = m8[n];
        return RMX_OK; }
    case RMX_A_STATE_SEQUENCE:
        if ((int)b->last_path.size() != d.N) return fail(RMX_EARG, "no Viterbi path computed yet");
        memcpy(dst, b->last_path.data(), (size_t)d.N * 8); return RMX_OK;
    case RMX_A_LOG_TRANSMAT: case RMX_A_CACHED_LOG_TRANSMAT: case RMX_A_JOINT_POSTERIOR_MARGINALS: {
        if (d.N < 2) return RMX_OK;
        const size_t cnt = (size_t)(d.N - 1) * d.S * d.S;
        double *tmp = nullptr;
        HIPCHK(hipMalloc((void **)&tmp, cnt * 8));
        // each snapshot with the plain tables of the transition model it was taken under
        if (id == RMX_A_JOINT_POSTERIOR_MARGINALS)
            hipLaunchKernelGGL(k_materialize_joint, dim3(d.N - 1), dim3(256), 0, b->stream, dev_for_model(b, b->lt_model[r]), r, b->lt_valid[r] ? 0 : 1, tmp);
        else
            hipLaunchKernelGGL(k_materialize_T, dim3(d.N - 1), dim3(256), 0, b->stream,
                               dev_for_model(b, id == RMX_A_LOG_TRANSMAT ? b->lt_model[r] : b->cached_model[r]), r, id == RMX_A_LOG_TRANSMAT ? 0 : 1,
                               (id == RMX_A_LOG_TRANSMAT && !b->lt_valid[r]) ? 1 : 0, tmp);
        hipError_t e = hipMemcpyAsync(dst, tmp, cnt * 8, hipMemcpyDeviceToHost, b->stream);
        hipStreamSynchronize(b->stream);
        hipFree(tmp);
        if (e != hipSuccess) return fail(RMX_EDEVICE, hipGetErrorString(e));
        return RMX_OK; }
    default: return fail(RMX_EARG, "bad array id");
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    return RMX_OK;
}

// calculate_log_transmat(out) (bpmodel.pyx:639-684): the dense (N-1) x S x S log transition array for the
// CURRENT p_breakpoint of restart r, into the caller's host array.  Neither the log_transmat snapshot of
// the last update_p_cn nor cached_log_transmat is touched.  Size warning: 8 (N-1) S^2 bytes.
int rmx_calculate_log_transmat(rmx_batch *b, int32_t r, double *dst) { BIND(b);
    if (!b || r < 0 || r >= b->R || !dst) return fail(RMX_EARG, "bad argument");
    const Dev &d = b->d;
    if (d.N < 2) return RMX_OK;
    const size_t cnt = (size_t)(d.N - 1) * d.S * d.S;
    double *tmp = nullptr, *pd = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, cnt * 8));
    if (hipMalloc((void **)&pd, std::max<size_t>((size_t)b->R * d.NBE * d.M * d.D, 1) * 8) != hipSuccess) { hipFree(tmp); return fail(RMX_EDEVICE, "out of device memory"); }
    if (d.NBE > 0) hipLaunchKernelGGL(k_brk_lut, dim3(d.NBE, 1), dim3(64), 0, b->stream, b->d, r, pd, (double *)nullptr, (double *)nullptr, 0);
    Dev d2 = b->d;
    d2.pd_lt = pd;
    hipLaunchKernelGGL(k_materialize_T, dim3(d.N - 1), dim3(256), 0, b->stream, d2, r, 0, 0, tmp);
    hipError_t e = hipMemcpyAsync(dst, tmp, cnt * 8, hipMemcpyDeviceToHost, b->stream);
    hipStreamSynchronize(b->stream);
    hipFree(tmp); hipFree(pd);
    if (e != hipSuccess) return fail(RMX_EDEVICE, hipGetErrorString(e));
    return RMX_OK;
}

// Host-only helper of the M-step's weighted segment sampling (BreakpointModel._create_sample with a private
// RNG stream): for k uniform draws u, the index each one selects from the cumulative distribution of p --
// numpy's  cdf = cumsum(p); cdf /= cdf[-1]; minimum(cdf.searchsorted(u, side='right'), n - 1)  with the same
// sequential accumulation, hence the same indices.  *positive receives count_nonzero(p > 0).  No device work,
// no Python objects: host threads run it concurrently.
int rmx_weighted_search(const double *p, int64_t n, const double *u, int32_t k, int64_t *out, int64_t *positive) {
    if (rmxh::weighted_search(p, n, u, k, out, positive)) return fail(RMX_EARG, "bad argument");
    return RMX_OK;
}

int rmx_weighted_sample_round(const double *w, int64_t n, int64_t stride, double norm, const double *u, int32_t k,
                              int64_t *found, int32_t *nfound, int32_t cap, int64_t *positive) {
    if (rmxh::weighted_sample_round(w, n, stride, norm, u, k, found, nfound, cap, positive)) return fail(RMX_EARG, "bad argument");
    return RMX_OK;
}

// p_outlier_total / p_outlier_allele of restarts [r0, r1) -- the weights of the M-step's samples (cn_model.py:323-352) -- in ONE
// transfer beside the batch stream, behind everything queued on it so far, into pinned host memory owned by
// the batch: 16 per-restart rmx_get_array calls on the batch stream put their staged pageable copies between the rounds of the h
// M-step that runs meanwhile.  *total / *allele: [r1 - r0][N][2], valid until the next call.
int rmx_fetch_indicators(rmx_batch *b, int32_t r0, int32_t r1, const double **total, const double **allele) { BIND(b);
    RANGE_CHECK();
    if (!total || !allele) return fail(RMX_EARG, "null argument");
    const Dev &d = b->d;
    const size_t per = (size_t)d.N * 2;
    if (!b->h_ind) {
        // ordinary (CPU-cached) pages, registered for DMA: the host reads every weight several times (sums, cumulative sums); memory
        // from hipHostMalloc is mapped uncached on this platform and made those reads ~50x slower
        const size_t bytes = sizeof(double) * 2 * (size_t)b->R * per;
        void *p_ = nullptr;
        if (posix_memalign(&p_, 4096, (bytes + 4095) & ~(size_t)4095) != 0) return fail(RMX_EDEVICE, "out of host memory");
        memset(p_, 0, bytes);
        if (hipHostRegister(p_, (bytes + 4095) & ~(size_t)4095, hipHostRegisterDefault) != hipSuccess) { free(p_); return fail(RMX_EDEVICE, "hipHostRegister failed"); }
        b->h_ind = (double *)p_;
    }
    // The copy is queued on the batch stream itself.  A copy stream of its own, or the sweep's second stream, looks free during the
    // M-step but is not: with the default four hardware queues the extra stream shares a queue with the OTHER restart group's main
    // stream and the 0.3 ms copy waits behind that group's whole sweep phase (measured: 19 ms of waiting, or -- on the second
    // stream -- the next sweeps 10 ms longer).  In order on the batch stream it delays one round of the h M-step by its own length.
    hipStream_t cs = b->stream;
    double *ht = b->h_ind + (size_t)r0 * per, *ha = b->h_ind + ((size_t)b->R + r0) * per;
    if (!b->ev_copy) HIPCHK(hipEventCreateWithFlags(&b->ev_copy, hipEventDisableTiming));
    {
        std::lock_guard<std::mutex> lk(b->mu);      // (not between the launches of a round in flight on another thread)
        HIPCHK(hipMemcpyAsync(ht, d.qt + (size_t)r0 * per, sizeof(double) * (size_t)(r1 - r0) * per, hipMemcpyDeviceToHost, cs));
        HIPCHK(hipMemcpyAsync(ha, d.qa + (size_t)r0 * per, sizeof(double) * (size_t)(r1 - r0) * per, hipMemcpyDeviceToHost, cs));
        HIPCHK(hipEventRecord(b->ev_copy, cs));
    }
    HIPCHK(hipEventSynchronize(b->ev_copy));      // the copies only: rounds queued behind them meanwhile are not waited for
    *total = ht; *allele = ha;
    return RMX_OK;
}

int rmx_get_state_table(rmx_batch *b, int32_t which, int64_t *dst) { BIND(b);
    const Dev &d = b->d;
    const int S = d.S, M = d.M;
    for (int n = 0; n < d.N; n++) {
        const int64_t *t = b->cn_classes.data() + (size_t)b->seg_class[n] * S * M * 2;
        for (int s = 0; s < S; s++) {
            if (which == 0) { for (int m = 0; m < M; m++) dst[((size_t)n * S + s) * M + m] = t[(s * M + m) * 2] + t[(s * M + m) * 2 + 1]; continue; }
            bool hdel = true, loh = false; int nsub = 0;
            for (int m = 0; m < M; m++) if (t[(s * M + m) * 2] || t[(s * M + m) * 2 + 1]) hdel = false;
            for (int a = 0; a < 2; a++) {
                int64_t sum = 0; for (int m = 0; m < M; m++) sum += t[(s * M + m) * 2 + a];
                if (sum == 0) loh = true;
                if (M > 1) { int64_t lo = t[(s * M + 1) * 2 + a], hi = lo; for (int m = 2; m < M; m++) { lo = std::min(lo, t[(s * M + m) * 2 + a]); hi = std::max(hi, t[(s * M + m) * 2 + a]); } if (hi != lo) nsub++; }
            }
            dst[(size_t)n * S + s] = which == 1 ? nsub : (which == 2 ? (hdel ? 1 : 0) : (loh ? 1 : 0));
        }
    }
    return RMX_OK;
}

// ---- coordinate updates -------------------------------------------------------------
static int do_framelogprob(rmx_batch *b, int r0, int r1) {
    int rc = ensure_tables(b, r0, r1);
    if (rc) return rc;
    ProfScope ps(b, KID_FRAMELOGPROB);
    if (use_strip(b)) {
        // per run of restarts: read the cache where it is current, otherwise evaluate (and fill it)
        int r = r0;
        while (r < r1) {
            const bool cur = b->use_cache && b->cache_stale[r] == 0;
            int e = r;
            while (e < r1 && (b->use_cache && b->cache_stale[e] == 0) == cur) e++;
            hipLaunchKernelGGL(cells_kernel(b, 0, CM_ALL, cur ? 2 : (b->use_cache ? 1 : 0)), strip_grid(b, e - r), dim3(256), 0, b->stream, b->d, r);
            if (b->use_cache) for (int i = r; i < e; i++) b->cache_stale[i] = 0;
            r = e;
        }
    }
    else hipLaunchKernelGGL(k_framelogprob, row_grid(b, r1 - r0), dim3(256), 0, b->stream, b->d, r0, b->G);
    HIPCHK(hipGetLastError());
    return RMX_OK;
}
// skip_frame: the frame log-probabilities of this sweep were already written by the fused pass of the
// previous sweep; fuse_next: the marginal pass also does the outlier / allele-swap updates and the next
// sweep's frame pass (k_cells MODE 3)
// update_p_cn in three parts: (1) frame pass, transition snapshot, forward-backward; (2) pairwise reductions at
// the breakend adjacencies; (3) marginals.  (2) and (3) only read what (1) wrote and write disjoint arrays
// (a fused marginal pass puts the next sweep's fe into the second buffer), so rmx_variational_update runs
// (2) -- and update_p_breakpoint behind it -- on a second stream next to (3).
// workgroups of a forward-backward launch that run side by side: one per CU (the weights fill the register file) -- the device's CU count (256 on an MI355X)
static inline int fb_wg_budget(const rmx_batch *b) { return b->opt[RMX_OPT_FB_WG_BUDGET] > 0 ? b->opt[RMX_OPT_FB_WG_BUDGET] : b->num_cus; }
static void (*fbm_kernel_for(int KB))(FbmArgs) {
    switch (KB) { case 8: return k_fbm<8>; case 16: return k_fbm<16>; case 28: return k_fbm<28>; case 36: return k_fbm<36>; case 42: return k_fbm<42>; case 44: return k_fbm<44>; }
    return nullptr;
}
// The workgroups of a k_fbm launch over restarts [r0, r1): per chain a shape -- 4 restarts per workgroup on the matrix cores, 2 or 1 on the vector
// ALU -- and per (chain, unit of that many restarts, direction) one work item.  A launch is a bundle of independent chains of dependent steps and
// lasts as long as its slowest workgroup; a step costs about 1 500 / 2 110 / 2 845 cycles with 1 / 2 / 4 restarts per workgroup (fb_launch_shapes),
// and a workgroup fills a CU.  So: the smallest T for which every chain has a shape with (segments x step cost) <= T while all workgroups together fit
// the chip at once, and for every chain the LARGEST such shape (fewest CUs).  Equal chains get one shape (23 chains x 8 restarts: two per
// workgroup, 184 workgroups); chromosomes of a real genome (lengths 5 : 1) get one restart per workgroup on the long ones and four on the
// short ones.  `pin` (option fb_nv) fixes the shape of every chain.  Items are ordered longest first.
// (`cost`: cycles per step with 1 / 2 / 4 restarts per workgroup at [1] / [2] / [4]; 0 = the kernel has no such shape)
static int fb_items_for(rmx_batch *b, int r0, int r1, int pin, const double (&cost)[5], rmx_batch::FbItems **out) {
    if ((pin == 1 || pin == 2 || pin == 4) && cost[pin] == 0.) pin = 4;
    auto key = std::make_tuple(r0, r1, pin * 4096 + fb_wg_budget(b));
    auto it = b->fb_items.find(key);
    if (it != b->fb_items.end()) { *out = &it->second; return RMX_OK; }
    std::vector<int> shapes;
    for (int nv : {4, 2, 1}) if (cost[nv] > 0.) shapes.push_back(nv);
    const int nc = (int)b->h_list_fast.size();
    auto units = [&](int nv) { return (r1 - 1) / nv - r0 / nv + 1; };
    std::vector<int> nv_of(nc, 4);
    if (pin == 1 || pin == 2 || pin == 4) std::fill(nv_of.begin(), nv_of.end(), pin);
    else {
        std::vector<double> cand;
        for (int c = 0; c < nc; c++) for (int nv : shapes) cand.push_back(b->h_chain_len[b->h_list_fast[c]] * cost[nv]);
        std::sort(cand.begin(), cand.end());
        for (double T : cand) {
            long wg = 0; bool ok = true;
            for (int c = 0; c < nc && ok; c++) {
                int pick = 0;
                for (int nv : shapes) if (b->h_chain_len[b->h_list_fast[c]] * cost[nv] <= T) { pick = nv; break; }      // largest shape within T
                if (!pick) ok = false; else { nv_of[c] = pick; wg += 2L * units(pick); }
            }
            if (ok && wg <= fb_wg_budget(b)) break;
            std::fill(nv_of.begin(), nv_of.end(), 4);      // (no T fits the chip in one round: four per workgroup everywhere)
        }
    }
    std::vector<int> order(nc);
    for (int c = 0; c < nc; c++) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->h_chain_len[b->h_list_fast[x]] * cost[nv_of[x]] > b->h_chain_len[b->h_list_fast[y]] * cost[nv_of[y]]; });
    std::vector<int4> items;
    rmx_batch::FbItems f;
    for (int c : order) {
        const int nv = nv_of[c];
        f.nv_min = std::min(f.nv_min, nv); f.nv_max = std::max(f.nv_max, nv);
        for (int u = r0 / nv; u <= (r1 - 1) / nv; u++) for (int dir = 0; dir < 2; dir++) items.push_back(make_int4(b->h_list_fast[c], u * nv, nv, dir));
    }
    f.n = (int)items.size();
    int rc = dalloc(b, &f.dev, items.size());
    if (rc) return rc;
    HIPCHK(hipMemcpy(f.dev, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice));
    *out = &(b->fb_items[key] = f);
    return RMX_OK;
}
static int p_cn_front(rmx_batch *b, int r0, int r1, bool skip_frame, bool snapshot_done = false) {
    int rc = skip_frame ? ensure_tables(b, r0, r1) : do_framelogprob(b, r0, r1);
    if (rc) return rc;
    // log_transmat snapshot := T(current p_breakpoint)   (bpmodel.pyx:939); snapshot_done: the previous sweep of the
    // same call already built it behind its update_p_breakpoint
    if (!snapshot_done && (rc = launch_brk_lut(b, r0, r1, b->d.pd_lt, b->d.pe_lt))) return rc;
    {
        ProfScope ps(b, KID_FB);
        const Dev &d = b->d;
        FbArgs a;
        a.S = d.S; a.SP = d.SP; a.M = d.M; a.D = d.D; a.C = d.C; a.N = d.N; a.NBE = d.NBE; a.cn_max = d.cn_max;
        a.r0 = r0; a.pen = d.pen;
        a.chain_start = d.chain_start; a.chain_end = d.chain_end; a.tclass = d.tclass; a.brk_slot = d.brk_slot;
        a.chain_tc = d.chain_tc; a.chain_cls = d.chain_cls; a.be_cls = d.be_cls; a.amat_lds = 0; a.pad_ = 0;
        a.fe = d.fe; a.Wf = d.Wf; a.Wb = d.Wb; a.pe_lt = d.pe_lt; a.af = d.af; a.ab = d.ab; a.tot = d.tot;
        a.fa = d.fa; a.fb = d.fb; a.mrow = d.mrow; a.err = d.err; a.dbg = b->d_dbg;
        bool fast = false;
        bool done_fast = false;
        // k_fbm: four restarts per workgroup on the matrix cores (a 4x4x4 MFMA carries four vectors), or two / one on the vector ALU where
        // that many workgroups still fit the chip in one round -- a step of the vector form is shorter (fewer products per CU and step), and a
        // launch is a chain of dependent steps.  Needs the clone-product tables for breakend steps.  Option fb_nv = 1 / 2 / 4 pins the shape;
        // fb_kernel = 3 selects the two-phase vector kernel below instead.
        if (b->fbv_rpt > 0 && b->n_fast > 0 && d.M <= 3 && (d.NBE == 0 || (d.pe2x_lt != nullptr && b->max_adist < 64 && b->opt[RMX_OPT_FB_BREAKEND_CODES])) &&
            b->opt[RMX_OPT_FB_KERNEL] != 3) {
            int KB = 0;
            for (int v : {8, 16, 28, 36, 42, 44}) if (4 * v >= d.S) { KB = v; break; }
            const int NCT = (d.S + 14) / 15;                          // waves: 15 state columns + the ones column each
            rmx_batch::FbItems *items = nullptr;
            static const double fbm_cost[5] = {0., 1500., 2110., 0., 2845.};      // cycles per step (profiles/r04_fb_launch_shapes.txt)
            if ((rc = fb_items_for(b, r0, r1, b->opt[RMX_OPT_FB_NV], fbm_cost, &items))) return rc;
            FbmArgs m;
            memset(&m, 0, sizeof m);
            m.S = d.S; m.SP = d.SP; m.M = d.M; m.D = d.D; m.C = d.C; m.N = d.N; m.NBE = d.NBE; m.cn_max = d.cn_max; m.r0 = r0; m.r1 = r1;
            m.PE2P = d.NBE > 0 ? b->pe2p : 0; m.SPC = NCT * 16; m.VR = std::max(4 * KB, ((NCT * 15 + 7) / 8) * 8); m.pen = d.pen;
            m.pad_ = d.tmodel;
            m.chain_start = d.chain_start; m.chain_end = d.chain_end; m.chain_list = d.chain_list_fast; m.chain_tc = d.chain_tc; m.chain_cls = d.chain_cls;
            m.be_n = d.be_n; m.chain_be = d.chain_be; m.fe = d.fe; m.Wf = d.Wf; m.Wb = d.Wb; m.pe2_lt = d.pe2x_lt; m.af = d.af; m.ab = d.ab; m.tot = d.tot;
            m.fa = d.fa; m.fb = d.fb; m.mrow = d.mrow; m.err = d.err; m.dbg = b->d_dbg;
            const size_t lds = (size_t)2 * m.VR * 4 * 8 + (size_t)2 * 4 * m.PE2P * 8 + (size_t)6 * m.PE2P * 8 + 64 * 8 + (size_t)(KB / 2) * 4 * m.SPC * 4 + (size_t)b->be_cap * 4 + 64;
            m.items = items->dev;
            void (*kf)(FbmArgs) = fbm_kernel_for(KB);
            if (kf && NCT <= 12 && lds <= kLdsBudget) {
                HIPCHK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kf, dim3(items->n), dim3(64 * NCT), lds, b->stream, m);
                done_fast = true; fast = true; b->last_fb_kernel = 1; b->last_fb_nv = items->nv_min; b->last_fb_nv_max = items->nv_max;
            }
        }
        if (!done_fast && b->fbv_rpt > 0 && b->n_fast > 0) {
            // multi-vector kernel: NV restarts per workgroup, as few as keep the grid within one wave
            // of workgroups over the 256 CUs
            const int nr = r1 - r0;
            int NV = 1;
            while (NV < 4 && (long)b->n_fast * 2 * ((nr + NV - 1) / NV) > 256) NV *= 2;
            { const int want = b->opt[RMX_OPT_FB_NV]; if (want == 1 || want == 2 || want == 4) NV = want; }
            FbvArgs v;
            v.S = d.S; v.SP = d.SP; v.M = d.M; v.D = d.D; v.C = d.C; v.N = d.N; v.NBE = d.NBE; v.cn_max = d.cn_max;
            v.r0 = r0; v.r1 = r1; v.pen = d.pen; v.pad_ = 0;
            v.G2 = (((d.S + 1) / 2 + 15) / 16) * 16;   // column pairs per row slice, whole DPP rows
            v.SPW = ((d.S + 63) / 64) * 64;
            v.chain_start = d.chain_start; v.chain_end = d.chain_end; v.tclass = d.tclass; v.brk_slot = d.brk_slot;
            v.chain_list = d.chain_list_fast; v.chain_tc = d.chain_tc; v.chain_cls = d.chain_cls; v.be_n = d.be_n; v.chain_be = d.chain_be; v.pe2_lt = d.pe2_lt; v.PE2P = b->pe2p; v.SPC = d.SP; v.pad2_ = 0; v.code_lds = 0;
            v.fe = d.fe; v.Wf = d.Wf; v.Wb = d.Wb; v.pe_lt = d.pe_lt; v.af = d.af; v.ab = d.ab; v.tot = d.tot;
            v.fa = d.fa; v.fb = d.fb; v.mrow = d.mrow; v.err = d.err; v.dbg = b->d_dbg;
            const int rpt = b->fbv_rpt;
            v.SPAD = ((std::max(d.S, FBV_P * rpt) + 7) / 8) * 8;
            const int mdp = (d.M * d.D + 1) & ~1;
            int nt = std::max(FBV_P * v.G2, NV * v.SPW);
            size_t lds = 0; const int blk = 0;      // no emission ring: publishing lanes load their emission value directly
            for (;;) {
                // breakend fast path: product tables + pair codes in LDS (the allele-distance matrix is then
                // not needed there)
                const size_t code_bytes = (size_t)FBV_P * rpt * d.SP * 2;
                const bool want_code = d.pe2_lt != nullptr && d.pcode != nullptr && d.M <= 3 && b->max_adist < 64 && (d.SP & 1) == 0 && b->opt[RMX_OPT_FB_BREAKEND_CODES];
                size_t base_ = ((size_t)NV * 2 * v.SPAD + (size_t)NV * FBV_P * d.SP + NV * 4 + 128) * 8 +
                               (((size_t)d.C * d.S * d.M + 15) & ~(size_t)15) + 64 + (size_t)b->be_cap * 4;
                size_t fixed;
                v.code_lds = (want_code && base_ + (size_t)NV * b->pe2p * 8 + code_bytes <= kLdsBudget) ? 1 : 0;
                if (v.code_lds) { fixed = base_ + (size_t)NV * b->pe2p * 8 + code_bytes; v.amat_lds = 0; }
                else {
                    fixed = base_ + (size_t)NV * mdp * 8;
                    v.amat_lds = (fixed + (size_t)d.S * d.S + 16 <= kLdsBudget) ? 1 : 0;
                    if (v.amat_lds) fixed += (((size_t)d.S * d.S + 15) & ~(size_t)15);
                }
                lds = fixed;
                if (lds <= kLdsBudget || NV == 1) break;
                NV /= 2; nt = std::max(FBV_P * v.G2, NV * v.SPW);
            }
            v.BLK = blk;
            fbv_kernel_t kf = fbv_kernel_for(rpt, NV, blk);
            if (kf && nt <= 768 && lds <= kLdsBudget) {
                HIPCHK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kf, dim3(b->n_fast, (nr + NV - 1) / NV, 2), dim3(nt), lds, b->stream, v);
                done_fast = true; fast = true; b->last_fb_kernel = 2; b->last_fb_nv = b->last_fb_nv_max = NV;
            }
        }
        if (!done_fast && b->fbv_rpt == 0 && b->fbk_ok && b->fbk_kmax < 64 && d.pe2x_lt && b->n_fast > 0 && b->opt[RMX_OPT_FB_KERNEL] == 0 && d.S <= 360) {
            // state grid too large for register-resident weights, matrix cores: 8-bit distances in registers, B operands
            // looked up in a 64-entry LDS table (k_fbq); fb_kernel = 3 selects the vector kernel k_fbk below instead
            const int KB = d.S <= 256 ? 64 : 90;
            const int NWq = (d.S + 29) / 30;
            // (as for k_fbm: a shape per chain from the work-item table; above 256 states -- 90 k-blocks -- two restarts per workgroup do not fit the
            // register file next to the 90 address registers (measured: spills, 12 200 cycles per step against 13 500 with four): one or four)
            static const double fbq_cost64[5] = {0., 5040., 8440., 0., 9520.}, fbq_cost90[5] = {0., 7150., 0., 0., 13550.};
            rmx_batch::FbItems *items = nullptr;
            if ((rc = fb_items_for(b, r0, r1, b->opt[RMX_OPT_FB_NV], KB == 64 ? fbq_cost64 : fbq_cost90, &items))) return rc;
            FbmArgs m;
            memset(&m, 0, sizeof m);
            m.S = d.S; m.SP = d.SP; m.M = d.M; m.D = d.D; m.C = d.C; m.N = d.N; m.NBE = d.NBE; m.cn_max = d.cn_max; m.r0 = r0; m.r1 = r1;
            m.PE2P = b->pe2p; m.SPC = 0; m.VR = std::max(4 * KB, ((NWq * 30 + 7) / 8) * 8); m.pen = d.pen;
            m.chain_start = d.chain_start; m.chain_end = d.chain_end; m.chain_list = d.chain_list_fast; m.chain_tc = d.chain_tc; m.chain_cls = d.chain_cls;
            m.be_n = d.be_n; m.chain_be = d.chain_be; m.fe = d.fe; m.Wf = d.Wf; m.Wb = d.Wb; m.pe2_lt = d.pe2x_lt; m.af = d.af; m.ab = d.ab; m.tot = d.tot;
            m.fa = d.fa; m.fb = d.fb; m.mrow = d.mrow; m.err = d.err; m.dbg = b->d_dbg;
            const size_t lds = (size_t)2 * m.VR * 4 * 8 + (size_t)2 * 4 * m.PE2P * 8 + 64 * 32 * 8 + 64 * 8 + (size_t)4 * KB * 4 + (size_t)b->be_cap * 4 + 64;
            m.items = items->dev;
            void (*kf)(FbmArgs, const double *, const uint32_t *, const uint32_t *) = KB == 64 ? k_fbq<64> : k_fbq<90>;
            if (NWq <= 12 && 4 * KB >= d.S && lds <= kLdsBudget) {
                HIPCHK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kf, dim3(items->n), dim3(64 * NWq), lds, b->stream, m,
                                   (const double *)b->d_wk, (const uint32_t *)b->d_cnpack, (const uint32_t *)b->d_totpack);
                done_fast = true; fast = true; b->last_fb_kernel = 4; b->last_fb_nv = items->nv_min; b->last_fb_nv_max = items->nv_max;
            }
        }
        if (!done_fast && b->fbv_rpt == 0 && b->fbk_ok && d.pe2_lt && b->n_fast > 0 && (b->opt[RMX_OPT_FB_KERNEL] == 0 || b->opt[RMX_OPT_FB_KERNEL] == 3)) {
            // state grid too large for register-resident weights: weights from packed copy numbers on the fly
            const int nr = r1 - r0;
            int NV = 1;
            while (NV < 4 && (long)b->n_fast * 2 * ((nr + NV - 1) / NV) > 256) NV *= 2;
            { const int want = b->opt[RMX_OPT_FB_NV]; if (want == 1 || want == 2 || want == 4) NV = want; }
            FbvArgs v;
            memset(&v, 0, sizeof v);
            v.S = d.S; v.SP = d.SP; v.M = d.M; v.D = d.D; v.C = d.C; v.N = d.N; v.NBE = d.NBE; v.cn_max = d.cn_max;
            v.r0 = r0; v.r1 = r1; v.pen = d.pen;
            // (round 5: two row slices per column pair above 512 states -- blocks of PP * G2 <= 1 024 threads up to 1 024 states)
            const int PP = d.S > 512 ? 2 : 4;
            const int gq = 64 / PP;      // whole waves per block: PP * G2 a multiple of 64
            v.G2 = (((d.S + 1) / 2 + gq - 1) / gq) * gq; v.SPW = ((d.S + 63) / 64) * 64;
            const int nch = (d.S + PP * 16 - 1) / (PP * 16);
            v.SPAD = PP * 16 * nch;
            v.PE2P = b->pe2p;
            v.chain_start = d.chain_start; v.chain_end = d.chain_end; v.tclass = d.tclass; v.brk_slot = d.brk_slot;
            v.chain_list = d.chain_list_fast; v.chain_tc = d.chain_tc; v.chain_cls = d.chain_cls; v.be_n = d.be_n; v.chain_be = d.chain_be; v.pe2_lt = d.pe2_lt;
            v.fe = d.fe; v.Wf = d.Wf; v.Wb = d.Wb; v.pe_lt = d.pe_lt; v.af = d.af; v.ab = d.ab; v.tot = d.tot;
            v.fa = d.fa; v.fb = d.fb; v.mrow = d.mrow; v.err = d.err; v.dbg = b->d_dbg;
            int nt = ((PP * v.G2 + v.SPW - 1) / v.SPW) * v.SPW;
            const bool m4 = d.M == 4;
            auto lds_of = [&](int nv_) { return ((size_t)nv_ * 2 * v.SPAD + (size_t)nv_ * PP * d.SP + nv_ * 4 + (size_t)nv_ * b->pe2p + FBK_WKN) * 8 + (size_t)(m4 ? 3 : 2) * v.SPAD * 4 + (size_t)b->be_cap * 4 + 64; };
            // (four clones: a breakend's table is D^3 entries per vector -- 27 KB at max_cn 6, 55 KB at 8 -- so fewer vectors per workgroup where four do not fit the LDS;
            //  phase 2 publishes the vectors in at most two passes of nt / SPW each)
            const int vpp = nt / v.SPW;
            while (NV > 1 && (lds_of(NV) > kLdsBudget || (NV + vpp - 1) / vpp > 2) && b->opt[RMX_OPT_FB_NV] == 0) NV /= 2;
            const size_t lds = lds_of(NV);
            // (round 4, late: blocks of up to 1 024 threads -- the kernel needs 59 / 82 / 117 registers at 1 / 2 / 4 vectors -- take it from 360 to 512 states)
            if (nt <= 1024 && lds <= kLdsBudget && (NV + vpp - 1) / vpp <= 2) {
                void (*kf)(FbvArgs, const double *, const uint32_t *, const uint32_t *, const uint32_t *) =
                    PP == 2 ? (m4 ? (NV == 1 ? k_fbk<1, 1024, true, 2> : (NV == 2 ? k_fbk<2, 1024, true, 2> : k_fbk<4, 1024, true, 2>))
                                  : (NV == 1 ? k_fbk<1, 1024, false, 2> : (NV == 2 ? k_fbk<2, 1024, false, 2> : k_fbk<4, 1024, false, 2>))) :
                    m4 ? (nt <= 768 ? (NV == 1 ? k_fbk<1, 768, true, 4> : (NV == 2 ? k_fbk<2, 768, true, 4> : k_fbk<4, 768, true, 4>)) : (NV == 1 ? k_fbk<1, 1024, true, 4> : (NV == 2 ? k_fbk<2, 1024, true, 4> : k_fbk<4, 1024, true, 4>)))
                       : (nt <= 768 ? (NV == 1 ? k_fbk<1, 768, false, 4> : (NV == 2 ? k_fbk<2, 768, false, 4> : k_fbk<4, 768, false, 4>)) : (NV == 1 ? k_fbk<1, 1024, false, 4> : (NV == 2 ? k_fbk<2, 1024, false, 4> : k_fbk<4, 1024, false, 4>)));
                HIPCHK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kf, dim3(b->n_fast, (nr + NV - 1) / NV, 2), dim3(nt), lds, b->stream, v, (const double *)b->d_wk, (const uint32_t *)b->d_cnpack, (const uint32_t *)b->d_totpack, (const uint32_t *)b->d_cnpack2);
                done_fast = true; fast = true; b->last_fb_kernel = 3; b->last_fb_nv = b->last_fb_nv_max = NV;
            }
        }
        if (!fast) { b->last_fb_kernel = 0; b->last_fb_nv = b->last_fb_nv_max = 1; }
        const int ngen = fast ? b->n_generic : d.NC;
        if (ngen > 0) {
            a.amat_lds = 0; a.P = b->fbG.P; a.BLK = b->fbG.BLK; a.SPAD = b->fbG.SPAD; a.chain_list = fast ? d.chain_list_generic : d.chain_list_all;
            hipLaunchKernelGGL(fb_kernel_for(0), dim3(ngen, r1 - r0, 2), dim3(b->fbG.NT), b->fbG_lds, b->stream, a);
        }
        HIPCHK(hipGetLastError());
    }
    for (int r = r0; r < r1; r++) b->lt_model[r] = b->d.tmodel;
    for (int r = r0; r < r1; r++) { if (!b->lt_valid[r]) { b->lt_valid[r] = 1; int one = 1; HIPCHK(hipMemcpyAsync(b->d_lt_valid + r, &one, 4, hipMemcpyHostToDevice, b->stream)); } }
    return RMX_OK;
}
static int p_cn_marginals(rmx_batch *b, int r0, int r1, bool fuse_next) {
    {
        ProfScope ps(b, KID_MARGINALS);
        if (use_strip(b)) hipLaunchKernelGGL(cells_kernel(b, fuse_next ? 3 : 1, CM_ALL, b->use_cache ? 2 : 0), strip_grid(b, r1 - r0), dim3(256), 0, b->stream, b->d, r0);   // the F pass made the cache current
        else hipLaunchKernelGGL(k_marginals<true>, row_grid(b, r1 - r0), dim3(256), 0, b->stream, b->d, r0, b->G);
        HIPCHK(hipGetLastError());
    }
    if (fuse_next && use_strip(b)) std::swap(b->d.fe, b->d.fe_alt);      // the fused pass left the next sweep's scaled emissions in the second buffer
    for (int r = r0; r < r1; r++) { b->ab_dirty[r] = 0; b->comp_dirty[r] = 0; b->comp_base[r] = 0; b->logz_dirty[r] = 1; b->sig_valid[r] = (!fuse_next && use_strip(b) && b->d.sig_cnt) ? 1 : 0; b->sig_epoch[r]++; }
    return RMX_OK;
}
static int do_update_p_cn(rmx_batch *b, int r0, int r1, bool skip_frame = false, bool fuse_next = false) {
    int rc;
    // pairwise reductions at breakend adjacencies (feed update_p_breakpoint and the ELBO) read this sweep's fa / fb / fe
    if ((rc = p_cn_front(b, r0, r1, skip_frame)) || (rc = launch_pairwise_breakends(b, r0, r1, 0))) return rc;
    return p_cn_marginals(b, r0, r1, fuse_next);
}
// snapshot: also write the NEXT sweep's log_transmat tables (T of the p_breakpoint computed here, bpmodel.pyx:939) -- the caller then
// skips its own launch_brk_lut(pd_lt)
static int do_update_p_breakpoint(rmx_batch *b, int r0, int r1, bool snapshot = false) {
    const Dev &d = b->d;
    if (d.K > 0 && d.NBE > 0) {
        // probabilities and the tables that follow from them in one launch
        ProfScope ps(b, KID_BRK_UPDATE);
        hipLaunchKernelGGL(k_brk_update_lut, dim3(d.K, r1 - r0), dim3(128), (size_t)d.B * 16, b->stream, b->d, r0, b->d.pd_cached,
                           snapshot ? b->d.pd_lt : (double *)nullptr, b->d.pe_lt, b->d.pe2_lt, b->pe2p);
        HIPCHK(hipGetLastError());
    } else {
        if (d.K > 0) {
            ProfScope ps(b, KID_BRK_UPDATE);
            hipLaunchKernelGGL(k_brk_update, dim3(d.K, r1 - r0), dim3(128), (size_t)d.B * 16, b->stream, b->d, r0);
            HIPCHK(hipGetLastError());
        }
        // cached_log_transmat := T(new p_breakpoint)  (bpmodel.pyx:985)
        int rc = launch_brk_lut(b, r0, r1, b->d.pd_cached, nullptr);
        if (rc) return rc;
        if (snapshot && (rc = launch_brk_lut(b, r0, r1, b->d.pd_lt, b->d.pe_lt))) return rc;
    }
    const double pti = plain_T_mean_sum(b);
    for (int r = r0; r < r1; r++) { b->plain_T_init[r] = pti; b->cached_model[r] = b->d.tmodel; }
    return RMX_OK;
}
static int do_indicator(rmx_batch *b, int r0, int r1, int which) {
    int rc = ensure_ab(b, r0, r1);
    if (rc) return rc;
    dim3 g((b->d.N + 255) / 256, r1 - r0);
    if (which == 0) { ProfScope ps(b, KID_OUTLIER_TOTAL); hipLaunchKernelGGL(k_update_outlier_total, g, dim3(256), 0, b->stream, b->d, r0); }
    else if (which == 1) { ProfScope ps(b, KID_OUTLIER_ALLELE); hipLaunchKernelGGL(k_update_outlier_allele, g, dim3(256), 0, b->stream, b->d, r0); }
    else { ProfScope ps(b, KID_ALLELE_SWAP); hipLaunchKernelGGL(k_update_allele_swap, g, dim3(256), 0, b->stream, b->d, r0); }
    HIPCHK(hipGetLastError());
    return RMX_OK;
}

int rmx_update_framelogprob(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b); RANGE_CHECK(); int rc = do_framelogprob(b, r0, r1); return rc ? rc : check_errors(b, r0, r1); }
int rmx_update_p_cn(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b); RANGE_CHECK(); int rc = do_update_p_cn(b, r0, r1); return rc ? rc : check_errors(b, r0, r1); }
int rmx_update_p_breakpoint(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b); RANGE_CHECK(); int rc = do_update_p_breakpoint(b, r0, r1); return rc ? rc : check_errors(b, r0, r1); }
int rmx_update_p_outlier_total(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b); RANGE_CHECK(); int rc = do_indicator(b, r0, r1, 0); return rc ? rc : check_errors(b, r0, r1); }
int rmx_update_p_outlier_allele(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b); RANGE_CHECK(); int rc = do_indicator(b, r0, r1, 1); return rc ? rc : check_errors(b, r0, r1); }
int rmx_update_p_allele_swap(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b); RANGE_CHECK(); int rc = do_indicator(b, r0, r1, 2); return rc ? rc : check_errors(b, r0, r1); }

int rmx_variational_update(rmx_batch *b, int32_t r0, int32_t r1, int32_t iters) { BIND(b);
    RANGE_CHECK();
    // Between two sweeps of this call everything from the marginals of sweep i to the frame
    // log-probabilities of sweep i+1 is local to a segment: one fused pass (k_cells MODE 3) instead of
    // marginals + update_p_outlier_total + update_p_outlier_allele + update_p_allele_swap + frame pass.
    const bool fusable = use_strip(b) && b->use_cache && b->opt[RMX_OPT_FUSE_SWEEPS];
    // the breakend branch of a sweep (pairwise reductions, update_p_breakpoint) next to its marginal pass
    const bool two_streams = use_strip(b) && b->d.NBE > 0 && b->opt[RMX_OPT_TWO_STREAMS];
    if (two_streams && !b->stream2) {
        if (b->opt[RMX_OPT_STREAM_POOL]) { if (pool_acquire(b->device, 1, &b->stream2, b->opt[RMX_OPT_CU_PARTITION]) != RMX_OK) return fail(RMX_EDEVICE, "hipStreamCreate failed"); b->pooled_stream2 = true; }
        else HIPCHK(create_stream_for(b->device, b->opt[RMX_OPT_CU_PARTITION], &b->stream2));
        HIPCHK(hipEventCreateWithFlags(&b->ev_fb, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b->ev_brk, hipEventDisableTiming));
    }
    bool snapshot_done = false;
    // pace_sweeps: a sweep's forward-backward point is reached in (device) real time -- not before the previous sweep's
    // forward-backward has finished -- instead of queueing the call's sweeps at once (what the restart groups
    // of a GPU gain from it depends on the state grid and the group size)
    const bool paced = b->opt[RMX_OPT_PACE_SWEEPS] != 0;
    if (paced && !b->ev_pace) HIPCHK(hipEventCreateWithFlags(&b->ev_pace, hipEventDisableTiming));
    for (int it = 0; it < iters; it++) {
        int rc;
        if (paced && it > 0) HIPCHK(hipEventSynchronize(b->ev_pace));
        const bool fused_in = fusable && it > 0, fuse_out = fusable && it + 1 < iters;
        if (!fused_in && (rc = do_indicator(b, r0, r1, 2))) return rc;
        if (two_streams) {
            if ((rc = p_cn_front(b, r0, r1, fused_in, snapshot_done))) return rc;
            if (paced) HIPCHK(hipEventRecord(b->ev_pace, b->stream));
            // breakend branch on the second stream: pairwise reductions -> p_breakpoint -> cached transition tables
            hipStream_t main_stream = b->stream;
            HIPCHK(hipEventRecord(b->ev_fb, main_stream));
            HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_fb, 0));
            b->stream = b->stream2;
            rc = launch_pairwise_breakends(b, r0, r1, 0);
            // the next sweep's transition snapshot is T(the p_breakpoint just computed): built with it, off the main stream
            snapshot_done = it + 1 < iters;
            if (!rc) rc = do_update_p_breakpoint(b, r0, r1, snapshot_done);
            b->stream = main_stream;
            if (rc) return rc;
            HIPCHK(hipEventRecord(b->ev_brk, b->stream2));
            if ((rc = p_cn_marginals(b, r0, r1, fuse_out))) return rc;
            HIPCHK(hipStreamWaitEvent(main_stream, b->ev_brk, 0));
        } else {
            if ((rc = do_update_p_cn(b, r0, r1, fused_in, fuse_out)) || (rc = do_update_p_breakpoint(b, r0, r1))) return rc;
            if (paced) HIPCHK(hipEventRecord(b->ev_pace, b->stream));
        }
        if (!fuse_out && ((rc = do_indicator(b, r0, r1, 0)) || (rc = do_indicator(b, r0, r1, 1)))) return rc;
    }
    return check_errors(b, r0, r1);
}

// ---- objectives ------------------------------------------------------------------------
static int elbo_parts(rmx_batch *b, int r0, int r1, bool exact_parts, double *out4 /* [nr][4] host */) {
    int rc = ensure_ab(b, r0, r1);
    if (rc) return rc;
    const int nr = r1 - r0;
    const double *full_plain = nullptr;
    if (exact_parts) {
        // energy and entropy individually also contain sum_plain joint*T, which cancels in the ELBO
        for (int r = r0; r < r1; r++) if (!b->lt_valid[r]) exact_parts = false;   // pre-update: both closed-form
    }
    std::vector<double> plain_host;
    double *d_fp = nullptr;
    if (exact_parts && !b->plain_list.empty()) {
        const int np = (int)b->plain_list.size();
        if (!b->d_plain_list) { if ((rc = dalloc(b, &b->d_plain_list, np)) || (rc = dalloc(b, &b->d_plain_jt, (size_t)b->R * np))) return rc;
            HIPCHK(hipMemcpy(b->d_plain_list, b->plain_list.data(), (size_t)np * 4, hipMemcpyHostToDevice)); }
        { ProfScope ps(b, KID_PAIRWISE);
          // (the joint belongs to the log_transmat snapshot: plain tables of the model each restart's snapshot was taken under)
          for (int r = r0; r < r1; r++)
              hipLaunchKernelGGL(k_pairwise, dim3(np, 1), dim3(256), 0, b->stream, dev_for_model(b, b->lt_model[r]), r, 0, (const int32_t *)b->d_plain_list,
                                 b->d_plain_jt + (size_t)(r - r0) * np, PairAux{}); }
        std::vector<double> jt((size_t)nr * np);
        HIPCHK(hipMemcpyAsync(jt.data(), b->d_plain_jt, jt.size() * 8, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        plain_host.assign(nr, 0.);
        for (int i = 0; i < nr; i++) for (int j = 0; j < np; j++) plain_host[i] += jt[(size_t)i * np + j];
        HIPCHK(hipMalloc((void **)&d_fp, nr * 8));
        HIPCHK(hipMemcpy(d_fp, plain_host.data(), nr * 8, hipMemcpyHostToDevice));
        full_plain = d_fp;
    }
    { ProfScope ps(b, KID_ELBO_SEG); hipLaunchKernelGGL(k_elbo_seg, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, b->d, r0, b->d_partial, b->d_be_e); }
    // plain_T_init may differ per restart only in exotic call orders: one launch per run of equal values
    for (int r = r0; r < r1;) {
        int e = r + 1;
        while (e < r1 && b->plain_T_init[e] == b->plain_T_init[r]) e++;
        ProfScope ps(b, KID_ELBO_FINAL);
        hipLaunchKernelGGL(k_elbo_final, dim3(e - r), dim3(256), 0, b->stream, b->d, r, b->d_partial + (size_t)(r - r0) * ELBO_BLOCKS * 3,
                           b->d_be_e + (size_t)(r - r0) * b->d.NBE, ELBO_BLOCKS,
                           (const int *)b->d_lt_valid, b->plain_T_init[r], full_plain ? full_plain + (r - r0) : nullptr, b->d_out4 + (size_t)(r - r0) * 4);
        r = e;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out4, b->d_out4, (size_t)nr * 32, hipMemcpyDeviceToHost, b->stream));
    rc = check_errors(b, r0, r1);
    if (d_fp) hipFree(d_fp);
    if (rc) return rc;
    for (int r = r0; r < r1; r++) if (b->lt_valid[r]) { b->logZ[r] = out4[(r - r0) * 4 + 3]; b->logz_dirty[r] = 0; }
    // A transition-model change between the two snapshots (cn_model.py:404 sets the model after the constructor cached
    // model-0 tables; bpmodel.pyx:939 vs :985): the entropy reads log_transmat (model of the last update_p_cn), the energy
    // cached_log_transmat (model of the last update_p_breakpoint / the constructor).  Their plain-adjacency terms no longer
    // cancel and the allele-flip term of the breakend adjacencies differs: both are summed explicitly here.
    for (int r = r0; r < r1; r++) {
        if (!b->lt_valid[r] || b->lt_model[r] == b->cached_model[r]) continue;
        const int lm = b->lt_model[r], cm = b->cached_model[r];
        if (!b->Tval_m[lm] || !b->Tval_m[cm]) return fail(RMX_EUNSUPPORTED, "transition tables of a previous model are not available");
        const Dev dl = dev_for_model(b, lm);
        const Dev &d = b->d;
        const int np = (int)b->plain_list.size();
        double e_corr = 0., h_corr = 0.;
        double *d_jt2 = nullptr;
        std::vector<double> jt(np), jt2(np);
        if (np > 0) {
            if (!b->d_plain_list) { if ((rc = dalloc(b, &b->d_plain_list, np)) || (rc = dalloc(b, &b->d_plain_jt, (size_t)b->R * np))) return rc;
                HIPCHK(hipMemcpy(b->d_plain_list, b->plain_list.data(), (size_t)np * 4, hipMemcpyHostToDevice)); }
            HIPCHK(hipMalloc((void **)&d_jt2, (size_t)np * 8));
            PairAux ax{}; ax.Tval2 = b->Tval_m[cm]; ax.jt2_out = d_jt2; ax.no_state = 1;
            hipLaunchKernelGGL(k_pairwise, dim3(np, 1), dim3(256), 0, b->stream, dl, r, 0, (const int32_t *)b->d_plain_list, b->d_plain_jt, ax);
            HIPCHK(hipMemcpyAsync(jt.data(), b->d_plain_jt, (size_t)np * 8, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipMemcpyAsync(jt2.data(), d_jt2, (size_t)np * 8, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
            hipFree(d_jt2);
            double sl = 0., sc = 0.;
            for (int j = 0; j < np; j++) { sl += jt[j]; sc += jt2[j]; }
            if (exact_parts) e_corr += sc - sl;              // both parts already contain the lt-model sum: the energy's becomes the cached model's
            else { e_corr += sc; h_corr += sl; }
        }
        if (d.NBE > 0) {
            // breakend adjacencies: the energy's allele-flip term under the cached model's table instead of the snapshot's
            double *d_ja2 = nullptr;
            HIPCHK(hipMalloc((void **)&d_ja2, (size_t)d.NBE * 8));
            PairAux ax{}; ax.af2 = b->af_m[cm]; ax.ja2_out = d_ja2; ax.no_state = 1;
            hipLaunchKernelGGL(k_pairwise, dim3(d.NBE, 1), dim3(256), 0, b->stream, dl, r, 0, (const int32_t *)nullptr, (double *)nullptr, ax);
            std::vector<double> ja2(d.NBE), ja(d.NBE);
            HIPCHK(hipMemcpyAsync(ja2.data(), d_ja2, (size_t)d.NBE * 8, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipMemcpyAsync(ja.data(), d.be_ja + (size_t)r * d.NBE, (size_t)d.NBE * 8, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
            hipFree(d_ja2);
            for (int sl_ = 0; sl_ < d.NBE; sl_++) if (b->tclass[b->be_n[sl_]] >= 0) e_corr += -d.pen * (ja2[sl_] - ja[sl_]);
        }
        double *o = out4 + (size_t)(r - r0) * 4;
        o[0] += e_corr; o[1] += h_corr; o[2] = o[0] - o[1];
    }
    return RMX_OK;
}
int rmx_calculate_elbo(rmx_batch *b, int32_t r0, int32_t r1, double *out) { BIND(b);
    RANGE_CHECK();
    std::vector<double> o4((size_t)(r1 - r0) * 4);
    int rc = elbo_parts(b, r0, r1, false, o4.data());
    if (rc) return rc;
    for (int i = 0; i < r1 - r0; i++) out[i] = o4[i * 4 + 2];
    return RMX_OK;
}
int rmx_calculate_elbo_begin(rmx_batch *b, int32_t r0, int32_t r1) { BIND(b);
    RANGE_CHECK();
    if (b->elbo_pending) return fail(RMX_EARG, "an ELBO is already pending: rmx_calculate_elbo_end first");
    const int nr = r1 - r0;
    if (!b->h_elbo) {
        HIPCHK(hipHostMalloc((void **)&b->h_elbo, ((size_t)b->R * 4 + (size_t)b->R) * 8, hipHostMallocDefault));      // [R][4] results, then R error words
        HIPCHK(hipEventCreateWithFlags(&b->ev_elbo, hipEventDisableTiming));
    }
    // the state in which the two transition snapshots belong to different models needs host-side sums between launches (elbo_parts): not deferred
    bool mixed = false;
    for (int r = r0; r < r1; r++) mixed |= b->lt_valid[r] && b->lt_model[r] != b->cached_model[r];
    b->elbo_r0 = r0; b->elbo_r1 = r1; b->elbo_sync = mixed;
    b->elbo_pending = true;
    if (mixed) return RMX_OK;
    int rc = ensure_ab(b, r0, r1);
    if (rc) { b->elbo_pending = false; return rc; }
    { ProfScope ps(b, KID_ELBO_SEG); hipLaunchKernelGGL(k_elbo_seg, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, b->d, r0, b->d_partial, b->d_be_e); }
    for (int r = r0; r < r1;) {
        int e = r + 1;
        while (e < r1 && b->plain_T_init[e] == b->plain_T_init[r]) e++;
        ProfScope ps(b, KID_ELBO_FINAL);
        hipLaunchKernelGGL(k_elbo_final, dim3(e - r), dim3(256), 0, b->stream, b->d, r, b->d_partial + (size_t)(r - r0) * ELBO_BLOCKS * 3,
                           b->d_be_e + (size_t)(r - r0) * b->d.NBE, ELBO_BLOCKS,
                           (const int *)b->d_lt_valid, b->plain_T_init[r], (const double *)nullptr, b->d_out4 + (size_t)(r - r0) * 4);
        r = e;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(b->h_elbo, b->d_out4, (size_t)nr * 32, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(b->h_elbo + (size_t)b->R * 4, b->d.err, sizeof(uint32_t) * b->R, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipEventRecord(b->ev_elbo, b->stream));
    return RMX_OK;
}
int rmx_calculate_elbo_end(rmx_batch *b, double *out) { BIND(b);
    if (!b || !out) return fail(RMX_EARG, "bad argument");
    if (!b->elbo_pending) return fail(RMX_EARG, "no ELBO pending");
    b->elbo_pending = false;
    const int r0 = b->elbo_r0, r1 = b->elbo_r1;
    if (b->elbo_sync) return rmx_calculate_elbo(b, r0, r1, out);
    HIPCHK(hipEventSynchronize(b->ev_elbo));
    const uint32_t *e = reinterpret_cast<const uint32_t *>(b->h_elbo + (size_t)b->R * 4);
    int first = -1;
    for (int r = r0; r < r1; r++) if (e[r]) { if (first < 0) { first = r; g_err_restarts.clear(); } g_err_restarts.push_back(r); }
    if (first >= 0) {
        std::lock_guard<std::mutex> lk(b->mu);
        HIPCHK(hipMemsetAsync(b->d.err + r0, 0, sizeof(uint32_t) * (size_t)(r1 - r0), b->stream));
        return translate_error(b, first, e[first]);
    }
    for (int r = r0; r < r1; r++) {
        const double *o = b->h_elbo + (size_t)(r - r0) * 4;
        out[r - r0] = o[2];
        // (logZ as of the ELBO's launch: valid only while no later update_p_cn has run -- the flag it would clear is left alone)
    }
    return RMX_OK;
}
int rmx_calculate_variational_energy(rmx_batch *b, int32_t r0, int32_t r1, double *out) { BIND(b);
    RANGE_CHECK();
    std::vector<double> o4((size_t)(r1 - r0) * 4);
    int rc = elbo_parts(b, r0, r1, true, o4.data());
    if (rc) return rc;
    for (int i = 0; i < r1 - r0; i++) out[i] = o4[i * 4 + 0];
    return RMX_OK;
}
int rmx_calculate_variational_entropy(rmx_batch *b, int32_t r0, int32_t r1, double *out) { BIND(b);
    RANGE_CHECK();
    std::vector<double> o4((size_t)(r1 - r0) * 4);
    int rc = elbo_parts(b, r0, r1, true, o4.data());
    if (rc) return rc;
    for (int i = 0; i < r1 - r0; i++) out[i] = o4[i * 4 + 1];
    return RMX_OK;
}

// mask -> index list on the device, cached per restart across the many evaluations of one M-step
static int set_sample(rmx_batch *b, int r, const int64_t *sample) {
    const Dev &d = b->d;
    std::vector<int64_t> &cache = b->sample_cache[r];
    if ((int)cache.size() == d.N && memcmp(cache.data(), sample, (size_t)d.N * 8) == 0) return RMX_OK;
    cache.assign(sample, sample + d.N);
    if (b->ev_lists) HIPCHK(hipEventSynchronize(b->ev_lists));       // a pending rmx_set_sample_lists scatter must not land after this upload
    std::vector<int32_t> idx;
    idx.reserve(256);
    for (int n = 0; n < d.N; n++) if (sample[n] != 0) idx.push_back(n);
    b->sample_count[r] = (int)idx.size(); b->sample_epoch[r]++;
    { int32_t c32 = (int32_t)idx.size(); HIPCHK(hipMemcpy(b->d_counts + r, &c32, 4, hipMemcpyHostToDevice)); }
    if (!idx.empty()) HIPCHK(hipMemcpy(b->d_sample + (size_t)r * d.N, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    return RMX_OK;
}
// queue one evaluation of E[ll] (and optionally its h-gradient) on restart r's current sample; the
// (1+MAXC) results land in `dst` (device)
static int queue_ell(rmx_batch *b, int r, bool grad, double *dst) {
    const Dev &d = b->d;
    const int W = 1 + RMX_MAX_CLONES;
    const int cnt = b->sample_count[r];
    double *partial = b->d_ell_partial + (size_t)r * std::max(d.N, ELBO_BLOCKS) * W;
    const int32_t *list = b->d_sample + (size_t)r * d.N;
    int rc;
    if (cnt == d.N && !grad) {
        if ((rc = ensure_ab(b, r, r + 1))) return rc;
        { ProfScope ps(b, KID_ELL_FULL); hipLaunchKernelGGL(k_ell_full, dim3(ELBO_BLOCKS), dim3(256), 0, b->stream, b->d, r, partial); }
        { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_ell_final, dim3(1), dim3(256), 0, b->stream, (const double *)partial, ELBO_BLOCKS, dst); }
    } else {
        if ((rc = ensure_tables(b, r, r + 1, false))) return rc;
        {
            ProfScope ps(b, KID_ELL_LIST);
            const int32_t r32 = r;
            // objective + gradient from the lists of states with posterior mass when they are current (the batched
            // evaluation of the lock-step h M-step asks the same question: both drivers sum the same terms)
            if (grad && ell_sparse_ok(b, 1, &r32)) hipLaunchKernelGGL(k_ell_list_sparse_grad, dim3((cnt + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK), dim3(256), 0, b->stream, b->d, r, list, cnt, partial);
            else if (grad) hipLaunchKernelGGL(k_ell_list<true>, dim3(cnt), ell_block(b), 0, b->stream, b->d, r, list, partial);
            else if (ell_sparse_ok(b, 1, &r32)) hipLaunchKernelGGL(k_ell_list_sparse_val, dim3((cnt + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK), dim3(256), 0, b->stream, b->d, r, list, cnt, partial);
            else hipLaunchKernelGGL(k_ell_list<false>, dim3(cnt), ell_block(b), 0, b->stream, b->d, r, list, partial);
        }
        { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_ell_final, dim3(1), dim3(256), 0, b->stream, (const double *)partial, cnt, dst); }
    }
    HIPCHK(hipGetLastError());
    return RMX_OK;
}

int rmx_set_sample(rmx_batch *b, int32_t r, const int64_t *sample) { BIND(b);
    if (!b || r < 0 || r >= b->R || !sample) return fail(RMX_EARG, "bad argument");
    return set_sample(b, r, sample);
}

int rmx_expected_log_likelihood(rmx_batch *b, int32_t r, const int64_t *sample, double *ell_out, double *partial_h_out) { BIND(b);
    if (!b || r < 0 || r >= b->R || !ell_out) return fail(RMX_EARG, "bad argument");
    const Dev &d = b->d;
    int rc;
    if (sample) { if ((rc = set_sample(b, r, sample))) return rc; }
    else if (b->sample_count[r] < 0) return fail(RMX_EARG, "no sample set for this restart");
    const int W = 1 + RMX_MAX_CLONES;
    if (b->sample_count[r] == 0) { *ell_out = 0.; if (partial_h_out) for (int m = 0; m < d.M; m++) partial_h_out[m] = 0.; return RMX_OK; }
    double *dst = b->d_ell_out + (size_t)r * 8;
    const size_t HW = (size_t)64 * W;
    double *hp = b->h_pinned + (size_t)r * HW;
    uint32_t *eh = reinterpret_cast<uint32_t *>(b->h_pinned + (size_t)b->R * HW) + r;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        if ((rc = queue_ell(b, r, partial_h_out != nullptr, dst))) return rc;
        HIPCHK(hipMemcpyAsync(hp, dst, W * 8, hipMemcpyDeviceToHost, b->stream));
        if ((rc = finish_restart(b, r, eh))) return rc;
    }
    HIPCHK(hipEventSynchronize(b->done_ev[r]));
    if (*eh) { uint32_t v = *eh; std::lock_guard<std::mutex> lk(b->mu); HIPCHK(hipMemsetAsync(b->d.err + r, 0, sizeof(uint32_t), b->stream)); return translate_error(b, r, v); }
    *ell_out = hp[0];
    if (partial_h_out) for (int m = 0; m < d.M; m++) partial_h_out[m] = hp[1 + m];
    return RMX_OK;
}

// E[ll] on restart r's current sample for a grid of values of one likelihood parameter, evaluated
// back to back with a single host round trip (the 20-point grid of scipy.optimize.brute in
// BreakpointModel.update_param, cn_model.py:553-558).  Leaves the parameter at values[G-1], as the
// sequential evaluation would.
int rmx_expected_ll_param_grid(rmx_batch *b, int32_t r, int32_t param_id, const double *values, int32_t G, double *out) { BIND(b);
    if (!b || r < 0 || r >= b->R || !values || !out || G < 1 || G > 64) return fail(RMX_EARG, "bad argument");
    if (param_id < 0 || param_id >= RMX_P_HMM_LOG_NORM_CONST) return fail(RMX_EARG, "bad param id");
    if (b->sample_count[r] < 0) return fail(RMX_EARG, "no sample set for this restart");
    const int W = 1 + RMX_MAX_CLONES;
    const size_t HW = (size_t)64 * W;
    double *gout = b->d_grid_out + (size_t)r * HW;
    double *hp = b->h_pinned + (size_t)r * HW;
    uint32_t *eh = reinterpret_cast<uint32_t *>(b->h_pinned + (size_t)b->R * HW) + r;
    int rc;
    if (b->sample_count[r] == 0) { for (int g = 0; g < G; g++) out[g] = 0.; return rmx_set_param(b, r, param_id, values[G - 1]); }
    {
        std::lock_guard<std::mutex> lk(b->mu);
        for (int g = 0; g < G; g++) {
            if ((rc = rmx_set_param(b, r, param_id, values[g]))) return rc;
            if ((rc = queue_ell(b, r, false, gout + (size_t)g * W))) return rc;
        }
        HIPCHK(hipMemcpyAsync(hp, gout, (size_t)G * W * 8, hipMemcpyDeviceToHost, b->stream));
        if ((rc = finish_restart(b, r, eh))) return rc;
    }
    HIPCHK(hipEventSynchronize(b->done_ev[r]));
    if (*eh) { uint32_t v = *eh; std::lock_guard<std::mutex> lk(b->mu); HIPCHK(hipMemsetAsync(b->d.err + r, 0, sizeof(uint32_t), b->stream)); return translate_error(b, r, v); }
    for (int g = 0; g < G; g++) out[g] = hp[(size_t)g * W];
    return RMX_OK;
}

// One candidate value of one likelihood parameter per listed restart (distinct restarts), each on
// its restart's current sample: the evaluation round of a lock-step optimiser.  Three launches and
// one host round trip for the whole list.
// shared tail of the batched objective calls: stage the listed restarts' parameters, rebuild their
// state tables, evaluate every restart's sample, reduce, copy back nout values per request
// The layout of k_gradflat_round for this request list: made (k_gradflat_setup, queued on the batch stream ahead of the round that needs it) when
// the requests, their samples, their lists of states with posterior mass or their likelihood parameters differ from what the current layout was
// made for -- i.e. once per h M-step; the L-BFGS-B rounds that follow change h only.  Call with b->mu held.
static bool gradflat_ready(rmx_batch *b, int nreq, const int32_t *restarts, const StageArgs &sa) {
    if (!b->gf_lay.upre) {
        const size_t R = (size_t)b->R;
        if (dalloc(b, &b->gf_lay.upre, R * (NM_MAX_SAMPLE + 1)) != RMX_OK || dalloc(b, &b->gf_lay.blk, R * (NM_MAX_SAMPLE + 2)) != RMX_OK ||
            dalloc(b, &b->gf_lay.nblk, R) != RMX_OK || dalloc(b, &b->gf_lay.k8, R * NM_MAX_SAMPLE * 8) != RMX_OK) { b->gf_lay.upre = nullptr; return false; }
        if (hipHostMalloc((void **)&b->gf_nblk_host, 16 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { b->gf_lay.upre = nullptr; return false; }
    }
    std::vector<double> sig;
    sig.reserve((size_t)nreq * (3 + RMX_P_HMM_LOG_NORM_CONST));
    for (int i = 0; i < nreq; i++) {
        const int r = restarts[i];
        sig.push_back((double)r); sig.push_back((double)b->sample_epoch[r]); sig.push_back((double)b->sig_epoch[r]);
        for (int k = 0; k < RMX_P_HMM_LOG_NORM_CONST; k++) sig.push_back(sa.rp[i].p[k]);
    }
    if (sig == b->gf_sig) return true;
    hipLaunchKernelGGL(k_gradflat_setup, dim3(nreq), dim3(256), 0, b->stream, b->d, sa, (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->gf_lay, b->gf_nblk_host);
    if (hipGetLastError() != hipSuccess) return false;
    b->gf_sig.swap(sig);
    b->gf_first = true;
    return true;
}
static int run_ell_batch(rmx_batch *b, int nreq, const int32_t *restarts, bool grad, double *out, int mask = CM_ALL) {
    const Dev &d = b->d;
    const long long t_in = now_ns();
    const int W = 1 + RMX_MAX_CLONES;
    const int nout = grad ? W : 1;
    int maxcnt = 0;
    // up to 16 requests travel in the kernel arguments and the results come back through host-pinned
    // memory the kernel writes directly: three launches and one stream wait per round, no copy kernels
    const bool by_value = nreq <= 16;
    StageArgs sa;
    RestartParams *hs = by_value ? sa.rp : (RestartParams *)b->h_batch;
    int32_t *hl = by_value ? sa.rlist : (int32_t *)((char *)b->h_batch + (size_t)b->R * sizeof(RestartParams));
    for (int i = 0; i < nreq; i++) {
        const int r = restarts[i];
        fill_logr(b->rp[r]);
        hs[i] = b->rp[r]; hl[i] = r;
        maxcnt = std::max(maxcnt, b->sample_count[r]);
    }
    for (int i = nreq; by_value && i < 16; i++) { hs[i] = hs[0]; hl[i] = hl[0]; }
    double *res = by_value ? b->h_pinned : b->d_batch_out;          // host-pinned memory is device-accessible
    uint32_t *eres = by_value ? b->h_err : nullptr;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        if (by_value) {
            ProfScope ps(b, KID_STATE_TABLES);
            hipLaunchKernelGGL(k_state_tables_list_v, dim3(d.C, nreq), dim3(256), 0, b->stream, b->d, sa, b->d_rlist, b->d_rp_stage);
        } else {
            HIPCHK(hipMemcpyAsync(b->d_rp_stage, hs, (size_t)nreq * sizeof(RestartParams), hipMemcpyHostToDevice, b->stream));
            HIPCHK(hipMemcpyAsync(b->d_rlist, hl, (size_t)nreq * 4, hipMemcpyHostToDevice, b->stream));
            ProfScope ps(b, KID_STATE_TABLES);
            hipLaunchKernelGGL(k_state_tables_list, dim3(d.C, nreq), dim3(256), 0, b->stream, b->d, (const int32_t *)b->d_rlist, (const RestartParams *)b->d_rp_stage);
        }
        const int pstride = std::max(d.N, ELBO_BLOCKS) * W;
        bool final_done = false;
        if (maxcnt > 0) {
            ProfScope ps(b, KID_ELL_LIST);
            if (grad && by_value && ell_sparse_ok(b, nreq, restarts) && b->opt[RMX_OPT_GRAD_KERNEL] == 0 && maxcnt <= NM_MAX_SAMPLE && gradflat_ready(b, nreq, restarts, sa)) {
                // the lane chains of the half-wave kernel flat over the threads (k_gradflat_round): the layout and the per-segment constants
                // were made when this M-step's first round came through (gradflat_ready), the rounds only evaluate
                int nb = 0;
                if (b->gf_first) nb = (maxcnt * SEGL + 255) / 256 + 1;      // (an upper bound until the setup's block counts have come back with the first round)
                else for (int i = 0; i < nreq; i++) nb = std::max(nb, (int)b->gf_nblk_host[i]);
                b->gf_first = false;
                hipLaunchKernelGGL(k_gradflat_round, dim3(std::max(nb, 1), nreq), dim3(256), 0, b->stream, b->d, (const int32_t *)b->d_rlist, (const RestartParams *)b->d_rp_stage,
                                   (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->gf_lay, b->d_ell_partial, pstride);
            }
            else if (grad && ell_sparse_ok(b, nreq, restarts) && b->opt[RMX_OPT_GRAD_KERNEL] != 1)
                hipLaunchKernelGGL(k_ell_list_batch_sparse_grad, dim3((maxcnt + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK, nreq), dim3(256), 0, b->stream, b->d, (const int32_t *)b->d_rlist,
                                   (const RestartParams *)b->d_rp_stage, (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->d_ell_partial, pstride);
            else if (grad && ell_sparse_ok(b, nreq, restarts)) {
                // (the final sums ride in the same launch: the block that finishes a request last reduces its partials)
                hipLaunchKernelGGL(k_ell_list_batch_sparse_grad_final, dim3((maxcnt + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK, nreq), dim3(256), 0, b->stream, b->d, (const int32_t *)b->d_rlist,
                                   (const RestartParams *)b->d_rp_stage, (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->d_ell_partial, pstride,
                                   b->d_done, res, nout, eres);
                final_done = true;
            }
            else if (grad) hipLaunchKernelGGL(k_ell_list_batch<true>, dim3(maxcnt, nreq), ell_block(b), 0, b->stream, b->d, (const int32_t *)b->d_rlist, (const RestartParams *)b->d_rp_stage,
                                         (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->d_ell_partial, pstride);
            else {
                void (*kf)(Dev, const int32_t *, const RestartParams *, const int32_t *, const int32_t *, double *, int) = k_ell_list_batch<false, CM_ALL>;
                const bool sparse = (mask == 1 || mask == 2 || mask == 4 || mask == 8 || mask == CM_ALL) && ell_sparse_ok(b, nreq, restarts);
                if (sparse && mask == CM_ALL) kf = k_ell_list_batch_sparse<CM_ALL>;
                switch (mask) {
                case 1: kf = sparse ? k_ell_list_batch_sparse<1> : k_ell_list_batch<false, 1>; break;
                case 2: kf = sparse ? k_ell_list_batch_sparse<2> : k_ell_list_batch<false, 2>; break;
                case 3: kf = k_ell_list_batch<false, 3>; break;
                case 4: kf = sparse ? k_ell_list_batch_sparse<4> : k_ell_list_batch<false, 4>; break;
                case 8: kf = sparse ? k_ell_list_batch_sparse<8> : k_ell_list_batch<false, 8>; break;
                case 12: kf = k_ell_list_batch<false, 12>; break;
                default: break;
                }
                hipLaunchKernelGGL(kf, sparse ? dim3((maxcnt + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK, nreq) : dim3(maxcnt, nreq), sparse ? dim3(256) : ell_block(b), 0, b->stream, b->d,
                                   (const int32_t *)b->d_rlist, (const RestartParams *)b->d_rp_stage,
                                   (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->d_ell_partial, pstride);
            }
        }
        if (!final_done) { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_ell_final_batch, dim3(nreq), dim3(256), 0, b->stream, b->d, (const int32_t *)b->d_rlist, (const int32_t *)b->d_counts,
                                                             (const double *)b->d_ell_partial, pstride, res, nout, eres); }
        HIPCHK(hipGetLastError());
        for (int i = 0; i < nreq; i++) { b->tables_dirty[restarts[i]] = 0; b->segc_dirty[restarts[i]] = 1; b->ab_dirty[restarts[i]] = 1; }   // comp_dirty / cache_stale: set by the callers' setters
        if (!by_value) HIPCHK(hipMemcpyAsync(b->h_pinned, b->d_batch_out, (size_t)nreq * nout * 8, hipMemcpyDeviceToHost, b->stream));
    }
    const long long t_launched = now_ns();
    if (by_value) {
        HIPCHK(hipStreamSynchronize(b->stream));
        const long long t_done = now_ns();
        b->t_launch_ns += t_launched - t_in; b->t_wait_ns += t_done - t_launched; b->n_rounds++;
        if (int rc_ = report_request_errors(b, nreq, eres, [&](int i) { return (int)restarts[i]; })) return rc_;
        for (int i = 0; i < nreq * nout; i++) out[i] = b->h_pinned[i];
        b->t_post_ns += now_ns() - t_done;
        return RMX_OK;
    } else {
        int rc = check_errors(b, 0, b->R);
        if (rc) return rc;
    }
    for (int i = 0; i < nreq * nout; i++) out[i] = b->h_pinned[i];
    return RMX_OK;
}
static int check_request_list(rmx_batch *b, int nreq, const int32_t *restarts, bool need_sample = true) {
    std::vector<char> seen(b->R, 0);
    for (int i = 0; i < nreq; i++) {
        const int r = restarts[i];
        if (r < 0 || r >= b->R || seen[r]) return fail(RMX_EARG, "restart list must hold distinct valid restarts");
        if (need_sample && b->sample_count[r] < 0) return fail(RMX_EARG, "no sample set for a listed restart");
        seen[r] = 1;
    }
    return RMX_OK;
}
int rmx_expected_ll_batch(rmx_batch *b, int32_t nreq, const int32_t *restarts, int32_t param_id, const double *values, double *out) { BIND(b);
    if (!b || nreq < 1 || nreq > b->R || !restarts || !values || !out) return fail(RMX_EARG, "bad argument");
    if (param_id < 0 || param_id >= RMX_P_HMM_LOG_NORM_CONST) return fail(RMX_EARG, "bad param id");
    int rc = check_request_list(b, nreq, restarts);
    if (rc) return rc;
    for (int i = 0; i < nreq; i++)
        if ((rc = rmx_set_param(b, restarts[i], param_id, values[i]))) return rc;
    return run_ell_batch(b, nreq, restarts, false, out);
}

// ---- lock-step 1-D parameter search ------------------------------------------------------------
// scipy.optimize.fmin (Nelder-Mead, one variable, xatol = fatol = 1e-4, maxiter = maxfun = 200) as a
// resumable state machine: same floating-point operations in the same order as scipy's
// _minimize_neldermead (python twin: remixt_amd/lockstep.py fmin_1d; tests compare the two).
namespace {
using rmxh::Nm1;
}  // namespace

// scipy.optimize.brute(nll, ranges=[(lo, hi)], Ns=G, full_output=True)[0] for one likelihood parameter
// of every listed restart in lock step (BreakpointModel.update_param, cn_model.py:553-561): grid of G
// values (np.mgrid[lo:hi:complex(G)], passed in), first arg-min, Nelder-Mead polish from the best grid
// point; nll(v) = -E[ll] on the restart's current sample, +inf outside [lo, hi] without touching the
// model (cn_model.py:542-543).  Each round of evaluations is one rmx_expected_ll_batch call.  The
// parameter is left at the last value evaluated for the restart (the reference's acceptance test
// looks at exactly that state); xopt[i] receives the optimiser's result for restarts[i].
int rmx_param_search(rmx_batch *b, int32_t nreq, const int32_t *restarts, int32_t param_id, double lo, double hi,
                     const double *grid, int32_t G, double *xopt) { BIND(b);
    if (!b || nreq < 1 || nreq > b->R || !restarts || !grid || G < 1 || !xopt) return fail(RMX_EARG, "bad argument");
    int rc;
    if ((rc = check_request_list(b, nreq, restarts))) return rc;
    // One likelihood parameter moves one or two of the four likelihood components (negbin_r_0: the
    // non-outlier total-count term, ...).  The rest of E[ll] is constant over the search: it is taken
    // once from a full evaluation at the first grid value, and every later evaluation computes only the
    // moving part.  (Differs from the full sum by rounding only; RMX_SEARCH_FULL=1 evaluates everything.)
    static const int comp_bits[RMX_P_HMM_LOG_NORM_CONST] = {CM_LT0, CM_LT1, CM_LT0 | CM_LT1, CM_LT0, CM_LT1, CM_LA0, CM_LA1, CM_LA0 | CM_LA1, CM_LA0, CM_LA1, 0, 0, 0};
    int mask = comp_bits[param_id] & CM_ALL;
    if (mask == 0 || b->opt[RMX_OPT_SEARCH_MODE] == 4) mask = CM_ALL;
    std::vector<double> vals(nreq), out(nreq), best(nreq, INFINITY), x0(nreq), cst(nreq, 0.);
    auto eval = [&](int n_, const int32_t *rl_, const double *v_, double *o_, const int *who) -> int {
        for (int i = 0; i < n_; i++)
            if (int e_ = rmx_set_param(b, rl_[i], param_id, v_[i])) return e_;
        int e_ = run_ell_batch(b, n_, rl_, false, o_, mask);
        if (e_) return e_;
        if (mask != CM_ALL) for (int i = 0; i < n_; i++) o_[i] += cst[who ? who[i] : i];
        return RMX_OK;
    };
    if (mask != CM_ALL) {
        std::vector<double> full(nreq), part(nreq);
        for (int i = 0; i < nreq; i++) vals[i] = grid[0];
        for (int i = 0; i < nreq; i++)
            if ((rc = rmx_set_param(b, restarts[i], param_id, vals[i]))) return rc;
        if ((rc = run_ell_batch(b, nreq, restarts, false, full.data(), CM_ALL))) return rc;
        if ((rc = run_ell_batch(b, nreq, restarts, false, part.data(), mask))) return rc;
        for (int i = 0; i < nreq; i++) cst[i] = full[i] - part[i];
    }
    // Table-free evaluation for the four standard parameters: candidates are evaluated against the
    // restart's device parameters and state tables with only the searched parameter (and the table
    // entries it feeds) overridden in registers -- no table rebuild, no host mirror update per candidate,
    // and the whole grid in ONE launch.
    const bool table_free = (mask == CM_LT0 || mask == CM_LT1 || mask == CM_LA0 || mask == CM_LA1) && nreq <= 16 && G <= 32 &&
                            (param_id == RMX_P_NEGBIN_R_0 || param_id == RMX_P_NEGBIN_R_1 || param_id == RMX_P_BETABIN_M_0 || param_id == RMX_P_BETABIN_M_1) &&
                            b->opt[RMX_OPT_SEARCH_MODE] != 2;
    std::vector<double> lastval(nreq, grid[0]);
    const bool sparse_search = ell_sparse_ok(b, nreq, restarts);
    auto search_eval = [&](int n_, const int *who, const double *v_, int Gz, bool per_request, double *o_) -> int {
        // who: indices into restarts[] (nullptr: all, in order); v_: Gz grid values or n_ per-request values
        const Dev &d = b->d;
        SearchVals sv;
        sv.per_request = per_request ? 1 : 0; sv.Gz = Gz; sv.pad0 = sv.pad1 = 0;
        const int nv_ = per_request ? n_ * Gz : Gz;
        if (per_request && Gz > 1) sv.per_request = 2;
        for (int i = 0; i < 64; i++) { sv.v[i] = i < nv_ ? v_[i] : v_[0]; sv.lv[i] = std::log(sv.v[i]); }
        int maxcnt = 0;
        for (int i = 0; i < 16; i++) { const int r_ = restarts[who ? who[i < n_ ? i : 0] : (i < n_ ? i : 0)]; sv.rlist[i] = r_; if (i < n_) maxcnt = std::max(maxcnt, b->sample_count[r_]); }
        {
            std::lock_guard<std::mutex> lk(b->mu);
            if (maxcnt > 0) {
                ProfScope ps(b, KID_ELL_LIST);
                void (*kf)(Dev, SearchVals, const int32_t *, const int32_t *, double *, int) =
                    sparse_search ? (mask == CM_LT0 ? k_ell_search_sparse<CM_LT0> : (mask == CM_LT1 ? k_ell_search_sparse<CM_LT1> : (mask == CM_LA0 ? k_ell_search_sparse<CM_LA0> : k_ell_search_sparse<CM_LA1>)))
                                  : (mask == CM_LT0 ? k_ell_search<CM_LT0> : (mask == CM_LT1 ? k_ell_search<CM_LT1> : (mask == CM_LA0 ? k_ell_search<CM_LA0> : k_ell_search<CM_LA1>)));
                hipLaunchKernelGGL(kf, sparse_search ? dim3((maxcnt + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK, n_, Gz) : dim3(maxcnt, n_, Gz), sparse_search ? dim3(256) : ell_block(b), 0, b->stream, b->d, sv,
                                   (const int32_t *)b->d_sample, (const int32_t *)b->d_counts, b->d_ell_partial, std::max(maxcnt, 1));
            }
            { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_ell_search_final, dim3(n_ * Gz), dim3(256), 0, b->stream, b->d, sv, (const int32_t *)b->d_counts, (const double *)b->d_ell_partial, std::max(maxcnt, 1), b->h_pinned, b->h_err); }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(b->stream));
        if (int rc_ = report_request_errors(b, n_, b->h_err, [&](int i) { return (int)sv.rlist[i]; })) return rc_;
        for (int i = 0; i < n_; i++)
            for (int g = 0; g < Gz; g++) o_[i * Gz + g] = b->h_pinned[i * Gz + g] + cst[who ? who[i] : i];
        return RMX_OK;
    };
    if (table_free) {
        std::vector<double> og((size_t)nreq * G);
        if ((rc = search_eval(nreq, nullptr, grid, G, false, og.data()))) return rc;
        for (int g = 0; g < G; g++)
            for (int i = 0; i < nreq; i++) { const double J = -og[(size_t)i * G + g]; if (g == 0 || J < best[i]) { best[i] = J; x0[i] = grid[g]; } }
        for (int i = 0; i < nreq; i++) lastval[i] = grid[G - 1];
    } else
    for (int g = 0; g < G; g++) {
        for (int i = 0; i < nreq; i++) vals[i] = grid[g];
        if ((rc = eval(nreq, restarts, vals.data(), out.data(), nullptr))) return rc;
        for (int i = 0; i < nreq; i++) { const double J = -out[i]; if (g == 0 || J < best[i]) { best[i] = J; x0[i] = grid[g]; } }   // np.argmin: first minimum
    }
    std::vector<Nm1> nm(nreq);
    std::vector<int> want;          // requests waiting for a device evaluation
    std::vector<int32_t> rl(nreq);
    // RMX_SEARCH_LOOKAHEAD=1: table-free rounds also evaluate, in the same launch, the points each
    // optimiser may ask for next (Nm1::lookahead); when the next request is one of them its value is
    // already here and the round trip is saved.  A value is a function of the point only, so the sequence
    // of (point, value) pairs every optimiser sees -- and with it the result and the restart's last
    // evaluated point -- is unchanged.  Off by default: the evaluation kernel is bound by FP64
    // transcendental throughput, not by launch latency, so 4 candidates per request cost more device time
    // (taken from the other restart group's sweeps) than the halved round count returns (measured on
    // MI355X at the benchmark shape: 69.3 ms per step with, 66.2 ms without).
    constexpr int LOOK = 4;
    const bool lookahead = table_free && b->opt[RMX_OPT_SEARCH_MODE] == 3;
    struct Seen { double x[LOOK], f[LOOK]; int n = 0; };
    std::vector<Seen> seen(nreq);
    auto pump = [&](int i, double f) {
        // advance optimiser i until it needs a device evaluation or finishes; out-of-bounds points are +inf at once
        while (nm[i].advance(x0[i], f)) {
            const double v = nm[i].req;
            if (v < lo || v > hi) { f = INFINITY; continue; }
            bool hit = false;
            for (int c = 0; c < seen[i].n && !hit; c++) if (seen[i].x[c] == v) { f = seen[i].f[c]; hit = true; }
            if (hit) { lastval[i] = v; continue; }
            want.push_back(i);
            return;
        }
    };
    for (int i = 0; i < nreq; i++) pump(i, 0.);
    std::vector<double> cand((size_t)nreq * LOOK), oc((size_t)nreq * LOOK);
    while (!want.empty()) {
        std::vector<int> cur;
        cur.swap(want);
        for (size_t k = 0; k < cur.size(); k++) { rl[k] = restarts[cur[k]]; vals[k] = nm[cur[k]].req; lastval[cur[k]] = vals[k]; }
        if (lookahead) {
            for (size_t k = 0; k < cur.size(); k++) {
                double la[3];
                const int nl = nm[cur[k]].lookahead(la);
                for (int c = 0; c < LOOK; c++) cand[k * LOOK + c] = vals[k];                       // unused slots repeat the pending point
                for (int c = 0; c < nl; c++) if (la[c] >= lo && la[c] <= hi) cand[k * LOOK + 1 + c] = la[c];
            }
            if ((rc = search_eval((int)cur.size(), cur.data(), cand.data(), LOOK, true, oc.data()))) return rc;
            for (size_t k = 0; k < cur.size(); k++) {
                Seen &sn = seen[cur[k]];
                sn.n = LOOK;
                for (int c = 0; c < LOOK; c++) { sn.x[c] = cand[k * LOOK + c]; sn.f[c] = -oc[k * LOOK + c]; }
                out[k] = oc[k * LOOK];
            }
        }
        else if (table_free) { if ((rc = search_eval((int)cur.size(), cur.data(), vals.data(), 1, true, out.data()))) return rc; }
        else if ((rc = eval((int)cur.size(), rl.data(), vals.data(), out.data(), cur.data()))) return rc;
        for (size_t k = 0; k < cur.size(); k++) pump(cur[k], -out[k]);
    }
    // the table-free evaluations did not touch the model: leave the parameter at the value of the restart's
    // last evaluation, as the sequential scipy run (and the table-rebuilding path) does
    if (table_free) for (int i = 0; i < nreq; i++) if ((rc = rmx_set_param(b, restarts[i], param_id, lastval[i]))) return rc;
    for (int i = 0; i < nreq; i++) xopt[i] = nm[i].xopt();
    return RMX_OK;
}

// The sample of restart r for parameter slot `slot` (0..3) of rmx_param_search_multi
int rmx_set_sample_slot(rmx_batch *b, int32_t r, int32_t slot, const int64_t *sample) { BIND(b);
    if (!b || r < 0 || r >= b->R || slot < 0 || slot > 3 || !sample) return fail(RMX_EARG, "bad argument");
    const Dev &d = b->d;
    int rc;
    if (!b->d_msample) {
        if ((rc = dalloc(b, &b->d_msample, (size_t)4 * b->R * d.N)) || (rc = dalloc(b, &b->d_mcounts, (size_t)4 * b->R))) return rc;
        b->msample_count.assign((size_t)4 * b->R, -1);
    }
    std::vector<int32_t> idx;
    idx.reserve(256);
    for (int n = 0; n < d.N; n++) if (sample[n] != 0) idx.push_back(n);
    const size_t q = (size_t)slot * b->R + r;
    if (b->ev_lists) HIPCHK(hipEventSynchronize(b->ev_lists));
    b->msample_count[q] = (int)idx.size();
    { int32_t c32 = (int32_t)idx.size(); HIPCHK(hipMemcpy(b->d_mcounts + q, &c32, 4, hipMemcpyHostToDevice)); }
    if (!idx.empty()) HIPCHK(hipMemcpy(b->d_msample + q * d.N, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    return RMX_OK;
}

int rmx_set_sample_lists(rmx_batch *b, int32_t nlists, const int32_t *restarts, const int32_t *slots, const int32_t *offsets, const int32_t *indices) { BIND(b);
    if (!b || nlists < 0 || (nlists > 0 && (!restarts || !slots || !offsets))) return fail(RMX_EARG, "bad argument");
    if (nlists == 0) return RMX_OK;
    const Dev &d = b->d;
    int rc;
    bool any_slot = false;
    if (offsets[0] != 0) return fail(RMX_EARG, "offsets must start at 0");
    for (int i = 0; i < nlists; i++) {
        if (restarts[i] < 0 || restarts[i] >= b->R || slots[i] < -1 || slots[i] > 3 || offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > d.N)
            return fail(RMX_EARG, "bad sample list");
        if (offsets[i + 1] > offsets[i] && !indices) return fail(RMX_EARG, "bad argument");
        for (int j = offsets[i]; j < offsets[i + 1]; j++)
            if (indices[j] < 0 || indices[j] >= d.N || (j > offsets[i] && indices[j] <= indices[j - 1])) return fail(RMX_EARG, "sample indices must be ascending segment indices");
        any_slot |= slots[i] >= 0;
    }
    if (any_slot && !b->d_msample) {
        if ((rc = dalloc(b, &b->d_msample, (size_t)4 * b->R * d.N)) || (rc = dalloc(b, &b->d_mcounts, (size_t)4 * b->R))) return rc;
        b->msample_count.assign((size_t)4 * b->R, -1);
    }
    const size_t total = (size_t)offsets[nlists], need = (size_t)4 * nlists + total;
    if (b->ev_lists) HIPCHK(hipEventSynchronize(b->ev_lists));      // the previous scatter has read the staging area
    else HIPCHK(hipEventCreateWithFlags(&b->ev_lists, hipEventDisableTiming));
    if (need > b->h_lists_cap) {
        if (b->h_lists) HIPCHK(hipHostFree(b->h_lists));
        b->h_lists = nullptr; b->h_lists_cap = 0;
        const size_t cap = std::max(need * 2, (size_t)4096);
        HIPCHK(hipHostMalloc((void **)&b->h_lists, cap * 4));
        b->h_lists_cap = cap;
    }
    int32_t *hd = b->h_lists, *body = b->h_lists + (size_t)4 * nlists;
    for (int i = 0; i < nlists; i++) {
        const int cnt = offsets[i + 1] - offsets[i], r = restarts[i];
        hd[4 * i] = r; hd[4 * i + 1] = slots[i]; hd[4 * i + 2] = cnt; hd[4 * i + 3] = offsets[i];
        if (slots[i] < 0) { b->sample_count[r] = cnt; b->sample_cache[r].clear(); b->sample_epoch[r]++; }
        else b->msample_count[(size_t)slots[i] * b->R + r] = cnt;
    }
    if (total) memcpy(body, indices, total * 4);
    hipLaunchKernelGGL(k_scatter_samples, dim3(nlists), dim3(256), 0, b->stream, (const int32_t *)hd, (const int32_t *)body, b->d_sample, b->d_counts,
                       b->d_msample, b->d_mcounts, d.N, b->R);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev_lists, b->stream));
    return RMX_OK;
}

// rmx_param_search for up to four of the standard likelihood parameters (negbin_r_0/1, betabin_M_0/1, each
// at most once) of every listed restart AT ONCE: the searches of one restart are independent -- each
// evaluates only the likelihood component its parameter moves, on its own sample (rmx_set_sample_slot, slot
// = position in param_ids), against a model nothing writes to -- so all of them share their evaluation
// rounds: one launch pair and one stream wait per round for nparams x nreq optimisers instead of nreq.
// The objective of a search is its component's part of -E[ll] alone (the full-sum form differs by a
// constant, i.e. by rounding only).  Nothing is written to the model: xopt / lastval [nparams][nreq]
// return the optimisers' results and the last point each one evaluated (what the reference's acceptance
// test looks at); the caller sets parameters.  RMX_EUNSUPPORTED when the request does not qualify (other
// parameters, lists of states with posterior mass not current, more than 64 optimisers, G > 20, ...):
// the caller falls back to rmx_param_search per parameter.
int rmx_param_search_multi(rmx_batch *b, int32_t nreq, const int32_t *restarts, int32_t nparams, const int32_t *param_ids,
                           const double *lo, const double *hi, const double *grids, int32_t G, double *xopt, double *lastval) { BIND(b);
    if (!b || nreq < 1 || nreq > b->R || !restarts || nparams < 1 || nparams > 4 || !param_ids || !lo || !hi || !grids || G < 1 || !xopt || !lastval)
        return fail(RMX_EARG, "bad argument");
    int rc;
    if ((rc = check_request_list(b, nreq, restarts, false))) return rc;      // (its samples live in the parameter slots)
    const Dev &d = b->d;
    const int Q = nreq * nparams;
    MultiVals mv;
    memset(&mv, 0, sizeof(mv));
    int used = 0;
    for (int j = 0; j < nparams; j++) {
        int bit = 0;
        switch (param_ids[j]) {
        case RMX_P_NEGBIN_R_0: bit = CM_LT0; break; case RMX_P_NEGBIN_R_1: bit = CM_LT1; break;
        case RMX_P_BETABIN_M_0: bit = CM_LA0; break; case RMX_P_BETABIN_M_1: bit = CM_LA1; break;
        default: return fail(RMX_EUNSUPPORTED, "rmx_param_search_multi: not one of the four standard parameters");
        }
        if (used & bit) return fail(RMX_EUNSUPPORTED, "rmx_param_search_multi: parameter listed twice");
        used |= bit; mv.maskbit[j] = bit;
        if (!(lo[j] > 0.)) return fail(RMX_EUNSUPPORTED, "rmx_param_search_multi: lower bound must be positive");
    }
    if (Q > 64 || G > RMX_MULTI_G || !ell_sparse_ok(b, nreq, restarts) || !b->d_msample)
        return fail(RMX_EUNSUPPORTED, "rmx_param_search_multi: request does not qualify");
    int maxcnt = 0;
    for (int j = 0; j < nparams; j++)
        for (int i = 0; i < nreq; i++) {
            const int c = b->msample_count[(size_t)j * b->R + restarts[i]];
            if (c < 0) return fail(RMX_EARG, "no sample set for a listed restart and parameter slot");
            maxcnt = std::max(maxcnt, c);
            mv.rlist[j * nreq + i] = (int16_t)restarts[i]; mv.slot[j * nreq + i] = (int8_t)j;
        }
    if ((size_t)Q * G > (size_t)(b->R + 1) * 64 * (1 + RMX_MAX_CLONES) || b->R > 32767)
        return fail(RMX_EUNSUPPORTED, "rmx_param_search_multi: staging too small for this request");
    const size_t need = (size_t)Q * G * std::max(maxcnt, 1);      // partial sums [request][candidate][sampled segment]
    if (b->mpartial_cap < need) {
        dfree(b, b->d_mpartial); b->d_mpartial = nullptr; b->mpartial_cap = 0;
        if ((rc = dalloc(b, &b->d_mpartial, need))) return rc;
        b->mpartial_cap = need;
    }
    for (int j = 0; j < nparams; j++)
        for (int g = 0; g < G; g++) { mv.gv[j][g] = grids[(size_t)j * G + g]; mv.glv[j][g] = std::log(mv.gv[j][g]); }
    // the candidates are evaluated against the restarts' device parameters and state tables: bring them up to date
    // first (a rolled-back h, a parameter set since the last pass: the one-at-a-time search does this through the
    // table rebuild of its first full evaluation)
    {
        std::lock_guard<std::mutex> lk(b->mu);
        // (maximal runs of consecutive restarts in one call: ensure_tables batches up to 16 stale restarts per launch -- the request list of a restart
        // group is its whole range, and eight one-restart launches were 0.8 ms of an M-step next to the other group's sweeps)
        for (int i = 0; i < nreq;) {
            int j = i + 1;
            while (j < nreq && restarts[j] == restarts[j - 1] + 1) j++;
            if ((rc = ensure_tables(b, restarts[i], restarts[j - 1] + 1, false))) return rc;
            i = j;
        }
    }
    // one round: requests cur[0..n) (indices q = j * nreq + i), their values vals[] (ignored in the grid stage)
    std::vector<double> out((size_t)Q * G);
    auto round = [&](int n_, const int *cur, const double *vals, bool grid_stage) -> int {
        MultiVals m2 = mv;
        m2.grid_stage = grid_stage ? 1 : 0; m2.Gz = grid_stage ? G : 1;
        int mc = 0;
        for (int k = 0; k < n_; k++) {
            const int q = cur[k];
            m2.rlist[k] = mv.rlist[q]; m2.slot[k] = mv.slot[q];
            m2.v[k] = grid_stage ? 1. : vals[k]; m2.lv[k] = std::log(m2.v[k]);
            mc = std::max(mc, b->msample_count[(size_t)mv.slot[q] * b->R + mv.rlist[q]]);
        }
        for (int k = n_; k < 64; k++) { m2.rlist[k] = m2.rlist[0]; m2.slot[k] = m2.slot[0]; m2.v[k] = m2.v[0]; m2.lv[k] = m2.lv[0]; }
        {
            std::lock_guard<std::mutex> lk(b->mu);
            if (mc > 0 && !grid_stage && b->opt[RMX_OPT_SEARCH_MODE] == 6) {
                // search_mode 6 (round 5, measured and NOT the default): a Nelder-Mead round whose objective kernel sums its own partials (the last block of a
                // request, behind a release fence per block).  One launch fewer per round -- and 5 % off the headline (397 against 418-421 EM it/s, two
                // alternating runs each on one box): an agent-scope release writes the XCD's dirty L2 lines back, and next to the other restart
                // group's forward-backward and marginal passes the L2s are full of their rows; 800 blocks x 52 rounds of that per M-step
                ProfScope ps(b, KID_ELL_LIST);
                hipLaunchKernelGGL(k_ell_search_multi_final, dim3((mc + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK, n_), dim3(256), 0, b->stream, b->d, m2, (const int32_t *)b->d_msample,
                                   (const int32_t *)b->d_mcounts, b->d_mpartial, std::max(mc, 1), b->d_done, b->h_pinned, b->h_err);
                HIPCHK(hipGetLastError());
            } else {
            if (mc > 0) {
                ProfScope ps(b, KID_ELL_LIST);
                hipLaunchKernelGGL(k_ell_search_multi, dim3((mc + SEG_PER_BLOCK - 1) / SEG_PER_BLOCK, n_, m2.Gz), dim3(256), 0, b->stream, b->d, m2, (const int32_t *)b->d_msample,
                                   (const int32_t *)b->d_mcounts, b->d_mpartial, std::max(mc, 1));
            }
            { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_ell_multi_final, dim3(n_ * m2.Gz), dim3(256), 0, b->stream, b->d, m2, (const int32_t *)b->d_mcounts,
                                                                 (const double *)b->d_mpartial, std::max(mc, 1), b->h_pinned, b->h_err); }
            HIPCHK(hipGetLastError());
            }
        }
        HIPCHK(hipStreamSynchronize(b->stream));
        if (int rc_ = report_request_errors(b, n_, b->h_err, [&](int k) { return (int)m2.rlist[k]; })) return rc_;
        for (int k = 0; k < n_ * m2.Gz; k++) out[k] = b->h_pinned[k];
        return RMX_OK;
    };
    std::vector<int> all(Q);
    for (int q = 0; q < Q; q++) all[q] = q;
    // search_mode 5 (the default since round 5): rounds the device drives (k_search_round / k_search_advance) -- the optimisers' state lives on
    // the device, a round is a kernel pair, and the host queues the grid round and a batch of rounds back to back without waiting in between;
    // then it looks at the finished flags and queues more while any optimiser still runs (at most Nm1::maxfun rounds).  Half the latency of
    // the rounds driven from the host below (1.15 against 2.5 ms for 8 restarts alone on the GPU).  Round 4 kept it for batches that have
    // the GPU to themselves (next to another group's sweeps 415-416 against 424-426 EM it/s for the host rounds); with round 5's flat trial
    // passes it wins there too -- 440.7 / 440.8 against 402-422 EM it/s, alternating runs on one box (profiles/r05_mstep_ab.txt): a restart
    // group's EM period is its sweeps plus its OWN M-step chain (the other group's sweeps hide it only while it is the shorter of the two).
    if ((b->opt[RMX_OPT_SEARCH_MODE] == 5 || b->opt[RMX_OPT_SEARCH_MODE] == 7) && maxcnt <= NM_MAX_SAMPLE) {
        if (!b->d_nm_state) {
            if ((rc = dalloc(b, &b->d_nm_state, (size_t)64)) ||
                (rc = dalloc(b, &b->nm_lay.pre, (size_t)64 * (NM_MAX_SAMPLE + 1))) || (rc = dalloc(b, &b->nm_lay.fix, (size_t)64 * NM_MAX_SAMPLE)) ||
                (rc = dalloc(b, &b->nm_lay.k1, (size_t)64 * NM_MAX_SAMPLE))) return rc;
        }
        NmArgs na;
        memset(&na, 0, sizeof(na));
        for (int j = 0; j < nparams; j++) {
            na.lo[j] = lo[j]; na.hi[j] = hi[j]; na.maskbit[j] = mv.maskbit[j];
            for (int g = 0; g < G; g++) { na.gv[j][g] = mv.gv[j][g]; na.glv[j][g] = mv.glv[j][g]; }
        }
        for (int q = 0; q < 64; q++) { na.rlist[q] = mv.rlist[q < Q ? q : 0]; na.slot[q] = mv.slot[q < Q ? q : 0]; }
        na.G = G; na.nreq = Q;
        // the requests' cells (one wait: the rounds' grid holds exactly the blocks that have cells)
        double *cells = b->h_pinned + 2 * 64;
        {
            std::lock_guard<std::mutex> lk(b->mu);
            ProfScope ps(b, KID_ELL_LIST);
            hipLaunchKernelGGL(k_search_setup, dim3(Q), dim3(256), 0, b->stream, b->d, na, (const int32_t *)b->d_msample, (const int32_t *)b->d_mcounts, b->nm_lay, cells);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(b->stream));
        for (int q = 0; q < Q; q++) na.blk0[q + 1] = na.blk0[q] + ((int)cells[q] + 255) / 256;
        for (int q = Q; q < 64; q++) na.blk0[q + 1] = na.blk0[Q];
        const int TB = std::max(na.blk0[Q], 1);
        uint32_t *done = b->h_err;                  // [Q]: 0 while the request's optimiser runs, then 1 + the restart's error word
        // search_mode 7: the whole search as ONE launch of resident blocks (k_search_persist) -- where every request has cells
        // (a block waits only for the blocks of its own request -- consecutive indices, dispatched in order --, so what must be resident together is one request's blocks)
        bool persist = b->opt[RMX_OPT_SEARCH_MODE] == 7 && na.blk0[Q] >= 1;
        for (int q = 0; q < Q && persist; q++) persist = na.blk0[q + 1] > na.blk0[q] && na.blk0[q + 1] - na.blk0[q] <= 128;
        b->last_search_blocks = na.blk0[Q]; b->last_search_persist = persist ? 1 : 0;
        if (persist) {
            const size_t need = (size_t)(G + Nm1::maxfun + 2) * TB;
            if (b->nm_partial_cap < need) {
                dfree(b, b->d_nm_partial); b->d_nm_partial = nullptr; b->nm_partial_cap = 0;
                if ((rc = dalloc(b, &b->d_nm_partial, need))) return rc;
                b->nm_partial_cap = need;
            }
            na.grid_stage = 0;
            {
                std::lock_guard<std::mutex> lk(b->mu);
                ProfScope ps(b, KID_ELL_LIST);
                HIPCHK(hipMemsetAsync(b->d_nm_partial, 0xff, need * 8, b->stream));      // "not yet published"
                hipLaunchKernelGGL(k_search_persist, dim3(TB), dim3(256), 0, b->stream, b->d, na, (const int32_t *)b->d_msample, (const int32_t *)b->d_mcounts,
                                   b->nm_lay, b->d_nm_partial, b->h_pinned);
                hipLaunchKernelGGL(k_search_flags, dim3(1), dim3(64), 0, b->stream, b->d, na, done);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipStreamSynchronize(b->stream));
            for (int q = 0; q < Q; q++) done[q] -= 1u;
            if (int rc_ = report_request_errors(b, Q, done, [&](int q) { return (int)mv.rlist[q]; })) return rc_;
            for (int q = 0; q < Q; q++) { xopt[q] = b->h_pinned[2 * q]; lastval[q] = b->h_pinned[2 * q + 1]; }
            return RMX_OK;
        }
        const size_t pneed = (size_t)G * TB;
        if (b->nm_partial_cap < pneed) {
            dfree(b, b->d_nm_partial); b->d_nm_partial = nullptr; b->nm_partial_cap = 0;
            if ((rc = dalloc(b, &b->d_nm_partial, pneed * 2))) return rc;
            b->nm_partial_cap = pneed * 2;
        }
        for (int q = 0; q < Q; q++) done[q] = 0;
        int queued = 0;
        bool all_done = false;
        while (!all_done) {
            const int batch = queued == 0 ? 52 : 12;      // (the slowest of 32 optimisers needs about 50 evaluations; a round too many is a kernel pair of blocks that return)
            {
                std::lock_guard<std::mutex> lk(b->mu);
                ProfScope ps(b, KID_ELL_LIST);
                if (queued == 0) {
                    na.grid_stage = 1;
                    hipLaunchKernelGGL(k_search_round, dim3(TB, G), dim3(256), 0, b->stream, b->d, na, (const int32_t *)b->d_msample, (const int32_t *)b->d_mcounts,
                                       b->nm_lay, b->d_nm_partial, (const NmState *)b->d_nm_state);
                    hipLaunchKernelGGL(k_search_advance, dim3(Q), dim3(256), 0, b->stream, b->d, na, (const double *)b->d_nm_partial, b->d_nm_state, b->h_pinned, done);
                    na.grid_stage = 0;
                }
                for (int k = 0; k < batch; k++) {
                    hipLaunchKernelGGL(k_search_round, dim3(TB, 1), dim3(256), 0, b->stream, b->d, na, (const int32_t *)b->d_msample, (const int32_t *)b->d_mcounts,
                                       b->nm_lay, b->d_nm_partial, (const NmState *)b->d_nm_state);
                    hipLaunchKernelGGL(k_search_advance, dim3(Q), dim3(256), 0, b->stream, b->d, na, (const double *)b->d_nm_partial, b->d_nm_state, b->h_pinned, done);
                }
                HIPCHK(hipGetLastError());
            }
            queued += batch;
            HIPCHK(hipStreamSynchronize(b->stream));
            all_done = true;
            for (int q = 0; q < Q; q++) all_done = all_done && done[q] != 0;
            if (queued > Nm1::maxfun + 16) return fail(RMX_EDEVICE, "rmx_param_search_multi: an optimiser did not finish");      // (cannot happen: Nm1 stops at maxfun evaluations)
        }
        for (int q = 0; q < Q; q++) done[q] -= 1u;
        if (int rc_ = report_request_errors(b, Q, done, [&](int q) { return (int)mv.rlist[q]; })) return rc_;
        for (int q = 0; q < Q; q++) { xopt[q] = b->h_pinned[2 * q]; lastval[q] = b->h_pinned[2 * q + 1]; }
        return RMX_OK;
    }
    if ((rc = round(Q, all.data(), nullptr, true))) return rc;
    std::vector<double> x0(Q), best(Q, INFINITY);
    for (int q = 0; q < Q; q++) {
        const int j = q / nreq;
        for (int g = 0; g < G; g++) { const double J = -out[(size_t)q * G + g]; if (g == 0 || J < best[q]) { best[q] = J; x0[q] = grids[(size_t)j * G + g]; } }   // np.argmin: first minimum
        lastval[q] = grids[(size_t)j * G + (G - 1)];
    }
    std::vector<Nm1> nm(Q);
    std::vector<int> want;
    auto pump = [&](int q, double f) {
        const int j = q / nreq;
        while (nm[q].advance(x0[q], f)) {
            const double v = nm[q].req;
            if (v < lo[j] || v > hi[j]) { f = INFINITY; continue; }      // +inf outside the bounds, nothing evaluated (cn_model.py:542-543)
            want.push_back(q);
            return;
        }
    };
    for (int q = 0; q < Q; q++) pump(q, 0.);
    std::vector<double> vals(Q);
    while (!want.empty()) {
        std::vector<int> cur;
        cur.swap(want);
        for (size_t k = 0; k < cur.size(); k++) { vals[k] = nm[cur[k]].req; lastval[cur[k]] = vals[k]; }
        if ((rc = round((int)cur.size(), cur.data(), vals.data(), false))) return rc;
        for (size_t k = 0; k < cur.size(); k++) pump(cur[k], -out[k]);
    }
    for (int q = 0; q < Q; q++) xopt[q] = nm[q].xopt();
    return RMX_OK;
}

// One candidate haploid-depth vector per listed restart: E[ll] and dE[ll]/dh on each restart's
// current sample (the objective / gradient pair of BreakpointModel.update_h, cn_model.py:484-498),
// the evaluation round of a lock-step L-BFGS-B.  h [nreq][M]; out [nreq][1 + RMX_MAX_CLONES].
int rmx_expected_ll_h_batch(rmx_batch *b, int32_t nreq, const int32_t *restarts, const double *h, double *out) { BIND(b);
    if (!b || nreq < 1 || nreq > b->R || !restarts || !h || !out) return fail(RMX_EARG, "bad argument");
    int rc = check_request_list(b, nreq, restarts);
    if (rc) return rc;
    for (int i = 0; i < nreq; i++)
        if ((rc = rmx_set_array(b, restarts[i], RMX_A_H, h + (size_t)i * b->d.M))) return rc;
    return run_ell_batch(b, nreq, restarts, true, out);
}

// Full-data E[ll] (sample of all ones, :1125-1157) for a restart range from the per-segment
// expectations (refreshed first if h / a parameter changed).
int rmx_expected_ll_full(rmx_batch *b, int32_t r0, int32_t r1, double *out) { BIND(b);
    RANGE_CHECK();
    int rc = ensure_ab(b, r0, r1);
    if (rc) return rc;
    const int nr = r1 - r0;
    { ProfScope ps(b, KID_ELL_FULL); hipLaunchKernelGGL(k_ell_full_batch, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, b->d, r0, b->d_partial); }
    { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_sum_partials, dim3(nr), dim3(256), 0, b->stream, (const double *)b->d_partial, ELBO_BLOCKS, b->d_out4); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(b->h_pinned, b->d_out4, (size_t)nr * 8, hipMemcpyDeviceToHost, b->stream));
    if ((rc = check_errors(b, r0, r1))) return rc;
    for (int i = 0; i < nr; i++) out[i] = b->h_pinned[i];
    return RMX_OK;
}

// Full-data E[ll] at parameter values that are on TRIAL: h / likelihood parameters were changed (the
// M-step's accept test `ell_after < ell_before`, cn_model.py:497-505, 563-569) but may be rolled back.
// The stale components of (A, B) are evaluated into a scratch copy -- the same values, reduced in the
// same order, as a refresh would produce -- while the restart's own (A, B), its cell cache and its
// staleness flags stay as they are.  A rejected value then costs no second pass over the cells:
// rmx_trial_rollback puts the old value back and declares (A, B) / cache current again.
// the stale components of (A, B) of restarts [r0, r1) at their current (trial) parameter values into the scratch copy d2
static int trial_pass(rmx_batch *b, int r0, int r1, Dev &d2) {
    b->trial_comp_r0 = b->trial_comp_r1 = -1;      // the scratch expectations are about to change
    b->scratch_r0 = b->scratch_r1 = -1;
    int rc = ensure_tables(b, r0, r1);
    if (rc) return rc;
    const Dev &d = b->d;
    const int nr = r1 - r0;
    const size_t RN0 = (size_t)r0 * d.N, cnt = (size_t)nr * d.N;
    HIPCHK(hipMemcpyAsync(b->d_A2 + RN0 * 2, d.A + RN0 * 2, cnt * 16, hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->d_Bv2 + RN0 * 4, d.Bv + RN0 * 4, cnt * 32, hipMemcpyDeviceToDevice, b->stream));
    d2 = b->d;
    d2.A = b->d_A2; d2.Bv = b->d_Bv2; d2.lc = nullptr;
    for (int r = r0; r < r1;) {
        const int mask = use_strip(b) ? cover_mask(b->comp_dirty[r] & 15) : (b->comp_dirty[r] ? 15 : 0);
        int e = r + 1;
        while (e < r1 && (use_strip(b) ? cover_mask(b->comp_dirty[e] & 15) : (b->comp_dirty[e] ? 15 : 0)) == mask) e++;
        if (mask) {
            ProfScope ps(b, KID_MARGINALS_AB);
            bool sparse = use_strip(b) && d.sig_cnt != nullptr;
            for (int i = r; i < e; i++) if (!b->sig_valid[i]) sparse = false;
            if (sparse) {
                void (*kf)(Dev, int) = nullptr;
                if (b->opt[RMX_OPT_TRIAL_KERNEL] == 0) {      // cells flat over the threads, segmented sum in list order (round 5)
                    switch (mask) {
                    case 1: kf = k_trial_flat<1>; break; case 2: kf = k_trial_flat<2>; break; case 3: kf = k_trial_flat<3>; break;
                    case 4: kf = k_trial_flat<4>; break; case 8: kf = k_trial_flat<8>; break; case 12: kf = k_trial_flat<12>; break;
                    default: kf = k_trial_flat<15>; break;
                    }
                    hipLaunchKernelGGL(kf, dim3((d.N + TF_SEGB - 1) / TF_SEGB, e - r), dim3(256), 0, b->stream, d2, r);
                } else {
                    switch (mask) {
                    case 1: kf = k_trial_sparse<1>; break; case 2: kf = k_trial_sparse<2>; break; case 3: kf = k_trial_sparse<3>; break;
                    case 4: kf = k_trial_sparse<4>; break; case 8: kf = k_trial_sparse<8>; break; case 12: kf = k_trial_sparse<12>; break;
                    default: kf = k_trial_sparse<15>; break;
                    }
                    hipLaunchKernelGGL(kf, dim3((d.N + TRIAL_SEG_PER_BLOCK - 1) / TRIAL_SEG_PER_BLOCK, e - r), dim3(256), 0, b->stream, d2, r);
                }
            }
            else if (use_strip(b)) hipLaunchKernelGGL(cells_kernel(b, 2, mask, 0), strip_grid(b, e - r), dim3(256), 0, b->stream, d2, r);
            else hipLaunchKernelGGL(k_marginals<false>, row_grid(b, e - r), dim3(256), 0, b->stream, d2, r, b->G);
        }
        r = e;
    }
    b->scratch_r0 = r0; b->scratch_r1 = r1;
    return RMX_OK;
}
int rmx_expected_ll_full_trial(rmx_batch *b, int32_t r0, int32_t r1, double *out) { BIND(b);
    RANGE_CHECK();
    Dev d2;
    int rc = trial_pass(b, r0, r1, d2);
    if (rc) return rc;
    const int nr = r1 - r0;
    // ... and, behind it in the same queue and the same transfer, the four component sums of these scratch expectations: where the h M-step
    // keeps its trial h they are the parameter M-steps' "before" values (rmx_expected_ll_components trial = 2), served from the host copy
    double *cpart = b->d_partial + (size_t)b->R * ELBO_BLOCKS, *cout = b->d_out4 + b->R;
    b->trial_comp_r0 = b->trial_comp_r1 = -1;
    { ProfScope ps(b, KID_ELL_FULL); hipLaunchKernelGGL(k_ell_full_batch, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, d2, r0, b->d_partial); }
    { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_sum_partials, dim3(nr), dim3(256), 0, b->stream, (const double *)b->d_partial, ELBO_BLOCKS, b->d_out4); }
    { ProfScope ps(b, KID_ELL_FULL); hipLaunchKernelGGL(k_ell_comp_batch, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, d2, r0, cpart); }
    { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_sum_partials, dim3(nr * 4), dim3(256), 0, b->stream, (const double *)cpart, ELBO_BLOCKS, cout); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(b->h_pinned, b->d_out4, (size_t)nr * 8, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(b->h_pinned + nr, cout, (size_t)nr * 32, hipMemcpyDeviceToHost, b->stream));
    if ((rc = check_errors(b, r0, r1))) return rc;
    for (int i = 0; i < nr; i++) out[i] = b->h_pinned[i];
    b->trial_comp.assign(b->h_pinned + nr, b->h_pinned + nr + (size_t)nr * 4);
    b->trial_comp_r0 = r0; b->trial_comp_r1 = r1;
    return RMX_OK;
}
// The full-data E[ll] of restarts [r0, r1) split into its four likelihood components (out[i][c]: NB total with u = 0 / 1, BB
// allele with v = 0 / 1 -- the parts negbin_r_0, negbin_r_1, betabin_M_0, betabin_M_1 move).  trial = 0: at the current
// values (like rmx_expected_ll_full); trial = 1: with the changed parameters on trial (like rmx_expected_ll_full_trial).
// With it the accept tests of the four standard parameters (cn_model.py:563-569, one after the other) need one pass over
// the cells instead of four: their components are disjoint, so E[ll] with parameter j on trial and the earlier ones
// decided is a sum of component values from the two calls.
int rmx_expected_ll_components(rmx_batch *b, int32_t r0, int32_t r1, int32_t trial, double *out) { BIND(b);
    RANGE_CHECK();
    if (!out) return fail(RMX_EARG, "bad argument");
    int rc;
    Dev d2 = b->d;
    if (trial == 2 && b->trial_comp_r0 == r0 && b->trial_comp_r1 == r1) {      // summed behind that trial pass already
        for (int i = 0; i < (r1 - r0) * 4; i++) out[i] = b->trial_comp[i];
        return RMX_OK;
    }
    if (trial == 3) {
        // per restart: the scratch expectations of the last rmx_expected_ll_full_trial over this range where the restart's own are stale (it
        // kept the trial h), its own where they are current (it was rolled back, or nothing changed) -- no pass over the cells for either
        if (b->trial_comp_r0 != r0 || b->trial_comp_r1 != r1) return fail(RMX_EUNSUPPORTED, "no trial pass over this range to take the expectations from");
        const int nr = r1 - r0;
        bool own = false;
        // (stale = some component is: ab_dirty alone is also raised by a table rebuild that changes no value, e.g. after a rollback)
        for (int r = r0; r < r1; r++) own |= b->comp_dirty[r] == 0;
        if (own) {
            { ProfScope ps(b, KID_ELL_FULL); hipLaunchKernelGGL(k_ell_comp_batch, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, d2, r0, b->d_partial); }
            { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_sum_partials, dim3(nr * 4), dim3(256), 0, b->stream, (const double *)b->d_partial, ELBO_BLOCKS, b->d_out4); }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(b->h_pinned, b->d_out4, (size_t)nr * 32, hipMemcpyDeviceToHost, b->stream));
            if ((rc = check_errors(b, r0, r1))) return rc;
        }
        for (int r = r0; r < r1; r++)
            for (int c = 0; c < 4; c++) out[(r - r0) * 4 + c] = b->comp_dirty[r] != 0 ? b->trial_comp[(size_t)(r - r0) * 4 + c] : b->h_pinned[(r - r0) * 4 + c];
        return RMX_OK;
    }
    if (trial == 2) {      // the scratch expectations of the last trial pass over this range, as they are
        if (b->scratch_r0 != r0 || b->scratch_r1 != r1) return fail(RMX_EUNSUPPORTED, "no trial pass over this range to take the expectations from");
        d2.A = b->d_A2; d2.Bv = b->d_Bv2;
    }
    else if (trial) { if ((rc = trial_pass(b, r0, r1, d2))) return rc; }
    else if ((rc = ensure_ab(b, r0, r1))) return rc;
    const int nr = r1 - r0;
    { ProfScope ps(b, KID_ELL_FULL); hipLaunchKernelGGL(k_ell_comp_batch, dim3(ELBO_BLOCKS, nr), dim3(256), 0, b->stream, d2, r0, b->d_partial); }
    { ProfScope ps(b, KID_ELL_FINAL); hipLaunchKernelGGL(k_sum_partials, dim3(nr * 4), dim3(256), 0, b->stream, (const double *)b->d_partial, ELBO_BLOCKS, b->d_out4); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(b->h_pinned, b->d_out4, (size_t)nr * 32, hipMemcpyDeviceToHost, b->stream));
    if ((rc = check_errors(b, r0, r1))) return rc;
    for (int i = 0; i < nr * 4; i++) out[i] = b->h_pinned[i];
    return RMX_OK;
}
// Undo a trial: param_id >= 0 puts likelihood parameter param_id of restart r back to values[0];
// param_id < 0 puts h back to values[0..M).  Only valid when (A, B) and the cell cache were current for
// exactly these values before the trial and nothing refreshed them since (no coordinate update, ELBO or
// rmx_expected_ll_full in between) -- which is the M-step's sequence: full E[ll], search on samples,
// rmx_expected_ll_full_trial, then accept (set the new value) or this.
int rmx_trial_rollback(rmx_batch *b, int32_t r, int32_t param_id, const double *values) { BIND(b);
    if (!b || r < 0 || r >= b->R || !values || param_id >= RMX_P_HMM_LOG_NORM_CONST) return fail(RMX_EARG, "bad argument");
    if (param_id >= 0) b->rp[r].p[param_id] = param_id == RMX_P_DIVERGENCE_WEIGHT ? std::fabs(values[0]) : values[0];
    else for (int m = 0; m < b->d.M; m++) b->rp[r].h[m] = values[m];
    b->tables_dirty[r] = 1; b->segc_dirty[r] = 1;      // the device tables hold the trial values
    if (param_id >= 0 && param_id < RMX_P_HMM_LOG_NORM_CONST) {
        // only this parameter's components become current again (others may be on trial at the same time: rmx_expected_ll_components)
        // (... unless they were stale already: an accepted h whose refresh is still pending, comp_base)
        const int back = kParamComponents[param_id] & ~b->comp_base[r];
        b->comp_dirty[r] &= ~back; b->cache_stale[r] &= ~(back & 15);
        b->ab_dirty[r] = b->comp_dirty[r] != 0;
    } else { b->ab_dirty[r] = 0; b->comp_dirty[r] = 0; b->comp_base[r] = 0; b->cache_stale[r] = 0; }
    return RMX_OK;
}

static int cell_probe(rmx_batch *b, int r, int n, int s, double out6[6]) {
    if (r < 0 || r >= b->R || n < 0 || n >= b->d.N || s < 0 || s >= b->d.S) return fail(RMX_EARG, "index out of range");
    int rc = ensure_tables(b, r, r + 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cell_probe, dim3(1), dim3(1), 0, b->stream, b->d, r, n, s, b->d_ell_out + (size_t)r * 8);
    HIPCHK(hipMemcpyAsync(b->h_pinned, b->d_ell_out + (size_t)r * 8, 48, hipMemcpyDeviceToHost, b->stream));
    if ((rc = check_errors(b, r, r + 1))) return rc;
    for (int i = 0; i < 6; i++) out6[i] = b->h_pinned[i];
    return RMX_OK;
}
int rmx_log_likelihood_total(rmx_batch *b, int32_t r, int32_t n, int32_t s, int32_t u, double *out) { BIND(b);
    double o[6]; int rc = cell_probe(b, r, n, s, o); if (rc) return rc; *out = o[u ? 1 : 0]; return RMX_OK;
}
int rmx_log_likelihood_allele(rmx_batch *b, int32_t r, int32_t n, int32_t s, int32_t v, int32_t w, double *out) { BIND(b);
    double o[6]; int rc = cell_probe(b, r, n, s, o); if (rc) return rc; *out = o[2 + (v ? 2 : 0) + (w ? 1 : 0)]; return RMX_OK;
}
int rmx_cell_quantity(rmx_batch *b, int32_t r, int32_t n, int32_t s, int32_t which, int32_t u, int32_t v, int32_t w, double *out) { BIND(b);
    if (!b || !out || r < 0 || r >= b->R || n < 0 || n >= b->d.N || s < 0 || s >= b->d.S || which < 0 || which > 6) return fail(RMX_EARG, "index out of range");
    static const int want[7] = {1, 1, 2, 2, 4, 8, 16}, off[7] = {0, 1, 5, 6, 10, 11, 15}, cnt_m[7] = {0, 1, 0, 1, 0, 1, 1};
    int rc = ensure_tables(b, r, r + 1, false);      // publishes h / the parameters to the device copy the probe reads
    if (rc) return rc;
    double *dst = b->d_grid_out + (size_t)r * 64 * (1 + RMX_MAX_CLONES);      // the restart's scratch row
    hipLaunchKernelGGL(k_cell_probe_h, dim3(1), dim3(1), 0, b->stream, b->d, r, n, s, u ? 1 : 0, v ? 1 : 0, w ? 1 : 0, want[which], dst);
    HIPCHK(hipMemcpyAsync(b->h_pinned, dst, 19 * 8, hipMemcpyDeviceToHost, b->stream));
    if ((rc = check_errors(b, r, r + 1))) return rc;
    const int cnt = cnt_m[which] ? b->d.M : 1;
    for (int i = 0; i < cnt; i++) out[i] = b->h_pinned[off[which] + i];
    return RMX_OK;
}

// ---- decoding -----------------------------------------------------------------------------
static int viterbi_P(int S) { int P = 1; while (S * P * 2 <= 1024 && P < 64) P *= 2; return P; }
static int viterbi_reg_P(int S) { int P = 1; while (S * P * 2 <= 768 && P < 64) P *= 2; return P; }

// Viterbi paths of restarts r0 .. r0+nr-1: forward lattices side by side (one workgroup each), then the trace-backs
static std::atomic<int> g_cluster_wgs[16];      // workgroups of lattice clusters (k_viterbi_sad_max<., true>) in flight per device
struct ClusterHold { std::atomic<int> *c = nullptr; int n = 0; ~ClusterHold() { if (c && n) c->fetch_sub(n); } };
static int viterbi_paths(rmx_batch *b, int r0, int nr, std::vector<int64_t> &paths, std::vector<double> &lps, int model, bool allow_cluster = true) {
    ClusterHold hold;      // (released when this call returns: it ends with a stream synchronisation)
    bool cluster_launched = false;
    // the lattice runs on the log_transmat snapshot: plain tables of the transition model it was taken under
    const Dev dv = dev_for_model(b, model);
    const bool cur_model = model == b->d.tmodel;
    const Dev &d = b->d;
    const int N = d.N, S = d.S, M = d.M;
    int rc;
    const int Pr = viterbi_reg_P(S), QPT = (S + Pr - 1) / Pr;
    const int QPT4 = ((QPT + 3) / 4) * 4;
    const int vopt = b->opt[RMX_OPT_VITERBI_PLAIN];      // 0: maxima forward + arg-maxima in the trace-back (round 5); 1: k_viterbi; 2: round 4's back-pointer lattices
    const bool reg = QPT <= 44 && vopt != 1;
    // code-table lattice (k_viterbi_code): V rows, breakend table, value table, S rows of P * QPT4 codes
    const size_t code_lds = b->vit_code_ok ? (size_t)(2 * (Pr * QPT4 + 4) + ((M * d.D + 1) & ~1) + 256) * 8 + (size_t)S * Pr * QPT4 : (size_t)1 << 30;
    const bool coded = !reg && cur_model && code_lds <= kLdsBudget && vopt != 1;
    const bool maxima = vopt == 0 && (reg || coded);
    const int SR = (S + 3) & ~3;
    // above 176 states (round 5): transition values of class 0 from the packed copies (k_fbk's closed form, verified at table build), workgroup clusters
    const int ca0 = b->tc_pairs.empty() ? 0 : b->tc_pairs[0].first, cb0 = b->tc_pairs.empty() ? 0 : b->tc_pairs[0].second;
    const int vcl = b->opt[RMX_OPT_VITERBI_CLUSTER];
    // (workgroups per restart the launch would get: with one -- the option, or too many restarts for clusters of two -- a grid whose codes fit keeps the code-table lattice)
    int Wcl = 1;
    const bool stall_test = vcl >= 100;      // (test hook: viterbi_cluster = 100 + W runs clusters of W with one member that never publishes its first row)
    if (vcl != 1 && allow_cluster) { const int want = vcl >= 100 ? vcl - 100 : vcl; Wcl = want >= 2 ? want : (S > 300 ? 8 : 4); while (Wcl > 1 && (long)nr * Wcl > 64) Wcl /= 2; }
    const bool sadmax = vopt == 0 && !reg && (!coded || Wcl > 1) && cur_model && b->fbk_ok && !b->tc_pairs.empty() && ca0 == cb0 && S <= 1024;
    if (maxima && reg && b->n_vit_special < 0) {
        std::vector<int32_t> sp;
        for (int n = 0; n + 1 < N; n++) if (!(b->tclass[n] == 0 && b->brk_slot[n] < 0)) sp.push_back(n);
        if ((rc = dalloc(b, &b->d_vit_special, std::max<size_t>(sp.size(), 1)))) return rc;
        if (!sp.empty()) HIPCHK(hipMemcpyAsync(b->d_vit_special, sp.data(), sp.size() * 4, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));      // (sp is a local)
        b->n_vit_special = (int)sp.size();
    }
    // (the list lives in the workgroup's LDS next to two lattice rows: a dataset with more special adjacencies than fit takes round 4's kernel)
    // breakend steps inside transition class 0 from LDS tables (totals, allele-flip bytes): the current model's tables only (d.ab has no per-model copy)
    const bool be_tab = cur_model && !b->tc_pairs.empty() && M <= 4;
    const bool reg_max = maxima && reg && (size_t)(2 * (Pr * ((QPT + 1) & ~1) + 44) + M * d.D + 2) * 8 + (size_t)b->n_vit_special * 4 + (be_tab ? (size_t)S * 8 + (size_t)S * SR : 0) + 64 <= kLdsBudget;
    if (b->vit_cap < nr) {
        dfree(b, b->d_final); dfree(b, b->d_path); dfree(b, b->d_logprob);
        b->d_final = nullptr; b->d_path = nullptr; b->d_logprob = nullptr; b->vit_cap = 0;
        if ((rc = dalloc(b, &b->d_final, (size_t)nr * S)) || (rc = dalloc(b, &b->d_path, (size_t)nr * N)) || (rc = dalloc(b, &b->d_logprob, nr))) return rc;
        b->vit_cap = nr;
    }
    const bool lattice_rows = (maxima && (reg_max || coded)) || sadmax;      // rows of the lattice kept (trace-back recomputes the arg-maxima) instead of back-pointers
    if (lattice_rows && b->vrow_cap < (size_t)nr * N * SR) {
        dfree(b, b->d_vrow); b->d_vrow = nullptr; b->vrow_cap = 0;
        if ((rc = dalloc(b, &b->d_vrow, (size_t)nr * N * SR))) return rc;
        b->vrow_cap = (size_t)nr * N * SR;
        HIPCHK(hipMemsetAsync(b->d_vrow, 0, b->vrow_cap * 8, b->stream));      // (the pads of a row stay 0: code 255 = -inf makes them lose every comparison)
    }
    // the trace-back in parallel (k_bp_all + k_chase_*): the lattice rows of this call, transition values of class 0 from the packed copies
    const bool par_tb = lattice_rows && vopt == 0 && b->opt[RMX_OPT_TRACEBACK] == 0 && cur_model && b->fbk_ok && !b->tc_pairs.empty() && ca0 == cb0 && N >= 2 && S <= 1024;
    if ((!lattice_rows || par_tb) && b->bp_cap < (size_t)nr * N * S) {
        dfree(b, b->d_bp); b->d_bp = nullptr; b->bp_cap = 0;
        if ((rc = dalloc(b, &b->d_bp, (size_t)nr * N * S))) return rc;
        b->bp_cap = (size_t)nr * N * S;
    }
    { ProfScope ps(b, KID_VITERBI);
      b->last_viterbi = sadmax ? 6 : (reg ? (reg_max ? 4 : 1) : (coded ? (maxima ? 5 : 2) : 3));
      b->last_viterbi_wgs = 1;
      if (sadmax) {
          // W workgroups per restart (each a share OW of the target states, the rows exchanged through memory step by step); all of them must be resident
          // at once: at most 64 per launch (restart groups decode next to each other) and 192 per device
          int W = Wcl;      // (measured, profiles/r05_large_grids.txt: a step's exchange costs 1.5-2 us; 251 states are fastest with 4 workgroups, 355 and above with 8)
          {
              if (W > 1) {      // every cluster workgroup in flight on the device must be resident: a process-wide count, W = 1 when it would pass 192 of the 256 CUs
                  const int want = nr * W, had = g_cluster_wgs[b->device & 15].fetch_add(want);
                  if (had + want > 192) { g_cluster_wgs[b->device & 15].fetch_sub(want); W = 1; } else { hold.c = &g_cluster_wgs[b->device & 15]; hold.n = want; }
              }
          }
          const int OW = W == 1 ? S : (((S + W - 1) / W + 63) / 64) * 64;
          const int SO = ((OW + 63) / 64) * 64, P = std::max(1, 1024 / SO);
          const int SQ = ((S + 4 * P - 1) / (4 * P)) * 4, SV = P * SQ;
          const size_t lds_ = (size_t)(2 * SV + (P > 1 ? P * SO : 0) + ((M * d.D + 1) & ~1)) * 8 + (size_t)(M == 4 ? 2 : 1) * SV * 4 + 16;
          auto kfs = W > 1 ? (M == 4 ? k_viterbi_sad_max<true, true> : k_viterbi_sad_max<false, true>) : (M == 4 ? k_viterbi_sad_max<true, false> : k_viterbi_sad_max<false, false>);
          if (!b->d_vflag) { if ((rc = dalloc(b, &b->d_vflag, (size_t)1))) return rc; }
          HIPCHK(hipMemsetAsync(b->d_vflag, 0, 4, b->stream));
          if (W > 1) HIPCHK(hipMemsetAsync(b->d_vrow, 0xff, (size_t)nr * N * SR * 8, b->stream));      // "not yet written" for the row exchange (k_viterbi_sad_max<., true>)
          b->last_viterbi_wgs = W;
          HIPCHK(hipFuncSetAttribute((const void *)kfs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));
          hipLaunchKernelGGL(kfs, W > 1 ? dim3(8 * W * ((nr + 7) / 8)) : dim3(nr), dim3(P * SO), lds_, b->stream, b->d, r0, P, SO, OW, SR, b->d_vrow, (const uint32_t *)b->d_cnpack, (const uint32_t *)b->d_cnpack2, -d.pen, ca0, W, nr, b->d_vflag, (stall_test && W > 1) ? 1 : 0);
          cluster_launched = W > 1;
      } else if (reg) {
          const int NT = ((S * Pr + 63) / 64) * 64;
#define VREG(Q) { if (reg_max) { const size_t lds_ = (size_t)(2 * (Pr * ((QPT + 1) & ~1) + Q) + ((M * d.D + 1) & ~1)) * 8 + (size_t)b->n_vit_special * 4 + (be_tab ? (size_t)S * 8 + (size_t)S * SR : 0) + 16; \
                    HIPCHK(hipFuncSetAttribute((const void *)k_viterbi_max<Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_)); \
                    hipLaunchKernelGGL(k_viterbi_max<Q>, dim3(nr), dim3(NT), lds_, b->stream, dv, r0, Pr, SR, b->d_vrow, (const int32_t *)b->d_vit_special, b->n_vit_special, be_tab ? 1 : 0, ca0, cb0, b->d_dbg); } \
                  else hipLaunchKernelGGL(k_viterbi_reg<Q>, dim3(nr), dim3(NT), (size_t)(2 * (Pr * QPT + Q) + M * d.D) * 8, b->stream, dv, r0, Pr, b->d_bp, b->d_final); }
          if (QPT <= 8) VREG(8) else if (QPT <= 16) VREG(16) else if (QPT <= 24) VREG(24) else if (QPT <= 32) VREG(32)
          else if (QPT <= 36) VREG(36) else if (QPT <= 40) VREG(40) else VREG(44)
#undef VREG
      } else if (coded) {
          const int NT = ((S * Pr + 63) / 64) * 64;
          if (maxima) {
              auto kfc = b->vit_mul_ok ? k_viterbi_code_max<true> : k_viterbi_code_max<false>;
              HIPCHK(hipFuncSetAttribute((const void *)kfc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)code_lds));
              hipLaunchKernelGGL(kfc, dim3(nr), dim3(NT), code_lds, b->stream, b->d, r0, Pr, QPT4,
                                 (const uint8_t *)b->d_vit_code, (const double *)b->d_vit_val, SR, b->d_vrow, b->vit_mul_ok ? -d.pen : 0.);
          } else {
              HIPCHK(hipFuncSetAttribute((const void *)k_viterbi_code, hipFuncAttributeMaxDynamicSharedMemorySize, (int)code_lds));
              hipLaunchKernelGGL(k_viterbi_code, dim3(nr), dim3(NT), code_lds, b->stream, b->d, r0, Pr, QPT4,
                                 (const uint8_t *)b->d_vit_code, (const double *)b->d_vit_val, b->d_bp, b->d_final);
          }
      } else {
          const int P = viterbi_P(S), NT = ((S * P + 63) / 64) * 64;
          hipLaunchKernelGGL(k_viterbi, dim3(nr), dim3(NT), (size_t)(2 * S + M * d.D) * 8, b->stream, dv, r0, P, b->d_bp, b->d_final);
      } }
    b->last_traceback = par_tb ? 1 : 0;
    if (par_tb) {
        const int SO = ((S + 63) / 64) * 64, P = std::max(1, 1024 / SO);
        const int NB = P * (P >= 4 ? 2 : (P == 1 ? 8 : 4));
        const int B = std::max(8, std::min(128, (96 * 1024) / (2 * S)));
        const int NBLK = (N - 1 + B - 1) / B;
        if (b->comp_cap < (size_t)nr * NBLK * S) { dfree(b, b->d_comp); b->d_comp = nullptr; b->comp_cap = 0; if ((rc = dalloc(b, &b->d_comp, (size_t)nr * NBLK * S))) return rc; b->comp_cap = (size_t)nr * NBLK * S; }
        if (b->ends_cap < (size_t)nr * NBLK) { dfree(b, b->d_ends); b->d_ends = nullptr; b->ends_cap = 0; if ((rc = dalloc(b, &b->d_ends, (size_t)nr * NBLK))) return rc; b->ends_cap = (size_t)nr * NBLK; }
        ProfScope ps(b, KID_BACKTRACE);
        const size_t lds_a = (size_t)NB * SR * 8 + (size_t)(M == 4 ? 2 : 1) * SR * 4 + 16, lds_c = (size_t)B * S * 2 + 16;
        auto ka = M == 4 ? k_bp_all<true> : k_bp_all<false>;
        HIPCHK(hipFuncSetAttribute((const void *)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
        hipLaunchKernelGGL(ka, dim3((N - 1 + NB - 1) / NB, nr), dim3(P * SO), lds_a, b->stream, b->d, r0, P, SO, NB, SR, (const double *)b->d_vrow,
                           (const uint32_t *)b->d_cnpack, (const uint32_t *)b->d_cnpack2, -d.pen, ca0, b->d_bp);
        HIPCHK(hipFuncSetAttribute((const void *)k_chase_compose, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        HIPCHK(hipFuncSetAttribute((const void *)k_chase_fill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        hipLaunchKernelGGL(k_chase_compose, dim3(NBLK, nr), dim3(256), lds_c, b->stream, N, S, B, (const uint16_t *)b->d_bp, b->d_comp);
        hipLaunchKernelGGL(k_chase_ends, dim3(nr), dim3(64), 0, b->stream, N, S, SR, NBLK, (const double *)b->d_vrow, (const uint16_t *)b->d_comp, b->d_ends, b->d_logprob);
        hipLaunchKernelGGL(k_chase_fill, dim3(NBLK, nr), dim3(256), lds_c, b->stream, N, S, B, (const uint16_t *)b->d_bp, (const int32_t *)b->d_ends, b->d_path);
    } else if (sadmax) {
        const int NG = (SR + 255) / 256;
        int rows = (int)((kLdsBudget - (size_t)2 * SR * 4 - 128) / ((size_t)SR * 8 + 4));
        rows = std::max(2, std::min(rows, 96) & ~1);
        const size_t lds = (size_t)rows * SR * 8 + (size_t)rows * 4 + (size_t)(M == 4 ? 2 : 1) * SR * 4 + 16;
        void (*kf)(Dev, int, int, const double *, const uint32_t *, const uint32_t *, double, int, int64_t *, double *, int) =
            M == 4 ? (NG <= 2 ? k_backtrace_sad<2, true> : (NG == 3 ? k_backtrace_sad<3, true> : k_backtrace_sad<4, true>))
                   : (NG <= 2 ? k_backtrace_sad<2, false> : (NG == 3 ? k_backtrace_sad<3, false> : k_backtrace_sad<4, false>));
        ProfScope ps(b, KID_BACKTRACE);
        HIPCHK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kf, dim3(nr), dim3(256), lds, b->stream, b->d, r0, SR, (const double *)b->d_vrow, (const uint32_t *)b->d_cnpack, (const uint32_t *)b->d_cnpack2,
                           -d.pen, ca0, b->d_path, b->d_logprob, rows);
    } else if (lattice_rows) {
        // the trace-back recomputes lattice[n, i] + log_transmat[n, i, state[n+1]] (bpmodel.pyx:1327-1331); class-0 plain adjacencies from the
        // code table of the CURRENT model (a snapshot of the other model goes through trans_value on the snapshot's tables)
        const bool tcode = cur_model && b->vit_code_ok;
        const size_t code_b = tcode ? (((size_t)S * SR + 7) & ~(size_t)7) : 0;
        const size_t be_b = (size_t)((M * d.D + 1) & ~1) * 8 + (size_t)S * 8 + (size_t)S * SR;
        // (the breakend-step tables only where they fit next to the code table and a few lattice rows: 355 states take the plain expression there)
        const bool be_bt = be_tab && (size_t)256 * 8 + code_b + be_b + (size_t)8 * (SR * 8 + 4) + 128 <= kLdsBudget;
        const size_t tabs = code_b + (be_bt ? be_b : 0);
        const size_t fixed = (size_t)256 * 8 + tabs + 64;
        int rows = (int)((kLdsBudget - fixed) / ((size_t)SR * 8 + 4));
        rows = std::max(2, std::min(rows, 96) & ~1);      // (even: the tables behind the per-row ints stay 8-byte aligned)
        const size_t lds = (size_t)rows * SR * 8 + 256 * 8 + (size_t)rows * 4 + tabs + 16;
        const bool mul = tcode && b->vit_mul_ok, two = SR > 256;
        auto kf = mul ? (two ? k_backtrace_max<true, true> : k_backtrace_max<true, false>) : (two ? k_backtrace_max<false, true> : k_backtrace_max<false, false>);
        ProfScope ps(b, KID_BACKTRACE);
        HIPCHK(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kf, dim3(nr), dim3(256), lds, b->stream, dv, r0, SR, (const double *)b->d_vrow, tcode ? (const uint8_t *)b->d_vit_code : nullptr,
                           (const double *)b->d_vit_val, mul ? -d.pen : 0., b->d_path, b->d_logprob, rows, be_bt ? 1 : 0, ca0, cb0);
    } else {
    int rows = std::max(1, std::min(256, (48 * 1024) / (2 * S)));
    { ProfScope ps(b, KID_BACKTRACE);
      hipLaunchKernelGGL(k_backtrace, dim3(nr), dim3(256), (size_t)rows * S * 2, b->stream, b->d, (const uint16_t *)b->d_bp, (const double *)b->d_final, b->d_path, b->d_logprob, rows); }
    }
    HIPCHK(hipGetLastError());
    paths.resize((size_t)nr * N); lps.resize(nr);
    HIPCHK(hipMemcpyAsync(paths.data(), b->d_path, (size_t)nr * N * 8, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(lps.data(), b->d_logprob, (size_t)nr * 8, hipMemcpyDeviceToHost, b->stream));
    uint32_t gave_up = 0;
    if (cluster_launched) HIPCHK(hipMemcpyAsync(&gave_up, b->d_vflag, 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (gave_up) {
        // a member of a lattice cluster waited for a row until its watchdog ran out (its partners were not resident, or gone): the decode is
        // repeated with one workgroup per restart, which waits for nobody
        b->cluster_timeouts++;
        if (hold.c && hold.n) { hold.c->fetch_sub(hold.n); hold.n = 0; }
        return viterbi_paths(b, r0, nr, paths, lps, model, false);
    }
    return RMX_OK;
}

int rmx_infer_cn_batch(rmx_batch *b, int32_t r0, int32_t nr, int64_t *cn_out, double *logprob_out) { BIND(b);
    if (!b || r0 < 0 || nr < 1 || r0 + nr > b->R || !cn_out) return fail(RMX_EARG, "bad argument");
    const Dev &d = b->d;
    const int N = d.N, S = d.S, M = d.M;
    std::vector<int64_t> paths; std::vector<double> lps;
    // maximal runs of restarts with a lattice; the others have framelogprob == 1 and log_transmat == 0
    // (bpmodel.pyx:557-558): every comparison ties, the first index wins
    std::vector<int64_t> all((size_t)nr * N, 0); std::vector<double> lp(nr, (double)N);
    for (int i = 0; i < nr;) {
        if (!b->lt_valid[r0 + i]) { i++; continue; }
        int j = i; while (j < nr && b->lt_valid[r0 + j] && b->lt_model[r0 + j] == b->lt_model[r0 + i]) j++;
        int rc = viterbi_paths(b, r0 + i, j - i, paths, lps, b->lt_model[r0 + i]); if (rc) return rc;
        memcpy(all.data() + (size_t)i * N, paths.data(), (size_t)(j - i) * N * 8);
        for (int k = i; k < j; k++) lp[k] = lps[k - i];
        i = j;
    }
    b->last_path.assign(all.end() - N, all.end());
    // bpmodel.pyx:1205-1210 (the allele "swap" there re-uses the flipped index on both sides: a plain gather)
    for (int i = 0; i < nr; i++) {
        if (logprob_out) logprob_out[i] = lp[i];
        for (int n = 0; n < N; n++) {
            const int64_t *t = b->cn_classes.data() + ((size_t)b->seg_class[n] * S + all[(size_t)i * N + n]) * M * 2;
            for (int k = 0; k < M * 2; k++) cn_out[((size_t)i * N + n) * M * 2 + k] = t[k];
        }
    }
    return RMX_OK;
}
int rmx_infer_cn(rmx_batch *b, int32_t r, int64_t *cn_out, double *logprob_out) { BIND(b);
    if (!b || r < 0 || r >= b->R || !cn_out) return fail(RMX_EARG, "bad argument");
    return rmx_infer_cn_batch(b, r, 1, cn_out, logprob_out);
}

// ---- module-level functions -------------------------------------------------------------------
int rmx_sum_product(const double *f, const double *T, double *alphas, double *betas, int32_t N, int32_t S, int32_t device) {
    if (!f || !T || !alphas || !betas || N < 1 || S < 1) return fail(RMX_EARG, "bad argument");
    if (S > 1024) return fail(RMX_EUNSUPPORTED, "S > 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RMX_EDEVICE, "no HIP device available (there is no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    double *df, *dT, *da, *db;
    const size_t ns = (size_t)N * S, nss = (size_t)std::max(N - 1, 1) * S * S;
    HIPCHK(hipMalloc((void **)&df, ns * 8)); HIPCHK(hipMalloc((void **)&dT, nss * 8)); HIPCHK(hipMalloc((void **)&da, ns * 8)); HIPCHK(hipMalloc((void **)&db, ns * 8));
    HIPCHK(hipMemcpy(df, f, ns * 8, hipMemcpyHostToDevice));
    if (N > 1) HIPCHK(hipMemcpy(dT, T, (size_t)(N - 1) * S * S * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sum_product_dense, dim3(1), dim3(((S + 63) / 64) * 64), (size_t)S * 8, 0, df, dT, da, db, N, S);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(alphas, da, ns * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(betas, db, ns * 8, hipMemcpyDeviceToHost));
    hipFree(df); hipFree(dT); hipFree(da); hipFree(db);
    for (size_t i = 0; i < ns; i++) if (alphas[i] != alphas[i] || betas[i] != betas[i]) return fail(RMX_EASSERT, "nan in alphas/betas");
    return RMX_OK;
}
int rmx_max_product(const double *f, const double *T, int64_t *ss, double *logprob, int32_t N, int32_t S, int32_t device) {
    if (!f || !T || !ss || N < 1 || S < 1) return fail(RMX_EARG, "bad argument");
    if (S > 1024) return fail(RMX_EUNSUPPORTED, "S > 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RMX_EDEVICE, "no HIP device available (there is no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    double *df, *dT, *dfin; uint16_t *dbp;
    const size_t ns = (size_t)N * S, nss = (size_t)std::max(N - 1, 1) * S * S;
    HIPCHK(hipMalloc((void **)&df, ns * 8)); HIPCHK(hipMalloc((void **)&dT, nss * 8)); HIPCHK(hipMalloc((void **)&dfin, (size_t)S * 8)); HIPCHK(hipMalloc((void **)&dbp, ns * 2));
    HIPCHK(hipMemcpy(df, f, ns * 8, hipMemcpyHostToDevice));
    if (N > 1) HIPCHK(hipMemcpy(dT, T, (size_t)(N - 1) * S * S * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_max_product_dense, dim3(1), dim3(((S + 63) / 64) * 64), (size_t)S * 8, 0, df, dT, dbp, dfin, N, S);
    HIPCHK(hipGetLastError());
    std::vector<uint16_t> bp(ns); std::vector<double> fin(S);
    HIPCHK(hipMemcpy(bp.data(), dbp, ns * 2, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(fin.data(), dfin, (size_t)S * 8, hipMemcpyDeviceToHost));
    hipFree(df); hipFree(dT); hipFree(dfin); hipFree(dbp);
    int mp = 0; double vm = fin[0];
    for (int i = 1; i < S; i++) if (fin[i] > vm) { vm = fin[i]; mp = i; }
    ss[N - 1] = mp;
    for (int n = N - 1; n >= 1; n--) ss[n - 1] = bp[(size_t)n * S + ss[n]];
    if (logprob) *logprob = vm;
    return RMX_OK;
}

// ---- measurement --------------------------------------------------------------------------------
int rmx_timer_start(rmx_batch *b) { BIND(b); HIPCHK(hipEventRecord(b->tm_a, b->stream)); return RMX_OK; }
int rmx_timer_stop(rmx_batch *b, double *ms) { BIND(b);
    HIPCHK(hipEventRecord(b->tm_b, b->stream)); HIPCHK(hipEventSynchronize(b->tm_b));
    float f = 0; HIPCHK(hipEventElapsedTime(&f, b->tm_a, b->tm_b)); *ms = f; return RMX_OK;
}
int rmx_profile_enable(rmx_batch *b, int32_t on) { BIND(b); prof_collect(b); b->prof = on < 0 ? 0 : (on > 2 ? 1 : on); return RMX_OK; }
int rmx_profile_get(rmx_batch *b, int32_t id, double *ms, int64_t *n) { BIND(b);
    if (id < 0 || id >= KID_COUNT) return fail(RMX_EARG, "bad kernel id");
    HIPCHK(hipStreamSynchronize(b->stream));
    prof_collect(b);
    *ms = b->prof_ms[id]; *n = b->prof_n[id];
    return RMX_OK;
}
int rmx_profile_reset(rmx_batch *b) { BIND(b); prof_collect(b); for (int i = 0; i < KID_COUNT; i++) { b->prof_ms[i] = 0; b->prof_n[i] = 0; } return RMX_OK; }
const char *rmx_kernel_name(int32_t id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : ""; }
int rmx_num_kernels(void) { return KID_COUNT; }

}  // extern "C"
