// rmx_kernels.h -- HIP kernels of the ReMixT variational-HMM hot path (gfx950).
//
// Design: the reference materialises three dense (N-1) x S x S
// float64 arrays per model (bpmodel.pyx:558-561).  Here nothing of that size
// exists: transitions are generated on the fly from per-class S x S tables held
// in registers / L2, the forward-backward recursion runs in the *scaled linear
// domain* (one FMA per transition term instead of one exp), telomeres split the
// genome into independent chains (one workgroup per chain x restart x direction),
// and the pairwise posterior is only ever formed at breakend adjacencies, where
// the breakpoint update and the ELBO actually need it.
#pragma once
#include "rmx_device.h"
#include "rmx_host.h"      // rmxh::Nm1 (k_param_search_nm)
// Lanes per segment in the kernels that walk the per-segment lists of states with posterior mass (mean 13 listed states, at most 32).
// The sampled objectives of the M-step rounds (<= 200 segments per request: one workgroup wave, latency-bound) keep half a wave per
// segment -- every listed state in one step; the trial passes over all segments (throughput-bound) take a quarter wave and a second
// step for the segments with more than 16 listed states: 81 % instead of 40 % of the lanes at work (h:trial 1.6 -> 0.94 ms).
#define SEGL 32
#define SEG_PER_BLOCK (256 / SEGL)
#define TRIAL_SEGL 16
#define RMX_CLUSTER_SPINS (1u << 22)      // polls of one element before a lattice cluster member gives up (each at least a memory round trip: seconds)
#define FBK_WKN 128      // entries of exp(-pen * k) per transition class (k = allele distance: < 64 up to max_cn 15, < 128 up to 31)
#define NM_MAX_SAMPLE 1024      // sampled segments per request in the layouts of the flat M-step kernels (the reference samples min(200, N / 10))
#define TRIAL_SEG_PER_BLOCK (256 / TRIAL_SEGL)

// =============================================================================
// per-restart tables
// =============================================================================

// state tables: depth, allele ratio, dispersion and lgamma(M p), lgamma(M (1-p))
// per (restart, class, state).  grid (C, nr), block 256.
__device__ __forceinline__ void state_tables_body(const Dev &d, int cls, int r, const RestartParams &rp) {
    __shared__ unsigned agg_sh;
    if (threadIdx.x == 0) agg_sh = 0u;
    __syncthreads();
    unsigned agg = 0u;
    for (int s = threadIdx.x; s < d.S; s += blockDim.x) {
        const int8_t *cn = d.cn + ((size_t)cls * d.S + s) * d.M * 2;
        const int8_t *tot = d.tot + ((size_t)cls * d.S + s) * d.M;
        double minor = 0., total = 0.;
        for (int m = 0; m < d.M; m++) {   // bpmodel.pyx:717-719 accumulation order
            minor += rp.h[m] * (double)cn[m * 2 + 0];
            total += rp.h[m] * (double)tot[m];
        }
        const unsigned sf = d.sflags[(size_t)cls * d.S + s];
        const bool hdel = sf & 1u, loh = sf & 2u;
        unsigned fl = 0;
        if (!d.nc && hdel) fl |= ST_HDEL_NB;
        double p;
        if (hdel) p = 0.;
        else {
            if (total <= 0.) { fl |= ST_E_TD; p = 0.5; }
            else p = minor / total;
        }
        double M0, M1;
        if (!d.nc && loh) {
            if (p == 0.) p = rp.p[RMX_P_BETABIN_LOH_P];
            else if (p == 1.) p = 1. - rp.p[RMX_P_BETABIN_LOH_P];
            else if (!(fl & ST_E_TD)) fl |= ST_E_LOH;
            M0 = rp.p[RMX_P_BETABIN_LOH_M_0]; M1 = rp.p[RMX_P_BETABIN_LOH_M_1];
            fl |= ST_LOH_M | ST_GZ_ALLELE;
        } else {
            M0 = rp.p[RMX_P_BETABIN_M_0]; M1 = rp.p[RMX_P_BETABIN_M_1];
        }
        if (p <= 0. || (1 - p) <= 0.) fl |= ST_E_BADP;
        const size_t base = ((size_t)r * d.C + cls);
        const size_t si = base * d.SP + s;
        d.stD[si] = total;
        d.stLogD[si] = total > 0. ? log(total) : 0.;
        d.stP[si] = p;
        d.stM[(base * 2 + 0) * d.SP + s] = M0;
        d.stM[(base * 2 + 1) * d.SP + s] = M1;
        const bool ok = !(fl & (ST_E_BADP | ST_E_TD | ST_E_LOH));
        d.stLg[(base * 4 + 0) * d.SP + s] = ok ? lgamma_pos(M0 * p) : 0.;
        d.stLg[(base * 4 + 1) * d.SP + s] = ok ? lgamma_pos(M0 * (1 - p)) : 0.;
        d.stLg[(base * 4 + 2) * d.SP + s] = ok ? lgamma_pos(M1 * p) : 0.;
        d.stLg[(base * 4 + 3) * d.SP + s] = ok ? lgamma_pos(M1 * (1 - p)) : 0.;
        d.stFlags[si] = fl;
        agg |= ((fl & ST_E_TD) ? 1u : 0u) | ((fl & ST_E_LOH) ? 2u : 0u) | (((fl & ST_E_BADP) && !(fl & (ST_E_TD | ST_E_LOH))) ? 4u : 0u);
    }
    if (agg) atomicOr(&agg_sh, agg);
    __syncthreads();
    if (threadIdx.x == 0) d.stFlagsAgg[(size_t)r * d.C + cls] = agg_sh;
}
__global__ void k_state_tables(Dev d, int r0) {
    const int cls = blockIdx.x, r = r0 + blockIdx.y;
    const RestartParams rp = d.rp[r];
    state_tables_body(d, cls, r, rp);
}
// single restart, parameters by value: also publishes them to d.rp[r] (one launch fewer per M-step evaluation)
__global__ void k_state_tables_one(Dev d, int r, RestartParams rp) {
    if (blockIdx.x == 0 && threadIdx.x == 0) d.rp[r] = rp;
    state_tables_body(d, blockIdx.x, r, rp);
}

// per-segment constants of the NB / BB log pmf.  grid (ceil(N/256), nr)
__device__ __forceinline__ double seg_const_value(const RestartParams &rp, double x, double y0, double ys, int i) {
    // i in 0..3: NB constants [u*2+var]; i in 4..7: BB constants [v*2+var]
    if (i < 4) {
        // (values selected, not an index: a parameter copy changed in registers -- the table-free searches -- must not need an indexable home in scratch)
        const double rr = i == 0 ? rp.p[RMX_P_NEGBIN_R_0] : (i == 1 ? rp.p[RMX_P_NEGBIN_HDEL_R_0] : (i == 2 ? rp.p[RMX_P_NEGBIN_R_1] : rp.p[RMX_P_NEGBIN_HDEL_R_1]));
        return lgamma_pos(x + rr) - lgamma_pos(x + 1) - lgamma_pos(rr);
    }
    const double MM = i == 4 ? rp.p[RMX_P_BETABIN_M_0] : (i == 5 ? rp.p[RMX_P_BETABIN_LOH_M_0] : (i == 6 ? rp.p[RMX_P_BETABIN_M_1] : rp.p[RMX_P_BETABIN_LOH_M_1]));
    return (lgamma_pos(ys + 1) - lgamma_pos(y0 + 1) - lgamma_pos(ys - y0 + 1)) - lgamma_pos(ys + MM) + lgamma_pos(MM);
}
__global__ void k_seg_const(Dev d, int r0) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = r0 + blockIdx.y;
    if (n >= d.N) return;
    const RestartParams &rp = d.rp[r];
    const double x = d.x[n], y0 = d.y[2 * (size_t)n], y1 = d.y[2 * (size_t)n + 1], ys = y0 + y1;
    double *sc = d.segc + (size_t)r * 8 * d.N + n;
#pragma unroll
    for (int i = 0; i < 8; i++) sc[(size_t)i * d.N] = seg_const_value(rp, x, y0, ys, i);
}

// =============================================================================
// emission fill: update_framelogprob (bpmodel.pyx:898-919) + per-row maximum
// grid (ceil(N / (256/G)), nr), block 256; G lanes cooperate on one segment row
// =============================================================================
__global__ void k_framelogprob(Dev d, int r0, int G) {
    const int r = r0 + blockIdx.y;
    const int rows = 256 / G;
    const int n = blockIdx.x * rows + threadIdx.x / G, gl = threadIdx.x % G;
    if (n >= d.N) return;
    const RestartParams &rp = d.rp[r];
    SegCtx sc; load_seg(d, r, n, sc);
    const int cls = d.seg_class[n];
    const double qt0 = d.qt[((size_t)r * d.N + n) * 2], qt1 = d.qt[((size_t)r * d.N + n) * 2 + 1];
    const double qa0 = d.qa[((size_t)r * d.N + n) * 2], qa1 = d.qa[((size_t)r * d.N + n) * 2 + 1];
    const double qs0 = d.qs[((size_t)r * d.N + n) * 2], qs1 = d.qs[((size_t)r * d.N + n) * 2 + 1];
    double *frow = d.f + rs_off(d, r, n);
    unsigned err = 0;
    double vmax = -INFINITY;
    for (int s = gl; s < d.S; s += G) {
        double LT[2], LA[4];
        cell_ll(d, rp, sc, r, cls, s, LT, LA, err);
        double f = 0.;
        f += qt0 * LT[0]; f += qt1 * LT[1];
        f += qa0 * qs0 * LA[0]; f += qa0 * qs1 * LA[1]; f += qa1 * qs0 * LA[2]; f += qa1 * qs1 * LA[3];
        f += cell_prior(d, rp, sc, cls, s);
        if (f != f) err |= RMX_ERR_NAN_F;
        frow[s] = f;
        vmax = f > vmax ? f : vmax;
    }
    vmax = group_max(vmax, G);
    if (gl == 0) d.fmax[(size_t)r * d.N + n] = vmax;
    // scaled linear-domain emissions for the forward-backward kernel (each lane re-reads its own stores)
    double *erow = d.fe + rs_off(d, r, n);
    for (int s = gl; s < d.SP; s += G) erow[s] = s < d.S ? exp_fast(frow[s] - vmax) : 0.;
    if (err) atomicOr(&d.err[r], err);
}

// =============================================================================
// forward-backward in the scaled linear domain
//
//   e[n,j]  = exp(f[n,j] - fmax[n])
//   fwd:  a~[k] = (a~[k-1] W) / max(a~[k-1]) * e[k]            (stored: fa)
//   bwd:  b~[k] = (W g[k+1]) / max(g[k+1]),  g[k] = e[k] * b~[k] (stored: fb)
//
// which equals sum_product (bpmodel.pyx:1213-1246) up to a per-row positive scale
// that cancels in every consumer (posterior marginals, pairwise marginals); the
// scales themselves (row maxima, fmax) are kept to rebuild hmm_log_norm_const.
//
// One workgroup per (chain, restart, direction).  Thread (o, p): output state o,
// slice p of the reduction index q.  Plain-adjacency weights W[q][o] come from the
// transition class's table; breakend adjacencies build their weights on the fly from
// the per-breakend distance tables (bpmodel.pyx:658-668).
// One barrier per step: the unnormalised vector and the per-wave maxima are
// published together and the division by the maximum is applied by the consumer.
// =============================================================================
#define FB_NBUF 3

struct FbLaunch { int P, NT, BLK, SPAD; };

// compact argument block: only what the recursion touches (keeps the kernel's SGPR footprint small)
struct FbArgs {
    int S, SP, M, D, C, N, NBE, cn_max, P, BLK, SPAD, r0, amat_lds, pad_;
    double pen;
    const int32_t *chain_start, *chain_end, *tclass, *brk_slot, *chain_list, *chain_tc, *chain_cls, *be_cls;
    const double *fe, *Wf, *Wb, *pe_lt;
    const int8_t *af, *ab, *tot;
    double *fa, *fb, *mrow;
    uint32_t *err;
    unsigned long long *dbg;   // optional: [0..3] = shader clock / 100 MHz wall clock at loop start and end of block (0,0,0)
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// raw workgroup barrier: LDS traffic of this wave retired, vector-memory traffic (result stores,
// LDS-DMA prefetch) left in flight
#define FB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// LDS-DMA through inline asm (cdna_hip_programming.md 5.7): hipcc must not see a pending LDS write or
// a pending store in the step loop, or it drains vmcnt(0) -- the result stores' HBM round trip -- in
// front of the LDS reads of every step.  All vector-memory traffic of the loop is therefore issued
// here and retired by the explicit waits at the block boundaries.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
// 8-byte global load whose completion the COMPILER does not track: no s_waitcnt is inserted for it
// anywhere (a tracked load pending over the step loop makes hipcc drain vmcnt(0) -- i.e. also the result
// stores' HBM round trip -- at the loop head).  The value is valid only after gwait8() on the same
// variable; nothing may read (or copy) it in between.
__device__ __forceinline__ void gload8(double &dst, const double *src) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
__device__ __forceinline__ void gwait8(double &v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory"); }
// ... placed behind the computation of `after` (the products of a step are non-volatile asm: without the tie the compiler is free to sink
// them below the wait -- it did, and every step then began with the load's HBM round trip)
__device__ __forceinline__ void gwait8_after(double &v, double &after) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v), "+v"(after) :: "memory"); }
__device__ __forceinline__ void gstore8(double *dst, double v) {
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(lptr_t)p; }

// k_fb is the general single-vector kernel: any chain (segments of different state-table classes, any S up
// to 1024), weights read from the tabulated S x S matrices in L2.  Chains of one class go to k_fbv /
// k_fbk below, which is where the time is spent.
//
// Thread t = p * S + o: slice p of the reduction index for output state o (slices are packed
// back to back, so a wave may straddle two slices).  A step is
//   phase 1 (all threads): partial[p][o] = sum_{q in slice p} vec[q] * W[q][o]        -> LDS
//   phase 2 (threads t < S): sum the P partials in fixed order, scale, multiply by the emission,
//           publish the new vector + its maximum, store the result row
// with one raw barrier after each phase.  Phase 2 runs on ceil(S/64) waves only, so the per-step
// bookkeeping is not replicated on every wave of the workgroup.
#ifdef RMX_FB_STAMPS
#define FB_STAMP(i_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); stamp_acc[i_] += t_ - stamp_last; stamp_last = t_; }
#else
#define FB_STAMP(i_)
#endif
template <int NTMAX>
__global__ __launch_bounds__(NTMAX) void k_fb(FbArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int chain = a.chain_list[blockIdx.x], r = a.r0 + blockIdx.y, dir = blockIdx.z;
    const int S = a.S, M = a.M, D = a.D, SP = a.SP;
    const int n0 = a.chain_start[chain], n1 = a.chain_end[chain], len = n1 - n0 + 1;
    const int t = threadIdx.x, NT = blockDim.x;
    const int PP = a.P;
    const int p = t / S, o = t - p * S;
    const bool act = p < PP;                            // thread takes part in phase 1
    const bool post = t < S;                            // thread takes part in phase 2
    const int QPT = (S + PP - 1) / PP;
    const int SPAD = a.SPAD, BLK = a.BLK;
    const int MDP = (M * D + 1) & ~1;                   // breakend table row, padded to 16 bytes
    // LDS carve-up (one array: see cdna_hip_programming.md 5, trap 4a)
    double *ebuf = (double *)smem_raw;                  // [NBUF][BLK][SP]   emission ring, filled by LDS-DMA
    double *vec = ebuf + (size_t)FB_NBUF * BLK * SP;    // [2][SPAD]
    double *part = vec + 2 * SPAD;                      // [P][SP] partial sums of phase 1
    double *red = part + (size_t)PP * SP;               // [4]  rotating per-step maxima (as u64)
    double *pel = red + 4;                              // [MDP]  exp(-pen*pd) of the current breakend (LDS-DMA)
    double *wa = pel + MDP;                             // [128] exp(-pen*a) for the allele-flip term
    int *meta = (int *)(wa + 128);                      // [NBUF][2][64]  tclass / brk_slot per step (LDS-DMA)
    int8_t *totl = (int8_t *)(meta + FB_NBUF * 2 * 64); // [C][S][M]
    for (int i = t; i < a.C * S * M; i += NT) totl[i] = a.tot[i];
    for (int i = t; i < 2 * SPAD; i += NT) vec[i] = 0.;   // tails [S, SPAD) stay zero: padded slices read them
    for (int i = t; i < 128; i += NT) wa[i] = exp(-a.pen * (double)i);
    for (int i = t; i < FB_NBUF * 2 * 64; i += NT) meta[i] = -1;
    unsigned long long *red64 = reinterpret_cast<unsigned long long *>(red);   // [3] used
    if (t < 4) red64[t] = 0ull;

    const size_t rbase = (size_t)r * a.N;
    const double *febase = a.fe + rbase * SP;
    double *outb = (dir == 0 ? a.fa : a.fb) + rbase * SP;
    double *mrow = a.mrow + rbase;
    const double *Wmat = dir == 0 ? a.Wf : a.Wb;

    __syncthreads();   // LDS initialisation done

#define ROW(k) (dir == 0 ? n0 + (k) : n1 - (k))
    // ---- prefetch ring: block b = steps [b*BLK, (b+1)*BLK).  A block's emission rows are one
    // contiguous span of BLK*SP doubles in HBM (ascending rows for either direction) and its
    // transition metadata one contiguous span of BLK ints; both are copied verbatim into the
    // block's ring slot by LDS-DMA, lane-linear per wave.  Step kk of a block sits at index kk
    // (forward) or BLK-1-kk (backward) of the span.
    const int nblk = (len + BLK - 1) / BLK;
    const int elems = BLK * SP / 2;                       // 16-byte elements per block
    const int wave_base = (t >> 6) << 6;
#define FB_ISSUE(b_)                                                                                                   \
    {                                                                                                                  \
        const int slot_ = (b_) % FB_NBUF;                                                                              \
        const int rs_ = dir == 0 ? n0 + (b_) * BLK : n1 - (b_) * BLK - (BLK - 1);                                      \
        for (int i0 = 0; i0 < elems; i0 += NT) {                                                                       \
            const int idx = i0 + t;                                                                                    \
            const int row_ = rs_ + (idx * 2) / SP;                                                                     \
            const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_addr(ebuf + (size_t)slot_ * BLK * SP + (size_t)(i0 + wave_base) * 2)); \
            if (idx < elems && row_ >= n0 && row_ <= n1) glds16(febase + (size_t)rs_ * SP + (size_t)idx * 2, dst_);   \
        }                                                                                                              \
        if (t < 64) {                                                                                                  \
            const int tn_ = (dir == 0 ? n0 + (b_) * BLK - 1 : n1 - (b_) * BLK - (BLK - 1)) + t;                        \
            const unsigned d0_ = __builtin_amdgcn_readfirstlane(lds_addr(meta + (slot_ * 2 + 0) * 64));                 \
            const unsigned d1_ = __builtin_amdgcn_readfirstlane(lds_addr(meta + (slot_ * 2 + 1) * 64));                 \
            if (t < BLK && tn_ >= n0 && tn_ < n1) { glds4(a.tclass + tn_, d0_); glds4(a.brk_slot + tn_, d1_); }        \
        }                                                                                                              \
    }

    FB_ISSUE(0)
    if (nblk > 1) FB_ISSUE(1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nblk > 2) FB_ISSUE(2)
    FB_BARRIER();

    // ---- step 0 --------------------------------------------------------------------
    const int rstep = dir == 0 ? SP : -SP;                       // row stride of one step, in doubles
    double *outp = outb + (size_t)ROW(0) * SP + o;                // this lane's output element of the current row
    double *mptr = mrow + ROW(0);
    const int npw = (S + 63) >> 6;                                // waves that take part in phase 2
    {
        double e0 = 0.;
        if (post) { e0 = ebuf[(size_t)(dir == 0 ? 0 : BLK - 1) * SP + o]; vec[o] = e0; gstore8(outp, (dir == 0) ? e0 : 1.0); }
        if ((t >> 6) < npw) {   // wave-uniform: the cross-lane maximum needs every lane of the wave
            const double wm = wave_max_nonneg(e0);
            if ((t & 63) == 0) lds_max_u64(&red64[0], (unsigned long long)__double_as_longlong(wm));
        }
    }
    FB_BARRIER();

    // incremental step bookkeeping (no integer division in the loop)
    int b = 0, kk = 0, slot = 0;          // block of step k, position inside it, ring slot b % FB_NBUF
    int cur = 0, nxt = 1;                 // vec double buffer
    int rc = 0;                           // red64 slot written during step k-1
    if (a.dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); a.dbg[4] = len; }
#ifdef RMX_FB_STAMPS
    unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif
    for (int k = 1; k < len; k++) {
        FB_STAMP(0)
        if (++kk == BLK) { kk = 0; b++; slot = slot + 1 == FB_NBUF ? 0 : slot + 1; }
        const int kidx = dir == 0 ? kk : BLK - 1 - kk;            // position of the step inside its span
        const double *vc = vec + cur * SPAD;
        const int bs = meta[(slot * 2 + 1) * 64 + kidx];
        // operands of phase 2 that are already final: fetched now, their latency (and the reciprocal)
        // hides under phase 1
        const int rn = rc + 1 == 3 ? 0 : rc + 1, rz = rn + 1 == 3 ? 0 : rn + 1;
        double inv = 0., e = 0.;
        if ((t >> 6) < npw) {
            double m;
            pow2_scale((unsigned)(red64[rc] >> 32), m, inv);
            if (post) e = ebuf[((size_t)slot * BLK + kidx) * SP + o];
            if (t == 0) { if (dir == 0) gstore8(mptr, m); red64[rz] = 0ull; }
        }
        FB_STAMP(1)
        // ============================ phase 1: partial products ============================
        double acc = 0.;
        if (bs < 0) {
            const double *Wt = Wmat + (size_t)meta[(slot * 2 + 0) * 64 + kidx] * S * S;
            if (act) for (int rr = 0; rr < QPT; rr++) { const int q = p * QPT + rr; if (q < S) acc = fma(vc[q], Wt[(size_t)q * S + o], acc); }
        } else {
            // ---- breakend adjacency: W[i][j] = prod_m exp(-pen*pd_m[d_m(i,j)]) * exp(-pen*a(i,j)) ----
            const int tc = meta[(slot * 2 + 0) * 64 + kidx];
            if (t < 64) {
                const unsigned dpe = __builtin_amdgcn_readfirstlane(lds_addr(pel));
                if (t * 2 < MDP) glds16(a.pe_lt + ((size_t)r * a.NBE + bs) * MDP + t * 2, dpe);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FB_BARRIER();
            int ca, cb;
            const int8_t *at;
            ca = a.be_cls[2 * bs]; cb = a.be_cls[2 * bs + 1];
            at = (dir == 0 ? a.af : a.ab) + (size_t)tc * S * S;
            // fwd: q = from-state (class ca), o = to-state (class cb); bwd: q = to-state (cb), o = from-state (ca)
            const int8_t *tq = totl + (size_t)(dir == 0 ? ca : cb) * S * M;
            const int8_t *to = totl + (size_t)(dir == 0 ? cb : ca) * S * M;
            const int sgn = dir == 0 ? 1 : -1;
            if (act) {
                int to_m[RMX_MAX_CLONES];
#pragma unroll
                for (int c = 0; c < RMX_MAX_CLONES; c++) to_m[c] = c < M ? (int)to[(size_t)o * M + c] : 0;
                for (int rr = 0; rr < QPT; rr++) {
                    const int q = p * QPT + rr;
                    if (q < S) {
                        double wv = wa[(int)at[(size_t)q * S + o]];
#pragma unroll
                        for (int c = 0; c < RMX_MAX_CLONES; c++)
                            if (c < M) { const int dd = sgn * ((int)tq[(size_t)q * M + c] - to_m[c]); wv *= pel[c * D + dd + a.cn_max + 1]; }
                        acc = fma(vc[q], wv, acc);
                    }
                }
            }
        }
        FB_STAMP(2)
        if (act) part[(size_t)p * SP + o] = acc;
        FB_BARRIER();
        FB_STAMP(3)
        // ============================ phase 2: combine, scale, publish =======================
        outp += rstep;
        if ((t >> 6) < npw) {   // wave-uniform
            double vecv = 0.;
            if (post) {
                // the first four partials are requested together (one LDS round trip, not four)
                const double s0 = part[o];
                const double s1 = PP > 1 ? part[(size_t)SP + o] : 0.;
                const double s2 = PP > 2 ? part[(size_t)2 * SP + o] : 0.;
                const double s3 = PP > 3 ? part[(size_t)3 * SP + o] : 0.;
                double sum = ((s0 + s1) + s2) + s3;
                for (int pp = 4; pp < PP; pp++) sum += part[(size_t)pp * SP + o];
                const double val = sum * inv;
                vecv = val * e;
                vec[nxt * SPAD + o] = vecv;
                gstore8(outp, (dir == 0) ? vecv : val);
                if (vecv != vecv) vecv = INFINITY;   // propagate a NaN as a detectable value
            }
            const double wm = wave_max_nonneg(vecv);
            if ((t & 63) == 0) lds_max_u64(&red64[rn], (unsigned long long)__double_as_longlong(wm));
        }
        mptr += (dir == 0 ? 1 : -1);
        if (kk == 0 && b >= 1) {
            // block boundary: block b+1 (requested one block ago) must have landed before its first
            // reader, BLK barriers from now; then request block b+2 into the slot block b-1 vacated
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (b + 2 < nblk) FB_ISSUE(b + 2)
        }
        cur ^= 1; nxt ^= 1; rc = rn;
        FB_STAMP(4)
        FB_BARRIER();
        FB_STAMP(5)
    }
#ifdef RMX_FB_STAMPS
    if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (t == 0 || t == NT - 64)) for (int i = 0; i < 6; i++) a.dbg[8 + (t == 0 ? 0 : 6) + i] = stamp_acc[i];
#endif
    if (a.dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
    if (t == 0) {
        // last row of the chain: its own maximum is not consumed by a later step
        double m, inv_;
        pow2_scale((unsigned)(red64[rc] >> 32), m, inv_);
        if (dir == 0) gstore8(mptr, m);
        if (!(m > 0.) || m == INFINITY) atomicOr(&a.err[r], RMX_ERR_NAN_AB);
    }
#undef ROW
#undef FB_ISSUE
}

// =============================================================================
// k_fbv: the production forward-backward kernel.  Same recursion as k_fb, but ONE workgroup advances
// NV restarts of the same (chain, direction) in lock step.  The plain-adjacency weights do not depend
// on the restart, so they are held in registers once (2 output columns x RPT rows per thread, P = 8
// row slices); the latency chain of a step (two barriers, the partial-sum hand-off, the maximum) is
// paid once for NV vectors instead of once per vector.
// The vector operand of the S x S product never comes from LDS broadcasts: the 16 lanes of a DPP row
// share one row slice p, lane n of the row loads a[p*RPT + n] (and a[p*RPT + 16 + n]) ONCE per step
// and vector, and every product term is `v_fmac_f64_dpp acc, a, w row_newbcast:n` -- the multiplier
// is lane n's register, broadcast inside the row by the DPP network.  Per step and wave that is
// 2*NV 8-byte LDS reads instead of RPT*NV/2 16-byte broadcast reads (11x less LDS traffic at S = 165),
// which leaves phase 1 bound by the FP64 FMA rate.
//   thread t (phase 1):  g = t % G2 (column pair 2g, 2g+1; G2 a multiple of 16), p = t / G2 (row slice), t < 8*G2
//   thread t (phase 2):  v = t / SPW (vector), o = t % SPW (state), SPW = ceil(S/64)*64
// Only chains whose segments share one state-table class come here (chain_tc >= 0).
// =============================================================================
// acc0[v] += a_v[row] * w0[row], acc1[v] += a_v[row] * w1[row] for row = RR..RPT-1 and every vector v, the
// multiplier taken from lane (row % 16) of the DPP row (register av[v][row / 16]).  gfx90a+ VOP2 DPP
// on 64-bit operands supports exactly this control (row_newbcast).  Rows ascend per accumulator (the
// summation order of the LDS-broadcast formulation) and the 2*NV accumulators give the FMA pipe
// independent chains.
// WAIT STATES (asm site 1 of 5; tools/asm_hazards.py checks the built code): the DPP source av[] comes from LDS reads (no VALU write,
// the s_waitcnt is the only dependency); the accumulators are asm-written and next read by compiler code -- fbv's row reduction, whose first
// consumer is a DPP v_mov.  A VALU write -> DPP read needs two states the compiler cannot know about here: the static check over the
// disassembly (CPU test tests/test_asm_hazards.py) is what enforces them for every build.
template <int RPT, int NV, int NA, int RR>
__device__ __forceinline__ void fbv_row_fma(const double (&av)[NV][NA], const double (&w0)[RPT], const double (&w1)[RPT],
                                            double (&acc0)[NV], double (&acc1)[NV]) {
    if constexpr (RR < RPT) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc0[v]) : "v"(av[v][RR / 16]), "v"(w0[RR]), "n"(RR % 16));
            asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc1[v]) : "v"(av[v][RR / 16]), "v"(w1[RR]), "n"(RR % 16));
        }
        fbv_row_fma<RPT, NV, NA, RR + 1>(av, w0, w1, acc0, acc1);
    }
}

// breakend-step counterpart of fbv_row_fma: the weight of (row, column) is wa[a] * tab_v[idx] with
// (idx | a << 10) the pair's 16-bit code (two adjacent columns = one 32-bit LDS read)
template <int RPT, int NV, int NA, int RR>
__device__ __forceinline__ void fbv_row_fma_be(const double (&av)[NV][NA], const unsigned short *crow, int SPC, const double *wa,
                                               const double *tab, int tabw, double (&acc0)[NV], double (&acc1)[NV]) {
    if constexpr (RR < RPT) {
        const unsigned cc = *reinterpret_cast<const unsigned *>(crow + (size_t)RR * SPC);
        const unsigned c0 = cc & 0xffffu, c1 = cc >> 16;
        const double wa0 = wa[c0 >> 10], wa1 = wa[c1 >> 10];
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const double x0 = wa0 * tab[(size_t)v * tabw + (c0 & 1023u)], x1 = wa1 * tab[(size_t)v * tabw + (c1 & 1023u)];
            asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc0[v]) : "v"(av[v][RR / 16]), "v"(x0), "n"(RR % 16));
            asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc1[v]) : "v"(av[v][RR / 16]), "v"(x1), "n"(RR % 16));
        }
        if (RR % 2 == 1) __builtin_amdgcn_sched_barrier(0);      // bound the look-ahead of the LDS reads (register budget)
        fbv_row_fma_be<RPT, NV, NA, RR + 1>(av, crow, SPC, wa, tab, tabw, acc0, acc1);
    }
}

#define FBV_P 8
#ifndef FBV_DEPTH
#define FBV_DEPTH 4
#endif
struct FbvArgs {
    int S, SP, M, D, C, N, NBE, cn_max, BLK, SPAD, r0, r1, amat_lds, G2, SPW, pad_;
    int code_lds, PE2P, SPC, pad2_;     // breakend fast path: code table in LDS, padded row length of pe2_lt, row stride of the code table
    double pen;
    const int32_t *chain_start, *chain_end, *tclass, *brk_slot, *chain_list, *chain_tc, *chain_cls, *be_n, *chain_be;
    const double *fe, *Wf, *Wb, *pe_lt, *pe2_lt;
    const int8_t *af, *ab, *tot;
    double *fa, *fb, *mrow;
    uint32_t *err;
    unsigned long long *dbg;
};

template <int RPT, int NV, int NTMAX>
__global__ __launch_bounds__(NTMAX) void k_fbv(FbvArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int chain = a.chain_list[blockIdx.x], dir = blockIdx.z;
    const int rg0 = a.r0 + blockIdx.y * NV;                 // first restart of this group
    const int nv = min(NV, a.r1 - rg0);                     // vectors actually present
    const int S = a.S, M = a.M, D = a.D, SP = a.SP, G2 = a.G2, SPW = a.SPW;
    const int n0 = a.chain_start[chain], n1 = a.chain_end[chain], len = n1 - n0 + 1;
    const int t = threadIdx.x, NT = blockDim.x;
    const int p = t / G2, g = t - p * G2;
    const int o0 = 2 * g, o1 = 2 * g + 1;
    const bool act = p < FBV_P;                             // phase-1 worker
    const int pv = t / SPW, po = t - pv * SPW;              // phase-2 role
    const bool post = pv < nv && po < S;
    const bool postwave = pv < nv;                          // wave-uniform (SPW is a multiple of 64)
    const int lane = t & 63;
    const int SPAD = a.SPAD;
    const int MDP = (M * D + 1) & ~1;
    // ---- LDS carve-up -----------------------------------------------------------------------
    double *vec = (double *)smem_raw;                           // [NV][2][SPAD]  the vectors, double-buffered by step parity
    double *part = vec + (size_t)NV * 2 * SPAD;                 // [NV][P][SP]    partial sums of phase 1
    double *red = part + (size_t)NV * FBV_P * SP;               // [NV][4]: [v][0] = 1/scale of the current vector, [v][1] = scale, [v][2 + buf] = largest high dword of vec[v][buf] (u32 in the slot's low half)
    unsigned *red32 = (unsigned *)red;
    // breakend steps: per-vector weight table(s) of the current breakend -- with code_lds the product over
    // the clones, indexed by the code of a state pair (PE2P doubles per vector), else one table per clone
    const int PELW = a.code_lds ? a.PE2P : MDP;
    double *pel = red + NV * 4;                                 // [NV][PELW]
    double *wa = pel + (size_t)NV * PELW;                       // [128]
    int8_t *totl = (int8_t *)(wa + 128);                        // [C][S][M]
    int8_t *atl = totl + ((a.C * S * M + 15) & ~15);            // [S][S] (amat_lds)
    unsigned short *codel = (unsigned short *)(atl + (a.amat_lds ? ((S * S + 15) & ~15) : 0));   // [8*RPT][SPC] (code_lds)
    int *bel = (int *)(codel + (a.code_lds ? (size_t)FBV_P * RPT * a.SPC : 0));                   // adjacencies of this chain's breakends
    for (int i = t; i < a.C * S * M; i += NT) totl[i] = a.tot[i];
    if (a.amat_lds) {
        const int8_t *src = (dir == 0 ? a.af : a.ab) + (size_t)a.chain_tc[chain] * S * S;
        for (int i = t; i < S * S; i += NT) atl[i] = src[i];
    }
    if (a.code_lds) {
        // code of the pair (row q -> column o) at a breakend adjacency of this chain and direction: the
        // index of the clone-product weight (differences of the tumour clones' totals; the normal clone's
        // is 0 inside a class) in the low 10 bits, the allele distance (< 64) above.  Rows / columns past S: 0.
        const int8_t *src = (dir == 0 ? a.af : a.ab) + (size_t)a.chain_tc[chain] * S * S;
        const int8_t *tg = a.tot + (size_t)a.chain_cls[chain] * S * M;
        const int off_ = a.cn_max + 1, sg_ = dir == 0 ? 1 : -1;
        for (int i = t; i < FBV_P * RPT * a.SPC; i += NT) {
            const int q = i / a.SPC, o = i - q * a.SPC;
            unsigned c_ = 0;
            if (q < S && o < S) {
                int idx = 0;
                for (int c = 1; c < M; c++) idx = idx * D + sg_ * ((int)tg[q * M + c] - (int)tg[o * M + c]) + off_;
                c_ = (unsigned)idx | ((unsigned)src[(size_t)q * S + o] << 10);
            }
            codel[i] = (unsigned short)c_;
        }
    }
    const int be_lo = a.chain_be[2 * chain], be_hi = a.chain_be[2 * chain + 1];
    for (int i = t; i < be_hi - be_lo; i += NT) bel[i] = a.be_n[be_lo + i];
    for (int i = t; i < NV * 2 * SPAD; i += NT) vec[i] = 0.;
    for (int i = t; i < 128; i += NT) wa[i] = exp(-a.pen * (double)i);
    if (t < NV * 4) red[t] = 0.;

    const double *Wmat = (dir == 0 ? a.Wf : a.Wb) + (size_t)a.chain_tc[chain] * S * S;
    // ---- stationary weights: 2 columns x RPT rows, consumed before the loop --------------------
    double w0[RPT], w1[RPT];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
        const int q = p * RPT + rr;
        w0[rr] = (act && q < S && o0 < S) ? Wmat[(size_t)q * S + o0] : 0.;
        w1[rr] = (act && q < S && o1 < S) ? Wmat[(size_t)q * S + o1] : 0.;
    }
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) { asm volatile("" ::"v"(w0[rr])); asm volatile("" ::"v"(w1[rr])); }
    int chain_cls_ = __builtin_amdgcn_readfirstlane(a.chain_cls[chain]);
    asm volatile("" : "+s"(chain_cls_));
    __syncthreads();

#define ROW(k) (dir == 0 ? n0 + (k) : n1 - (k))
    // Adjacency crossed by step k (between rows ROW(k-1) and ROW(k)): n0 + k - 1 forward, n1 - k backward.
    // Every adjacency of such a chain has transition class chain_tc; the breakend ones are the slot
    // interval chain_be[chain] of be_n (ascending, copied to LDS above), walked in step order, so a plain
    // step pays one scalar compare and nothing is fetched per step.
#define ADJ(k) (dir == 0 ? n0 + (k) - 1 : n1 - (k))
    const int tc = a.chain_tc[chain];
    const int be_step = dir == 0 ? 1 : -1;
    int be_i = dir == 0 ? be_lo : be_hi - 1;                                   // slot of the next breakend step
    int be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2;   // its adjacency
    // breakend slot of step k_ (-1: plain adjacency), advancing the walk past it
#define BE_SLOT(k_, bs_)                                                                                   \
    int bs_ = -1;                                                                                          \
    if (ADJ(k_) == be_adj) {                                                                               \
        bs_ = be_i; be_i += be_step;                                                                       \
        be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2; \
    }
    // ---- step 0 ---------------------------------------------------------------------------------
    // The emission value a publishing lane needs in step k is one 8-byte global load (coalesced over
    // the wave), requested a whole step ahead of its use.
    const int rstep = dir == 0 ? SP : -SP;
    // (lanes without a publishing role get a valid in-range address: their loads are issued too, unused)
    const size_t lane_off = ((size_t)(rg0 + (postwave ? pv : 0)) * a.N + ROW(0)) * SP + (po < S ? po : S - 1);
    double *outp = (dir == 0 ? a.fa : a.fb) + lane_off;
    const double *eptr = a.fe + lane_off;
    if (postwave) {      // wave-uniform
        double e0 = 0.;
        if (post) {
            e0 = *eptr;
            vec[(size_t)pv * 2 * SPAD + po] = e0;
            gstore8(outp, (dir == 0) ? e0 : 1.0);
        }
        const unsigned wm = wave_max_u32((unsigned)__double2hiint(e0));
        if (lane == 0) lds_max_u32(&red32[(pv * 4 + 2) * 2], wm);
    }
    eptr += rstep;
    const int8_t *tcl = totl + (size_t)chain_cls_ * S * M;
    const int sgn = dir == 0 ? 1 : -1;
    FB_BARRIER();

    // lanes 0..NV-1: scale of the vector in buffer buf_ (complete since the last barrier) and its exact
    // reciprocal, the scale to mrow (forward), the other buffer's accumulator cleared for this step
#define FBV_SCALE(buf_, row_)                                                                              \
    if (t < NV) {                                                                                          \
        double m_, i_;                                                                                     \
        pow2_scale(red32[(t * 4 + 2 + (buf_)) * 2], m_, i_);                                               \
        red[t * 4] = i_; red[t * 4 + 1] = m_;                                                              \
        red32[(t * 4 + 2 + ((buf_) ^ 1)) * 2] = 0u;                                                        \
        if (dir == 0 && t < nv) gstore8(a.mrow + (size_t)(rg0 + t) * a.N + (row_), m_);                    \
    }

#ifdef RMX_FB_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif
    if (a.dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); a.dbg[4] = len; }
    constexpr int NVX = NV;
    const int vX = 0;
    for (int k = 1; k < len; k++) {
        FB_STAMP(0)
        BE_SLOT(k, bs)
        const int cb = (k - 1) & 1, nb = k & 1;       // buffers of the vectors of steps k-1 and k
        // phase-2 operand that is already final: requested now, latency hidden under phase 1
        double e;
        gload8(e, eptr);
        eptr += rstep;
        FB_STAMP(1)
        // ============================ phase 1 ============================
        double acc0[NV], acc1[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) { acc0[v] = 0.; acc1[v] = 0.; }
        FBV_SCALE(cb, ROW(k - 1))
        if (bs < 0) {
            if (act) {      // wave-uniform: 8*G2 is a multiple of 128
                constexpr int NA = (RPT + 15) / 16;
                double av[NV][NA];
                const double *vb0 = vec + (size_t)cb * SPAD + p * RPT + (lane & 15);
#pragma unroll
                for (int v = 0; v < NV; v++)
#pragma unroll
                    for (int h = 0; h < NA; h++) av[v][h] = vb0[(size_t)v * 2 * SPAD + 16 * h];
                fbv_row_fma<RPT, NV, NA, 0>(av, w0, w1, acc0, acc1);
            }
        } else {
            // ---- breakend adjacency: restart-specific weights prod_m pe_m[d_m] * exp(-pen*a) ----
            if (a.code_lds) {
                // fast path: the clone product comes as one table per vector (k_brk_lut), the pair's table index
                // and allele distance as one 16-bit code from LDS; the vector operand is broadcast inside the DPP
                // row exactly as on a plain step
                if (t < (a.PE2P + 1) / 2) {          // (waves 0..2 at most: wave-granular s_waitcnt below)
                    for (int v = 0; v < NVX; v++) {
                        const unsigned dpe = __builtin_amdgcn_readfirstlane(lds_addr(pel + (size_t)(vX + v) * PELW) + (unsigned)(((t >> 6) << 6) * 16));
                        if (vX + v < nv && t * 2 < a.PE2P) glds16(a.pe2_lt + ((size_t)(rg0 + vX + v) * a.NBE + bs) * a.PE2P + t * 2, dpe);
                    }
                }
                if (t < 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                FB_BARRIER();
                if (act) {
                    constexpr int NA = (RPT + 15) / 16;
                    double av[NVX][NA];
                    const double *vb0 = vec + ((size_t)vX * 2 + cb) * SPAD + p * RPT + (lane & 15);
#pragma unroll
                    for (int v = 0; v < NVX; v++)
#pragma unroll
                        for (int h = 0; h < NA; h++) av[v][h] = vb0[(size_t)v * 2 * SPAD + 16 * h];
                    const unsigned short *crow = codel + (size_t)(p * RPT) * a.SPC + (o0 < a.SPC ? o0 : a.SPC - 2);
                    fbv_row_fma_be<RPT, NVX, NA, 0>(av, crow, a.SPC, wa, pel + (size_t)vX * PELW, PELW, acc0, acc1);
                }
            } else {
                if (t < 64) {
                    for (int v = 0; v < NVX; v++) {
                        const unsigned dpe = __builtin_amdgcn_readfirstlane(lds_addr(pel + (size_t)(vX + v) * MDP));
                        if (vX + v < nv && t * 2 < MDP) glds16(a.pe_lt + ((size_t)(rg0 + vX + v) * a.NBE + bs) * MDP + t * 2, dpe);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                FB_BARRIER();
                const int8_t *at = a.amat_lds ? atl : ((dir == 0 ? a.af : a.ab) + (size_t)tc * S * S);
                if (act) {
                    for (int rr = 0; rr < RPT; rr++) {
                        const int q = p * RPT + rr;
                        if (q >= S) break;
#pragma unroll
                        for (int col = 0; col < 2; col++) {
                            const int o = col == 0 ? o0 : o1;
                            if (o < S) {
                                const double wbase = wa[(int)at[(size_t)q * S + o]];
                                int dd[RMX_MAX_CLONES];
#pragma unroll
                                for (int c = 0; c < RMX_MAX_CLONES; c++) dd[c] = c < M ? sgn * ((int)tcl[(size_t)q * M + c] - (int)tcl[(size_t)o * M + c]) + a.cn_max + 1 : 0;
#pragma unroll
                                for (int v = 0; v < NVX; v++) {
                                    if (vX + v < nv) {
                                        double wv = wbase;
#pragma unroll
                                        for (int c = 0; c < RMX_MAX_CLONES; c++) if (c < M) wv *= pel[(size_t)(vX + v) * MDP + c * D + dd[c]];
                                        const double x = vec[((size_t)(vX + v) * 2 + cb) * SPAD + q];
                                        if (col == 0) acc0[v] = fma(x, wv, acc0[v]); else acc1[v] = fma(x, wv, acc1[v]);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        FB_STAMP(2)
        if (act && o0 < SP) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                double2 pr; pr.x = acc0[v]; pr.y = acc1[v];
                *reinterpret_cast<double2 *>(part + ((size_t)v * FBV_P + p) * SP + o0) = pr;
            }
        }
        FB_BARRIER();
        FB_STAMP(3)
        // ============================ phase 2 ============================
        outp += rstep;
        unsigned vmax_in = 0u;
        gwait8(e);        // a step old by now, like this wave's previous result store
        if (post) {
            const double inv = red[pv * 4];
            const double *pp_ = part + (size_t)pv * FBV_P * SP + po;
            const double s0 = pp_[0], s1 = pp_[SP], s2 = pp_[2 * SP], s3 = pp_[3 * SP];
            const double s4 = pp_[4 * SP], s5 = pp_[5 * SP], s6 = pp_[6 * SP], s7 = pp_[7 * SP];
            const double sum = ((((((s0 + s1) + s2) + s3) + s4) + s5) + s6) + s7;
            FB_STAMP(6)
            const double val = sum * inv;
            const double vecv = val * e;
            vec[((size_t)pv * 2 + nb) * SPAD + po] = vecv;
            gstore8(outp, (dir == 0) ? vecv : val);
            vmax_in = (unsigned)__double2hiint(vecv);
        }
        FB_STAMP(7)
        if (postwave) {
            const unsigned wm = wave_max_u32(vmax_in);
            if (lane == 0) lds_max_u32(&red32[(pv * 4 + 2 + nb) * 2], wm);
        }
        FB_STAMP(8)
        FB_BARRIER();
        FB_STAMP(5)
    }
    if (a.dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
#ifdef RMX_FB_STAMPS
    if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (t == 0 || t == NT - 64)) for (int i = 0; i < 10; i++) a.dbg[8 + (t == 0 ? 0 : 10) + i] = stamp_acc[i];
#endif
    // last row of each chain: its scale is not consumed by a later step, but hmm_log_norm_const and
    // the vanishing-row check need it
    FBV_SCALE((len - 1) & 1, ROW(len - 1))
    if (t < nv) { const double m = red[t * 4 + 1]; if (!(m > 0.) || m == INFINITY) atomicOr(&a.err[rg0 + t], RMX_ERR_NAN_AB); }
#undef FBV_SCALE
#undef ROW
#undef ADJ
#undef BE_SLOT
}

__global__ void k_fill_f64(double *p, size_t n, double v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// =============================================================================
// k_fbm: the production forward-backward kernel for S <= 176 -- the S x S product of a step on the FP64 matrix cores.
//
// One workgroup advances FOUR restarts of one (chain, direction) in lock step.  The plain-adjacency weights do not
// depend on the restart, so a step is the 4 x S by S x S product  out[i][o] = sum_q a_i[q] W[q][o]  (i = restart).
// `v_mfma_f64_4x4x4f64` is four independent 4x4x4 products ("blocks"; lane 16 k + 4 b + r holds element [r][k] of block
// b's A and [k][r] of its B, lane 16 r + 4 b + c element [r][c] of its result): with the vectors as the A operand of all
// four blocks and four column groups of W as the B operands, one instruction multiplies 4 restarts x 4 reduction rows x
// 16 output columns.  Measured (tools/micro/mfma64_bench.hip): 16 FMA / cycle / SIMD, 20 % above what v_fmac_f64
// reaches, at 1/4 of the issue slots of the vector formulation and without cross-lane broadcasts.
//   * wave w < NW = ceil(S / 15) owns output columns 15 w .. 15 w + 14 for the WHOLE reduction index: W[:, cols] sits in
//     its registers as KB = ceil(S / 4) B operands (lane 16 k + c holds W[4 kb + k][15 w + c]), so a column's sum is finished
//     inside one lane -- no partial sums through LDS, no cross-lane traffic, ONE barrier per step; four accumulators
//     (k-blocks mod 4) keep the matrix pipe's dependent-issue latency covered and are added in a fixed order;
//   * the 16th column of every wave's tile is an all-ones column: its result is sum_q a_i[q], the sum of the PREVIOUS row,
//     for free (165 states = 11 waves x 15 columns).  Any positive per-row factor is a legal scale; the one
//     used is the power of two 2^e <= sum < 2^(e+1): exact, and available to the whole DPP row through one row_newbcast --
//     no maximum reduction, no LDS atomics, no second barrier;
//   * the A operand of k-block kb (lane 16 k + 4 b + i holds a_i[4 kb + k]) is one conflict-free ds_read_b64 (a broadcast
//     over b) from the restart-interleaved vector image in LDS; k-blocks are stored in pairs so that one ds_read_b128 brings a
//     lane its elements of two; the reads are issued through untracked asm FBM_DEPTH pairs ahead of the MFMAs that consume
//     them, into a ring of registers, and retired by counted lgkmcnt waits (LDS returns in order);
//   * result lane 16 i + c holds out_i[15 w + c]: it scales by that power of two, multiplies by its emission value (one
//     global load, issued at the top of the step through untracked asm), stores the row and publishes the new vector
//     element; wave 0's sum lanes store the forward scales for hmm_log_norm_const;
//   * the three waves of a SIMD run at different priorities, so that one wave's result code overlaps the others' products
//     instead of all three leaving the matrix pipe idle together (tools/micro/fbm_loop_bench.hip);
//   * breakend steps (2 % of the steps): the weights are restart-specific, W_i[q][o] = W[q][o] * tab2_i[idx(q,o)] (the plain weight
//     times an entry of the clone-product table of k_brk_lut, rescaled in LDS when it has landed: LDS-DMA, a run of plain steps
//     ahead).  Four restarts with four different B operands leave an MFMA one useful row in four, so these steps run on the vector
//     ALU: lane (kq, c) multiplies its resident weights w[kb] by the table entries (16-bit row offsets per pair in LDS) and by its
//     rows' vector elements of all four restarts; the four lanes of a column are added at the end.
// Summation order is fixed: repeated runs are bit-identical.
// grid (chains of one state-table class, ceil(restarts / 4), 2 directions), block 64 NW.
// =============================================================================
struct FbmArgs {
    int S, SP, M, D, C, N, NBE, cn_max, r0, r1, PE2P, SPC, VR, pad_;      // pad_: the transition model of the tables (0 / 1)
    double pen;
    const int32_t *chain_start, *chain_end, *chain_list, *chain_tc, *chain_cls, *be_n, *chain_be;
    const double *fe, *Wf, *Wb, *pe2_lt;
    const int8_t *af, *ab, *tot;
    double *fa, *fb, *mrow;
    uint32_t *err;
    unsigned long long *dbg;
    const int4 *items;      // k_fbm: one workgroup each -- {chain, first restart of the unit, restarts per workgroup (1 / 2 / 4), direction}
};
// maximum over the 16 lanes of a DPP row (every lane of the row gets it)
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    return v;
}
#define FBM_NV 4
#define FBM_DEPTH 4          // A operands in flight, in PAIRS of k-blocks
#define FBM_RING (FBM_DEPTH + 1)
typedef double fbm_d2 __attribute__((ext_vector_type(2)));
// LDS read / wait the COMPILER does not track (see gload8): the value is valid only after fbm_wait<younger reads in flight>
template <int OFF> __device__ __forceinline__ void fbm_rd(fbm_d2 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT> __device__ __forceinline__ void fbm_wait(fbm_d2 &x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(CNT) : "memory"); }
// position (in doubles) of vector element (state q, restart i) in the LDS image: k-blocks in pairs, [pair][k][i][parity], so
// that one 16-byte read gives a lane its A elements of two consecutive k-blocks
__device__ __forceinline__ int fbm_pos(int q, int i) { return ((((q >> 3) * 4 + (q & 3)) * 4 + i) << 1) + ((q >> 2) & 1); }
// the products of a plain step: pair PI of KB / 2 pairs of k-blocks.  The A operands arrive through a ring of FBM_RING
// registers: pair PI + FBM_DEPTH is requested into the slot whose MFMAs were issued one pair ago.
template <int PI, int KB> struct fbm_chain {
    static __device__ __forceinline__ void run(fbm_d2 (&ring)[FBM_RING], const double (&w)[KB], unsigned addr, double (&acc)[4]) {
        constexpr int NP = KB / 2;
        constexpr int younger = (NP - 1 - PI) < (FBM_DEPTH - 1) ? (NP - 1 - PI) : (FBM_DEPTH - 1);
        fbm_wait<younger>(ring[PI % FBM_RING]);
        acc[(2 * PI) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[PI % FBM_RING].x, w[2 * PI], acc[(2 * PI) & 3], 0, 0, 0);
        acc[(2 * PI + 1) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(ring[PI % FBM_RING].y, w[2 * PI + 1], acc[(2 * PI + 1) & 3], 0, 0, 0);
        if constexpr (PI + FBM_DEPTH < NP) fbm_rd<(PI + FBM_DEPTH) * 256>(ring[(PI + FBM_DEPTH) % FBM_RING], addr);
        if constexpr (PI + 1 < NP) fbm_chain<PI + 1, KB>::run(ring, w, addr, acc);
    }
    template <int I> static __device__ __forceinline__ void fill(fbm_d2 (&ring)[FBM_RING], unsigned addr) {
        fbm_rd<I * 256>(ring[I], addr);
        if constexpr (I + 1 < FBM_DEPTH && I + 1 < KB / 2) fill<I + 1>(ring, addr);
    }
};

// The same products on the vector ALU, for workgroups that carry NV = 1 or 2 restarts (k_fbm<KB, NV>): lane (kq, c) owns rows 4 kb + kq of
// column c -- its B-operand registers -- and accumulates acc_j += a_j[4 kb + kq] w[kb].  The vector elements never come as LDS
// broadcasts (one read per k-block is bound by LDS latency: a pair of FMAs is ten cycles, a read a hundred): lane c of a DPP row holds
// a_j[4 (16 g + c) + kq] for g < ceil(KB / 16) -- 3 eight-byte reads per restart and step -- and the FMA of k-block kb takes its multiplier
// from lane kb % 16 of the row through its own DPP operand (row_newbcast).  Even and odd k-blocks go to two accumulators per restart.
// WAIT STATES (asm site 2 of 5): the accumulators written here reach a matrix instruction (the row-group reduce of the vector forms) only
// through a compiler-visible v_add_f64 (even + odd k-blocks, fbw_chain / fbw_be): VALU -> VALU is interlocked, and the compiler pads its own
// v_add -> v_mfma.  The DPP source `a` is an LDS read's destination.  tools/asm_hazards.py verifies both on the built code.
template <int J> __device__ __forceinline__ void fbm_vfma(double &acc, const double a, const double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(x), "n"(J));
}
#define FBW_G(KB_) (((KB_) + 15) / 16)
template <int KBI, int KB, int NV> struct fbw_chain {
    static __device__ __forceinline__ void run(const double (&av)[NV][FBW_G(KB)], const double (&w)[KB], double (&acc)[NV][2]) {
        fbm_vfma<KBI % 16>(acc[0][KBI & 1], av[0][KBI / 16], w[KBI]);
        if constexpr (NV > 1) fbm_vfma<KBI % 16>(acc[1][KBI & 1], av[1][KBI / 16], w[KBI]);
        if constexpr (KBI + 1 < KB) fbw_chain<KBI + 1, KB, NV>::run(av, w, acc);
    }
};
// position (in doubles) of state q in a restart's plain vector image: 64-blocks, inside a block the four row groups kq one after the
// other -- lane (kq, c) reads 64 g + 16 kq + c, a wave consecutive doubles
__device__ __forceinline__ int fbw_pos(int q) { return 64 * (q >> 6) + 16 * (q & 3) + ((q >> 2) & 15); }
// a breakend step's products in that form: pair P of k-blocks; the weight of (row, column) is the resident plain weight times the
// restart's rescaled clone-product entry (tab2: rows of a quad's four restarts, 48 bytes apart; the pair's two 16-bit row offsets in one word)
template <int P, int KB, int NV> struct fbw_be {
    static __device__ __forceinline__ void run(const double (&av)[NV][FBW_G(KB)], const double (&w)[KB], const unsigned *cw, const int pstride,
                                               const char *tbb, double (&acc)[NV][2]) {
        const unsigned c_ = cw[(size_t)P * pstride];
        const char *r0_ = tbb + (c_ & 0xffffu), *r1_ = tbb + (c_ >> 16);
        double t0[NV], t1[NV];
        if constexpr (NV == 2) {
            const double2 u0_ = *reinterpret_cast<const double2 *>(r0_), u1_ = *reinterpret_cast<const double2 *>(r1_);
            t0[0] = u0_.x; t0[1] = u0_.y; t1[0] = u1_.x; t1[1] = u1_.y;
        } else { t0[0] = *reinterpret_cast<const double *>(r0_); t1[0] = *reinterpret_cast<const double *>(r1_); }
        constexpr int k0 = 2 * P, k1 = 2 * P + 1;
        fbm_vfma<k0 % 16>(acc[0][0], av[0][k0 / 16], w[k0] * t0[0]);
        if constexpr (NV > 1) fbm_vfma<k0 % 16>(acc[1][0], av[1][k0 / 16], w[k0] * t0[1]);
        fbm_vfma<k1 % 16>(acc[0][1], av[0][k1 / 16], w[k1] * t1[0]);
        if constexpr (NV > 1) fbm_vfma<k1 % 16>(acc[1][1], av[1][k1 / 16], w[k1] * t1[1]);
        if constexpr (P + 1 < KB / 2) fbw_be<P + 1, KB, NV>::run(av, w, cw, pstride, tbb, acc);
    }
};

// one workgroup: chain `chain`, direction `dir`, the unit of NV restarts that starts at restart `rg0` (a multiple of NV)
template <int KB, int NV>
__device__ __forceinline__ void fbm_body(const FbmArgs &a, const int chain, const int rg0, const int dir) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    static_assert(KB % 2 == 0 && KB / 2 >= FBM_DEPTH, "k-blocks come in pairs; ring no deeper than the chain");
    static_assert(NV == 1 || NV == 2 || NV == 4, "restarts per workgroup");
    // restarts in absolute units of NV (NV u .. NV u + NV - 1) inside absolute quads (4 g .. 4 g + 3: the interleave of the breakend
    // tables); of a unit, those inside [r0, r1).  The unit's restarts take slots 0 .. NV - 1 of the vector image.
    const int quad = rg0 >> 2, I0 = rg0 & 3;
    const int v_lo = max(a.r0 - rg0, 0), v_hi = min(a.r1 - rg0, NV);          // present: v_lo <= i < v_hi
    const int S = a.S, SP = a.SP, M = a.M, D = a.D, VR = a.VR, SPC = a.SPC;
    const int n0 = a.chain_start[chain], n1 = a.chain_end[chain], len = n1 - n0 + 1;
    constexpr int G = FBW_G(KB), VRP = 64 * G;                  // (NV < 4) groups of 16 k-blocks, doubles of a restart's plain vector image
    const int t = threadIdx.x, NT = blockDim.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), NW = NT >> 6;
    const int kq = lane >> 4, c16 = lane & 15, ib = lane & 3;   // operand roles: k inside the k-block, column inside the tile, restart of the A element
    const int id = lane >> 4;                                   // result role: restart (numerically kq)
    const bool is_sum = c16 == 15;                              // the tile's 16th column is the all-ones column: its result is the vector's sum
    const int col = wave * 15 + c16;                            // (state columns 15 w .. 15 w + 14)
    const double sel0 = ib == 0 ? 1.0 : 0.0, sel1 = ib == 1 ? 1.0 : 0.0;   // (NV < 4) A operands that pick restart slot 0 / 1 as result row
    (void)sel0; (void)sel1;
    // ---- LDS carve-up ---------------------------------------------------------------------------
    double *vec = (double *)smem_raw;                           // [2][VR][4]   vectors, restart-interleaved, double-buffered by step parity
    double *tab = vec + (size_t)2 * VR * 4;                     // [2][PE2P][4] clone-product weights of the current and the next breakend, restart-interleaved (LDS-DMA)
    double *wa = tab + (size_t)2 * 4 * a.PE2P;                  // [64]         exp(+pen * kt), kt = sum of the absolute total differences a table entry stands for
    double *tab2 = wa + 64;                                     // [PE2P][6]    this breakend step's rescaled table, rows of four restarts at a 48-byte stride (16 bank slots instead of 8)
    unsigned *codel = (unsigned *)(tab2 + (size_t)6 * a.PE2P);  // [KB / 2][4][SPC] pair codes of (row q -> column o), two k-blocks per word
    int *bel = (int *)(codel + (size_t)(KB / 2) * 4 * SPC);     // adjacencies of this chain's breakends
    const int be_lo = a.chain_be[2 * chain], be_hi = a.chain_be[2 * chain + 1];
    if (be_hi > be_lo) {
        // code of the pair (row q -> column o) at a breakend adjacency of this chain and direction: 48 x the index of the
        // clone-product weight (differences of the tumour clones' totals; the normal clone's is 0 inside a class) = the byte offset of
        // its row of four restarts in the interleaved table; word [p][k][o] holds rows 4 (2 p) + k and 4 (2 p + 1) + k.
        // Rows / columns past S: 0.  (The allele distance is not needed: the step multiplies the PLAIN weight, below.)
        const int8_t *tg = a.tot + (size_t)a.chain_cls[chain] * S * M;
        const int off_ = a.cn_max + 1, sg_ = dir == 0 ? 1 : -1;
        const unsigned ones_code = (unsigned)(M == 2 ? D : D * D);      // table entry n2 holds 1 (k_brk_lut)
        for (int i = t; i < (KB / 2) * 4 * SPC; i += NT) {
            const int ls = i % SPC, pk_ = i / SPC, k_ = pk_ & 3, p_ = pk_ >> 2;      // lane slot 16 w + c: state column 15 w + c, c = 15: the ones column
            const int o = (ls >> 4) * 15 + (ls & 15);
            unsigned word = 0;
            for (int h = 0; h < 2; h++) {
                const int q = 4 * (2 * p_ + h) + k_;
                if ((ls & 15) == 15) word |= (ones_code * 48u) << (16 * h);
                else if (q < S && o < S) {
                    int idx = 0;
                    for (int c = 1; c < M; c++) idx = idx * D + sg_ * ((int)tg[q * M + c] - (int)tg[o * M + c]) + off_;
                    word |= ((unsigned)idx * 48u) << (16 * h);
                }
            }
            codel[i] = word;
        }
        for (int i = t; i < be_hi - be_lo; i += NT) bel[i] = a.be_n[be_lo + i];
    }
    for (int i = t; i < 2 * VR * 4; i += NT) vec[i] = 0.;
    for (int i = t; i < 64; i += NT) wa[i] = exp(a.pen * (double)i);      // exp(+pen kt): table entry -> factor on the plain weight (breakend steps)
    // ---- stationary weights: B operands of this wave's 15 columns and the ones column, every k-block ---------------
    double w[KB];
    {
        const double *Wmat = (dir == 0 ? a.Wf : a.Wb) + (size_t)a.chain_tc[chain] * S * S;
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            const int q = 4 * kb + kq;
            w[kb] = q < S ? (is_sum ? 1.0 : (col < S ? Wmat[(size_t)q * S + col] : 0.)) : 0.;
        }
#pragma unroll
        for (int kb = 0; kb < KB; kb++) asm volatile("" : "+v"(w[kb]));      // their loads retire here, not inside the step loop
    }
    __syncthreads();

#define ROW(k) (dir == 0 ? n0 + (k) : n1 - (k))
#define ADJ(k) (dir == 0 ? n0 + (k) - 1 : n1 - (k))
    const int be_step = dir == 0 ? 1 : -1;
    int be_i = dir == 0 ? be_lo : be_hi - 1;                                   // slot of the next breakend step
    int be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2;   // its adjacency
    const int rstep = dir == 0 ? SP : -SP;
    const bool mine = !is_sum && (NV == 4 ? col < VR : (col < VRP && id < NV));      // this lane publishes a vector element (possibly a padding column: zero)
    const bool present = id >= v_lo && id < v_hi;
    const bool live = !is_sum && col < S && present;            // ... of an existing column and restart
    const size_t lane_off = ((size_t)(rg0 + (present ? id : v_lo)) * a.N + ROW(0)) * SP + (col < S ? col : S - 1);
    double *outp = (dir == 0 ? a.fa : a.fb) + lane_off;
    // The emission value a result lane needs in step k: one 8-byte global load issued at the top of the step through untracked asm, consumed after
    // the products.  (Round 4 measured the alternative -- rows staged three steps ahead in an LDS ring by LDS-DMA from one wave, retired by counted
    // vmcnt -- and dropped it: with one restart per workgroup a step without ANY emission load is 80 cycles shorter than with this load, the ring's
    // transfers cost the issuing wave 60 cycles each, and every shape got slower: 2 950 -> 3 410 cycles per step at four restarts per workgroup.)
    const double *eptr = a.fe + lane_off + rstep;
#define FBM_EGET(e_, k_) double e_; gload8(e_, eptr); eptr += rstep;
    // vector image: NV = 4 the MFMA A-operand layout (fbm_pos), buffers VR * 4 doubles apart; NV < 4 one plain image of VRP doubles per restart (fbw_pos)
    const int vbuf = NV == 4 ? VR * 4 : NV * VRP;                // doubles between the two buffers (step parity)
    double *vput = vec + (NV == 4 ? fbm_pos(mine ? col : 0, id) : (mine ? id * VRP + fbw_pos(col) : 0));      // this lane's element of the vector image (buffer 0)
    const unsigned ap0 = lds_addr(vec + (kq * 4 + ib) * 2);      // (NV = 4) A operands of k-blocks 0 and 1 (buffer 0); pair p: + 256 p bytes
    const double *avp = vec + 16 * kq + c16;                     // (NV < 4) this lane's vector elements: restart j, group g at + j VRP + 64 g
    const bool scribe = dir == 0 && wave == 0 && is_sum && present;   // this lane records the forward scales of restart id
    double *mptr = a.mrow + (size_t)(rg0 + (present ? id : v_lo)) * a.N + ROW(0);
    // the three waves of a SIMD (w, w + 4, w + 8) at different issue priorities
    if (wave < 4) __builtin_amdgcn_s_setprio(2); else if (wave < 8) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    // The clone-product tables of a breakend step (k_brk_lut: PE2P doubles per restart) travel by LDS-DMA into one of two
    // buffers a whole run of plain steps ahead of the step that reads them: requested right after the previous breakend
    // step (the first one here), retired by the vmcnt(0) every wave executes in each step, published by the steps' barriers.
    const int NCH = (a.PE2P * 32 + 1023) >> 10;                  // 1 KiB pieces of a quad's interleaved table
#define FBM_FETCH(slot_, buf_)                                                                                                     \
    for (int c_ = wave; c_ < NCH; c_ += NW) {                                                                                      \
        const int e16_ = c_ * 64 + lane;                                              /* 16-byte element of the table */          \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_addr(tab + (size_t)(buf_) * 4 * a.PE2P) + (unsigned)(c_ * 1024)); \
        if (e16_ * 2 < a.PE2P * 4) glds16(a.pe2_lt + ((size_t)quad * a.NBE + (slot_)) * a.PE2P * 4 + e16_ * 2, dst_);             \
    }
    int be_buf = 0;                                              // buffer holding the tables of breakend slot be_i
    if (be_adj >= 0) FBM_FETCH(be_i, 0)
    // ---- step 0 ------------------------------------------------------------------------------------------------
    {
        double e0 = 0.;
        if (live) { e0 = a.fe[lane_off]; gstore8(outp, (dir == 0) ? e0 : 1.0); }
        if (mine) *vput = live ? e0 : 0.;
    }
    FB_BARRIER();
    if (a.dbg && t == 0 && blockIdx.x == 0) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); a.dbg[4] = len; }
    // the tail of a step, common to plain and breakend steps: `sum` = this lane's product (lane 15 of every DPP row: the
    // sum of the previous row, from the ones column, whose power of two is this step's scale -- broadcast inside the
    // row, no LDS, no reduction)
#define FBM_FINISH(sum_, e_, k_)                                                                                                   \
    {                                                                                                                              \
        const unsigned hs_ = (unsigned)__builtin_amdgcn_update_dpp(0, __double2hiint(sum_), 0x15F, 0xf, 0xf, false);   /* row_newbcast:15 */ \
        double m_, inv_;                                                                                                           \
        pow2_scale(hs_, m_, inv_);                                                                                                 \
        outp += rstep;                                                                                                             \
        double val_ = (sum_) * inv_;                                                                                               \
        gwait8_after(e_, val_);        /* issued at the top of the step */                                                         \
        const double vecv_ = val_ * (e_);                                                                                          \
        if (mine) vput[(size_t)((k_) & 1) * vbuf] = live ? vecv_ : 0.;      /* (first: its LDS round trip runs under the stores' issue) */ \
        if (live) gstore8(outp, (dir == 0) ? vecv_ : val_);                                                                        \
        if (scribe) gstore8(mptr, m_);                           /* the scale of row k-1 for hmm_log_norm_const */                \
        mptr += dir == 0 ? 1 : -1;                                                                                                 \
        PST(1)                                                                                                                     \
        FB_BARRIER();                                                                                                              \
    }
    // Plain steps run in their own tight loops between the chain's breakend steps: nothing of the breakend code (tracked
    // loads, a second barrier, the walk over the chain's breakends) is inside them -- with it in the same loop body every
    // plain step paid ~750 cycles for the compiler's conservative waits at the merge.
#ifdef RMX_FB_PSTAMPS
    // diagnostic build (tools/fb_only.py PSTAMPS): cycles a wave spends from the top of a plain step to the end of its products (0), from
    // there to its arrival at the barrier (1), and in the barrier (2)
    unsigned long long pst_acc[3] = {0, 0, 0}, pst_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_last) :: "memory");
#define PST(i_) { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); pst_acc[i_] += t_ - pst_last; pst_last = t_; }
#else
#define PST(i_)
#endif
    int k = 1;
    while (k < len) {
        const int k_be = be_adj >= 0 ? (dir == 0 ? be_adj - n0 + 1 : n1 - be_adj) : len;      // step that crosses the next breakend adjacency
        const int k_stop = k_be < len ? k_be : len;
        for (; k < k_stop; k++) {
            PST(2)
            FBM_EGET(e, k)
            double sum;
            if constexpr (NV == 4) {
                fbm_d2 ring[FBM_RING];
                const unsigned apc = ap0 + (unsigned)((k - 1) & 1) * (unsigned)(VR * 32);
                fbm_chain<0, KB>::template fill<0>(ring, apc);
                double acc[4] = {0., 0., 0., 0.};
                fbm_chain<0, KB>::run(ring, w, apc, acc);
                sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            } else {
                double av[NV][G], acc[NV][2];
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc[j][0] = acc[j][1] = 0.;
#pragma unroll
                    for (int g = 0; g < G; g++) av[j][g] = avp[(size_t)((k - 1) & 1) * vbuf + j * VRP + 64 * g];
                }
                fbw_chain<0, KB, NV>::run(av, w, acc);
                // the four row groups kq of a column, and the move to the result lanes (lane 16 i + c: restart slot i), in one matrix
                // instruction per restart: A = (row i == slot j) for every k, B = this lane's partial sum  ->  D[i][c] = [i == j] sum_kq p_j[kq][c]
                sum = __builtin_amdgcn_mfma_f64_4x4x4f64(sel0, acc[0][0] + acc[0][1], 0., 0, 0, 0);
                if constexpr (NV > 1) sum = __builtin_amdgcn_mfma_f64_4x4x4f64(sel1, acc[1][0] + acc[1][1], sum, 0, 0, 0);
            }
            PST(0)
            FBM_FINISH(sum, e, k)
        }
        if (k < len) {
            // ---- breakend step: its tables were requested a run of plain steps ago; every wave has retired its own requests
            // (vmcnt(0) in each step) unless the previous step was a breakend step too
            const double *tb = tab + (size_t)be_buf * 4 * a.PE2P;
#ifdef RMX_FB_STAMPS
            unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_last;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FB_BARRIER();
            FB_STAMP(0)
            be_i += be_step;
            be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2;
            be_buf ^= 1;
            if (be_adj >= 0) FBM_FETCH(be_i, be_buf)            // the next breakend's tables, into the other buffer
            FB_STAMP(1)
            // tab -> tab2 (rescaled, restrided): W_i[q][o] = exp(-pen (k - kt)) tab_i[ix] = W[q][o] tab2_i[ix] with tab2_i[ix] = tab_i[ix] exp(+pen kt(ix)),
            // kt = the absolute total differences ix stands for -- the PLAIN weight, resident in this lane's registers, times one entry
            {
                const double *tw = tab + (size_t)(be_buf ^ 1) * 4 * a.PE2P;      // (be_buf was flipped above: this step's table)
                const int n2_ = M == 2 ? D : D * D, toff_ = a.cn_max + 1;
                for (int e_ = t; e_ <= n2_; e_ += NT) {      // (entry n2: the ones columns' 1)
                    const int da_ = M == 3 ? abs(e_ / D - toff_) : abs(e_ - toff_), db_ = M == 3 ? abs(e_ % D - toff_) : 0;
                    const int kt_ = a.pad_ == 0 ? da_ + db_ : (da_ != 0) + (db_ != 0);      // transition model 0: |d|, 1: [d != 0] (bpmodel.pyx:606-616)
                    const double f_ = e_ < n2_ ? wa[kt_ & 63] : 1.0;
                    const double2 *row = reinterpret_cast<const double2 *>(tw + (size_t)e_ * 4);
                    double2 r01 = row[0], r23 = row[1];
                    r01.x *= f_; r01.y *= f_; r23.x *= f_; r23.y *= f_;
                    double2 *dst = reinterpret_cast<double2 *>(tab2 + (size_t)e_ * 6);
                    dst[0] = r01; dst[1] = r23;
                }
            }
            FB_BARRIER();
            FBM_EGET(e, k)
            // The weights differ per restart, so the four rows of an MFMA cannot share a B operand (four MFMAs per k-block, each with one
            // useful row): the products run on the vector ALU instead.  Lane (kq, c) owns rows 4 kb + kq of column c -- exactly its B-operand
            // registers w[kb] -- and accumulates acc_i += a_i[q] (w[kb] tab2_i[ix(q, c)]) for the four restarts; the four lanes of a column
            // (kq = 0..3) are added at the end in a fixed order.  19 000 -> ~8 000 cycles per breakend step.
            double sum;
            const unsigned *cw = codel + (size_t)kq * SPC + (wave * 16 + c16);
            const char *tbb = reinterpret_cast<const char *>(tab2) + I0 * 8;       // (the unit's restarts inside the quad's rows)
            (void)tb;
            if constexpr (NV < 4) {
                double av[NV][G], acc[NV][2];
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc[j][0] = acc[j][1] = 0.;
#pragma unroll
                    for (int g = 0; g < G; g++) av[j][g] = avp[(size_t)((k - 1) & 1) * vbuf + j * VRP + 64 * g];
                }
                fbw_be<0, KB, NV>::run(av, w, cw, 4 * SPC, tbb, acc);
                // (as in a plain step.  The accumulators are written by inline-asm FMAs the hazard recognizer cannot see into: a matrix
                // instruction must not take one of them as an operand directly -- it read a stale value now and then; the additions are
                // ordinary vector instructions, whose hazards against the matrix instruction the compiler covers)
                sum = __builtin_amdgcn_mfma_f64_4x4x4f64(sel0, acc[0][0] + acc[0][1], 0., 0, 0, 0);
                if constexpr (NV > 1) sum = __builtin_amdgcn_mfma_f64_4x4x4f64(sel1, acc[1][0] + acc[1][1], sum, 0, 0, 0);
            } else {
            double acc[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) acc[j] = 0.;
            // all row offsets first (one LDS round trip), then the pairs with the next pair's operands requested before this pair's
            // products: a pair is otherwise two dependent LDS round trips (offsets, then rows) in front of 4 NV FMAs
            constexpr int NPAIR = KB / 2, CH0 = (NPAIR + 1) / 2;      // the offsets in two chunks (register budget)
            unsigned cpk[CH0];
            // vector elements: the MFMA A-operand image (lane (kq, c) reads a_{c mod 4}[4 kb + kq], two k-blocks per 16 bytes); restart i's
            // element reaches all 16 lanes of the row through the FMA's own DPP operand (row_newbcast:i) -- 16 bytes of LDS per pair
            // instead of 64
            const fbm_d2 *apd = reinterpret_cast<const fbm_d2 *>(vec + (size_t)((k - 1) & 1) * VR * 4 + (kq * 4 + ib) * 2);     // pair p: + 16 p
            fbm_d2 an;
            double tn[2][NV];                                          // table entries of the unit's restarts: rows of k-block 2 p and 2 p + 1
#define FBM_BE_LOAD(p_, c_)                                                                                                        \
            {                                                                                                                      \
                an = apd[(p_) * 16];                                                                                               \
                const char *r0_ = tbb + ((c_) & 0xffffu), *r1_ = tbb + ((c_) >> 16);                                               \
                if constexpr (NV == 4) {                                                                                           \
                    const double2 u0_ = *reinterpret_cast<const double2 *>(r0_), u1_ = *reinterpret_cast<const double2 *>(r0_ + 16); \
                    const double2 u2_ = *reinterpret_cast<const double2 *>(r1_), u3_ = *reinterpret_cast<const double2 *>(r1_ + 16); \
                    tn[0][0] = u0_.x; tn[0][1] = u0_.y; tn[0][2] = u1_.x; tn[0][3] = u1_.y;                                        \
                    tn[1][0] = u2_.x; tn[1][1] = u2_.y; tn[1][2] = u3_.x; tn[1][3] = u3_.y;                                        \
                } else if constexpr (NV == 2) {                                                                                    \
                    const double2 u0_ = *reinterpret_cast<const double2 *>(r0_), u2_ = *reinterpret_cast<const double2 *>(r1_);    \
                    tn[0][0] = u0_.x; tn[0][1] = u0_.y; tn[1][0] = u2_.x; tn[1][1] = u2_.y;                                        \
                } else {                                                                                                           \
                    tn[0][0] = *reinterpret_cast<const double *>(r0_); tn[1][0] = *reinterpret_cast<const double *>(r1_);          \
                }                                                                                                                  \
            }
#pragma unroll
            for (int c0 = 0; c0 < NPAIR; c0 += CH0) {
#pragma unroll
                for (int u = 0; u < CH0; u++) if (c0 + u < NPAIR) cpk[u] = cw[(size_t)(c0 + u) * 4 * SPC];
                FBM_BE_LOAD(c0, cpk[0])
#pragma unroll
                for (int u = 0; u < CH0; u++) {
                    const int p = c0 + u;
                    if (p >= NPAIR) break;
                    const double ax = an.x, ay = an.y;                                 // k-blocks 2 p and 2 p + 1
                    double t0[NV], t1[NV];                                             // rows of k-block 2 p and 2 p + 1
#pragma unroll
                    for (int j = 0; j < NV; j++) { t0[j] = tn[0][j]; t1[j] = tn[1][j]; }
                    if (u + 1 < CH0 && p + 1 < NPAIR) FBM_BE_LOAD(p + 1, cpk[u + 1])
                    const double w0 = w[2 * p], w1 = w[2 * p + 1];
                    double xs[NV], ys[NV];
#pragma unroll
                    for (int j = 0; j < NV; j++) { xs[j] = w0 * t0[j]; ys[j] = w1 * t1[j]; }
                    fbm_vfma<0>(acc[0], ax, xs[0]);
                    if constexpr (NV > 1) fbm_vfma<1>(acc[1], ax, xs[1]);
                    if constexpr (NV > 2) { fbm_vfma<2>(acc[2], ax, xs[2]); fbm_vfma<3>(acc[3], ax, xs[3]); }
                    fbm_vfma<0>(acc[0], ay, ys[0]);
                    if constexpr (NV > 1) fbm_vfma<1>(acc[1], ay, ys[1]);
                    if constexpr (NV > 2) { fbm_vfma<2>(acc[2], ay, ys[2]); fbm_vfma<3>(acc[3], ay, ys[3]); }
                }
            }
#undef FBM_BE_LOAD
            // the four row groups of a column: ((kq 0 + kq 1) + (kq 2 + kq 3)), the same value in all four lanes (a + b == b + a)
#pragma unroll
            for (int j = 0; j < NV; j++) { acc[j] += __shfl_xor(acc[j], 16); acc[j] += __shfl_xor(acc[j], 32); }
            sum = acc[0];                                                              // result lane (i, c) keeps restart slot i
#pragma unroll
            for (int j = 1; j < NV; j++) if (id == j) sum = acc[j];
            }
            FB_STAMP(2)
            FBM_FINISH(sum, e, k)
            FB_STAMP(3)
#ifdef RMX_FB_STAMPS
            if (a.dbg && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0) { for (int i = 0; i < 4; i++) a.dbg[8 + (wave >> 2) * 6 + i] += stamp_acc[i]; if (wave == 0) a.dbg[5] += 1; }
#endif
            k++;
        }
    }
#undef FBM_FINISH
#undef FBM_FETCH
#undef FBM_EGET
#ifdef RMX_FB_PSTAMPS
    if (a.dbg && blockIdx.x == 0 && lane == 0) {
        const int slot = wave == 0 ? 0 : (wave == 4 ? 1 : (wave == 8 ? 2 : (wave == 3 ? 3 : (wave == 7 ? 4 : (wave == NW - 1 ? 5 : -1)))));
        if (slot >= 0) for (int i = 0; i < 3; i++) a.dbg[8 + slot * 3 + i] = pst_acc[i];
    }
#endif
#undef PST
    if (a.dbg && t == 0 && blockIdx.x == 0) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
    // last row of each chain: its scale is not consumed by a later step, but the vanishing-row check needs the row's sum
    if (wave == 0) {
        const double *vb = vec + (size_t)((len - 1) & 1) * vbuf;
        double ps = 0.;
        for (int q = c16; q < S; q += 16) ps += vb[NV == 4 ? fbm_pos(q, id) : (id < NV ? id : 0) * VRP + fbw_pos(q)];
        ps = group_sum(ps, 16);
        if (c16 == 0 && present) {
            double m_, inv;
            pow2_scale((unsigned)__double2hiint(ps), m_, inv);
            if (dir == 0) gstore8(a.mrow + (size_t)(rg0 + id) * a.N + ROW(len - 1), m_);
            if (!(m_ > 0.) || m_ == INFINITY) atomicOr(&a.err[rg0 + id], RMX_ERR_NAN_AB);
        }
    }
#undef ROW
#undef ADJ
}
// grid (work items): every workgroup takes its chain, unit of restarts and direction from a.items -- and with them its SHAPE: the host gives a
// long chain fewer restarts per workgroup (shorter steps) than a short one, so that the workgroups of a launch end together (rmx_api.hip
// fb_items).  The three shapes are three bodies in one kernel; a workgroup runs one of them from start to end.
template <int KB>
__global__ __launch_bounds__(768) void k_fbm(FbmArgs a) {
    const int4 it = a.items[blockIdx.x];
    if (it.z == 4) fbm_body<KB, 4>(a, it.x, it.y, it.w);
    else if (it.z == 2) fbm_body<KB, 2>(a, it.x, it.y, it.w);
    else fbm_body<KB, 1>(a, it.x, it.y, it.w);
}

// =============================================================================
// k_fbq: the matrix-core forward-backward kernel for state grids whose S x S weights do not fit the register file
// (176 < S <= 360: 355 states at the reference's default max_copy_number = 12, the "~400 states" of BASELINE's metric).
//
// Same step as k_fbm -- four restarts of one (chain, direction) per workgroup, out[i][o] = sum_q a_i[q] W[q][o] on
// v_mfma_f64_4x4x4f64, column-owner waves, the ones column for the row scale, one barrier per step -- but the B operands are
// not resident: a plain-adjacency weight is exp(-pen k) with k = min(SAD(cn_q, cn_o), SAD(cn_q, swap_alleles(cn_o))) < 64
// (k_fbk's closed form, verified by the host against the tabulated log-weights: fbk_ok), so what a wave keeps in registers is
// the SMALL INTEGER k of its column tiles for the whole reduction index (KB 16-bit fields per lane and tile instead of KB doubles),
// and every B operand is one ds_read_b64 from a table of exp(-pen k) in LDS.  The table is stored 32 times, entry k of copy j at
// byte 256 k + 8 j, and lane l reads copy l mod 32: whatever the k's of a wave are, the 32 lanes of an LDS lane group hit 32
// different bank pairs (measured with one copy: +5 200 cycles per step of bank conflicts on top of 10 400; tools/fbq_variants.sh).
// Per MFMA the LDS moves 512 B for the B operand and 256 B for the A operand (one ds_read_b128 feeds two k-blocks of BOTH tiles of
// the wave): 3 of the 4 LDS cycles a CU has per MFMA at the matrix pipe's full rate (MI355X_MICROARCH.md, LDS: 256 B per clock).
//   * wave w owns TWO tiles of 15 state columns + a ones column: columns 30 w .. 30 w + 29 (12 waves at 355 states);
//   * all LDS reads of the plain steps go through untracked asm into a ring, FBQ_DEPTH pairs of k-blocks ahead of the
//     MFMAs that consume them (5 reads per pair: the A pair and four table lookups), retired by counted lgkmcnt waits;
//   * breakend steps (2 % of the steps): W_i[q][o] = W[q][o] tab2_i[idx(tot_q - tot_o)] -- the plain weight (the same lookup) times an
//     entry of the quad's interleaved clone-product table (LDS-DMA a run of plain steps ahead, rescaled in LDS when it has landed;
//     its byte address is one v_mad of the row's and the column's totals index) -- on the vector ALU, as in k_fbm.
// Summation order is fixed: repeated runs are bit-identical.
// grid (chains of one state-table class, ceil(restarts / 4), 2 directions), block 64 ceil(S / 30).
// =============================================================================
#define FBQ_DEPTH 2
#define FBQ_RING (FBQ_DEPTH + 1)
struct fbq_slot { fbm_d2 a; double b[4]; };      // A operands of a pair of k-blocks; B operands [k-block of the pair][tile]
__device__ __forceinline__ void fbq_rd64(double &dst, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
template <int CNT> __device__ __forceinline__ void fbq_wait(fbq_slot &x) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(x.a), "+v"(x.b[0]), "+v"(x.b[1]), "+v"(x.b[2]), "+v"(x.b[3]) : "n"(CNT) : "memory");
}
// A lane keeps the distance k of (row 4 kb + kq, its column) as the 16-bit LDS address of ITS copy of exp(-pen k) -- 256 k + 8 (lane
// mod 32), the table at LDS address 0 -- two k-blocks per register: one VALU operation per B-operand address.
#define FBQ_ADDR(c_, kb_) (((kb_) & 1) ? ((c_)[(kb_) >> 1] >> 16) : ((c_)[(kb_) >> 1] & 0xffffu))
template <int KB> struct fbq_chain {
    static constexpr int NP = KB / 2, NW32 = KB / 2;
    // requests of pair P: the A pair, then the table entries of k-blocks 2 P, 2 P + 1 for tiles 0 and 1
    template <int P> static __device__ __forceinline__ void issue(fbq_slot &s, unsigned apc, unsigned wt, const unsigned (&c0)[NW32], const unsigned (&c1)[NW32]) {
        constexpr int k0 = 2 * P, k1 = 2 * P + 1;
        (void)wt;      // the table sits at LDS address 0 (checked at kernel entry): the field is the address
        // The four address extractions back to back, THEN the reads: a vector instruction between two FP64 MFMAs costs ~9 cycles
        // of matrix-pipe time, four in a row ~5 each (tools/micro/mfma64_bench.hip, "beside other work")
        unsigned a0 = FBQ_ADDR(c0, k0), a1 = FBQ_ADDR(c1, k0), a2 = FBQ_ADDR(c0, k1), a3 = FBQ_ADDR(c1, k1);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        __builtin_amdgcn_sched_barrier(0);
        fbm_rd<P * 256>(s.a, apc);
        fbq_rd64(s.b[0], a0);
        fbq_rd64(s.b[1], a1);
        fbq_rd64(s.b[2], a2);
        fbq_rd64(s.b[3], a3);
    }
    template <int P> static __device__ __forceinline__ void run(fbq_slot (&ring)[FBQ_RING], unsigned apc, unsigned wt, const unsigned (&c0)[NW32], const unsigned (&c1)[NW32],
                                                                double (&acc)[4]) {
        constexpr int younger = (NP - 1 - P) < (FBQ_DEPTH - 1) ? (NP - 1 - P) : (FBQ_DEPTH - 1);
        fbq_slot &s = ring[P % FBQ_RING];
        fbq_wait<5 * younger>(s);
        acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(s.a.x, s.b[0], acc[0], 0, 0, 0);      // tile 0, even k-block
        acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(s.a.x, s.b[1], acc[1], 0, 0, 0);      // tile 1, even k-block
        acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(s.a.y, s.b[2], acc[2], 0, 0, 0);      // tile 0, odd k-block
        acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(s.a.y, s.b[3], acc[3], 0, 0, 0);      // tile 1, odd k-block
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P + FBQ_DEPTH < NP) issue<P + FBQ_DEPTH>(ring[(P + FBQ_DEPTH) % FBQ_RING], apc, wt, c0, c1);
        if constexpr (P + 1 < NP) run<P + 1>(ring, apc, wt, c0, c1, acc);
    }
    template <int I> static __device__ __forceinline__ void fill(fbq_slot (&ring)[FBQ_RING], unsigned apc, unsigned wt, const unsigned (&c0)[NW32], const unsigned (&c1)[NW32]) {
        issue<I>(ring[I], apc, wt, c0, c1);
        if constexpr (I + 1 < FBQ_DEPTH && I + 1 < NP) fill<I + 1>(ring, apc, wt, c0, c1);
    }
};

// k_fbq with NV = 1 or 2 restarts per workgroup: the same lookups, the products on the vector ALU (k_fbm's fbw_chain: plain vector images,
// the multiplier of k-block kb from lane kb % 16 of the DPP row).  Per k-block and tile one address extraction, one table lookup and NV FMAs;
// the lookups are ordinary (tracked) LDS loads, a few k-blocks ahead of their FMAs (the scheduling barriers bound the look-ahead).
#define FBQW_DEPTH 2
#define FBQW_RING (FBQW_DEPTH + 1)
struct fbqw_slot { double b[4]; };               // B operands [k-block of the pair][tile]
template <int CNT> __device__ __forceinline__ void fbqw_wait(fbqw_slot &x) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x.b[0]), "+v"(x.b[1]), "+v"(x.b[2]), "+v"(x.b[3]) : "n"(CNT) : "memory");
}
template <int KB, int NV> struct fbqw_chain {
    static constexpr int NP = KB / 2;
    // the four lookups of pair P (k-blocks 2 P, 2 P + 1, tiles 0 and 1) through untracked reads, FBQW_DEPTH pairs ahead of the FMAs that consume
    // them, retired by counted waits -- left to the compiler the reads were waited for one by one (a third of the FMAs behind an lgkmcnt(0))
    template <int P> static __device__ __forceinline__ void issue(fbqw_slot &s, const unsigned (&c0)[KB / 2], const unsigned (&c1)[KB / 2]) {
        constexpr int k0 = 2 * P, k1 = 2 * P + 1;
        unsigned a0 = FBQ_ADDR(c0, k0), a1 = FBQ_ADDR(c1, k0), a2 = FBQ_ADDR(c0, k1), a3 = FBQ_ADDR(c1, k1);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        __builtin_amdgcn_sched_barrier(0);
        fbq_rd64(s.b[0], a0); fbq_rd64(s.b[1], a1); fbq_rd64(s.b[2], a2); fbq_rd64(s.b[3], a3);
    }
    template <int P> static __device__ __forceinline__ void run(fbqw_slot (&ring)[FBQW_RING], const double (&av)[NV][FBW_G(KB)], const unsigned (&c0)[KB / 2],
                                                                const unsigned (&c1)[KB / 2], double (&acc0)[NV][2], double (&acc1)[NV][2]) {
        constexpr int younger = (NP - 1 - P) < (FBQW_DEPTH - 1) ? (NP - 1 - P) : (FBQW_DEPTH - 1);
        constexpr int k0 = 2 * P, k1 = 2 * P + 1;
        fbqw_slot &s = ring[P % FBQW_RING];
        fbqw_wait<4 * younger>(s);
        // (one restart: even and odd k-blocks in two accumulators per tile; two restarts: one accumulator per tile and restart -- four
        // independent FMA chains either way, and the register file has no room for eight)
        constexpr int O = NV > 1 ? 0 : 1;
        fbm_vfma<k0 % 16>(acc0[0][0], av[0][k0 / 16], s.b[0]);
        if constexpr (NV > 1) fbm_vfma<k0 % 16>(acc0[1][0], av[1][k0 / 16], s.b[0]);
        fbm_vfma<k0 % 16>(acc1[0][0], av[0][k0 / 16], s.b[1]);
        if constexpr (NV > 1) fbm_vfma<k0 % 16>(acc1[1][0], av[1][k0 / 16], s.b[1]);
        fbm_vfma<k1 % 16>(acc0[0][O], av[0][k1 / 16], s.b[2]);
        if constexpr (NV > 1) fbm_vfma<k1 % 16>(acc0[1][O], av[1][k1 / 16], s.b[2]);
        fbm_vfma<k1 % 16>(acc1[0][O], av[0][k1 / 16], s.b[3]);
        if constexpr (NV > 1) fbm_vfma<k1 % 16>(acc1[1][O], av[1][k1 / 16], s.b[3]);
        // (the FMAs are plain asm: pinned between the wait above and the slot's next request below by their operands, and here)
        // WAIT STATES (asm site 3 of 5): fbm_vfma's products at 355 states; the accumulators meet the matrix instruction of the row-group reduce behind
        // compiler-visible additions (site 2's rule); the DPP sources are LDS reads' destinations.  tools/asm_hazards.py checks the built code.
        asm volatile("" : "+v"(acc0[0][0]), "+v"(acc0[0][O]), "+v"(acc1[0][0]), "+v"(acc1[0][O]));
        if constexpr (NV > 1) asm volatile("" : "+v"(acc0[1][0]), "+v"(acc1[1][0]));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P + FBQW_DEPTH < NP) issue<P + FBQW_DEPTH>(ring[(P + FBQW_DEPTH) % FBQW_RING], c0, c1);
        if constexpr (P + 1 < NP) run<P + 1>(ring, av, c0, c1, acc0, acc1);
    }
    template <int I> static __device__ __forceinline__ void fill(fbqw_slot (&ring)[FBQW_RING], const unsigned (&c0)[KB / 2], const unsigned (&c1)[KB / 2]) {
        issue<I>(ring[I], c0, c1);
        if constexpr (I + 1 < FBQW_DEPTH && I + 1 < NP) fill<I + 1>(ring, c0, c1);
    }
};
// ... and a breakend step's: the plain weight (the same lookup) times the restart's entry of the quad's rescaled clone-product table, whose row is
// sgn 32 (U_q - U_o) + const bytes into the table (k_fbq's scheme), the unit's restarts at + 8 I0
template <int KBI, int KB, int NV> struct fbqw_be {
    static __device__ __forceinline__ void run(const double (&av)[NV][FBW_G(KB)], const char *wtb, const unsigned (&c0)[KB / 2], const unsigned (&c1)[KB / 2],
                                               const int *uplq, const int umul, const int uoff0, const int uoff1, const char *tbb,
                                               double (&acc0)[NV][2], double (&acc1)[NV][2]) {
        const double w0 = *reinterpret_cast<const double *>(wtb + FBQ_ADDR(c0, KBI)), w1 = *reinterpret_cast<const double *>(wtb + FBQ_ADDR(c1, KBI));
        const int uq = __mul24(uplq[4 * KBI], umul);
        const double *r0 = reinterpret_cast<const double *>(tbb + (uq + uoff0)), *r1 = reinterpret_cast<const double *>(tbb + (uq + uoff1));
        fbm_vfma<KBI % 16>(acc0[0][KBI & 1], av[0][KBI / 16], w0 * r0[0]);
        if constexpr (NV > 1) fbm_vfma<KBI % 16>(acc0[1][KBI & 1], av[1][KBI / 16], w0 * r0[1]);
        fbm_vfma<KBI % 16>(acc1[0][KBI & 1], av[0][KBI / 16], w1 * r1[0]);
        if constexpr (NV > 1) fbm_vfma<KBI % 16>(acc1[1][KBI & 1], av[1][KBI / 16], w1 * r1[1]);
        if constexpr (KBI % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        if constexpr (KBI + 1 < KB) fbqw_be<KBI + 1, KB, NV>::run(av, wtb, c0, c1, uplq, umul, uoff0, uoff1, tbb, acc0, acc1);
    }
};

template <int KB, int NV>
__device__ __forceinline__ void fbq_body(const FbmArgs &a, const double *wk, const uint32_t *cnpack, const uint32_t *totpack, const int chain, const int rg0, const int dir) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    static_assert(KB % 2 == 0 && KB / 2 >= FBQ_DEPTH, "k-blocks come in pairs; ring no deeper than the chain");
    constexpr int NW32 = KB / 2;
    static_assert(NV == 1 || NV == 2 || NV == 4, "restarts per workgroup");
    constexpr int G = FBW_G(KB), VRP = 64 * G;                  // (NV < 4) groups of 16 k-blocks, doubles of a restart's plain vector image
    // restarts in absolute units of NV inside absolute quads (k_fbm); the unit's restarts take slots 0 .. NV - 1 of the vector image
    const int quad = rg0 >> 2, I0 = rg0 & 3;
    const int v_lo = max(a.r0 - rg0, 0), v_hi = min(a.r1 - rg0, NV);
    const int S = a.S, SP = a.SP, M = a.M, D = a.D, VR = a.VR;
    const int n0 = a.chain_start[chain], n1 = a.chain_end[chain], len = n1 - n0 + 1;
    const int t = threadIdx.x, NT = blockDim.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), NW = NT >> 6;
    const int kq = lane >> 4, c16 = lane & 15, ib = lane & 3, id = lane >> 4;
    const bool is_sum = c16 == 15;
    const int col0 = (2 * wave) * 15 + c16, col1 = (2 * wave + 1) * 15 + c16;      // this lane's state column in tile 0 / tile 1
    const int cls = a.chain_cls[chain];
    // ---- LDS carve-up ---------------------------------------------------------------------------
    double *wtab = (double *)smem_raw;                          // [64][32]     exp(-pen k), 32 copies (entry k of copy j at 32 k + j), at LDS address 0
    double *vec = wtab + 64 * 32;                                // [2][VR][4]   vectors, restart-interleaved, double-buffered by step parity
    double *tab = vec + (size_t)2 * VR * 4;                     // [2][PE2P][4] clone-product weights of the current and the next breakend (LDS-DMA)
    int *upl = (int *)(tab + (size_t)2 * 4 * a.PE2P);           // [4 KB]       32 U_q of the row states, U = index of the state's tumour totals in the clone-product table's order (0 past S)
    double *wup = (double *)(upl + 4 * KB);                     // [64]         exp(+pen kt): turns a table entry into the factor on the PLAIN weight (below)
    int *bel = (int *)(wup + 64);                               // adjacencies of this chain's breakends
    const int be_lo = a.chain_be[2 * chain], be_hi = a.chain_be[2 * chain + 1];
    for (int i = t; i < be_hi - be_lo; i += NT) bel[i] = a.be_n[be_lo + i];
    for (int i = t; i < 4 * KB; i += NT) {
        const uint32_t tp = i < S ? totpack[(size_t)cls * S + i] : 0u;
        upl[i] = 32 * (M == 3 ? (int)(tp & 0xff) * D + (int)((tp >> 8) & 0xff) : (int)(tp & 0xff));
    }
    for (int i = t; i < 64; i += NT) wup[i] = exp(a.pen * (double)i);
    for (int i = t; i < 2 * VR * 4; i += NT) vec[i] = 0.;
    for (int i = t; i < 64 * 32; i += NT) wtab[i] = wk[(size_t)a.chain_tc[chain] * FBK_WKN + (i >> 5)];
    // ---- this lane's 8-bit distances: rows 4 kb + kq against its two columns, every k-block ---------------------
    // (rows past S multiply vector elements that are always 0, columns past S are never published: their codes only have to
    // be valid table indices; the ones columns' code is 0: weight exp(0) = 1)
    auto swap_alleles = [](uint32_t x) { return ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu); };
    const uint32_t co0 = (!is_sum && col0 < S) ? cnpack[(size_t)cls * S + col0] : 0u, co1 = (!is_sum && col1 < S) ? cnpack[(size_t)cls * S + col1] : 0u;
    const uint32_t co0s = swap_alleles(co0), co1s = swap_alleles(co1);
    const uint32_t to0 = (!is_sum && col0 < S) ? totpack[(size_t)cls * S + col0] : 0u, to1 = (!is_sum && col1 < S) ? totpack[(size_t)cls * S + col1] : 0u;
    unsigned c0[NW32], c1[NW32];
#pragma unroll
    for (int i = 0; i < NW32; i++) { c0[i] = 0u; c1[i] = 0u; }
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const int q = 4 * kb + kq;
        const uint32_t cq = q < S ? cnpack[(size_t)cls * S + q] : 0u;
        unsigned k0 = min(__builtin_amdgcn_sad_u8(cq, co0, 0u), __builtin_amdgcn_sad_u8(cq, co0s, 0u));
        unsigned k1 = min(__builtin_amdgcn_sad_u8(cq, co1, 0u), __builtin_amdgcn_sad_u8(cq, co1s, 0u));
        if (is_sum || q >= S || col0 >= S) k0 = 0u;
        if (is_sum || q >= S || col1 >= S) k1 = 0u;
        c0[kb >> 1] |= (((k0 & 63u) << 8) | ((unsigned)(lane & 31) << 3)) << (16 * (kb & 1));
        c1[kb >> 1] |= (((k1 & 63u) << 8) | ((unsigned)(lane & 31) << 3)) << (16 * (kb & 1));
    }
#pragma unroll
    for (int i = 0; i < NW32; i++) asm volatile("" : "+v"(c0[i]), "+v"(c1[i]));      // their loads retire here, not inside the step loop
    __syncthreads();

#define ROW(k) (dir == 0 ? n0 + (k) : n1 - (k))
    const int be_step = dir == 0 ? 1 : -1;
    int be_i = dir == 0 ? be_lo : be_hi - 1;
    int be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2;
    const int rstep = dir == 0 ? SP : -SP;
    const bool present = id >= v_lo && id < v_hi;
    const bool mine0 = !is_sum && (NV == 4 ? col0 < VR : (col0 < VRP && id < NV)), mine1 = !is_sum && (NV == 4 ? col1 < VR : (col1 < VRP && id < NV));
    const bool live0 = !is_sum && col0 < S && present, live1 = !is_sum && col1 < S && present;
    const size_t row_off = ((size_t)(rg0 + (present ? id : v_lo)) * a.N + ROW(0)) * SP;
    const size_t off0 = row_off + (col0 < S ? col0 : S - 1), off1 = row_off + (col1 < S ? col1 : S - 1);
    double *outp0 = (dir == 0 ? a.fa : a.fb) + off0, *outp1 = (dir == 0 ? a.fa : a.fb) + off1;
    const double *eptr0 = a.fe + off0, *eptr1 = a.fe + off1;
    const int vbuf = NV == 4 ? VR * 4 : NV * VRP;                // doubles between the two buffers (step parity)
    double *vput0 = vec + (NV == 4 ? fbm_pos(mine0 ? col0 : 0, id) : (mine0 ? id * VRP + fbw_pos(col0) : 0));
    double *vput1 = vec + (NV == 4 ? fbm_pos(mine1 ? col1 : 0, id) : (mine1 ? id * VRP + fbw_pos(col1) : 0));
    const unsigned ap0 = lds_addr(vec + (kq * 4 + ib) * 2);
    const double *avp = vec + 16 * kq + c16;                     // (NV < 4) this lane's vector elements: restart j, group g at + j VRP + 64 g
    const double sel0 = ib == 0 ? 1.0 : 0.0, sel1 = ib == 1 ? 1.0 : 0.0;   // (NV < 4) A operands that pick restart slot 0 / 1 as result row
    (void)sel0; (void)sel1; (void)avp; (void)I0;
    const unsigned wt = lds_addr(wtab);
    if (wt != 0u) {      // (uniform) the lookups address the table from LDS address 0: this kernel must not have static LDS in front of it
        if (t == 0) atomicOr(&a.err[rg0 + v_lo], RMX_ERR_NAN_AB);
        return;
    }
    const bool scribe = dir == 0 && wave == 0 && is_sum && present;
    double *mptr = a.mrow + (size_t)(rg0 + (present ? id : v_lo)) * a.N + ROW(0);
    if (wave < 4) __builtin_amdgcn_s_setprio(2); else if (wave < 8) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    const int NCH = (a.PE2P * 32 + 1023) >> 10;
#define FBQ_FETCH(slot_, buf_)                                                                                                     \
    for (int c_ = wave; c_ < NCH; c_ += NW) {                                                                                      \
        const int e16_ = c_ * 64 + lane;                                                                                           \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_addr(tab + (size_t)(buf_) * 4 * a.PE2P) + (unsigned)(c_ * 1024)); \
        if (e16_ * 2 < a.PE2P * 4) glds16(a.pe2_lt + ((size_t)quad * a.NBE + (slot_)) * a.PE2P * 4 + e16_ * 2, dst_);             \
    }
    int be_buf = 0;
    if (be_adj >= 0) FBQ_FETCH(be_i, 0)
    // ---- step 0 ------------------------------------------------------------------------------------------------
    {
        double e0 = 0., e1 = 0.;
        if (live0) { e0 = *eptr0; gstore8(outp0, (dir == 0) ? e0 : 1.0); }
        if (live1) { e1 = *eptr1; gstore8(outp1, (dir == 0) ? e1 : 1.0); }
        if (mine0) *vput0 = live0 ? e0 : 0.;
        if (mine1) *vput1 = live1 ? e1 : 0.;
    }
    eptr0 += rstep; eptr1 += rstep;
    FB_BARRIER();
    if (a.dbg && t == 0 && blockIdx.x == 0) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); a.dbg[4] = len; }
    // tail of a step (k_fbm's FBM_FINISH for two tiles): tile 0's ones column holds the sum of the previous row
#define FBQ_FINISH(s0_, s1_, e0_, e1_, k_)                                                                                         \
    {                                                                                                                              \
        const unsigned hs_ = (unsigned)__builtin_amdgcn_update_dpp(0, __double2hiint(s0_), 0x15F, 0xf, 0xf, false);   /* row_newbcast:15 */ \
        double m_, inv_;                                                                                                           \
        pow2_scale(hs_, m_, inv_);                                                                                                 \
        outp0 += rstep; outp1 += rstep;                                                                                            \
        double val0_ = (s0_) * inv_, val1_ = (s1_) * inv_;                                                                         \
        gwait8_after(e0_, val0_); gwait8_after(e1_, val1_);      /* (behind the products: see gwait8_after) */                     \
        const double vv0_ = val0_ * (e0_), vv1_ = val1_ * (e1_);                                                                   \
        if (live0) gstore8(outp0, (dir == 0) ? vv0_ : val0_);                                                                      \
        if (live1) gstore8(outp1, (dir == 0) ? vv1_ : val1_);                                                                      \
        if (mine0) vput0[(size_t)((k_) & 1) * vbuf] = live0 ? vv0_ : 0.;                                                           \
        if (mine1) vput1[(size_t)((k_) & 1) * vbuf] = live1 ? vv1_ : 0.;                                                           \
        if (scribe) gstore8(mptr, m_);                                                                                             \
        mptr += dir == 0 ? 1 : -1;                                                                                                 \
        FB_BARRIER();                                                                                                              \
    }
    // Breakend steps.  W_i[q][o] = exp(-pen (k - kt)) tab_i[ix] with kt = SAD(tot_q, tot_o) and ix the index of the total differences in
    // the clone-product table; kt is a function of ix, so tab2_i[ix] = tab_i[ix] exp(+pen kt(ix)) -- one pass over the 730-entry table
    // when it has landed -- leaves W_i[q][o] = W[q][o] tab2_i[ix]: the PLAIN weight (the same lookup as in a plain step) times one table
    // entry whose byte address is sgn 32 (U_q - U_o) + const: one v_mad per B operand instead of a dozen integer operations.
    const int sgn = dir == 0 ? 1 : -1, toff = a.cn_max + 1;
    const int ones_idx = M == 2 ? D : D * D;                       // table entry n2 holds 1 (k_brk_lut)
    const int ucst = M == 3 ? toff * (D + 1) : toff;
    auto u_of = [&](uint32_t tp) { return M == 3 ? (int)(tp & 0xff) * D + (int)((tp >> 8) & 0xff) : (int)(tp & 0xff); };
    const int umul = is_sum ? 0 : sgn;                             // the ones columns read entry n2 whatever the row
    const int uoff0 = is_sum ? 32 * ones_idx : 32 * (ucst - sgn * u_of(to0)), uoff1 = is_sum ? 32 * ones_idx : 32 * (ucst - sgn * u_of(to1));
    int k = 1;
    while (k < len) {
        const int k_be = be_adj >= 0 ? (dir == 0 ? be_adj - n0 + 1 : n1 - be_adj) : len;
        const int k_stop = k_be < len ? k_be : len;
        for (; k < k_stop; k++) {
            double e0, e1;
            gload8(e0, eptr0); gload8(e1, eptr1);
            eptr0 += rstep; eptr1 += rstep;
            // (the lookup addresses derive from loop-invariant registers: without this the compiler hoists all 4 KB of them out of
            // the step loop and spills them)
#pragma unroll
            for (int i = 0; i < NW32; i++) asm volatile("" : "+v"(c0[i]), "+v"(c1[i]));
            double s0, s1;
            if constexpr (NV == 4) {
                double acc[4] = {0., 0., 0., 0.};
                fbq_slot ring[FBQ_RING];
                const unsigned apc = ap0 + (unsigned)((k - 1) & 1) * (unsigned)(VR * 32);
                fbq_chain<KB>::template fill<0>(ring, apc, wt, c0, c1);
                fbq_chain<KB>::template run<0>(ring, apc, wt, c0, c1, acc);
                s0 = acc[0] + acc[2]; s1 = acc[1] + acc[3];
            } else {
                double av[NV][G], acc0[NV][2], acc1[NV][2];
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc0[j][0] = acc0[j][1] = acc1[j][0] = acc1[j][1] = 0.;
#pragma unroll
                    for (int g = 0; g < G; g++) av[j][g] = avp[(size_t)((k - 1) & 1) * vbuf + j * VRP + 64 * g];
                }
                fbqw_slot ring[FBQW_RING];
                fbqw_chain<KB, NV>::template fill<0>(ring, c0, c1);      // (the table sits at LDS address 0: the fields are the addresses)
                fbqw_chain<KB, NV>::template run<0>(ring, av, c0, c1, acc0, acc1);
                // (the row groups kq of a column and the move to the result lanes in one matrix instruction per restart, as in k_fbm; the
                // additions in front keep the inline-asm accumulators away from the matrix instruction's operands)
                s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel0, acc0[0][0] + acc0[0][1], 0., 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel0, acc1[0][0] + acc1[0][1], 0., 0, 0, 0);
                if constexpr (NV > 1) {
                    s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel1, acc0[1][0] + acc0[1][1], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel1, acc1[1][0] + acc1[1][1], s1, 0, 0, 0);
                }
            }
            FBQ_FINISH(s0, s1, e0, e1, k)
        }
        if (k < len) {
            // ---- breakend step ----------------------------------------------------------------------------------
            const double *tb = tab + (size_t)be_buf * 4 * a.PE2P;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FB_BARRIER();
            be_i += be_step;
            be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2;
            be_buf ^= 1;
            if (be_adj >= 0) FBQ_FETCH(be_i, be_buf)
            double e0, e1;
            gload8(e0, eptr0); gload8(e1, eptr1);
            eptr0 += rstep; eptr1 += rstep;
            // tab -> tab2 in place (every entry once; the pad entry n2 stays 1), then a barrier: the products read all of it
            {
                double *tw = tab + (size_t)(be_buf ^ 1) * 4 * a.PE2P;      // (be_buf was flipped above: this step's table)
                for (int e_ = t; e_ < ones_idx; e_ += NT) {
                    const int kt_ = M == 3 ? abs(e_ / D - toff) + abs(e_ % D - toff) : abs(e_ - toff);
                    const double f_ = wup[kt_ & 63];
                    double2 *row = reinterpret_cast<double2 *>(tw + (size_t)e_ * 4);
                    double2 r01 = row[0], r23 = row[1];
                    r01.x *= f_; r01.y *= f_; r23.x *= f_; r23.y *= f_;
                    row[0] = r01; row[1] = r23;
                }
            }
            FB_BARRIER();
            // Four restarts with four different weights leave an MFMA one useful row in four: these steps run on the vector ALU (as
            // k_fbm's).  Lane (kq, c) owns rows 4 kb + kq of its two columns; restart i's vector element reaches the 16 lanes of the row
            // through the FMA's DPP operand; the four lanes of a column are added at the end in a fixed order.
            double s0, s1;
            if constexpr (NV < 4) {
                double av[NV][G], acc0[NV][2], acc1[NV][2];
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc0[j][0] = acc0[j][1] = acc1[j][0] = acc1[j][1] = 0.;
#pragma unroll
                    for (int g = 0; g < G; g++) av[j][g] = avp[(size_t)((k - 1) & 1) * vbuf + j * VRP + 64 * g];
                }
                fbqw_be<0, KB, NV>::run(av, reinterpret_cast<const char *>(wtab), c0, c1, upl + kq, umul, uoff0, uoff1,
                                        reinterpret_cast<const char *>(tb) + I0 * 8, acc0, acc1);
                s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel0, acc0[0][0] + acc0[0][1], 0., 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel0, acc1[0][0] + acc1[0][1], 0., 0, 0, 0);
                if constexpr (NV > 1) {
                    s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel1, acc0[1][0] + acc0[1][1], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(sel1, acc1[1][0] + acc1[1][1], s1, 0, 0, 0);
                }
            } else {
            double acc0[FBM_NV] = {0., 0., 0., 0.}, acc1[FBM_NV] = {0., 0., 0., 0.};
            const fbm_d2 *apc = reinterpret_cast<const fbm_d2 *>(vec + (size_t)((k - 1) & 1) * VR * 4 + (kq * 4 + ib) * 2);     // pair p: + 16 p
            const char *tbb = reinterpret_cast<const char *>(tb);
            const char *wtb = reinterpret_cast<const char *>(wtab);
// WAIT STATES (asm site 4 of 5): breakend steps of k_fbq on the vector ALU; accumulators are folded by compiler-visible adds and __shfl_xor
// before anything else reads them; the DPP source is an LDS read's destination (checked on the built code: tools/asm_hazards.py)
#define FBQ_BE_FMA(acc_, a_, x_, i_) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #i_ " row_mask:0xf bank_mask:0xf" : "+v"(acc_) : "v"(a_), "v"(x_))
#define FBQ_BE_TILE(acc_, c_, uoff_, ak_)                                                                                          \
            {                                                                                                                      \
                const double wv = *reinterpret_cast<const double *>(wtb + FBQ_ADDR(c_, kb));                                       \
                const char *row = tbb + (__mul24(uq, umul) + (uoff_));                                                             \
                const double2 t01 = *reinterpret_cast<const double2 *>(row), t23 = *reinterpret_cast<const double2 *>(row + 16);   \
                const double x0 = wv * t01.x, x1 = wv * t01.y, x2 = wv * t23.x, x3 = wv * t23.y;                                   \
                FBQ_BE_FMA(acc_[0], ak_, x0, 0); FBQ_BE_FMA(acc_[1], ak_, x1, 1); FBQ_BE_FMA(acc_[2], ak_, x2, 2); FBQ_BE_FMA(acc_[3], ak_, x3, 3); \
            }
#pragma unroll
            for (int p = 0; p < KB / 2; p++) {      // fully unrolled: the address registers need compile-time indices
                const fbm_d2 av = apc[p * 16];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int kb = 2 * p + h;
                    const int uq = upl[4 * kb + kq];
                    const double ak = h ? av.y : av.x;
                    FBQ_BE_TILE(acc0, c0, uoff0, ak)
                    FBQ_BE_TILE(acc1, c1, uoff1, ak)
                }
                __builtin_amdgcn_sched_barrier(0);      // a pair of k-blocks at a time (register budget)
            }
#undef FBQ_BE_TILE
#undef FBQ_BE_FMA
#pragma unroll
            for (int i = 0; i < FBM_NV; i++) {
                acc0[i] += __shfl_xor(acc0[i], 16); acc0[i] += __shfl_xor(acc0[i], 32);
                acc1[i] += __shfl_xor(acc1[i], 16); acc1[i] += __shfl_xor(acc1[i], 32);
            }
            s0 = id == 0 ? acc0[0] : (id == 1 ? acc0[1] : (id == 2 ? acc0[2] : acc0[3]));
            s1 = id == 0 ? acc1[0] : (id == 1 ? acc1[1] : (id == 2 ? acc1[2] : acc1[3]));
            }
            FBQ_FINISH(s0, s1, e0, e1, k)
            k++;
        }
    }
#undef FBQ_FINISH
#undef FBQ_FETCH
    if (a.dbg && t == 0 && blockIdx.x == 0) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
    if (wave == 0) {
        const double *vb = vec + (size_t)((len - 1) & 1) * vbuf;
        double ps = 0.;
        for (int q = c16; q < S; q += 16) ps += vb[NV == 4 ? fbm_pos(q, id) : (id < NV ? id : 0) * VRP + fbw_pos(q)];
        ps = group_sum(ps, 16);
        if (c16 == 0 && present) {
            double m_, inv;
            pow2_scale((unsigned)__double2hiint(ps), m_, inv);
            if (dir == 0) gstore8(a.mrow + (size_t)(rg0 + id) * a.N + ROW(len - 1), m_);
            if (!(m_ > 0.) || m_ == INFINITY) atomicOr(&a.err[rg0 + id], RMX_ERR_NAN_AB);
        }
    }
#undef ROW
}
// grid (work items), as k_fbm: a workgroup's chain, unit of restarts, shape and direction come from a.items (above 256 states the shapes are four and one)
template <int KB>
__global__ __launch_bounds__(768) void k_fbq(FbmArgs a, const double *wk, const uint32_t *cnpack, const uint32_t *totpack) {
    const int4 it = a.items[blockIdx.x];
    if (it.z == 4) fbq_body<KB, 4>(a, wk, cnpack, totpack, it.x, it.y, it.w);
    else if (KB <= 64 && it.z == 2) fbq_body<(KB <= 64 ? KB : 64), 2>(a, wk, cnpack, totpack, it.x, it.y, it.w);
    else fbq_body<KB, 1>(a, wk, cnpack, totpack, it.x, it.y, it.w);
}

// =============================================================================
// k_fbk: forward-backward for state grids whose S x S weight matrix does not fit the register file
// (S > 176, e.g. 355 states at max_cn = 12).  Same step structure, vector operand broadcast (DPP
// row_newbcast), scaling, emission loads and breakend walk as k_fbv, but the plain-adjacency weight
// is never stored: with the default transition model (|d|) the log-weight of (q -> o) is
//     -pen * min( SAD(cn_q, cn_o), SAD(cn_q, swap_alleles(cn_o)) )
// (bpmodel.pyx:648-684: the total-copy terms cancel against the allele term's correction), the copies
// of the tumour clones' alleles packed one byte each -- two v_sad_u8 and a v_min per pair, then one
// LDS lookup exp(-pen*k) (<= 64 entries).  M <= 3; the host verifies the identity against the
// tabulated log-weights of the class before it selects this kernel.
//   thread t (phase 1):  g = t % G2 (column pair), p = t / G2 (row slice, PP slices of 16*NCH rows)
//   thread t (phase 2):  vector t / SPW + pass * (NT / SPW), state t % SPW
// =============================================================================
// M4 (round 4): four clones -- the third tumour clone's allele copies in a second packed word (cnpack2), its total in the third byte of totpack, the
// clone-product table of a breakend D^3 entries (loaded by as many transfers per thread as it takes)
// PP (round 5): row slices per column pair -- 4 up to 512 states (blocks of PP * G2 <= 1 024 threads), 2 from there to 1 024 states
template <int NV, int NTMAX, bool M4, int PP>
__global__ __launch_bounds__(NTMAX) void k_fbk(FbvArgs a, const double *wk, const uint32_t *cnpack, const uint32_t *totpack, const uint32_t *cnpack2) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int chain = a.chain_list[blockIdx.x], dir = blockIdx.z;
    const int rg0 = a.r0 + blockIdx.y * NV;
    const int nv = min(NV, a.r1 - rg0);
    const int S = a.S, M = a.M, D = a.D, SP = a.SP, G2 = a.G2, SPW = a.SPW;
    const int n0 = a.chain_start[chain], n1 = a.chain_end[chain], len = n1 - n0 + 1;
    const int t = threadIdx.x, NT = blockDim.x;
    const int p = t / G2, g = t - p * G2;
    const int o0 = 2 * g, o1 = 2 * g + 1;
    const bool act = p < PP;
    const int VPP = NT / SPW;                               // vectors published per pass of phase 2
    const int NPASS = (NV + VPP - 1) / VPP;
    const int pv0 = t / SPW, po = t - pv0 * SPW;
    const int lane = t & 63;
    const int SPAD = a.SPAD;                                // = PP * 16 * NCH (rows padded per slice)
    const int NCH = SPAD / (PP * 16);                    // 16-row chunks per slice
    const int cls = a.chain_cls[chain];
    // ---- LDS carve-up -----------------------------------------------------------------------
    double *vec = (double *)smem_raw;                           // [NV][2][SPAD]
    double *part = vec + (size_t)NV * 2 * SPAD;                 // [NV][PP][SP]
    double *red = part + (size_t)NV * PP * SP;               // [NV][4]
    unsigned *red32 = (unsigned *)red;
    double *pel = red + NV * 4;                                 // [NV][PE2P]  clone-product weights of the current breakend
    double *wtab = pel + (size_t)NV * a.PE2P;                   // [FBK_WKN] exp(-pen * k)
    uint32_t *cnl = (uint32_t *)(wtab + FBK_WKN);                    // [SPAD] packed allele copies of the row states (0 past S)
    uint32_t *tpl = cnl + SPAD;                                 // [SPAD] packed totals
    uint32_t *cnl2 = tpl + SPAD;                                // [SPAD] (M4) packed allele copies of the third tumour clone
    int *bel = (int *)(cnl2 + (M4 ? SPAD : 0));                 // adjacencies of this chain's breakends
    for (int i = t; i < SPAD; i += NT) { cnl[i] = i < S ? cnpack[(size_t)cls * S + i] : 0u; tpl[i] = i < S ? totpack[(size_t)cls * S + i] : 0u; if (M4) cnl2[i] = i < S ? cnpack2[(size_t)cls * S + i] : 0u; }
    const int be_lo = a.chain_be[2 * chain], be_hi = a.chain_be[2 * chain + 1];
    for (int i = t; i < be_hi - be_lo; i += NT) bel[i] = a.be_n[be_lo + i];
    for (int i = t; i < NV * 2 * SPAD; i += NT) vec[i] = 0.;
    for (int i = t; i < FBK_WKN; i += NT) wtab[i] = wk[(size_t)a.chain_tc[chain] * FBK_WKN + i];
    if (t < NV * 4) red[t] = 0.;
    // this thread's two columns: allele copies as they are and with the alleles swapped, totals
    auto swap_alleles = [](uint32_t x) { return ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu); };
    const uint32_t co0 = o0 < S ? cnpack[(size_t)cls * S + o0] : 0u, co1 = o1 < S ? cnpack[(size_t)cls * S + o1] : 0u;
    const uint32_t co0s = swap_alleles(co0), co1s = swap_alleles(co1);
    const uint32_t cb0 = (M4 && o0 < S) ? cnpack2[(size_t)cls * S + o0] : 0u, cb1 = (M4 && o1 < S) ? cnpack2[(size_t)cls * S + o1] : 0u;
    const uint32_t cb0s = swap_alleles(cb0), cb1s = swap_alleles(cb1);
    const uint32_t to0 = o0 < S ? totpack[(size_t)cls * S + o0] : 0u, to1 = o1 < S ? totpack[(size_t)cls * S + o1] : 0u;
    __syncthreads();

#define ROW(k) (dir == 0 ? n0 + (k) : n1 - (k))
#define ADJ(k) (dir == 0 ? n0 + (k) - 1 : n1 - (k))
    const int be_step = dir == 0 ? 1 : -1;
    int be_i = dir == 0 ? be_lo : be_hi - 1;
    int be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2;
#define BE_SLOT(k_, bs_)                                                                                   \
    int bs_ = -1;                                                                                          \
    if (ADJ(k_) == be_adj) {                                                                               \
        bs_ = be_i; be_i += be_step;                                                                       \
        be_adj = (be_i >= be_lo && be_i < be_hi) ? __builtin_amdgcn_readfirstlane(bel[be_i - be_lo]) : -2; \
    }
    // ---- step 0 ---------------------------------------------------------------------------------
    const int rstep = dir == 0 ? SP : -SP;
    const size_t vstride = (size_t)VPP * a.N * SP;          // between the vectors of consecutive passes
    const bool pubwave = pv0 < VPP;                         // (always true for NT == VPP * SPW)
    const size_t lane_off = ((size_t)(rg0 + (pv0 < nv ? pv0 : 0)) * a.N + ROW(0)) * SP + (po < S ? po : S - 1);
    double *outp = (dir == 0 ? a.fa : a.fb) + lane_off;
    const double *eptr = a.fe + lane_off;
    // (the second pass's vector of a ragged unit -- fewer restarts than the workgroup shape holds -- does not exist: its emission request, issued every step
    //  whether or not the value is used, then takes the first pass's address; found by the fuzz as a read two restarts past the end of the array)
    const size_t e1_off = (pv0 + VPP < nv) ? vstride : 0;
    for (int ps = 0; ps < NPASS; ps++) {
        const int pv = pv0 + ps * VPP;
        if (pv < NV && pubwave) {      // wave-uniform
            double e0 = 0.;
            if (pv < nv && po < S) {
                e0 = eptr[(size_t)ps * vstride];
                vec[(size_t)pv * 2 * SPAD + po] = e0;
                gstore8(outp + (size_t)ps * vstride, (dir == 0) ? e0 : 1.0);
            }
            const unsigned wm = wave_max_u32((unsigned)__double2hiint(e0));
            if (lane == 0) lds_max_u32(&red32[(pv * 4 + 2) * 2], wm);
        }
    }
    eptr += rstep;
    const int sgn = dir == 0 ? 1 : -1, off = a.cn_max + 1;
    FB_BARRIER();

#define FBV_SCALE(buf_, row_)                                                                              \
    if (t < NV) {                                                                                          \
        double m_, i_;                                                                                     \
        pow2_scale(red32[(t * 4 + 2 + (buf_)) * 2], m_, i_);                                               \
        red[t * 4] = i_; red[t * 4 + 1] = m_;                                                              \
        red32[(t * 4 + 2 + ((buf_) ^ 1)) * 2] = 0u;                                                        \
        if (dir == 0 && t < nv) gstore8(a.mrow + (size_t)(rg0 + t) * a.N + (row_), m_);                    \
    }
    // one 16-row chunk of phase 1: rows q0 .. q0+15 of this thread's slice against its two columns;
    // BE: breakend step (weight = exp(-pen * allele distance) * clone-product table entry)
#define FBK_ROW(rr_, BE_)                                                                                  \
    {                                                                                                      \
        const uint32_t cq_ = cnl[q0 + (rr_)];                                                              \
        const uint32_t cq2_ = M4 ? cnl2[q0 + (rr_)] : 0u;                                                  \
        const unsigned k0_ = min(__builtin_amdgcn_sad_u8(cq_, co0, M4 ? __builtin_amdgcn_sad_u8(cq2_, cb0, 0u) : 0u), __builtin_amdgcn_sad_u8(cq_, co0s, M4 ? __builtin_amdgcn_sad_u8(cq2_, cb0s, 0u) : 0u)); \
        const unsigned k1_ = min(__builtin_amdgcn_sad_u8(cq_, co1, M4 ? __builtin_amdgcn_sad_u8(cq2_, cb1, 0u) : 0u), __builtin_amdgcn_sad_u8(cq_, co1s, M4 ? __builtin_amdgcn_sad_u8(cq2_, cb1s, 0u) : 0u)); \
        double w0_, w1_;                                                                                   \
        if (!(BE_)) { w0_ = wtab[k0_]; w1_ = wtab[k1_]; }                                                  \
        else {                                                                                             \
            const uint32_t tq_ = tpl[q0 + (rr_)];                                                          \
            const unsigned a0_ = k0_ - __builtin_amdgcn_sad_u8(tq_, to0, 0u), a1_ = k1_ - __builtin_amdgcn_sad_u8(tq_, to1, 0u); \
            int i0_ = sgn * ((int)(tq_ & 0xff) - (int)(to0 & 0xff)) + off, i1_ = sgn * ((int)(tq_ & 0xff) - (int)(to1 & 0xff)) + off; \
            if (M >= 3) { i0_ = i0_ * D + sgn * ((int)((tq_ >> 8) & 0xff) - (int)((to0 >> 8) & 0xff)) + off;  \
                          i1_ = i1_ * D + sgn * ((int)((tq_ >> 8) & 0xff) - (int)((to1 >> 8) & 0xff)) + off; } \
            if (M4)     { i0_ = i0_ * D + sgn * ((int)((tq_ >> 16) & 0xff) - (int)((to0 >> 16) & 0xff)) + off; \
                          i1_ = i1_ * D + sgn * ((int)((tq_ >> 16) & 0xff) - (int)((to1 >> 16) & 0xff)) + off; } \
            w0_ = wtab[a0_]; w1_ = wtab[a1_];                                                              \
            _Pragma("unroll") for (int v = 0; v < NV; v++) { wbe0[v] = w0_ * pel[(size_t)v * a.PE2P + i0_]; wbe1[v] = w1_ * pel[(size_t)v * a.PE2P + i1_]; } \
        }                                                                                                  \
        _Pragma("unroll") for (int v = 0; v < NV; v++) {                                                   \
            const double x0_ = (BE_) ? wbe0[v] : w0_, x1_ = (BE_) ? wbe1[v] : w1_;                         \
            asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #rr_ " row_mask:0xf bank_mask:0xf" : "+v"(acc0[v]) : "v"(av[v]), "v"(x0_)); \
            asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #rr_ " row_mask:0xf bank_mask:0xf" : "+v"(acc1[v]) : "v"(av[v]), "v"(x1_)); \
        }                                                                                                  \
    }
// WAIT STATES (asm site 5 of 5): k_fbk's products (FBK_ROW: v_fmac_f64_dpp on weights formed from packed copy numbers by compiler code); the
// accumulators are read next by phase 2's compiler code (LDS writes after additions), the DPP source av[] is an LDS read's destination; as for
// the other four sites the built code is checked by tools/asm_hazards.py (tests/test_asm_hazards.py)
#define FBK_CHUNK(BE_)                                                                                     \
    FBK_ROW(0, BE_) FBK_ROW(1, BE_) FBK_ROW(2, BE_) FBK_ROW(3, BE_) FBK_ROW(4, BE_) FBK_ROW(5, BE_) FBK_ROW(6, BE_) FBK_ROW(7, BE_) \
    FBK_ROW(8, BE_) FBK_ROW(9, BE_) FBK_ROW(10, BE_) FBK_ROW(11, BE_) FBK_ROW(12, BE_) FBK_ROW(13, BE_) FBK_ROW(14, BE_) FBK_ROW(15, BE_)

    if (a.dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { a.dbg[0] = clock64(); a.dbg[1] = wall_clock64(); a.dbg[4] = len; }
    for (int k = 1; k < len; k++) {
        BE_SLOT(k, bs)
        const int cb = (k - 1) & 1, nb = k & 1;
        double e[2] = {0., 0.};                                 // NPASS <= 2
        gload8(e[0], eptr);
        if (NPASS > 1) gload8(e[1], eptr + e1_off);
        eptr += rstep;
        // ============================ phase 1 ============================
        double acc0[NV], acc1[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) { acc0[v] = 0.; acc1[v] = 0.; }
        FBV_SCALE(cb, ROW(k - 1))
        if (bs >= 0) {
            // clone-product tables of this breakend, one per vector (k_brk_lut), by LDS-DMA
            for (int e0 = 0; e0 < (a.PE2P + 1) / 2; e0 += NT) {      // (one round up to 2 NT entries: all grids of two or three clones)
                const int e = e0 + t;
                if (e - (t & 63) < (a.PE2P + 1) / 2) {                 // wave-uniform: the wave's first element is inside
                    for (int v = 0; v < NV; v++) {
                        const unsigned dpe = __builtin_amdgcn_readfirstlane(lds_addr(pel + (size_t)v * a.PE2P) + (unsigned)(((e >> 6) << 6) * 16));
                        if (v < nv && e * 2 < a.PE2P) glds16(a.pe2_lt + ((size_t)(rg0 + v) * a.NBE + bs) * a.PE2P + e * 2, dpe);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FB_BARRIER();
        }
        if (act) {      // wave-uniform: PP * G2 is a multiple of 64
            double wbe0[NV], wbe1[NV];
            for (int h = 0; h < NCH; h++) {
                const int q0 = (p * NCH + h) * 16;
                double av[NV];
#pragma unroll
                for (int v = 0; v < NV; v++) av[v] = vec[((size_t)v * 2 + cb) * SPAD + q0 + (lane & 15)];
                if (bs < 0) { FBK_CHUNK(false) } else { FBK_CHUNK(true) }
            }
        }
        if (act && o0 < SP) {
#pragma unroll
            for (int v = 0; v < NV; v++) {
                double2 pr; pr.x = acc0[v]; pr.y = acc1[v];
                *reinterpret_cast<double2 *>(part + ((size_t)v * PP + p) * SP + o0) = pr;
            }
        }
        FB_BARRIER();
        // ============================ phase 2 ============================
        outp += rstep;
        gwait8(e[0]);
        if (NPASS > 1) gwait8(e[1]);
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            if (ps >= NPASS) break;
            const int pv = pv0 + ps * VPP;
            if (pv < NV) {           // wave-uniform
                unsigned vmax_in = 0u;
                if (pv < nv && po < S) {
                    const double inv = red[pv * 4];
                    const double *pp_ = part + (size_t)pv * PP * SP + po;
                    double sum = pp_[0] + pp_[SP];
                    if (PP == 4) sum = (sum + pp_[2 * SP]) + pp_[3 * SP];
                    const double val = sum * inv;
                    const double vecv = val * e[ps];
                    vec[((size_t)pv * 2 + nb) * SPAD + po] = vecv;
                    gstore8(outp + (size_t)ps * vstride, (dir == 0) ? vecv : val);
                    vmax_in = (unsigned)__double2hiint(vecv);
                }
                const unsigned wm = wave_max_u32(vmax_in);
                if (lane == 0) lds_max_u32(&red32[(pv * 4 + 2 + nb) * 2], wm);
            }
        }
        FB_BARRIER();
    }
    if (a.dbg && t == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) { a.dbg[2] = clock64(); a.dbg[3] = wall_clock64(); }
    FBV_SCALE((len - 1) & 1, ROW(len - 1))
    if (t < nv) { const double m = red[t * 4 + 1]; if (!(m > 0.) || m == INFINITY) atomicOr(&a.err[rg0 + t], RMX_ERR_NAN_AB); }
#undef FBK_CHUNK
#undef FBK_ROW
#undef FBV_SCALE
#undef ROW
#undef ADJ
#undef BE_SLOT
}

// =============================================================================
// posterior marginals + per-segment likelihood expectations
//   post[n,s]  = softmax(alpha+beta) (bpmodel.pyx:948-950) == fa*fb / sum
//   A[n,u]     = sum_s post * LT_u          B[n,vw] = sum_s post * LA_vw
//   rowPF[n]   = sum_s post * f             rowPP[n] = sum_s post * prior
//   rowZ[n]    = this row's share of hmm_log_norm_const (bpmodel.pyx:946)
// The three indicator updates (bpmodel.pyx:987-1042), the likelihood part of the
// ELBO (:1076-1109) and the full-data E[ll] (:1125-1157) are all linear in
// (A, B), so the (N x S) likelihood cells are visited once here instead of once
// per consumer.
// =============================================================================
template <bool COMPUTE_POST>
__global__ void k_marginals(Dev d, int r0, int G) {
    const int r = r0 + blockIdx.y;
    const int rows = 256 / G;
    const int n = blockIdx.x * rows + threadIdx.x / G, gl = threadIdx.x % G;
    if (n >= d.N) return;
    const RestartParams &rp = d.rp[r];
    SegCtx sc; load_seg(d, r, n, sc);
    const int cls = d.seg_class[n];
    const size_t ro = rs_off(d, r, n);
    double *post = d.post + ro;
    const double *frow = d.f + ro;
    unsigned err = 0;
    double sum = 0.;
    if (COMPUTE_POST) {
        const double *fa = d.fa + ro, *fb = d.fb + ro;
        for (int s = gl; s < d.S; s += G) sum += fa[s] * fb[s];
        sum = group_sum(sum, G);
        if (!(sum > 0.) || sum != sum || sum == INFINITY) err |= RMX_ERR_NAN_POST;
        double s2 = 0.;
        for (int s = gl; s < d.S; s += G) { const double y = fa[s] * fb[s] / sum; post[s] = y; s2 += y; }
        s2 = group_sum(s2, G);
        for (int s = gl; s < d.S; s += G) post[s] = post[s] / s2;   // second renormalisation of _exp_normalize
    }
    double a0 = 0., a1 = 0., b0 = 0., b1 = 0., b2 = 0., b3 = 0., pf = 0., pp = 0.;
    for (int s = gl; s < d.S; s += G) {
        double LT[2], LA[4];
        cell_ll(d, rp, sc, r, cls, s, LT, LA, err);
        const double ps = post[s];
        a0 += ps * LT[0]; a1 += ps * LT[1];
        b0 += ps * LA[0]; b1 += ps * LA[1]; b2 += ps * LA[2]; b3 += ps * LA[3];
        pf += ps * frow[s];
        pp += ps * cell_prior(d, rp, sc, cls, s);
    }
    a0 = group_sum(a0, G); a1 = group_sum(a1, G);
    b0 = group_sum(b0, G); b1 = group_sum(b1, G); b2 = group_sum(b2, G); b3 = group_sum(b3, G);
    pf = group_sum(pf, G); pp = group_sum(pp, G);
    if (gl == 0) {
        const size_t rn = (size_t)r * d.N + n;
        d.A[rn * 2] = a0; d.A[rn * 2 + 1] = a1;
        d.Bv[rn * 4] = b0; d.Bv[rn * 4 + 1] = b1; d.Bv[rn * 4 + 2] = b2; d.Bv[rn * 4 + 3] = b3;
        d.rowPF[rn] = pf; d.rowPP[rn] = pp;
        if (COMPUTE_POST) d.rowZ[rn] = d.fmax[rn] + (d.chain_end_flag[n] ? log(sum) : log(d.mrow[rn]));
    }
    if (err) atomicOr(&d.err[r], err);
}

// =============================================================================
// Strip kernels for the (segment x state) likelihood passes, S > 32.
// A wave owns a strip of consecutive segments; lane l owns states l, l+64, ... (NS per lane) and keeps
// their table entries in registers for the whole strip; everything that depends on the segment only
// is wave-uniform and arrives through scalar loads (the segment index is made an SGPR).
//   MODE 0: update_framelogprob (+ row maximum, scaled emissions)          -- k_framelogprob
//   MODE 1: posterior marginals + (A, B, PF, PP, row share of log Z)        -- k_marginals<true>
//   MODE 2: refresh of the components of (A, B) selected by MASK            -- k_marginals<false>
//   MODE 3: MODE 1, then -- everything below is local to the segment -- update_p_outlier_total,
//           update_p_outlier_allele, the NEXT sweep's update_p_allele_swap and update_framelogprob from
//           the cached cell values (CACHE 2 only; their second read hits L2): between two sweeps of one
//           rmx_variational_update call the six cached planes are streamed once instead of twice
// grid (ceil(N / (4*RPW)), nr), block 256.
// =============================================================================
#define STRIP_RPW 8
// CACHE: 0 = evaluate the cells, 1 = evaluate and store the evaluated components in the cell cache
// d.lc, 2 = take all six values from the cache (no transcendental at all).  The six values depend on
// (h, likelihood parameters, masks) only, i.e. they are constant across the variational sweeps of one
// EM iteration; the host tracks per restart which components of the cache are current.
template <int NS, int MODE, int MASK, int CACHE>
__global__ __launch_bounds__(256) void k_cells(Dev d, int r0) {
    const int r = r0 + blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nbeg = (blockIdx.x * 4 + wave) * STRIP_RPW;
    if (nbeg >= d.N) return;
    const int nend = min(d.N, nbeg + STRIP_RPW);
    const RestartParams &rp = d.rp[r];
    const int S = d.S;
    StateRegs st[NS];
    int cur_cls = -1;
    auto nsub_of = [&](int cls_, int s_) { return (double)((d.sflags[(size_t)cls_ * d.S + s_] >> 2) & 3); };
    unsigned err = 0;
    const double divw = rp.p[RMX_P_DIVERGENCE_WEIGHT];
    const size_t plane = (size_t)d.N * d.SP;
    double *lcr = d.lc ? d.lc + (size_t)r * 6 * plane : nullptr;
    // one cell through the cache policy
    auto cell = [&](const SegCtx &sc, const StateRegs &stx, size_t off, double LT[2], double LA[4]) {
        if (CACHE == 2) {
            LT[0] = lcr[off]; LT[1] = lcr[plane + off];
            LA[0] = lcr[2 * plane + off]; LA[1] = lcr[3 * plane + off]; LA[2] = lcr[4 * plane + off]; LA[3] = lcr[5 * plane + off];
        } else {
            cell_ll_regs<MASK & CM_ALL>(rp, sc, stx, LT, LA, err);
            if (CACHE == 1) {
                if (MASK & CM_LT0) lcr[off] = LT[0];
                if (MASK & CM_LT1) lcr[plane + off] = LT[1];
                if (MASK & CM_LA0) { lcr[2 * plane + off] = LA[0]; lcr[3 * plane + off] = LA[1]; }
                if (MASK & CM_LA1) { lcr[4 * plane + off] = LA[2]; lcr[5 * plane + off] = LA[3]; }
            }
        }
    };
    double lp_prior[4] = {0., 0., 0., 0.};
    if (MODE == 3) {
        const double pt_ = rp.p[RMX_P_PRIOR_OUTLIER_TOTAL], pa_ = rp.p[RMX_P_PRIOR_OUTLIER_ALLELE];
        lp_prior[0] = log_prior(1. - pt_); lp_prior[1] = log_prior(pt_); lp_prior[2] = log_prior(1. - pa_); lp_prior[3] = log_prior(pa_);
    }
    // posterior passes: the forward / backward rows of the NEXT segment of the strip are requested while this one is processed
    // (the pass is a chain of dependent steps per segment: rows -> sum -> six planes -> sums -> indicators -> write)
    constexpr bool PREFETCH = (MODE == 1 || MODE == 3) && NS >= 2 && NS <= 4;      // (where the extra registers do not cost a wave per SIMD)
    double pfa[PREFETCH ? NS : 1], pfb[PREFETCH ? NS : 1];
    if (PREFETCH) {
        const size_t ro0 = ((size_t)r * d.N + nbeg) * d.SP;
#pragma unroll
        for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; pfa[k] = s < S ? d.fa[ro0 + s] : 0.; pfb[k] = s < S ? d.fb[ro0 + s] : 0.; }
    }
    for (int n = nbeg; n < nend; n++) {     // n is wave-uniform (SGPR)
        const int cls = d.seg_class[n];
        if (CACHE != 2 && cls != cur_cls) {
#pragma unroll
            for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; load_state_regs(d, r, cls, s < S ? s : S - 1, st[k]); }   // clamped: lanes past S are masked at use
            cur_cls = cls;
        }
        SegCtx sc; load_seg(d, r, n, sc);
        const size_t rn = (size_t)r * d.N + n;
        const size_t ro = rn * d.SP;
        if (MODE == 0) {
            const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
            const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
            double fv[NS];
            double vmax = -INFINITY;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int s = lane + 64 * k;
                fv[k] = -INFINITY;
                if (s < S) {
                    double LT[2], LA[4];
                    cell(sc, st[k], (size_t)n * d.SP + s, LT, LA);
                    double f = 0.;
                    f += qt0 * LT[0]; f += qt1 * LT[1];
                    f += qa0 * qs0 * LA[0]; f += qa0 * qs1 * LA[1]; f += qa1 * qs0 * LA[2]; f += qa1 * qs1 * LA[3];
                    f += -1.0 * (CACHE == 2 ? nsub_of(cls, s) : st[k].nsub) * sc.l * divw;
                    if (f != f) err |= RMX_ERR_NAN_F;
                    d.f[ro + s] = f;
                    fv[k] = f;
                    vmax = fmax(vmax, f);
                }
                __builtin_amdgcn_sched_barrier(0);   // one cell at a time: keeps the register footprint (occupancy) in check
            }
            vmax = group_max(vmax, 64);
            if (lane == 0) d.fmax[rn] = vmax;
#pragma unroll
            for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; if (s < d.SP) d.fe[ro + s] = s < S ? exp_fast(fv[k] - vmax) : 0.; }
        } else {
            constexpr bool M1 = MODE == 1 || MODE == 3;
            // planes of the cell cache kept in a wave-private LDS stash between the pass's two reads of them (expectations, then the next sweep's
            // frame values): all six up to 192 states (36 KB per block); three of six up to 384 (round 4: at 355 states the second read of all six missed
            // L2 -- 512 waves per XCD x 17 KB -- and the pass moved 1.46 x its algorithmic bytes); the other three are read again
            constexpr int NSTASH = MODE != 3 ? 0 : (NS <= 3 ? 6 : (NS <= 6 ? 3 : 0));
            constexpr bool STASH = NSTASH > 0;
            __shared__ double lsm[STASH ? 4 : 1][STASH ? NS * NSTASH * 64 : 1];
            double pv[NS];
            double sum = 0.;
            if (M1) {
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const int s = lane + 64 * k;
                    pv[k] = PREFETCH ? pfa[PREFETCH ? k : 0] * pfb[PREFETCH ? k : 0] : (s < S ? d.fa[ro + s] * d.fb[ro + s] : 0.);
                    sum += pv[k];
                }
                if (PREFETCH && n + 1 < nend) {
#pragma unroll
                    for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; pfa[PREFETCH ? k : 0] = s < S ? d.fa[ro + d.SP + s] : 0.; pfb[PREFETCH ? k : 0] = s < S ? d.fb[ro + d.SP + s] : 0.; }
                }
                sum = group_sum(sum, 64);
                if (!(sum > 0.) || sum != sum || sum == INFINITY) err |= RMX_ERR_NAN_POST;
                double s2 = 0.;
#pragma unroll
                for (int k = 0; k < NS; k++) { pv[k] = pv[k] / sum; s2 += pv[k]; }
                s2 = group_sum(s2, 64);
#pragma unroll
                // (a fused pass keeps the posterior in registers: nothing reads d.post before the last, unfused, sweep of the call writes it)
                for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; pv[k] = pv[k] / s2; if (MODE != 3 && s < S) d.post[ro + s] = pv[k]; }   // second renormalisation of _exp_normalize
                if (MODE == 1 && d.sig_cnt) {
                    // states with posterior mass, in state order, for the sparse trial passes of the M-step
                    int base_ = 0;
#pragma unroll
                    for (int k = 0; k < NS; k++) {
                        const int s = lane + 64 * k;
                        const bool sg_ = s < S && pv[k] >= RMX_POST_EPS;
                        const unsigned long long bal_ = __ballot(sg_);
                        const int pos_ = base_ + __popcll(bal_ & ((1ull << lane) - 1ull));
                        if (sg_ && pos_ < RMX_SIGK) d.sig_idx[rn * RMX_SIGK + pos_] = (uint16_t)s;
                        base_ += __popcll(bal_);
                    }
                    if (lane == 0) d.sig_cnt[rn] = base_ <= RMX_SIGK ? (uint8_t)base_ : (uint8_t)255;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; pv[k] = s < S ? d.post[ro + s] : 0.; }
            }
            double a0 = 0., a1 = 0., b0 = 0., b1 = 0., b2 = 0., b3 = 0., pf = 0., pp = 0.;
            double fq[6] = {0., 0., 0., 0., 0., 0.};      // MODE 1: the indicators this sweep's frame log-probabilities were built from
            if (MODE == 1) { fq[0] = d.qt[rn * 2]; fq[1] = d.qt[rn * 2 + 1]; fq[2] = d.qa[rn * 2]; fq[3] = d.qa[rn * 2 + 1]; fq[4] = d.qs[rn * 2]; fq[5] = d.qs[rn * 2 + 1]; }
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int s = lane + 64 * k;
                // (trial passes -- expectations only, nothing cached: a group of 64 states without posterior mass adds nothing)
                if (MODE == 2 && CACHE == 0 && !(MASK & 16) && !__any(pv[k] >= RMX_POST_EPS)) { if (s < S) cell_static_errors<MASK & CM_ALL>(sc, st[k].fl, err); continue; }
                if (s < S) {
                    double LT[2], LA[4];
                    cell(sc, st[k], (size_t)n * d.SP + s, LT, LA);
                    const double ps = pv[k];
                    if (STASH) {      // planes 0, 1: LT; 2 .. 5: LA -- the first NSTASH of them
#pragma unroll
                        for (int q_ = 0; q_ < 2; q_++) if (q_ < NSTASH) lsm[wave][(k * NSTASH + q_) * 64 + lane] = LT[q_];
#pragma unroll
                        for (int q_ = 0; q_ < 4; q_++) if (2 + q_ < NSTASH) lsm[wave][(k * NSTASH + 2 + q_) * 64 + lane] = LA[q_];
                    }
                    a0 += ps * LT[0]; a1 += ps * LT[1];
                    b0 += ps * LA[0]; b1 += ps * LA[1]; b2 += ps * LA[2]; b3 += ps * LA[3];
                    if (MODE == 1) {
                        // this sweep's frame log-probability from the values in registers (update_framelogprob's operations on the
                        // indicators it read): the fused passes in front of this one do not write f, this pass does
                        double f = 0.;
                        f += fq[0] * LT[0]; f += fq[1] * LT[1];
                        f += fq[2] * fq[4] * LA[0]; f += fq[2] * fq[5] * LA[1]; f += fq[3] * fq[4] * LA[2]; f += fq[3] * fq[5] * LA[3];
                        f += -1.0 * (CACHE == 2 ? nsub_of(cls, s) : st[k].nsub) * sc.l * divw;
                        d.f[ro + s] = f;
                        pf += ps * f;
                    }
                    if (MODE == 2 && (MASK & 16)) pf += ps * d.f[ro + s];
                    if (MODE == 1 || (MODE == 2 && (MASK & 16))) pp += ps * (-1.0 * (CACHE == 2 ? nsub_of(cls, s) : st[k].nsub) * sc.l * divw);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MASK & CM_LT0) { a0 = group_sum(a0, 64); if (lane == 0) d.A[rn * 2] = a0; }
            if (MASK & CM_LT1) { a1 = group_sum(a1, 64); if (lane == 0) d.A[rn * 2 + 1] = a1; }
            if (MASK & CM_LA0) { b0 = group_sum(b0, 64); b1 = group_sum(b1, 64); if (lane == 0) { d.Bv[rn * 4] = b0; d.Bv[rn * 4 + 1] = b1; } }
            if (MASK & CM_LA1) { b2 = group_sum(b2, 64); b3 = group_sum(b3, 64); if (lane == 0) { d.Bv[rn * 4 + 2] = b2; d.Bv[rn * 4 + 3] = b3; } }
            // (a fused pass is never the last of its call: the ELBO terms PF, PP, Z of its sweep are not read by anyone)
            if (MODE == 1 || (MODE == 2 && (MASK & 16))) {
                pf = group_sum(pf, 64); pp = group_sum(pp, 64);
                if (lane == 0) { d.rowPF[rn] = pf; d.rowPP[rn] = pp; }
            }
            if (MODE == 1 && lane == 0) d.rowZ[rn] = d.fmax[rn] + (d.chain_end_flag[n] ? log(sum) : log(d.mrow[rn]));
            if (MODE == 3) {
                // lane 0's sums are the ones the stand-alone kernels would read back from (A, B)
                auto b0_ = [](double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); };
                a0 = b0_(a0); a1 = b0_(a1); b0 = b0_(b0); b1 = b0_(b1); b2 = b0_(b2); b3 = b0_(b3);
                // The three updates are two-element _exp_normalize's (bpmodel.pyx:120-128): exp, exp, log, exp,
                // exp each.  Evaluated one after the other in every lane they would be a serial chain of ~15
                // transcendentals per segment; here independent evaluations sit in different lanes of ONE
                // call (lanes 0,1: outlier-total, lanes 2,3: outlier-allele) and are gathered with
                // v_readlane -- same functions on the same arguments, hence the same bits.
                auto lane_of = [&](double v, int l_) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l_), __builtin_amdgcn_readlane(__double2loint(v), l_)); };
                // update_p_outlier_total (bpmodel.pyx:987-1003)
                double lpt0 = lp_prior[0], lpt1 = lp_prior[1];
                lpt0 += a0; lpt1 += a1;
                // update_p_outlier_allele (:1005-1023), with this sweep's allele-swap indicator
                const double qs0o = d.qs[rn * 2], qs1o = d.qs[rn * 2 + 1];
                double lpa0, lpa1;
                outlier_allele_logits(lp_prior[2], lp_prior[3], qs0o, qs1o, b0, b1, b2, b3, lpa0, lpa1);
                const double vmt = lpt0 > lpt1 ? lpt0 : lpt1, vma = lpa0 > lpa1 ? lpa0 : lpa1;       // _max: strict > from -inf
                const double lp_l = lane == 0 ? lpt0 : (lane == 1 ? lpt1 : (lane == 2 ? lpa0 : lpa1));
                const double vm_l = lane < 2 ? vmt : vma;
                const double ex = exp_fast(lp_l - vm_l);
                double pst = 0.; pst += lane_of(ex, 0); pst += lane_of(ex, 1);
                double psa = 0.; psa += lane_of(ex, 2); psa += lane_of(ex, 3);
                const double lg = fast_log_pos(lane < 2 ? pst : psa);
                const double norm_l = lg + vm_l;
                const double y_l = exp_fast(lp_l - norm_l);
                double qt0 = lane_of(y_l, 0), qt1 = lane_of(y_l, 1), qa0 = lane_of(y_l, 2), qa1 = lane_of(y_l, 3);
                { const double st_ = qt0 + qt1; qt0 /= st_; qt1 /= st_; const double sa_ = qa0 + qa1; qa0 /= sa_; qa1 /= sa_; }
                // the next sweep's update_p_allele_swap (:1025-1042)
                double lps0, lps1;
                allele_swap_logits(qa0, qa1, b0, b1, b2, b3, lps0, lps1);
                const double vms = lps0 > lps1 ? lps0 : lps1;
                const double lps_l = (lane & 1) ? lps1 : lps0;
                const double exs = exp_fast(lps_l - vms);
                double pss = 0.; pss += lane_of(exs, 0); pss += lane_of(exs, 1);
                const double norms = fast_log_pos(pss) + vms;
                const double ys_l = exp_fast(lps_l - norms);
                double qs0 = lane_of(ys_l, 0), qs1 = lane_of(ys_l, 1);
                { const double ss_ = qs0 + qs1; qs0 /= ss_; qs1 /= ss_; }
                if (lane == 0) {
                    d.qt[rn * 2] = qt0; d.qt[rn * 2 + 1] = qt1; d.qa[rn * 2] = qa0; d.qa[rn * 2 + 1] = qa1;
                    d.qs[rn * 2] = qs0; d.qs[rn * 2 + 1] = qs1;
                }
                // the next sweep's update_framelogprob (:898-919) -- MODE 0 on the values in registers
                double fv[NS];
                double vmax = -INFINITY;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const int s = lane + 64 * k;
                    fv[k] = -INFINITY;
                    if (s < S) {
                        // the six values were streamed a moment ago for the expectations: up to 192 states they
                        // wait in a wave-private LDS stash (36 KB per block); beyond, they are read again (L2 /
                        // MALL).  Keeping 6 x NS doubles live in registers would cost a wave per SIMD of occupancy.
                        double LT[2], LA[4];
                        if (STASH) {
                            const size_t off_ = (size_t)n * d.SP + s;
#pragma unroll
                            for (int q_ = 0; q_ < 2; q_++) LT[q_] = q_ < NSTASH ? lsm[wave][(k * NSTASH + q_) * 64 + lane] : lcr[q_ * plane + off_];
#pragma unroll
                            for (int q_ = 0; q_ < 4; q_++) LA[q_] = 2 + q_ < NSTASH ? lsm[wave][(k * NSTASH + 2 + q_) * 64 + lane] : lcr[(2 + q_) * plane + off_];
                        } else cell(sc, st[k], (size_t)n * d.SP + s, LT, LA);
                        double f = 0.;
                        f += qt0 * LT[0]; f += qt1 * LT[1];
                        f += qa0 * qs0 * LA[0]; f += qa0 * qs1 * LA[1]; f += qa1 * qs0 * LA[2]; f += qa1 * qs1 * LA[3];
                        f += -1.0 * nsub_of(cls, s) * sc.l * divw;
                        if (f != f) err |= RMX_ERR_NAN_F;
                        fv[k] = f;      // (not stored: the call's last, unfused marginal pass writes the plane)
                        vmax = fmax(vmax, f);
                    }
                }
                vmax = group_max(vmax, 64);
                if (lane == 0) d.fmax[rn] = vmax;
#pragma unroll
                for (int k = 0; k < NS; k++) { const int s = lane + 64 * k; if (s < d.SP) d.fe_alt[ro + s] = s < S ? exp_fast(fv[k] - vmax) : 0.; }
            }
        }
    }
    if (err) atomicOr(&d.err[r], err);
}

// =============================================================================
// k_trial_sparse: the (A, B) expectations of the components in MASK at the restarts' CURRENT parameters
// into d.A / d.Bv (the caller passes a Dev whose A / Bv point at scratch), from the per-segment lists of
// states with posterior mass (built by the last marginal pass): ~13 of 165 states per segment carry all
// of the posterior, the others cannot move the rounded sums (RMX_POST_EPS).  A quarter wave per segment (TRIAL_SEGL);
// segments whose list overflowed (count 255) walk all states.  State-table error flags are reported for
// every state.  grid (ceil(N / 16), nr), block 256.
// =============================================================================
template <int MASK>
__global__ __launch_bounds__(256) void k_trial_sparse(Dev d, int r0) {
    const int r = r0 + blockIdx.y;
    const int n = blockIdx.x * TRIAL_SEG_PER_BLOCK + (threadIdx.x / TRIAL_SEGL), j = threadIdx.x & (TRIAL_SEGL - 1);
    if (n >= d.N) return;
    const RestartParams &rp = d.rp[r];
    SegCtx sc; load_seg(d, r, n, sc);
    const int cls = d.seg_class[n];
    const size_t rn = (size_t)r * d.N + n;
    const double *post = d.post + rn * d.SP;
    const int cnt = d.sig_cnt[rn];
    unsigned err = 0;
    double a0 = 0., a1 = 0., b0 = 0., b1 = 0., b2 = 0., b3 = 0.;
    auto one = [&](int s) {
        StateRegs st; load_state_regs(d, r, cls, s, st);
        double LT[2], LA[4];
        cell_ll_regs<MASK>(rp, sc, st, LT, LA, err);
        const double ps = post[s];
        a0 += ps * LT[0]; a1 += ps * LT[1];
        b0 += ps * LA[0]; b1 += ps * LA[1]; b2 += ps * LA[2]; b3 += ps * LA[3];
    };
    if (cnt == 255) { for (int s = j; s < d.S; s += TRIAL_SEGL) one(s); }
    else {
        for (int jj = j; jj < cnt; jj += TRIAL_SEGL) one((int)d.sig_idx[rn * RMX_SIGK + jj]);
        if ((MASK & (CM_LA0 | CM_LA1)) && j == 0) table_static_errors<MASK>(sc, d.stFlagsAgg[(size_t)r * d.C + cls], err);
    }
    if (MASK & CM_LT0) { a0 = group_sum(a0, TRIAL_SEGL); if (j == 0) d.A[rn * 2] = a0; }
    if (MASK & CM_LT1) { a1 = group_sum(a1, TRIAL_SEGL); if (j == 0) d.A[rn * 2 + 1] = a1; }
    if (MASK & CM_LA0) { b0 = group_sum(b0, TRIAL_SEGL); b1 = group_sum(b1, TRIAL_SEGL); if (j == 0) { d.Bv[rn * 4] = b0; d.Bv[rn * 4 + 1] = b1; } }
    if (MASK & CM_LA1) { b2 = group_sum(b2, TRIAL_SEGL); b3 = group_sum(b3, TRIAL_SEGL); if (j == 0) { d.Bv[rn * 4 + 2] = b2; d.Bv[rn * 4 + 3] = b3; } }
    if (err) atomicOr(&d.err[r], err);
}

// =============================================================================
// k_trial_flat (round 5): k_trial_sparse's sums with the (segment, listed state) CELLS laid out flat over the threads.  A quarter wave per
// segment runs as many steps as its longest list asks for (mean 13 listed states, p99 36, an overflowed list 165) with the lanes past a list's
// end idle, and a cell is 8 lgamma + 2 log1p of FP64: the pass took 0.47 ms for 0.07 ms of arithmetic.  Here a block owns TF_SEGB consecutive
// segments of one restart: the list lengths are prefix-summed in LDS, thread t evaluates cells t, t + 256, ... (segment by binary search in the
// 32 prefix sums, segment constants from LDS), the six products post * ll go to LDS, and one thread per (segment, component) adds its segment's
// products IN LIST ORDER (state order) -- a fixed order that depends on nothing but the segment's own list, so a restart's sums do not depend on
// the launch's range or on chunking (TF_CAP cells per chunk).  grid (ceil(N / TF_SEGB), nr), block 256.
// =============================================================================
#define TF_SEGB 32
#define TF_CAP 512
template <int MASK>
__global__ __launch_bounds__(256) void k_trial_flat(Dev d, int r0) {
    __shared__ int offs[TF_SEGB + 1];
    __shared__ int cnts[TF_SEGB];
    __shared__ double segv[15][TF_SEGB];          // x, l, logl, y0, y1, mask_t, mask_a, segc[0..7]
    __shared__ int segcls[TF_SEGB];
    __shared__ double prod[6][TF_CAP];
    const int r = r0 + blockIdx.y, nb = blockIdx.x * TF_SEGB, tid = threadIdx.x;
    const int nseg = min(TF_SEGB, d.N - nb);
    const RestartParams &rp = d.rp[r];
    if (tid < TF_SEGB) {
        int c = 0;
        if (tid < nseg) { c = d.sig_cnt[(size_t)r * d.N + nb + tid]; segcls[tid] = d.seg_class[nb + tid]; }
        cnts[tid] = c;
    }
    for (int t = tid; t < 15 * TF_SEGB; t += 256) {
        const int k = t / TF_SEGB, i = t % TF_SEGB;
        double v = 0.;
        if (i < nseg) {
            const int n = nb + i;
            switch (k) {
            case 0: v = d.x[n]; break; case 1: v = d.l[n]; break; case 2: v = d.logl[n]; break;
            case 3: v = d.y[2 * (size_t)n]; break; case 4: v = d.y[2 * (size_t)n + 1]; break;
            case 5: v = (double)d.mask_t[n]; break; case 6: v = (double)d.mask_a[n]; break;
            default: v = d.segc[((size_t)r * 8 + (k - 7)) * d.N + n]; break;
            }
        }
        segv[k][i] = v;
    }
    __syncthreads();
    if (tid == 0) {
        int a = 0;
        for (int i = 0; i < TF_SEGB; i++) { offs[i] = a; a += cnts[i] == 255 ? d.S : cnts[i]; }
        offs[TF_SEGB] = a;
    }
    __syncthreads();
    const int total = offs[TF_SEGB];
    unsigned err = 0;
    // owner of (segment oi, component ok): the first 6 * TF_SEGB threads
    const int oi = tid / 6, ok = tid % 6;
    const bool owner = tid < 6 * TF_SEGB && oi < nseg;
    const bool okm = ok == 0 ? (MASK & CM_LT0) : (ok == 1 ? (MASK & CM_LT1) : (ok < 4 ? (MASK & CM_LA0) : (MASK & CM_LA1)));
    double acc = 0.;
    for (int c0 = 0; c0 < total; c0 += TF_CAP) {
        const int c1 = min(total, c0 + TF_CAP);
        for (int c = c0 + tid; c < c1; c += 256) {
            int lo = 0, hi = TF_SEGB;                       // largest i with offs[i] <= c
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= c) lo = mid; else hi = mid; }
            const int i = lo, jj = c - offs[i];
            const size_t rn = (size_t)r * d.N + nb + i;
            const int s = cnts[i] == 255 ? jj : (int)d.sig_idx[rn * RMX_SIGK + jj];
            SegCtx sc;
            sc.x = segv[0][i]; sc.l = segv[1][i]; sc.logl = segv[2][i]; sc.y0 = segv[3][i]; sc.y1 = segv[4][i]; sc.ys = sc.y0 + sc.y1;
            sc.mt = (int)segv[5][i]; sc.ma = (int)segv[6][i];
#pragma unroll
            for (int q = 0; q < 4; q++) { sc.cnb[q] = segv[7 + q][i]; sc.cbb[q] = segv[11 + q][i]; }
            StateRegs st; load_state_regs(d, r, segcls[i], s, st);
            double LT[2], LA[4];
            cell_ll_regs<MASK>(rp, sc, st, LT, LA, err);
            const double ps = d.post[rn * d.SP + s];
            const int cc = c - c0;
            if (MASK & CM_LT0) prod[0][cc] = ps * LT[0];
            if (MASK & CM_LT1) prod[1][cc] = ps * LT[1];
            if (MASK & CM_LA0) { prod[2][cc] = ps * LA[0]; prod[3][cc] = ps * LA[1]; }
            if (MASK & CM_LA1) { prod[4][cc] = ps * LA[2]; prod[5][cc] = ps * LA[3]; }
        }
        __syncthreads();
        if (owner && okm) {
            const int a = max(offs[oi], c0), b = min(offs[oi + 1], c1);
            for (int c = a; c < b; c++) acc += prod[ok][c - c0];
        }
        __syncthreads();
    }
    if (owner) {
        const size_t rn = (size_t)r * d.N + nb + oi;
        if (okm) { if (ok < 2) d.A[rn * 2 + ok] = acc; else d.Bv[rn * 4 + (ok - 2)] = acc; }
        if (ok == 0 && (MASK & (CM_LA0 | CM_LA1)) && cnts[oi] != 255) {
            SegCtx sc; sc.ma = (int)segv[6][oi]; sc.ys = segv[3][oi] + segv[4][oi];
            table_static_errors<MASK>(sc, d.stFlagsAgg[(size_t)r * d.C + segcls[oi]], err);
        }
    }
    if (err) atomicOr(&d.err[r], err);
}

// =============================================================================
// O(N) indicator updates from (A, B)
// =============================================================================
__global__ void k_update_outlier_total(Dev d, int r0) {   // bpmodel.pyx:987-1003
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = r0 + blockIdx.y;
    if (n >= d.N) return;
    const size_t rn = (size_t)r * d.N + n;
    const double prior = d.rp[r].p[RMX_P_PRIOR_OUTLIER_TOTAL];
    double lp0 = log_prior(1. - prior), lp1 = log_prior(prior);
    lp0 += d.A[rn * 2]; lp1 += d.A[rn * 2 + 1];
    double y0, y1; exp_normalize2(lp0, lp1, y0, y1);
    d.qt[rn * 2] = y0; d.qt[rn * 2 + 1] = y1;
}
__global__ void k_update_outlier_allele(Dev d, int r0) {  // bpmodel.pyx:1005-1023
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = r0 + blockIdx.y;
    if (n >= d.N) return;
    const size_t rn = (size_t)r * d.N + n;
    const double prior = d.rp[r].p[RMX_P_PRIOR_OUTLIER_ALLELE];
    const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
    double lp0, lp1;
    outlier_allele_logits(log_prior(1. - prior), log_prior(prior), qs0, qs1, d.Bv[rn * 4 + 0], d.Bv[rn * 4 + 1], d.Bv[rn * 4 + 2], d.Bv[rn * 4 + 3], lp0, lp1);
    double y0, y1; exp_normalize2(lp0, lp1, y0, y1);
    d.qa[rn * 2] = y0; d.qa[rn * 2 + 1] = y1;
}
__global__ void k_update_allele_swap(Dev d, int r0) {     // bpmodel.pyx:1025-1042
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = r0 + blockIdx.y;
    if (n >= d.N) return;
    const size_t rn = (size_t)r * d.N + n;
    const double qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
    double lp0, lp1;
    allele_swap_logits(qa0, qa1, d.Bv[rn * 4 + 0], d.Bv[rn * 4 + 1], d.Bv[rn * 4 + 2], d.Bv[rn * 4 + 3], lp0, lp1);
    double y0, y1; exp_normalize2(lp0, lp1, y0, y1);
    d.qs[rn * 2] = y0; d.qs[rn * 2 + 1] = y1;
}

// =============================================================================
// breakend distance tables: pd[m][d] = sum_b p_brk[k,b] * g(d - orient*brk[b,m])
// (bpmodel.pyx:659-664), including the reference's wrap-around aliasing of the
// two outermost slots d = +-(cn_max+1) (buffer of length 2(cn_max+1), :600).
// grid (NBE, nr), block 64 (>= M*D threads looped)
// =============================================================================
// tables of breakend slot `slot` of restart r from the breakpoint's probabilities pb[0..B) (global or LDS): dst (and dst2 when given)
// [M][D]; exp_base / prod_base as below.  Every thread of the block calls it; `pes` is block-shared scratch [RMX_MAX_CLONES * 64].
__device__ __forceinline__ void brk_lut_body(const Dev &d, int r, int slot, const double *pb, double *dst_base, double *dst2_base, double *exp_base,
                                             double *prod_base, int PE2P, double *pes) {
    const int n = d.be_n[slot], orient = d.brk_orient[n];
    double *dst = dst_base + ((size_t)r * d.NBE + slot) * d.M * d.D;
    double *dst2 = dst2_base ? dst2_base + ((size_t)r * d.NBE + slot) * d.M * d.D : nullptr;
    for (int i = threadIdx.x; i < d.M * d.D; i += blockDim.x) {
        const int m = i / d.D, dd = i % d.D;
        const int dv = dd - (d.cn_max + 1);
        double acc = 0.;
        if (dd == 0 || dd == d.D - 1) {
            // both ends share one slot in the reference: accumulate d=-(cn_max+1) then d=+(cn_max+1)
            for (int b = 0; b < d.B; b++) acc += pb[b] * g_transition(d.tmodel, -(d.cn_max + 1) - orient * d.brk_states[b * d.M + m]);
            for (int b = 0; b < d.B; b++) acc += pb[b] * g_transition(d.tmodel, (d.cn_max + 1) - orient * d.brk_states[b * d.M + m]);
        } else {
            for (int b = 0; b < d.B; b++) acc += pb[b] * g_transition(d.tmodel, dv - orient * d.brk_states[b * d.M + m]);
        }
        dst[i] = acc;
        if (dst2) dst2[i] = acc;
        if (exp_base) { const double ev = exp(-d.pen * acc); exp_base[((size_t)r * d.NBE + slot) * ((d.M * d.D + 1) & ~1) + i] = ev; if (prod_base && d.D <= 64) pes[m * 64 + dd] = ev; }
    }
    if (exp_base && prod_base && d.D <= 64) {
        // product over the clones for the forward-backward kernels' breakend steps: entry
        // (d_1 [, d_2 [, d_3]]) = pe_0[normal-clone difference of the two classes] * pe_1[d_1] [* pe_2[d_2] [* pe_3[d_3]]]  (four clones: k_fbk only)
        __syncthreads();
        double *pr = prod_base + ((size_t)r * d.NBE + slot) * PE2P;
        // (normal clone: every state of a class has the same total, so its difference is a property of the adjacency)
        const int d0 = (int)d.tot[(size_t)d.be_cls[2 * slot] * d.S * d.M] - (int)d.tot[(size_t)d.be_cls[2 * slot + 1] * d.S * d.M];
        const double p0 = pes[d0 + d.cn_max + 1];
        const int n2 = d.M == 2 ? d.D : (d.M == 3 ? d.D * d.D : d.D * d.D * d.D);
        for (int i = threadIdx.x; i < PE2P; i += blockDim.x) {
            double v = 0.;
            if (i < n2) v = d.M == 2 ? p0 * pes[64 + i] : (d.M == 3 ? (p0 * pes[64 + i / d.D]) * pes[128 + i % d.D]
                                                                     : ((p0 * pes[64 + i / (d.D * d.D)]) * pes[128 + (i / d.D) % d.D]) * pes[192 + i % d.D]);
            pr[i] = v;
            if (d.pe2x_lt) d.pe2x_lt[((((size_t)(r >> 2) * d.NBE + slot) * PE2P + i) << 2) + (r & 3)] = i == n2 ? 1.0 : v;   // (entry n2, the pad: weight of k_fbm's ones column)
        }
    }
}
__global__ void k_brk_lut(Dev d, int r0, double *dst_base, double *exp_base, double *prod_base, int PE2P) {
    __shared__ double pes[RMX_MAX_CLONES * 64];
    const int slot = blockIdx.x, r = r0 + blockIdx.y;
    const int k = d.brk_idx[d.be_n[slot]];
    brk_lut_body(d, r, slot, d.pbrk + ((size_t)r * d.K + k) * d.B, dst_base, nullptr, exp_base, prod_base, PE2P, pes);
}

// log transition value of adjacency (n, n+1) for states (i, j), reference
// accumulation order (bpmodel.pyx:648-684).  pd: the breakend table or nullptr.
__device__ __forceinline__ double trans_value(const Dev &d, int n, int i, int j, const double *pd) {
    const int tc = d.tclass[n];
    if (tc < 0) return 0.;
    if (pd == nullptr) return d.Tval[((size_t)tc * d.S + i) * d.S + j];
    const int ca = d.seg_class[n], cb = d.seg_class[n + 1];
    double T = 0.;
    for (int c = 0; c < d.M; c++) {
        const int dd = (int)d.tot[((size_t)ca * d.S + i) * d.M + c] - (int)d.tot[((size_t)cb * d.S + j) * d.M + c];
        T += -d.pen * pd[c * d.D + dd + d.cn_max + 1];
    }
    T += -d.pen * (double)d.af[((size_t)tc * d.S + i) * d.S + j];
    return T;
}

// =============================================================================
// pairwise posterior at selected adjacencies (bpmodel.pyx:954-960), reduced on
// the fly to what its consumers need:
//   hist[m][d] = sum_{i,j: tot_i,m - tot_j,m = d} joint[i,j]   (:628-633)
//   jt         = sum joint * log_transmat                      (:1052)
//   ja         = sum joint * allele-flip term
// mode 0: joint from fa/fb/f ; mode 1: uniform joint (state before the first
// update_p_cn, :566-567) with log_transmat == 0.
// grid (nlist, nr), block 256.  list == nullptr -> breakend slots.
// =============================================================================
// aux (optional, all null by default): a second plain table / allele-flip table (the tables of ANOTHER transition
// model, for the ELBO in the state where log_transmat and cached_log_transmat belong to different models):
//   jt2_out = sum joint * Tval2 (plain adjacencies: list given), ja2_out = sum joint * af2; no_state: do not write
//   hist / be_jt / be_ja (the run only produces the aux sums)
struct PairAux { const double *Tval2; const int8_t *af2; double *jt2_out; double *ja2_out; int no_state; int pad_; };
__global__ void k_pairwise(Dev d, int r0, int mode, const int32_t *list, double *jt_out, PairAux aux) {
    __shared__ double hw[4][RMX_MAX_CLONES * 64];
    __shared__ double gvec[1024];
    __shared__ double scratch[8];
    const int r = r0 + blockIdx.y;
    const int slot = list ? -1 : (int)blockIdx.x;
    const int n = list ? list[blockIdx.x] : d.be_n[blockIdx.x];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, wv = t >> 6;
    for (int i = t; i < 4 * RMX_MAX_CLONES * 64; i += 256) (&hw[0][0])[i] = 0.;
    const int bs = d.brk_slot[n];
    const double *pd = (bs >= 0) ? d.pd_lt + ((size_t)r * d.NBE + bs) * M * D : nullptr;
    const int tc = d.tclass[n];
    const int ca = d.seg_class[n], cb = d.seg_class[n + 1];
    const double *fa = d.fa + rs_off(d, r, n);
    if (mode == 0) {
        const double *fb = d.fb + rs_off(d, r, n + 1), *fe = d.fe + rs_off(d, r, n + 1);
        for (int j = t; j < S; j += 256) gvec[j] = fe[j] * fb[j];
    }
    __syncthreads();
    double z = 0., jt = 0., ja = 0., jt2 = 0., ja2 = 0.;
    for (int idx = t; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx - i * S;
        double T = 0., J;
        if (mode == 0) { T = trans_value(d, n, i, j, pd); J = fa[i] * exp(T) * gvec[j]; }
        else J = 1.0;
        z += J; jt += J * T;
        if (tc >= 0) {
            ja += J * (double)d.af[((size_t)tc * S + i) * S + j];
            if (aux.Tval2) jt2 += J * aux.Tval2[((size_t)tc * S + i) * S + j];
            if (aux.af2) ja2 += J * (double)aux.af2[((size_t)tc * S + i) * S + j];
            if (slot >= 0 && !aux.no_state)
                for (int c = 0; c < M; c++) {
                    const int dd = (int)d.tot[((size_t)ca * S + i) * M + c] - (int)d.tot[((size_t)cb * S + j) * M + c];
                    atomicAdd(&hw[wv][c * 64 + dd + d.cn_max + 1], J);
                }
        } else if (slot >= 0 && !aux.no_state) {
            for (int c = 0; c < M; c++) {
                const int dd = (int)d.tot[((size_t)ca * S + i) * M + c] - (int)d.tot[((size_t)cb * S + j) * M + c];
                atomicAdd(&hw[wv][c * 64 + dd + d.cn_max + 1], J);
            }
        }
    }
    z = block_sum<256>(z, scratch);
    __shared__ double zsh;
    if (t == 0) zsh = z;
    jt = block_sum<256>(jt, scratch);
    __shared__ double jtsh;
    if (t == 0) jtsh = jt;
    ja = block_sum<256>(ja, scratch);
    if (aux.jt2_out) { jt2 = block_sum<256>(jt2, scratch); }
    if (aux.ja2_out) { ja2 = block_sum<256>(ja2, scratch); }
    __syncthreads();
    const double zz = zsh;
    if (t == 0) {
        if (slot >= 0 && !aux.no_state) {
            d.be_jt[(size_t)r * d.NBE + slot] = jtsh / zz;
            d.be_ja[(size_t)r * d.NBE + slot] = ja / zz;
        }
        if (jt_out) jt_out[(size_t)(r - r0) * gridDim.x + blockIdx.x] = jtsh / zz;
        if (aux.jt2_out) aux.jt2_out[(size_t)(r - r0) * gridDim.x + blockIdx.x] = jt2 / zz;
        if (aux.ja2_out) aux.ja2_out[(size_t)(r - r0) * gridDim.x + blockIdx.x] = ja2 / zz;
    }
    if (slot >= 0 && !aux.no_state)
        for (int i = t; i < M * D; i += 256) {
            const int c = i / D, dd = i % D;
            const double v = ((hw[0][c * 64 + dd] + hw[1][c * 64 + dd]) + hw[2][c * 64 + dd]) + hw[3][c * 64 + dd];
            d.hist[((size_t)r * d.NBE + slot) * M * D + i] = v / zz;
        }
}

// =============================================================================
// k_pairwise_sp: the pairwise reductions at a breakend adjacency over the state pairs that CAN carry mass (round 3).
//
// joint[i][j] = fa_n[i] W_r[i][j] g_{n+1}[j] has the posteriors as its marginals: sum_j joint[i][j] / Z = post_n[i] and sum_i joint[i][j] / Z =
// post_{n+1}[j], so joint[i][j] / Z <= min(post_n[i], post_{n+1}[j]): a pair with a row or a column below RMX_POST_EPS (1e-30) of the
// posterior mass cannot move a rounded sum (all dropped pairs together: < S^2 1e-30).  The block forms both posteriors from the rows it needs
// anyway (fa, fb of segments n and n+1), compacts the states above the threshold (~13 of 165, ~20 of 355) in state / column order, and walks
// only their product: a few hundred pairs instead of 27 000 (165 states) or 126 000 (355 states: the dense kernel takes 7.3 ms per sweep there).
// Fixed order everywhere, the scheme of k_pairwise_be2 on the lists: a thread per listed row walks the listed columns in the order of their
// tumour-clone totals and flushes a run of equal totals into its private LDS bins; passes of 64 rows; one thread per histogram bin folds a
// pass's rows in list order.  Any list length is handled (a flat posterior -- short segments with few reads -- makes it the dense product,
// at the dense kernel's cost per pair).  Outputs as k_pairwise_be2: hist [M][D], be_ja, be_jt.  M in {2, 3}, pair codes as k_pairwise_be2.
// grid (NBE, nr), block 256 or 64 (round 4, option pairwise_kernel 4: one wave per adjacency -- a dozen of its threads have a listed row and its barriers are
// free; measured within the noise of the 256-thread block at 165 and 355 states, as was a variant with 16 rows per pass whose 11 KB of LDS fit next to
// four workgroups of the marginal pass: removing the kernel altogether would gain 2.5 % of the step), dynamic LDS.
// =============================================================================
#define PSP_ROWS 64         // rows per pass (a thread per row with private histogram bins in LDS)
__global__ __launch_bounds__(256) void k_pairwise_sp(Dev d, int r0, int PE2P, int SPC) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double scratch[16];
    __shared__ double tot_n, tot_m;
    __shared__ int NI, NJ;
    const int r = r0 + blockIdx.y, slot = blockIdx.x;
    const int n = d.be_n[slot];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, lane = t & 63, wave = t >> 6;
    const int S8 = (S + 7) & ~7;
    const int NB = d.cn_max + 2;                          // totals 0 .. cn_max + 1
    const int tc = d.tclass[n];
    const int ca = d.seg_class[n], cb = d.seg_class[n + 1];
    double *fav = (double *)smem_raw;                     // [S8] fa of segment n, state order
    double *pn = fav + S8;                                // [S8] fa * fb of segment n
    double *gv = pn + S8;                                 // [S8] fe * fb of segment n + 1, column (jord) order
    double *pm = gv + S8;                                 // [S8] fa * fb of segment n + 1, column order
    double *bins = pm + S8;                               // [PSP_ROWS][M - 1][NB] private bins of a pass's rows
    double *zrow = bins + (size_t)PSP_ROWS * 2 * NB;      // [PSP_ROWS] row sums, [PSP_ROWS] row sums of joint * allele distance
    double *hacc = zrow + 2 * PSP_ROWS;                   // [M * D + 2] histogram, total mass, allele-distance sum: accumulated over the passes
    double *wa = hacc + ((M * D + 2 + 1) & ~1);           // [64] exp(-pen a)
    int *li = (int *)(wa + 64);                           // [S8] rows above the threshold
    int *lj = li + S8;                                    // [S8] columns (positions in jord order) above the threshold
    int *lm = lj + S8;                                    // [S8] per listed column: t1 | t2 << 8 | run flags << 16 (as jmeta, flags for the LIST)
    int *rtot = lm + S8;                                  // [PSP_ROWS] totals of a pass's rows, a byte per clone (the fold reads them from here, not through a chain of global loads)
    const double *fa0 = d.fa + rs_off(d, r, n), *fb0 = d.fb + rs_off(d, r, n);
    const double *fa1 = d.fa + rs_off(d, r, n + 1), *fb1 = d.fb + rs_off(d, r, n + 1), *fe1 = d.fe + rs_off(d, r, n + 1);
    double sn = 0., sm = 0.;
    for (int s = t; s < S8; s += NT) {
        double a0 = 0., p0 = 0., g1 = 0., p1 = 0.;
        if (s < S) {
            a0 = fa0[s]; p0 = a0 * fb0[s];
            const int j = d.jord[(size_t)cb * S + s];
            const double b1 = fb1[j];
            g1 = fe1[j] * b1; p1 = fa1[j] * b1;
        }
        fav[s] = a0; pn[s] = p0; gv[s] = g1; pm[s] = p1;
        sn += p0; sm += p1;
    }
    for (int i = t; i < 64; i += NT) wa[i] = tc >= 0 ? exp(-d.pen * (double)i) : 1.0;
    for (int i = t; i < M * D + 2; i += NT) hacc[i] = 0.;
    sn = group_sum(sn, 64); sm = group_sum(sm, 64);
    if (lane == 0) { scratch[wave] = sn; scratch[8 + wave] = sm; }
    __syncthreads();
    if (t == 0) { double x = 0., y = 0.; for (int w_ = 0; w_ < NT / 64; w_++) { x += scratch[w_]; y += scratch[8 + w_]; } tot_n = x; tot_m = y; }
    __syncthreads();
    // compaction in index order: wave 0 the rows, wave 1 the columns (ballot + prefix count per group of 64)
    // (a block of one wave -- round 4: an adjacency per wave -- does both lists)
    for (int which = wave; which < 2; which += NT >> 6) {
        const double *pp = which == 0 ? pn : pm;
        const double thr = RMX_POST_EPS * (which == 0 ? tot_n : tot_m);
        int *lst = which == 0 ? li : lj;
        int base = 0;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const bool keep = s < S && pp[s] >= thr && pp[s] > 0.;
            const unsigned long long bal = __ballot(keep);
            if (keep) lst[base + __popcll(bal & ((1ull << lane) - 1ull))] = s;
            base += __popcll(bal);
        }
        if (lane == 0) { if (which == 0) NI = base; else NJ = base; }
    }
    __syncthreads();
    const int ni = NI, nj = NJ;
    // totals of the listed columns and the run flags OF THE LIST: bit 16 = last listed column of a (t1, t2) run, bit 17 = of a t1 run
    for (int k = t; k < nj; k += NT) {
        const int m_ = d.jmeta[(size_t)cb * S + lj[k]] & 0xffff;
        const int mn = k + 1 < nj ? (d.jmeta[(size_t)cb * S + lj[k + 1]] & 0xffff) : -1;
        int fl = 0;
        if (mn < 0 || mn != m_) fl |= 1;
        if (mn < 0 || (mn & 0xff) != (m_ & 0xff)) fl |= 2;
        lm[k] = m_ | (fl << 16);
    }
    __syncthreads();
    const uint16_t *prow = d.pcode + (size_t)(tc >= 0 ? tc : 0) * S8 * SPC;
    const double *tg = d.pe2_lt + ((size_t)r * d.NBE + slot) * PE2P;
    const int8_t *tota = d.tot + (size_t)ca * S * M;
    const int off = d.cn_max + 1;
    const int t0b = (int)d.tot[(size_t)cb * S * M];       // the normal clone's total is the same for every column of the class
    for (int u0 = 0; u0 < ni; u0 += PSP_ROWS) {
        const int nu = (ni - u0) < PSP_ROWS ? (ni - u0) : PSP_ROWS;
        if (t < nu) {
            double *mybins = bins + (size_t)t * 2 * NB;
            for (int i = 0; i < (M - 1) * NB; i++) mybins[i] = 0.;
            const int i = li[u0 + t];
            const double fai = fav[i];
            { int pk = 0; for (int c = 0; c < M; c++) pk |= ((int)tota[i * M + c] & 0xff) << (8 * c); rtot[t] = pk; }
            double z = 0., ja = 0., acc1 = 0., acc2 = 0.;
            // chunks of 16 listed columns: their 16 codes are requested together, then their 16 table entries -- two memory round trips
            // per chunk instead of two per column
            for (int k0 = 0; k0 < nj; k0 += 16) {
                unsigned cc[16]; double tw[16];
#pragma unroll
                for (int u = 0; u < 16; u++) { const int k = k0 + u; cc[u] = (tc >= 0 && k < nj) ? (unsigned)prow[(size_t)lj[k] * SPC + i] : 0u; }
#pragma unroll
                for (int u = 0; u < 16; u++) tw[u] = (tc >= 0 && k0 + u < nj) ? tg[cc[u] & 1023u] : 1.0;
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int k = k0 + u;
                    if (k < nj) {
                        const int a = (int)(cc[u] >> 10);
                        const double w = tc >= 0 ? wa[a] * tw[u] : 1.0;
                        const double J = fai * w * gv[lj[k]];
                        z += J; ja += J * (double)a;
                        acc2 += J;
                        const int m_ = lm[k];
                        if (m_ & 0x10000) {                            // last listed column of a (t1, t2) run
                            if (M == 3) mybins[1 * NB + ((m_ >> 8) & 0xff)] += acc2;
                            acc1 += acc2; acc2 = 0.;
                            if (m_ & 0x20000) { mybins[0 * NB + (m_ & 0xff)] += acc1; acc1 = 0.; }   // last listed column of a t1 run
                        }
                    }
                }
            }
            zrow[t] = z; zrow[PSP_ROWS + t] = ja;
        }
        __syncthreads();
        // fold this pass's rows into the histogram, one thread per bin, rows in list order; total mass and allele-distance sum likewise
        for (int tt = t; tt < M * D + 2; tt += NT) {
            double acc = hacc[tt];
            if (tt < M * D) {
                const int c = tt / D, dv = tt % D - off;
                for (int u = 0; u < nu; u++) {
                    const int tj = ((rtot[u] >> (8 * c)) & 0xff) - dv;      // the column total this bin pairs with the row's
                    if (c == 0) { if (tj == t0b) acc += zrow[u]; }
                    else if (tj >= 0 && tj < NB) acc += bins[((size_t)u * 2 + (c - 1)) * NB + tj];
                }
            } else if (tt == M * D) { for (int u = 0; u < nu; u++) acc += zrow[u]; }
            else { for (int u = 0; u < nu; u++) acc += zrow[PSP_ROWS + u]; }
            hacc[tt] = acc;
        }
        __syncthreads();
    }
    const double zz = hacc[M * D];
    double *hist = d.hist + ((size_t)r * d.NBE + slot) * M * D;
    for (int tt = t; tt < M * D; tt += NT) { const double hv = hacc[tt] / zz; hist[tt] = hv; hacc[tt] = hv; }      // (entry tt is this thread's alone; zz = hacc[M * D] stays)
    __syncthreads();
    if (t < 64) {
        double jt = 0.;
        if (tc >= 0) {
            const double *pd = d.pd_lt + ((size_t)r * d.NBE + slot) * M * D;
            for (int i = t; i < M * D; i += 64) jt += hacc[i] * (-d.pen * pd[i]);
        }
        jt = group_sum(jt, 64);
        if (t == 0) {
            const double jaz = hacc[M * D + 1] / zz;
            d.be_ja[(size_t)r * d.NBE + slot] = jaz;
            if (tc >= 0) jt += -d.pen * jaz;
            d.be_jt[(size_t)r * d.NBE + slot] = jt;
        }
    }
}

// =============================================================================
// k_pairwise_be2: the production form of k_pairwise for breakend adjacencies (mode 0).  Thread i owns row i of the
// pairwise posterior joint[i][j] = fa[i] * W[i][j] * g[j] and walks j, accumulating into PRIVATE LDS bins indexed by the
// column state's totals (no atomics between threads, fixed order); the expectation of log_transmat is rebuilt from the
// histogram: sum joint*T = -pen * (sum_c sum_d hist_c[d] * pd_c[d] + sum joint*a).  The weight of a state pair is taken from the per-breakend clone-product
// table (k_brk_lut) through a precomputed 16-bit pair code, and with the columns walked in the order of
// their tumour-clone totals (t1, t2): the histogram contribution of a whole run of equal totals is
// flushed to the thread's private bins once per run (about S/3 + 9 LDS adds per row at M = 3 instead
// of 3 S).  M in {2, 3}.
// grid (NBE, nr), block NT = ceil(S/64)*64, dynamic LDS.
// =============================================================================
__global__ void k_pairwise_be2(Dev d, int r0, int PE2P, int SPC) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ double scratch[16];
    __shared__ double zsh, jash;
    const int r = r0 + blockIdx.y, slot = blockIdx.x;
    const int n = d.be_n[slot];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x;
    const int NB = d.cn_max + 2;                          // totals 0 .. cn_max+1
    const int tc = d.tclass[n];
    const int ca = d.seg_class[n], cb = d.seg_class[n + 1];
    double *gvec = (double *)smem_raw;                    // [S8] fe*fb of segment n+1, in jord order, zero-padded
    double *tab = gvec + ((S + 7) & ~7);                  // [PE2P] clone-product weights of this breakend (1 for a telomere)
    double *wa = tab + PE2P;                              // [128]
    double *bins = wa + 128;                              // [NT][M-1][NB] tumour clones; [NT] row sums behind them (the normal clone's single bin)
    double *zrow = bins + (size_t)NT * (M - 1) * NB;
    double *fold = zrow + NT;                             // [3][M*D] partial sums of the fold
    int *jm = (int *)(fold + 3 * M * D);                  // [S8] jmeta of class cb, zero-padded
    int *toti = jm + ((S + 7) & ~7);                      // [S] totals of the row states (class ca), one byte per clone
    // every global read of the prologue is issued before the first is consumed (one round trip, not a chain of them):
    // the column order is applied afterwards, LDS to LDS
    const double *fb = d.fb + rs_off(d, r, n + 1), *fe = d.fe + rs_off(d, r, n + 1);
    double *glin = bins + ((S + 7) & ~7);                 // [S] fe*fb in state order (parked, like the column order, in the not yet used bins area)
    const int S8_ = (S + 7) & ~7;
    const double *tg = d.pe2_lt + ((size_t)r * d.NBE + slot) * PE2P;
    {
        double gl_[4]; int jo_[4], jm_[4], tk_[4]; double tb_[8];
#pragma unroll
        for (int u = 0; u < 4; u++) {           // S <= 4 NT (NT >= 64, S <= 1024 but this kernel is only selected for S <= 255: pe2 tables)
            const int jj = t + u * NT;
            gl_[u] = jj < S ? fe[jj] * fb[jj] : 0.;
            jo_[u] = jj < S ? d.jord[(size_t)cb * S + jj] : 0;
            jm_[u] = jj < S ? d.jmeta[(size_t)cb * S + jj] : 0;
            int pk = 0;
            if (jj < S) for (int c = 0; c < M; c++) pk |= ((int)d.tot[((size_t)ca * S + jj) * M + c] & 0xff) << (8 * c);
            tk_[u] = pk;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = t + u * NT; tb_[u] = (tc >= 0 && i < PE2P) ? tg[i] : 1.0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int jj = t + u * NT;
            if (jj < S) { glin[jj] = gl_[u]; toti[jj] = tk_[u]; }
            if (jj < S8_) { jm[jj] = jj < S ? jm_[u] : 0; zrow[jj < NT ? jj : 0] = 0.; }
            if (jj < S8_) ((int *)(bins))[jj] = jo_[u];       // the column order, parked in the (not yet used) bins area
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = t + u * NT; if (i < PE2P) tab[i] = tb_[u]; }
        for (int i = t + 8 * NT; i < PE2P; i += NT) tab[i] = tc >= 0 ? tg[i] : 1.0;
    }
    for (int i = t; i < 128; i += NT) wa[i] = tc >= 0 ? exp(-d.pen * (double)i) : 1.0;
    __syncthreads();
    for (int jj = t; jj < S8_; jj += NT) gvec[jj] = jj < S ? glin[((int *)(bins))[jj]] : 0.;
    __syncthreads();
    double *mybins = bins + (size_t)t * (M - 1) * NB;     // plane c-1 for tumour clone c
    for (int i = 0; i < (M - 1) * NB; i++) mybins[i] = 0.;
    zrow[t] = 0.;
    __syncthreads();
    double z = 0., ja = 0.;
    if (t < S) {
        const double fai = d.fa[rs_off(d, r, n) + t];
        const int S8 = (S + 7) & ~7;                       // code rows / gvec / jm are padded to a multiple of 8 columns (zeros)
        const uint16_t *crow = d.pcode + (size_t)(tc >= 0 ? tc : 0) * S8 * SPC + t;
        double acc1 = 0., acc2 = 0.;
        // chunks of 8 columns, the next chunk's codes requested before the current one is consumed
        unsigned cn_[8], cc_[8];
#pragma unroll
        for (int u = 0; u < 8; u++) cn_[u] = tc >= 0 ? (unsigned)crow[(size_t)u * SPC] : 0u;
        for (int j0 = 0; j0 < S8; j0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) cc_[u] = cn_[u];
            if (j0 + 8 < S8) {
#pragma unroll
                for (int u = 0; u < 8; u++) cn_[u] = tc >= 0 ? (unsigned)crow[(size_t)(j0 + 8 + u) * SPC] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int jj = j0 + u;
                const unsigned c_ = cc_[u];
                const int a = (int)(c_ >> 10);
                const double w = wa[a] * tab[c_ & 1023u];
                const double J = fai * w * gvec[jj];
                z += J; ja += J * (double)a;
                acc2 += J;
                const int m_ = jm[jj];                         // wave-uniform
                if (m_ & 0x10000) {                            // last column of a (t1, t2) run
                    if (M == 3) unsafeAtomicAdd(&mybins[1 * NB + ((m_ >> 8) & 0xff)], acc2);   // private address: fire-and-forget ds_add_f64
                    acc1 += acc2; acc2 = 0.;
                    if (m_ & 0x20000) { unsafeAtomicAdd(&mybins[0 * NB + (m_ & 0xff)], acc1); acc1 = 0.; }   // last column of a t1 run
                }
            }
        }
        // normal clone: one total for every column, the whole row lands in one bin
        zrow[t] = z;
    }
    // deterministic block sums
    z = group_sum(z, 64); ja = group_sum(ja, 64);
    if ((t & 63) == 0) { scratch[t >> 6] = z; scratch[8 + (t >> 6)] = ja; }
    __syncthreads();
    if (t == 0) { double zz = 0., aa = 0.; for (int w_ = 0; w_ < NT / 64; w_++) { zz += scratch[w_]; aa += scratch[8 + w_]; } zsh = zz; jash = aa; }
    __syncthreads();
    const double zz = zsh;
    // fold the private bins: hist[c][d] = sum_i bins[i][c][tot_i,c - d].  Three threads per (c, d), each over
    // every third row, combined in fixed order; row totals come from LDS
    double *hist = d.hist + ((size_t)r * d.NBE + slot) * M * D;
    const int t0b = (int)d.tot[(size_t)cb * S * M];
    for (int tt = t; tt < 3 * M * D; tt += NT) {
        const int i = tt / 3, part_ = tt - i * 3;
        const int c = i / D, dv = i % D - (d.cn_max + 1);
        double acc = 0.;
        for (int row = part_; row < S; row += 3) {
            const int tj = ((toti[row] >> (8 * c)) & 0xff) - dv;
            if (c == 0) { if (tj == t0b) acc += zrow[row]; }
            else if (tj >= 0 && tj < NB) acc += bins[((size_t)row * (M - 1) + (c - 1)) * NB + tj];
        }
        fold[part_ * M * D + i] = acc;
    }
    __syncthreads();
    double *hl = fold;                                      // fold[0][i] becomes hist[i]
    for (int i = t; i < M * D; i += NT) {       // (entry i of the three partial rows is touched by this thread only)
        const double v = ((fold[i] + fold[M * D + i]) + fold[2 * M * D + i]) / zz;
        hl[i] = v;
        hist[i] = v;
    }
    __syncthreads();
    if (t < 64) {
        // sum joint*T = -pen * (sum_c sum_d hist_c[d] * pd_c[d] + sum joint*a)
        double jt = 0.;
        if (tc >= 0) {
            const double *pd = d.pd_lt + ((size_t)r * d.NBE + slot) * M * D;
            for (int i = t; i < M * D; i += 64) jt += hl[i] * (-d.pen * pd[i]);
        }
        jt = group_sum(jt, 64);
        if (t == 0) {
            d.be_ja[(size_t)r * d.NBE + slot] = jash / zz;
            if (tc >= 0) jt += -d.pen * (jash / zz);
            d.be_jt[(size_t)r * d.NBE + slot] = jt;
        }
    }
}

// =============================================================================
// update_p_breakpoint (bpmodel.pyx:964-985, 618-637) from the breakend histograms
// grid (K, nr), block 128
// =============================================================================
// the update of breakpoint k's probabilities into pb (global) and, normalised, into y (LDS); lp, y: [B] each
__device__ __forceinline__ void brk_update_body(const Dev &d, int r, int k, double *lp, double *y) {
    const int B = d.B, M = d.M, D = d.D;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        double acc = 0.;
        for (int e = d.bk_ptr[k]; e < d.bk_ptr[k + 1]; e++) {
            const int slot = d.bk_slots[e];
            const int n = d.be_n[slot], orient = d.brk_orient[n];
            const double *h = d.hist + ((size_t)r * d.NBE + slot) * M * D;
            const double mult = -d.pen;
            for (int c = 0; c < M; c++) {
                const int bv = d.brk_states[b * M + c];
                const double edge = h[c * D] + h[c * D + D - 1];   // aliased outer slots (see k_brk_lut)
                for (int dd = 0; dd < D; dd++) {
                    const double pdv = (dd == 0 || dd == D - 1) ? edge : h[c * D + dd];
                    acc += mult * pdv * g_transition(d.tmodel, (dd - (d.cn_max + 1)) - orient * bv);
                }
            }
        }
        lp[b] = acc;
    }
    __syncthreads();
    // _exp_normalize (bpmodel.pyx:120-128), every thread runs the same sequential sums
    double vmax = -INFINITY;
    for (int b = 0; b < B; b++) if (lp[b] > vmax) vmax = lp[b];
    double ps = 0.;
    for (int b = 0; b < B; b++) ps += exp(lp[b] - vmax);
    const double norm = log(ps) + vmax;
    for (int b = threadIdx.x; b < B; b += blockDim.x) y[b] = exp(lp[b] - norm);
    __syncthreads();
    double s = 0.;
    for (int b = 0; b < B; b++) s += y[b];
    __syncthreads();
    double *pb = d.pbrk + ((size_t)r * d.K + k) * B;
    for (int b = threadIdx.x; b < B; b += blockDim.x) { const double v = y[b] / s; pb[b] = v; y[b] = v; }
}
__global__ void k_brk_update(Dev d, int r0) {
    extern __shared__ double lp[];   // [B] then [B]
    brk_update_body(d, r0 + blockIdx.y, blockIdx.x, lp, lp + d.B);
}
// update_p_breakpoint and the tables that follow from it in ONE launch: breakpoint k's new probabilities stay in LDS and feed the
// distance tables of its breakend slots -- cached_log_transmat's (cached_base) and, when lt_base is given, the next sweep's
// log_transmat snapshot with its exponentials and clone products (what two k_brk_lut launches wrote).  Same arithmetic per output.
// grid (K, nr), block 128, dynamic LDS 2 B doubles
__global__ void k_brk_update_lut(Dev d, int r0, double *cached_base, double *lt_base, double *exp_base, double *prod_base, int PE2P) {
    extern __shared__ double lp[];   // [B] then [B]
    __shared__ double pes[RMX_MAX_CLONES * 64];
    const int k = blockIdx.x, r = r0 + blockIdx.y;
    double *y = lp + d.B;
    brk_update_body(d, r, k, lp, y);
    __syncthreads();
    for (int e = d.bk_ptr[k]; e < d.bk_ptr[k + 1]; e++) {
        brk_lut_body(d, r, d.bk_slots[e], y, cached_base, lt_base, lt_base ? exp_base : nullptr, lt_base ? prod_base : nullptr, PE2P, pes);
        __syncthreads();
    }
}

// =============================================================================
// ELBO pieces (bpmodel.pyx:1044-1123)
// =============================================================================
#define ELBO_BLOCKS 256
// per-block partials: [r][ELBO_BLOCKS][3] (energy, entropy, log Z) over segments; be_e [nr][NBE]: energy term of every breakend slot
__global__ void k_elbo_seg(Dev d, int r0, double *partial, double *be_e) {
    __shared__ double scratch[8];
    const int r = r0 + blockIdx.y;
    const RestartParams &rp = d.rp[r];
    const double pt = rp.p[RMX_P_PRIOR_OUTLIER_TOTAL], pa = rp.p[RMX_P_PRIOR_OUTLIER_ALLELE];
    const double l1t = log(1. - pt), l0t = log(pt), l1a = log(1. - pa), l0a = log(pa);
    double en = 0., ent = 0., z = 0.;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < d.N; n += gridDim.x * blockDim.x) {
        const size_t rn = (size_t)r * d.N + n;
        const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
        const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
        z += d.rowZ[rn];
        double e = d.rowPP[rn];
        e += qt0 * d.A[rn * 2] + qt1 * d.A[rn * 2 + 1];
        e += qt0 * l1t + qt1 * l0t;
        e += qa0 * qs0 * d.Bv[rn * 4] + qa0 * qs1 * d.Bv[rn * 4 + 1] + qa1 * qs0 * d.Bv[rn * 4 + 2] + qa1 * qs1 * d.Bv[rn * 4 + 3];
        e += qa0 * l1a + qa1 * l0a;
        en += e;
        double h = -d.rowZ[rn] + d.rowPF[rn];
        h += xlogx(qt0) + xlogx(qt1) + xlogx(qa0) + xlogx(qa1) + xlogx(qs0) + xlogx(qs1);
        ent += h;
    }
    en = block_sum<256>(en, scratch);
    ent = block_sum<256>(ent, scratch);
    z = block_sum<256>(z, scratch);
    if (threadIdx.x == 0) {
        partial[((size_t)(r - r0) * gridDim.x + blockIdx.x) * 3] = en;
        partial[((size_t)(r - r0) * gridDim.x + blockIdx.x) * 3 + 1] = ent;
        partial[((size_t)(r - r0) * gridDim.x + blockIdx.x) * 3 + 2] = z;
    }
    // transition factor of the energy at breakend adjacencies, one thread per slot (the sum over a slot's table in index order)
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < d.NBE; slot += gridDim.x * blockDim.x) {
        double e = 0.;
        if (d.tclass[d.be_n[slot]] >= 0) {      // telomere: T == 0
            const double *h = d.hist + ((size_t)r * d.NBE + slot) * d.M * d.D;
            const double *pc = d.pd_cached + ((size_t)r * d.NBE + slot) * d.M * d.D;
            for (int i = 0; i < d.M * d.D; i++) e += h[i] * (-d.pen * pc[i]);
            e += -d.pen * d.be_ja[(size_t)r * d.NBE + slot];
        }
        be_e[(size_t)(r - r0) * d.NBE + slot] = e;
    }
}
// out[(r-r0)*4 + {0,1,2,3}] = energy, entropy, elbo, hmm_log_norm_const
// lt_valid: update_p_cn has run (joint/log_transmat are live) ; plain_T_init: sum over plain
// adjacencies of mean(cached_log_transmat) for the pre-update state ; full_plain: optional
// sum over plain adjacencies of joint*T (adds to both energy and entropy when exact parts are requested)
__global__ void k_elbo_final(Dev d, int r0, const double *partial, const double *be_e, int nblk, const int *lt_valid, double plain_T_init,
                             const double *full_plain, double *out) {
    __shared__ double scratch[8];
    const int r = r0 + blockIdx.x, t = threadIdx.x;
    double en = 0., ent = 0., z = 0.;
    if (t == 0) for (int i = 0; i < nblk; i++) {
        const double *pp = partial + ((size_t)blockIdx.x * nblk + i) * 3;
        en += pp[0]; ent += pp[1]; z += pp[2];
    }
    // entropy of q(brk)
    double hb = 0.;
    for (int i = t; i < d.K * d.B; i += 256) hb += xlogx(d.pbrk[(size_t)r * d.K * d.B + i]);
    hb = block_sum<256>(hb, scratch);
    // transition factors at breakend adjacencies
    double ec = 0., jt = 0.;
    for (int slot = t; slot < d.NBE; slot += 256) {
        if (d.tclass[d.be_n[slot]] < 0) continue;   // telomere: T == 0
        ec += be_e[(size_t)blockIdx.x * d.NBE + slot];
        jt += d.be_jt[(size_t)r * d.NBE + slot];
    }
    ec = block_sum<256>(ec, scratch);
    jt = block_sum<256>(jt, scratch);
    if (t == 0) {
        const bool live = lt_valid[r] != 0;
        double energy = en + ec, entropy = ent + hb + (live ? jt : 0.);
        if (!live) energy += plain_T_init;
        if (full_plain) { energy += full_plain[blockIdx.x]; entropy += full_plain[blockIdx.x]; }
        out[blockIdx.x * 4 + 0] = energy; out[blockIdx.x * 4 + 1] = entropy; out[blockIdx.x * 4 + 2] = energy - entropy;
        out[blockIdx.x * 4 + 3] = z;
    }
}

// =============================================================================
// M-step objectives on a list of segments (bpmodel.pyx:1125-1195)
// grid (nlist), block 256: one segment per block; partial [nlist][1+M]
// =============================================================================
// one sampled segment per block: E[ll] (and d/dh when GRAD) of segment n under parameters rp -> prow[1+MAXC]
// MASK (CM_* bits, GRAD == false only): restrict the sum to those likelihood components -- the part of
// the objective that moves during the search over one likelihood parameter
// OVR: the state-table entries that depend on the searched beta-binomial precision (M, lgamma(M p),
// lgamma(M (1-p))) are recomputed from rp here instead of being read from the restart's tables, which
// therefore need no rebuild per candidate value
// one state's terms of E[ll] (and of d/dh when GRAD) under parameters rp: acc += ..., g[m] += ...
template <bool GRAD, int MASK, bool OVR>
__device__ __forceinline__ void ell_state_terms(const Dev &d, const RestartParams &rp, const SegCtx &sc, int r, int cls, int s, double ps_,
                                                double qt0, double qt1, double qa0, double qa1, double qs0, double qs1,
                                                double &acc, double (&g)[RMX_MAX_CLONES], unsigned &err) {
    double LT[2], LA[4];
    if (MASK == CM_ALL) cell_ll(d, rp, sc, r, cls, s, LT, LA, err);
    else {
        StateRegs st_; load_state_regs(d, r, cls, s, st_);
        if (OVR && (MASK & (CM_LA0 | CM_LA1)) && !(st_.fl & ST_LOH_M)) {
            const bool ok_ = !(st_.fl & (ST_E_BADP | ST_E_TD | ST_E_LOH));      // as state_tables_body
            if (MASK & CM_LA0) { const double M_ = rp.p[RMX_P_BETABIN_M_0]; st_.M0 = M_; st_.lgA0 = ok_ ? lgamma_pos(M_ * st_.p) : 0.; st_.lgB0 = ok_ ? lgamma_pos(M_ * (1 - st_.p)) : 0.; }
            if (MASK & CM_LA1) { const double M_ = rp.p[RMX_P_BETABIN_M_1]; st_.M1 = M_; st_.lgA1 = ok_ ? lgamma_pos(M_ * st_.p) : 0.; st_.lgB1 = ok_ ? lgamma_pos(M_ * (1 - st_.p)) : 0.; }
        }
        cell_ll_regs<MASK>(rp, sc, st_, LT, LA, err);
    }
    const double ps = ps_;
    if (MASK & CM_LT0) acc += ps * qt0 * LT[0];
    if (MASK & CM_LT1) acc += ps * qt1 * LT[1];
    if (MASK & CM_LA0) { acc += ps * qa0 * qs0 * LA[0]; acc += ps * qa0 * qs1 * LA[1]; }
    if (MASK & CM_LA1) { acc += ps * qa1 * qs0 * LA[2]; acc += ps * qa1 * qs1 * LA[3]; }
    if (GRAD) {
        const size_t si = ((size_t)r * d.C + cls) * d.SP + s;
        const unsigned fl = d.stFlags[si];
        const int8_t *cn = d.cn + ((size_t)cls * d.S + s) * d.M * 2;
        const int8_t *tot = d.tot + ((size_t)cls * d.S + s) * d.M;
        // total part (bpmodel.pyx:778-807)
        if (sc.mt && !(fl & ST_HDEL_NB)) {
            const double mu = d.stD[si] * sc.l;
            const double r0_ = rp.p[RMX_P_NEGBIN_R_0], r1_ = rp.p[RMX_P_NEGBIN_R_1];
            const double pm0 = sc.x / mu - (r0_ + sc.x) / (r0_ + mu), pm1 = sc.x / mu - (r1_ + sc.x) / (r1_ + mu);
            if (pm0 != pm0 || pm1 != pm1) err |= RMX_ERR_NAN_GRAD;
            for (int m = 0; m < d.M; m++) {
                const double base = sc.l * (double)tot[m];
                g[m] += ps * qt0 * (base * pm0); g[m] += ps * qt1 * (base * pm1);
            }
        }
        // allele part (bpmodel.pyx:855-896)
        if (sc.ma && !(fl & ST_GZ_ALLELE)) {
            double minor = 0., total = 0.;
            for (int m = 0; m < d.M; m++) { minor += rp.h[m] * (double)cn[m * 2]; total += rp.h[m] * (double)tot[m]; }
            if (total <= 0.) err |= RMX_ERR_TOTAL_DEPTH;
            else if (sc.ys != 0.) {
                const double p = minor / total;
                if (p <= 0. || (1 - p) <= 0.) err |= RMX_ERR_BAD_P;
                else {
                    double pp[4];
#pragma unroll
                    for (int v = 0; v < 2; v++) {
                        const double Mv = v == 0 ? rp.p[RMX_P_BETABIN_M_0] : rp.p[RMX_P_BETABIN_M_1];
                        const double dg_a = digamma_as103(Mv * p, err), dg_b = digamma_as103(Mv * (1 - p), err);
#pragma unroll
                        for (int w = 0; w < 2; w++) {
                            const double k = w == 0 ? sc.y0 : sc.y1;
                            pp[v * 2 + w] = (Mv * digamma_as103(k + Mv * p, err) + (-Mv) * digamma_as103(sc.ys - k + Mv * (1 - p), err)
                                             - Mv * dg_a - (-Mv) * dg_b);
                            if (pp[v * 2 + w] != pp[v * 2 + w]) err |= RMX_ERR_NAN_GRAD;
                        }
                    }
                    for (int m = 0; m < d.M; m++) {
                        const double base = ((double)cn[m * 2] * total - minor * (double)tot[m]) / (total * total);
                        g[m] += ps * qa0 * qs0 * (base * pp[0]); g[m] += ps * qa0 * qs1 * (base * pp[1]);
                        g[m] += ps * qa1 * qs0 * (base * pp[2]); g[m] += ps * qa1 * qs1 * (base * pp[3]);
                    }
                }
            }
        }
    }
}
template <bool GRAD, int MASK = CM_ALL, bool OVR = false>
__device__ __forceinline__ void ell_segment(const Dev &d, const RestartParams &rp, int r, int n, double *prow) {
    __shared__ double scratch[8];
    __shared__ double segk[8];
    SegCtx sc;
    sc.x = d.x[n]; sc.l = d.l[n]; sc.logl = d.logl[n]; sc.y0 = d.y[2 * (size_t)n]; sc.y1 = d.y[2 * (size_t)n + 1]; sc.ys = sc.y0 + sc.y1;
    sc.mt = d.mask_t[n]; sc.ma = d.mask_a[n];
    // the per-segment constants depend on the parameters being optimised: evaluate them here
    // for this block's segment only (the full [R][8][N] table is rebuilt lazily, not per evaluation)
    if (threadIdx.x < 8) segk[threadIdx.x] = seg_const_value(rp, sc.x, sc.y0, sc.ys, threadIdx.x);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) { sc.cnb[i] = segk[i]; sc.cbb[i] = segk[4 + i]; }
    const int cls = d.seg_class[n];
    const size_t rn = (size_t)r * d.N + n;
    const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
    const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
    const double *post = d.post + rs_off(d, r, n);
    unsigned err = 0;
    double acc = 0., g[RMX_MAX_CLONES] = {0., 0., 0., 0.};
    for (int s0 = 0; s0 < d.S; s0 += blockDim.x) {      // blocks of min(256, 64 * ceil(S / 64)) threads
        const int s = s0 + threadIdx.x;
        const double ps_ = s < d.S ? post[s] : 0.;
        if (!GRAD && !__any(ps_ >= RMX_POST_EPS)) {      // wave-uniform: no posterior mass in this group of 64 states
            if (s < d.S) cell_static_errors<MASK>(sc, d.stFlags[((size_t)r * d.C + cls) * d.SP + s], err);      // (its state-table errors still count)
            continue;
        }
        if (s >= d.S) continue;
        ell_state_terms<GRAD, MASK, OVR>(d, rp, sc, r, cls, s, ps_, qt0, qt1, qa0, qa1, qs0, qs1, acc, g, err);
    }
    acc = block_sum_rt(acc, scratch);
    if (threadIdx.x == 0) prow[0] = acc;
    if (GRAD)
        for (int m = 0; m < RMX_MAX_CLONES; m++) {
            const double gm = block_sum_rt(g[m], scratch);
            if (threadIdx.x == 0) prow[1 + m] = gm;
        }
    if (err) atomicOr(&d.err[r], err);
}
// The same objective for ONE likelihood component from the segment's list of states with posterior mass
// (d.sig_idx / d.sig_cnt, built by the last marginal pass; ~13 of 165 states, at most RMX_SIGK = 32): HALF A
// WAVE per sampled segment instead of one block, a lane per listed state; the states off the list cannot move the
// rounded sum (RMX_POST_EPS) and only report their state-table error flags.  A segment whose list
// overflowed (count 255) walks all states.  No block-level synchronisation: the half-waves of a block work
// on different segments.
template <int MASK, bool OVR>
__device__ __forceinline__ void ell_segment_sparse(const Dev &d, const RestartParams &rp, int r, int n, double *prow) {
    const int lane = threadIdx.x & (SEGL - 1);      // SEGL lanes per segment
    SegCtx sc;
    sc.x = d.x[n]; sc.l = d.l[n]; sc.logl = d.logl[n]; sc.y0 = d.y[2 * (size_t)n]; sc.y1 = d.y[2 * (size_t)n + 1]; sc.ys = sc.y0 + sc.y1;
    sc.mt = d.mask_t[n]; sc.ma = d.mask_a[n];
    // lanes 0..7 hold the eight per-segment constants; only the family the components in MASK read is evaluated
    // (the two families are divergent branches of 3 and 5 lgamma's: a negative-binomial search skips the longer one)
    double k_ = 0.;
    if ((MASK & (CM_LT0 | CM_LT1)) && (lane & 7) < 4) k_ = seg_const_value(rp, sc.x, sc.y0, sc.ys, lane & 7);
    if ((MASK & (CM_LA0 | CM_LA1)) && (lane & 7) >= 4) k_ = seg_const_value(rp, sc.x, sc.y0, sc.ys, lane & 7);
#pragma unroll
    for (int i = 0; i < 4; i++) { sc.cnb[i] = __shfl(k_, i, SEGL); sc.cbb[i] = __shfl(k_, 4 + i, SEGL); }
    const int cls = d.seg_class[n];
    const size_t rn = (size_t)r * d.N + n;
    const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
    const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
    const double *post = d.post + rs_off(d, r, n);
    unsigned err = 0;
    double acc = 0.;
    auto one = [&](int s) {
        StateRegs st_; load_state_regs(d, r, cls, s, st_);
        if (OVR && (MASK & (CM_LA0 | CM_LA1)) && !(st_.fl & ST_LOH_M)) {
            const bool ok_ = !(st_.fl & (ST_E_BADP | ST_E_TD | ST_E_LOH));      // as state_tables_body
            if (MASK & CM_LA0) { const double M_ = rp.p[RMX_P_BETABIN_M_0]; st_.M0 = M_; st_.lgA0 = ok_ ? lgamma_pos(M_ * st_.p) : 0.; st_.lgB0 = ok_ ? lgamma_pos(M_ * (1 - st_.p)) : 0.; }
            if (MASK & CM_LA1) { const double M_ = rp.p[RMX_P_BETABIN_M_1]; st_.M1 = M_; st_.lgA1 = ok_ ? lgamma_pos(M_ * st_.p) : 0.; st_.lgB1 = ok_ ? lgamma_pos(M_ * (1 - st_.p)) : 0.; }
        }
        double LT[2], LA[4];
        cell_ll_regs<MASK>(rp, sc, st_, LT, LA, err);
        const double ps = post[s];
        if (MASK & CM_LT0) acc += ps * qt0 * LT[0];
        if (MASK & CM_LT1) acc += ps * qt1 * LT[1];
        if (MASK & CM_LA0) { acc += ps * qa0 * qs0 * LA[0]; acc += ps * qa0 * qs1 * LA[1]; }
        if (MASK & CM_LA1) { acc += ps * qa1 * qs0 * LA[2]; acc += ps * qa1 * qs1 * LA[3]; }
    };
    const int cnt = d.sig_cnt[rn];
    if (cnt == 255) { for (int s = lane; s < d.S; s += SEGL) one(s); }
    else {
        for (int jj = lane; jj < cnt; jj += SEGL) one((int)d.sig_idx[rn * RMX_SIGK + jj]);
        if (lane == 0) table_static_errors<MASK>(sc, d.stFlagsAgg[(size_t)r * d.C + cls], err);
    }
    acc = group_sum(acc, SEGL);
    if (lane == 0) prow[0] = acc;
    if (err) atomicOr(&d.err[r], err);
}
// E[ll] and d/dh of one sampled segment from its list of states with posterior mass: half a wave per segment,
// a lane per listed state (ell_segment_sparse with the gradient terms; all four likelihood components)
__device__ __forceinline__ void ell_segment_sparse_grad(const Dev &d, const RestartParams &rp, int r, int n, double *prow) {
    const int lane = threadIdx.x & (SEGL - 1);
    SegCtx sc;
    sc.x = d.x[n]; sc.l = d.l[n]; sc.logl = d.logl[n]; sc.y0 = d.y[2 * (size_t)n]; sc.y1 = d.y[2 * (size_t)n + 1]; sc.ys = sc.y0 + sc.y1;
    sc.mt = d.mask_t[n]; sc.ma = d.mask_a[n];
    const double k_ = seg_const_value(rp, sc.x, sc.y0, sc.ys, lane & 7);
#pragma unroll
    for (int i = 0; i < 4; i++) { sc.cnb[i] = __shfl(k_, i, SEGL); sc.cbb[i] = __shfl(k_, 4 + i, SEGL); }
    const int cls = d.seg_class[n];
    const size_t rn = (size_t)r * d.N + n;
    const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
    const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
    const double *post = d.post + rs_off(d, r, n);
    unsigned err = 0;
    double acc = 0., g[RMX_MAX_CLONES] = {0., 0., 0., 0.};
    const int cnt = d.sig_cnt[rn];
    if (cnt == 255) { for (int s = lane; s < d.S; s += SEGL) ell_state_terms<true, CM_ALL, false>(d, rp, sc, r, cls, s, post[s], qt0, qt1, qa0, qa1, qs0, qs1, acc, g, err); }
    else {
        for (int jj = lane; jj < cnt; jj += SEGL) { const int s = (int)d.sig_idx[rn * RMX_SIGK + jj]; ell_state_terms<true, CM_ALL, false>(d, rp, sc, r, cls, s, post[s], qt0, qt1, qa0, qa1, qs0, qs1, acc, g, err); }
        if (lane == 0) table_static_errors<CM_ALL>(sc, d.stFlagsAgg[(size_t)r * d.C + cls], err);
    }
    acc = group_sum(acc, SEGL);
    if (lane == 0) prow[0] = acc;
#pragma unroll
    for (int m = 0; m < RMX_MAX_CLONES; m++) { const double gm = group_sum(g[m], SEGL); if (lane == 0) prow[1 + m] = gm; }
    if (err) atomicOr(&d.err[r], err);
}
// value only (all four components): the same per-state terms and summation as ell_segment_sparse_grad's value
__global__ __launch_bounds__(256) void k_ell_list_sparse_val(Dev d, int r, const int32_t *list, int count, double *partial) {
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    if (i >= count) return;
    ell_segment_sparse<CM_ALL, false>(d, d.rp[r], r, list[i], partial + (size_t)i * (1 + RMX_MAX_CLONES));
}
// grid (ceil(count / 8)), block 256: restart r's sampled segments, half a wave each
__global__ __launch_bounds__(256) void k_ell_list_sparse_grad(Dev d, int r, const int32_t *list, int count, double *partial) {
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    if (i >= count) return;
    ell_segment_sparse_grad(d, d.rp[r], r, list[i], partial + (size_t)i * (1 + RMX_MAX_CLONES));
}
// grid (ceil(maxcount / 8), nreq), block 256
__global__ __launch_bounds__(256) void k_ell_list_batch_sparse_grad(Dev d, const int32_t *rlist, const RestartParams *stage, const int32_t *samples, const int32_t *counts,
                                                                    double *partial, int pstride) {
    const int r = rlist[blockIdx.y];
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    if (i >= counts[r]) return;
    const int n = samples[(size_t)r * d.N + i];
    ell_segment_sparse_grad(d, stage[blockIdx.y], r, n, partial + (size_t)r * pstride + (size_t)i * (1 + RMX_MAX_CLONES));
}
// The same with the deterministic final sum of k_ell_final_batch folded in: the block that finishes a request's partials last
// (ticket from a per-request counter, which it resets) reduces them in k_ell_final_batch's order and writes out[j][0..nout) and the
// request's error word to host-visible memory -- one launch and one kernel boundary fewer per round of the lock-step h M-step.
__global__ __launch_bounds__(256) void k_ell_list_batch_sparse_grad_final(Dev d, const int32_t *rlist, const RestartParams *stage, const int32_t *samples,
                                                                          const int32_t *counts, double *partial, int pstride, unsigned *done,
                                                                          double *out, int nout, uint32_t *err_out) {
    __shared__ double scratch[8];
    __shared__ int last;
    const int r = rlist[blockIdx.y];
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    const int cnt = counts[r];
    if (i < cnt) ell_segment_sparse_grad(d, stage[blockIdx.y], r, samples[(size_t)r * d.N + i], partial + (size_t)r * pstride + (size_t)i * (1 + RMX_MAX_CLONES));
    // ONE release per block, behind the barrier that orders the block's partial sums before it: a release fence writes the L2's dirty lines
    // back (the XCDs' L2s are not coherent with each other), and next to another restart group's sweep kernels that is what a fence by all
    // 256 threads of all 200 blocks was paying for
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); last = atomicAdd(&done[blockIdx.y], 1u) == gridDim.x - 1; }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) { done[blockIdx.y] = 0; if (err_out) err_out[blockIdx.y] = __hip_atomic_load(&d.err[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    const int W = 1 + RMX_MAX_CLONES;
    for (int c = 0; c < nout; c++) {
        double a = 0.;
        for (int k = threadIdx.x; k < cnt; k += 256) a += __hip_atomic_load(&partial[(size_t)r * pstride + (size_t)k * W + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = block_sum<256>(a, scratch);
        if (threadIdx.x == 0) out[(size_t)blockIdx.y * nout + c] = a;
    }
}
// ---- the h M-step's objective + gradient with the work laid out flat (round 5) -----------------------------------------------------
// k_ell_list_batch_sparse_grad gives every sampled segment half a wave, a lane per listed state: 13 of 32 lanes work on average, a request
// is 25 blocks, and next to the other restart group's forward-backward launch (184 of 256 CUs held whole) the 200 blocks of a round crowd
// onto the 72 free CUs four to a CU -- 180-250 us per round in the bench's trace against 76 alone, thirteen rounds per M-step.  Here the
// LANE CHAINS of that kernel (segment i, lane l: listed states l, l + 32, ...) are the units of work, laid out flat over the threads of as
// few blocks as hold them (whole segments per block, at most 256 units), and a half-wave then adds a segment's units with the SAME
// group_sum the half-wave kernel uses (absent lanes contribute the same zeros): every per-segment partial sum is bit-identical to
// ell_segment_sparse_grad's, the final sums are k_ell_final_batch's.  The per-segment constants (functions of the likelihood parameters,
// not of h) and the block layout are made once per M-step by k_gradflat_setup, not in every round.
struct StageArgs { int32_t rlist[16]; RestartParams rp[16]; };      // up to 16 requests travel by value in the kernel arguments
struct GradFlatLayout {
    int32_t *upre;                // [R][NM_MAX_SAMPLE + 1]  units before sampled segment i
    int32_t *blk;                 // [R][NM_MAX_SAMPLE + 2]  first sampled segment of block b; blk[nblk] = cnt
    int32_t *nblk;                // [R]
    double *k8;                   // [R][NM_MAX_SAMPLE][8]   seg_const_value(rp, ., k)
};
// grid (nreq), block 256
__global__ __launch_bounds__(256) void k_gradflat_setup(Dev d, StageArgs sa, const int32_t *samples, const int32_t *counts, GradFlatLayout lay, int32_t *nblk_host) {
    __shared__ int un[NM_MAX_SAMPLE + 1];
    const int r = sa.rlist[blockIdx.x];
    const RestartParams &rp = sa.rp[blockIdx.x];
    const int cnt = counts[r];
    const int32_t *smp = samples + (size_t)r * d.N;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int n = smp[i];
        const int c = d.sig_cnt[(size_t)r * d.N + n];
        const int cells = c == 255 ? d.S : c;
        un[i + 1] = cells < 1 ? 1 : (cells < SEGL ? cells : SEGL);      // (lane 0 always exists: it reports the table's error flags)
    }
    for (int t = threadIdx.x; t < cnt * 8; t += 256) {
        const int i = t >> 3, k = t & 7, n = smp[i];
        const double x = d.x[n], y0 = d.y[2 * (size_t)n], ys = y0 + d.y[2 * (size_t)n + 1];
        lay.k8[((size_t)r * NM_MAX_SAMPLE + i) * 8 + k] = seg_const_value(rp, x, y0, ys, k);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *up = lay.upre + (size_t)r * (NM_MAX_SAMPLE + 1), *bl = lay.blk + (size_t)r * (NM_MAX_SAMPLE + 2);
        int a = 0, nb = 0, inblk = 0;
        up[0] = 0;
        if (cnt > 0) bl[nb++] = 0;
        for (int i = 0; i < cnt; i++) {
            const int u = un[i + 1];
            if (inblk + u > 256) { bl[nb++] = i; inblk = 0; }      // whole segments per block
            inblk += u; a += u; up[i + 1] = a;
        }
        bl[nb] = cnt;
        lay.nblk[r] = nb;
        nblk_host[blockIdx.x] = nb;
    }
}
// grid (blocks, nreq), block 256: block b of request j evaluates the units of the sampled segments blk[b] .. blk[b + 1] - 1
__global__ __launch_bounds__(256) void k_gradflat_round(Dev d, const int32_t *rlist, const RestartParams *stage, const int32_t *samples, const int32_t *counts,
                                                        GradFlatLayout lay, double *partial, int pstride) {
    __shared__ int lpre[257];
    __shared__ double vals[1 + RMX_MAX_CLONES][256];
    const int r = rlist[blockIdx.y], tid = threadIdx.x;
    if ((int)blockIdx.x >= lay.nblk[r]) return;
    const RestartParams &rp = stage[blockIdx.y];
    const int32_t *bl = lay.blk + (size_t)r * (NM_MAX_SAMPLE + 2), *up = lay.upre + (size_t)r * (NM_MAX_SAMPLE + 1);
    const int i0 = bl[blockIdx.x], i1 = bl[blockIdx.x + 1], nseg = i1 - i0;
    const int u0 = up[i0];
    for (int k = tid; k <= nseg; k += 256) lpre[k] = up[i0 + k] - u0;
    __syncthreads();
    const int nunits = lpre[nseg];
    const int32_t *smp = samples + (size_t)r * d.N;
    unsigned err = 0;
    double acc = 0., g[RMX_MAX_CLONES] = {0., 0., 0., 0.};
    if (tid < nunits) {
        int lo = 0, hi = nseg;                                  // the segment k with lpre[k] <= tid < lpre[k + 1] (segments without units are skipped)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lpre[mid] <= tid) lo = mid; else hi = mid; }
        const int i = i0 + lo, lane = tid - lpre[lo], n = smp[i];
        SegCtx sc;
        sc.x = d.x[n]; sc.l = d.l[n]; sc.logl = d.logl[n]; sc.y0 = d.y[2 * (size_t)n]; sc.y1 = d.y[2 * (size_t)n + 1]; sc.ys = sc.y0 + sc.y1;
        sc.mt = d.mask_t[n]; sc.ma = d.mask_a[n];
        const double *k8 = lay.k8 + ((size_t)r * NM_MAX_SAMPLE + i) * 8;
#pragma unroll
        for (int q = 0; q < 4; q++) { sc.cnb[q] = k8[q]; sc.cbb[q] = k8[4 + q]; }
        const int cls = d.seg_class[n];
        const size_t rn = (size_t)r * d.N + n;
        const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
        const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
        const double *post = d.post + rs_off(d, r, n);
        const int cnt_s = d.sig_cnt[rn];
        if (cnt_s == 255) { for (int s_ = lane; s_ < d.S; s_ += SEGL) ell_state_terms<true, CM_ALL, false>(d, rp, sc, r, cls, s_, post[s_], qt0, qt1, qa0, qa1, qs0, qs1, acc, g, err); }
        else {
            for (int jj = lane; jj < cnt_s; jj += SEGL) { const int s_ = (int)d.sig_idx[rn * RMX_SIGK + jj]; ell_state_terms<true, CM_ALL, false>(d, rp, sc, r, cls, s_, post[s_], qt0, qt1, qa0, qa1, qs0, qs1, acc, g, err); }
            if (lane == 0) table_static_errors<CM_ALL>(sc, d.stFlagsAgg[(size_t)r * d.C + cls], err);
        }
    }
    vals[0][tid] = acc;
#pragma unroll
    for (int m = 0; m < RMX_MAX_CLONES; m++) vals[1 + m][tid] = g[m];
    __syncthreads();
    // a half-wave per segment: lane l takes unit l's sums (zero past the segment's units, as the idle lanes of the half-wave kernel hold) and
    // group_sum adds them in that kernel's order
    const int hw = tid / SEGL, l = tid & (SEGL - 1);
    for (int k = hw; k < nseg; k += 256 / SEGL) {
        const int ub = lpre[k], nu = lpre[k + 1] - ub;
        double *prow = partial + (size_t)r * pstride + (size_t)(i0 + k) * (1 + RMX_MAX_CLONES);
#pragma unroll
        for (int c = 0; c < 1 + RMX_MAX_CLONES; c++) {
            double v = l < nu ? vals[c][ub + l] : 0.;
            v = group_sum(v, SEGL);
            if (l == 0) prow[c] = v;
        }
    }
    if (err) atomicOr(&d.err[r], err);
}
template <bool GRAD>
__global__ void k_ell_list(Dev d, int r, const int32_t *list, double *partial) {
    ell_segment<GRAD>(d, d.rp[r], r, list[blockIdx.x], partial + (size_t)blockIdx.x * (1 + RMX_MAX_CLONES));
}
// deterministic final sum over nlist partials -> out[1+MAXC].  grid 1, block 256
__global__ void k_ell_final(const double *partial, int nlist, double *out) {
    __shared__ double scratch[8];
    const int W = 1 + RMX_MAX_CLONES;
    for (int c = 0; c < W; c++) {
        double a = 0.;
        for (int i = threadIdx.x; i < nlist; i += 256) a += partial[(size_t)i * W + c];
        a = block_sum<256>(a, scratch);
        if (threadIdx.x == 0) out[c] = a;
    }
}
// full-data E[ll] from (A, B): grid ELBO_BLOCKS, block 256 -> partial[blk*(1+MAXC)]
__global__ void k_ell_full(Dev d, int r, double *partial) {
    __shared__ double scratch[8];
    double acc = 0.;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < d.N; n += gridDim.x * blockDim.x) {
        const size_t rn = (size_t)r * d.N + n;
        const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
        const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
        acc += qt0 * d.A[rn * 2] + qt1 * d.A[rn * 2 + 1];
        acc += qa0 * qs0 * d.Bv[rn * 4] + qa0 * qs1 * d.Bv[rn * 4 + 1] + qa1 * qs0 * d.Bv[rn * 4 + 2] + qa1 * qs1 * d.Bv[rn * 4 + 3];
    }
    acc = block_sum<256>(acc, scratch);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * (1 + RMX_MAX_CLONES)] = acc;
}

// ---- batched M-step objective: one candidate parameter set per listed restart -------------------
// (the restarts' optimisers advance in lock step on the host; each round costs three launches for
// all restarts instead of three per restart)
// grid (C, nreq): publish the staged parameters of request i to restart rlist[i] and rebuild its tables
__global__ void k_state_tables_list(Dev d, const int32_t *rlist, const RestartParams *stage) {
    const int r = rlist[blockIdx.y];
    const RestartParams rp = stage[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) d.rp[r] = rp;
    state_tables_body(d, blockIdx.x, r, rp);
}
// The same with the request list and the candidate parameters passed BY VALUE in the kernel arguments
// (<= 16 requests: 2.5 KB of the 4 KB argument segment): no host-to-device staging copies -- each one
// is a copy kernel serialised on the stream -- in front of every evaluation round.  The list is also
// written to the device staging buffers for the kernels that follow on the stream.
__global__ void k_state_tables_list_v(Dev d, StageArgs sa, int32_t *rlist_dev, RestartParams *stage_dev) {
    const int r = sa.rlist[blockIdx.y];
    const RestartParams rp = sa.rp[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.rp[r] = rp; rlist_dev[blockIdx.y] = r; stage_dev[blockIdx.y] = rp; }
    state_tables_body(d, blockIdx.x, r, rp);
}
// M-step sample lists from a host-pinned staging area to their homes: header [nlists][4] = (restart, slot, count, offset of the
// list in `indices`); slot -1: the restart's current sample, 0..3: parameter slot.  grid (nlists), block 256
__global__ void k_scatter_samples(const int32_t *header, const int32_t *indices, int32_t *sample, int32_t *counts, int32_t *msample, int32_t *mcounts,
                                  int N, int R) {
    const int32_t *hd = header + 4 * blockIdx.x;
    const int r = hd[0], slot = hd[1], cnt = hd[2], off = hd[3];
    int32_t *dst = slot < 0 ? sample + (size_t)r * N : msample + ((size_t)slot * R + r) * N;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) dst[i] = indices[off + i];
    if (threadIdx.x == 0) { if (slot < 0) counts[r] = cnt; else mcounts[(size_t)slot * R + r] = cnt; }
}
// the same without the staging copies: the tables of up to 16 restarts whose parameters changed, one launch (ensure_tables)
__global__ void k_state_tables_many(Dev d, StageArgs sa) {
    const int r = sa.rlist[blockIdx.y];
    const RestartParams rp = sa.rp[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) d.rp[r] = rp;
    state_tables_body(d, blockIdx.x, r, rp);
}
// grid (maxcount, nreq): block (i, j) evaluates sampled segment i of restart rlist[j]
template <bool GRAD, int MASK = CM_ALL>
__global__ void k_ell_list_batch(Dev d, const int32_t *rlist, const RestartParams *stage, const int32_t *samples, const int32_t *counts,
                                 double *partial, int pstride) {
    const int r = rlist[blockIdx.y];
    if ((int)blockIdx.x >= counts[r]) return;
    const int n = samples[(size_t)r * d.N + blockIdx.x];
    // stage[j] is identical to d.rp[r]; read from the stage to stay independent of launch order
    ell_segment<GRAD, MASK>(d, stage[blockIdx.y], r, n, partial + (size_t)r * pstride + (size_t)blockIdx.x * (1 + RMX_MAX_CLONES));
}
// grid (ceil(maxcount / 8), nreq), block 256: half-wave w of block (i, j) evaluates sampled segment 8 i + w of restart rlist[j]
template <int MASK>
__global__ __launch_bounds__(256) void k_ell_list_batch_sparse(Dev d, const int32_t *rlist, const RestartParams *stage, const int32_t *samples, const int32_t *counts,
                                                               double *partial, int pstride) {
    const int r = rlist[blockIdx.y];
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    if (i >= counts[r]) return;
    const int n = samples[(size_t)r * d.N + i];
    ell_segment_sparse<MASK, false>(d, stage[blockIdx.y], r, n, partial + (size_t)r * pstride + (size_t)i * (1 + RMX_MAX_CLONES));
}
// grid (nreq): deterministic sum of restart rlist[j]'s partials -> out[j * nout + c], c < nout (1 = value only, 1+MAXC = value and d/dh)
// err_out (optional, host-visible): the restart's error word, so that the host needs no separate copy
__global__ void k_ell_final_batch(Dev d, const int32_t *rlist, const int32_t *counts, const double *partial, int pstride, double *out, int nout,
                                  uint32_t *err_out) {
    __shared__ double scratch[8];
    const int r = rlist[blockIdx.x];
    if (err_out && threadIdx.x == 0) err_out[blockIdx.x] = d.err[r];
    const int W = 1 + RMX_MAX_CLONES;
    for (int c = 0; c < nout; c++) {
        double a = 0.;
        for (int i = threadIdx.x; i < counts[r]; i += 256) a += partial[(size_t)r * pstride + (size_t)i * W + c];
        a = block_sum<256>(a, scratch);
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * nout + c] = a;
    }
}
// ---- parameter search without table rebuilds ------------------------------------------------------
// Candidate values of ONE of negbin_r_0 / negbin_r_1 / betabin_M_0 / betabin_M_1 (MASK = the likelihood
// component it moves: 1 / 2 / 4 / 8) for the listed restarts: block (i, j, g) evaluates sampled segment i
// of restart rlist[j] with the parameter set to v[g] (grid stage: the same Gz values for every restart)
// or v[j] (per_request: one value per restart, Gz = 1).  Everything else comes from d.rp[r] and the
// restart's state tables; nothing is written back, so a whole grid is ONE launch.
// per_request: 0 = v[] is a grid of Gz values shared by all requests; 1 = one value per request (Gz = 1);
// 2 = Gz values per request, v[req * Gz + gz] (a request's pending point and the points its optimiser may ask
// for next, evaluated in the same launch)
struct SearchVals { double v[64], lv[64]; int32_t rlist[16]; int32_t per_request, Gz, pad0, pad1; };
template <int MASK>
__global__ void k_ell_search(Dev d, SearchVals sv, const int32_t *samples, const int32_t *counts, double *partial, int maxcnt) {
    const int req = blockIdx.y, gz = blockIdx.z;
    const int r = sv.rlist[req];
    if ((int)blockIdx.x >= counts[r]) return;
    const int n = samples[(size_t)r * d.N + blockIdx.x];
    RestartParams rp = d.rp[r];
    const int vi = sv.per_request == 2 ? req * sv.Gz + gz : (sv.per_request ? req : gz);
    if (MASK == CM_LT0) { rp.p[RMX_P_NEGBIN_R_0] = sv.v[vi]; rp.logr[0] = sv.lv[vi]; }
    if (MASK == CM_LT1) { rp.p[RMX_P_NEGBIN_R_1] = sv.v[vi]; rp.logr[1] = sv.lv[vi]; }
    if (MASK == CM_LA0) rp.p[RMX_P_BETABIN_M_0] = sv.v[vi];
    if (MASK == CM_LA1) rp.p[RMX_P_BETABIN_M_1] = sv.v[vi];
    ell_segment<false, MASK, true>(d, rp, r, n, partial + ((size_t)(req * sv.Gz + gz) * maxcnt + blockIdx.x));
}
// grid (ceil(maxcount / 8), nreq, Gz), block 256: half a wave per sampled segment (ell_segment_sparse)
template <int MASK>
__global__ __launch_bounds__(256) void k_ell_search_sparse(Dev d, SearchVals sv, const int32_t *samples, const int32_t *counts, double *partial, int maxcnt) {
    const int req = blockIdx.y, gz = blockIdx.z;
    const int r = sv.rlist[req];
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    if (i >= counts[r]) return;
    const int n = samples[(size_t)r * d.N + i];
    RestartParams rp = d.rp[r];
    const int vi = sv.per_request == 2 ? req * sv.Gz + gz : (sv.per_request ? req : gz);
    if (MASK == CM_LT0) { rp.p[RMX_P_NEGBIN_R_0] = sv.v[vi]; rp.logr[0] = sv.lv[vi]; }
    if (MASK == CM_LT1) { rp.p[RMX_P_NEGBIN_R_1] = sv.v[vi]; rp.logr[1] = sv.lv[vi]; }
    if (MASK == CM_LA0) rp.p[RMX_P_BETABIN_M_0] = sv.v[vi];
    if (MASK == CM_LA1) rp.p[RMX_P_BETABIN_M_1] = sv.v[vi];
    ell_segment_sparse<MASK, true>(d, rp, r, n, partial + ((size_t)(req * sv.Gz + gz) * maxcnt + i));
}
// ---- all four standard parameter searches of a restart group in the same rounds -------------------------
// The four likelihood parameters move disjoint components of E[ll] (negbin_r_0: LT0, negbin_r_1: LT1,
// betabin_M_0: LA0, betabin_M_1: LA1), each search evaluates only its own component, and the restart's
// model is not touched while candidates are evaluated: the four searches of a restart are independent of
// each other and advance together.  A request is a (restart, parameter slot) pair; slot j has its own
// sample per restart (samples [4][R][N], counts [4][R]).  grid_stage: candidate gz of every request is
// gv[slot][gz]; otherwise request q evaluates v[q] (Gz = 1).
#define RMX_MULTI_G 20
struct MultiVals {
    double v[64], lv[64];
    double gv[4][RMX_MULTI_G], glv[4][RMX_MULTI_G];
    int16_t rlist[64];
    int8_t slot[64];
    int32_t maskbit[4];
    int32_t grid_stage, Gz, pad0, pad1;
};
// grid (ceil(maxcount / 8), nreq, Gz), block 256: half a wave per sampled segment
__global__ __launch_bounds__(256) void k_ell_search_multi(Dev d, MultiVals mv, const int32_t *samples, const int32_t *counts, double *partial, int maxcnt) {
    const int req = blockIdx.y, gz = blockIdx.z;
    const int r = mv.rlist[req], sl = mv.slot[req];
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    if (i >= counts[sl * d.R + r]) return;
    const int n = samples[((size_t)sl * d.R + r) * d.N + i];
    const double v = mv.grid_stage ? mv.gv[sl][gz] : mv.v[req], lv = mv.grid_stage ? mv.glv[sl][gz] : mv.lv[req];
    double *prow = partial + ((size_t)(req * mv.Gz + gz) * maxcnt + i);
    // (a copy of the restart's parameters per case, changed at a constant index: one copy changed under a switch lives in scratch)
    switch (mv.maskbit[sl]) {
    case CM_LT0: { RestartParams rp = d.rp[r]; rp.p[RMX_P_NEGBIN_R_0] = v; rp.logr[0] = lv; ell_segment_sparse<CM_LT0, true>(d, rp, r, n, prow); break; }
    case CM_LT1: { RestartParams rp = d.rp[r]; rp.p[RMX_P_NEGBIN_R_1] = v; rp.logr[1] = lv; ell_segment_sparse<CM_LT1, true>(d, rp, r, n, prow); break; }
    case CM_LA0: { RestartParams rp = d.rp[r]; rp.p[RMX_P_BETABIN_M_0] = v; ell_segment_sparse<CM_LA0, true>(d, rp, r, n, prow); break; }
    default:     { RestartParams rp = d.rp[r]; rp.p[RMX_P_BETABIN_M_1] = v; ell_segment_sparse<CM_LA1, true>(d, rp, r, n, prow); break; }
    }
}
// grid (nreq * Gz): fixed-order sum of the partials of (request, candidate), as k_ell_search_final
__global__ void k_ell_multi_final(Dev d, MultiVals mv, const int32_t *counts, const double *partial, int maxcnt, double *out, uint32_t *err_out) {
    __shared__ double scratch[8];
    const int req = blockIdx.x / mv.Gz;
    const int r = mv.rlist[req], cnt = counts[mv.slot[req] * d.R + r];
    if (err_out && threadIdx.x == 0 && blockIdx.x % mv.Gz == 0) err_out[req] = d.err[r];
    double a = 0.;
    for (int i = threadIdx.x; i < cnt; i += 256) a += partial[(size_t)blockIdx.x * maxcnt + i];
    a = block_sum<256>(a, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}
// k_ell_search_multi with k_ell_multi_final folded in (round 5; the Nelder-Mead rounds, Gz = 1): the block that finishes a request's partial sums
// last (ticket from a per-request counter, which it resets: the scheme of k_ell_list_batch_sparse_grad_final) adds them in k_ell_multi_final's
// order -- the same bits -- and writes the request's value and error word to host-visible memory: one launch and one kernel boundary fewer
// in each of the ~50 rounds of an M-step.  Selected by search_mode 6 only: the release fence every block pays (an L2 write-back on a chip whose
// XCDs' L2s are not coherent) cost the headline 5 % next to the other restart group's sweeps.  grid (ceil(maxcount / 8), nreq), block 256.
__global__ __launch_bounds__(256) void k_ell_search_multi_final(Dev d, MultiVals mv, const int32_t *samples, const int32_t *counts, double *partial, int maxcnt,
                                                                unsigned *done, double *out, uint32_t *err_out) {
    __shared__ double scratch[8];
    __shared__ int last;
    const int req = blockIdx.y;
    const int r = mv.rlist[req], sl = mv.slot[req];
    const int i = blockIdx.x * SEG_PER_BLOCK + (threadIdx.x / SEGL);
    const int cnt = counts[sl * d.R + r];
    if (i < cnt) {
        const int n = samples[((size_t)sl * d.R + r) * d.N + i];
        const double v = mv.v[req], lv = mv.lv[req];
        double *prow = partial + ((size_t)req * maxcnt + i);
        switch (mv.maskbit[sl]) {
        case CM_LT0: { RestartParams rp = d.rp[r]; rp.p[RMX_P_NEGBIN_R_0] = v; rp.logr[0] = lv; ell_segment_sparse<CM_LT0, true>(d, rp, r, n, prow); break; }
        case CM_LT1: { RestartParams rp = d.rp[r]; rp.p[RMX_P_NEGBIN_R_1] = v; rp.logr[1] = lv; ell_segment_sparse<CM_LT1, true>(d, rp, r, n, prow); break; }
        case CM_LA0: { RestartParams rp = d.rp[r]; rp.p[RMX_P_BETABIN_M_0] = v; ell_segment_sparse<CM_LA0, true>(d, rp, r, n, prow); break; }
        default:     { RestartParams rp = d.rp[r]; rp.p[RMX_P_BETABIN_M_1] = v; ell_segment_sparse<CM_LA1, true>(d, rp, r, n, prow); break; }
        }
    }
    __syncthreads();      // (one release per block, behind the barrier that orders the block's partial sums before it)
    if (threadIdx.x == 0) { __threadfence(); last = atomicAdd(&done[req], 1u) == gridDim.x - 1; }
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) { done[req] = 0; if (err_out) err_out[req] = __hip_atomic_load(&d.err[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    double a = 0.;
    for (int k = threadIdx.x; k < cnt; k += 256) a += __hip_atomic_load(&partial[(size_t)req * maxcnt + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = block_sum<256>(a, scratch);
    if (threadIdx.x == 0) out[req] = a;
}
// ---- those searches in rounds the DEVICE drives (round 4; search_mode 5) ---------------------------------------------
// The rounds above cost a launch pair, a stream wait and the host's optimiser step each, fifty times per M-step, and their kernel
// is as long as its slowest half-wave: a sampled segment whose list of states overflowed walks six cells per lane, and the
// samples of the outlier dispersions are weighted towards exactly those segments (30 us for 10 us of work).  Here
//   * the optimisers' state (rmxh::Nm1, the state machine the host drives in the round-based path) lives in device memory and a
//     round is a kernel pair with nothing in between (back-to-back kernels of a stream start within a microsecond of each other):
//     k_search_round evaluates, k_search_advance (a block per request) sums the blocks' partial sums in a fixed order, feeds the
//     value to the request's optimiser and leaves the next point in its state -- or, when it is finished, the result in
//     host-visible memory.  The host queues the grid round and a batch of rounds back to back without waiting, then looks at the
//     finished flags (a round of a finished request is blocks that return).  (One kernel per round with a last-block ticket was
//     measured first: a release fence per block -- an L2 write-back -- made the round longer than the kernel pair.);
//   * a request's (sampled segment, listed state) CELLS are laid out flat, one per thread (k_search_setup: prefix sums of the
//     segments' cell counts, and the candidate-free parts of the per-segment constants); a block first evaluates the candidate's
//     constant for the few segments its 256 cells belong to (LDS), then its cells.
// Same per-cell expressions as ell_segment_sparse (cell_ll_regs and the register override of the betabin tables); the sum runs
// over blocks of cells instead of segments and log(candidate) of the Nelder-Mead points is the device's: values differ from the
// round-based path's by rounding.  Points outside [lo, hi] are +inf without an evaluation (cn_model.py:542-543).
struct NmState {
    rmxh::Nm1 nm;
    double x0, last, v, lv;
    int32_t done, pad0;
};
struct NmArgs {
    double lo[4], hi[4];
    double gv[4][RMX_MULTI_G], glv[4][RMX_MULTI_G];
    int16_t rlist[64];
    int8_t slot[64];
    int32_t maskbit[4];
    int32_t G, grid_stage;
    int32_t blk0[65];             // blocks of 256 cells before request q (k_search_round's grid is the requests' blocks one after the other)
    int32_t nreq;
};
struct NmLayout {                 // per request q, rows of pitch NM_MAX_SAMPLE (+ 1)
    int32_t *pre;                 // [64][NM_MAX_SAMPLE + 1] cells before sampled segment i; pre[cnt] = the request's cells
    double *fix, *k1;             // [64][NM_MAX_SAMPLE] candidate-free part of the component's constant; constant of the other dispersion
};
__device__ __forceinline__ int nm_const_index(int mask) { return mask == CM_LT0 ? 0 : (mask == CM_LT1 ? 2 : (mask == CM_LA0 ? 4 : 6)); }
// grid (nreq), block 256
__global__ __launch_bounds__(256) void k_search_setup(Dev d, NmArgs na, const int32_t *samples, const int32_t *counts, NmLayout lay, double *cells_out) {
    __shared__ int cl[NM_MAX_SAMPLE + 1];
    const int req = blockIdx.x, r = na.rlist[req], sl = na.slot[req], mask = na.maskbit[sl];
    const int cnt = counts[sl * d.R + r];
    const int32_t *smp = samples + ((size_t)sl * d.R + r) * d.N;
    const bool nb = (mask & (CM_LT0 | CM_LT1)) != 0;
    const RestartParams &rp = d.rp[r];
    unsigned err = 0;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int n = smp[i];
        const int c = d.sig_cnt[(size_t)r * d.N + n];
        cl[i + 1] = c == 255 ? d.S : c;
        const double x = d.x[n], y0 = d.y[2 * (size_t)n], ys = y0 + d.y[2 * (size_t)n + 1];
        lay.k1[(size_t)req * NM_MAX_SAMPLE + i] = seg_const_value(rp, x, y0, ys, nm_const_index(mask) + 1);
        lay.fix[(size_t)req * NM_MAX_SAMPLE + i] = nb ? lgamma_pos(x + 1) : (lgamma_pos(ys + 1) - lgamma_pos(y0 + 1) - lgamma_pos(ys - y0 + 1));
        // the table flags of the states off the list (ell_segment_sparse reports them per evaluation)
        if (c != 255 && !nb) { SegCtx sc; sc.ma = d.mask_a[n]; sc.ys = ys; table_static_errors<CM_LA0>(sc, d.stFlagsAgg[(size_t)r * d.C + d.seg_class[n]], err); }
    }
    if (threadIdx.x == 0) cl[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < cnt; i++) cl[i + 1] += cl[i];
    __syncthreads();
    for (int i = threadIdx.x; i <= cnt; i += 256) lay.pre[(size_t)req * (NM_MAX_SAMPLE + 1) + i] = cl[i];
    if (threadIdx.x == 0) cells_out[req] = (double)cl[cnt];      // (host-visible: the host sizes the rounds' grid from it)
    if (err) atomicOr(&d.err[r], err);
}
template <int MASK>
__device__ __forceinline__ double nm_cell(const Dev &d, int r, int n, int s, double v, double lv, double k0, double k1, unsigned &err) {
    RestartParams rp = d.rp[r];
    if (MASK == CM_LT0) { rp.p[RMX_P_NEGBIN_R_0] = v; rp.logr[0] = lv; }
    if (MASK == CM_LT1) { rp.p[RMX_P_NEGBIN_R_1] = v; rp.logr[1] = lv; }
    if (MASK == CM_LA0) rp.p[RMX_P_BETABIN_M_0] = v;
    if (MASK == CM_LA1) rp.p[RMX_P_BETABIN_M_1] = v;
    SegCtx sc;
    sc.x = d.x[n]; sc.l = d.l[n]; sc.logl = d.logl[n]; sc.y0 = d.y[2 * (size_t)n]; sc.y1 = d.y[2 * (size_t)n + 1]; sc.ys = sc.y0 + sc.y1;
    sc.mt = d.mask_t[n]; sc.ma = d.mask_a[n];
#pragma unroll
    for (int k = 0; k < 4; k++) { sc.cnb[k] = 0.; sc.cbb[k] = 0.; }
    if (MASK == CM_LT0) { sc.cnb[0] = k0; sc.cnb[1] = k1; }
    if (MASK == CM_LT1) { sc.cnb[2] = k0; sc.cnb[3] = k1; }
    if (MASK == CM_LA0) { sc.cbb[0] = k0; sc.cbb[1] = k1; }
    if (MASK == CM_LA1) { sc.cbb[2] = k0; sc.cbb[3] = k1; }
    const size_t rn = (size_t)r * d.N + n;
    StateRegs st_; load_state_regs(d, r, d.seg_class[n], s, st_);
    if ((MASK & (CM_LA0 | CM_LA1)) && !(st_.fl & ST_LOH_M)) {
        const bool ok_ = !(st_.fl & (ST_E_BADP | ST_E_TD | ST_E_LOH));      // as state_tables_body
        if (MASK & CM_LA0) { st_.M0 = v; st_.lgA0 = ok_ ? lgamma_pos(v * st_.p) : 0.; st_.lgB0 = ok_ ? lgamma_pos(v * (1 - st_.p)) : 0.; }
        if (MASK & CM_LA1) { st_.M1 = v; st_.lgA1 = ok_ ? lgamma_pos(v * st_.p) : 0.; st_.lgB1 = ok_ ? lgamma_pos(v * (1 - st_.p)) : 0.; }
    }
    double LT[2], LA[4];
    cell_ll_regs<MASK>(rp, sc, st_, LT, LA, err);
    const double ps = d.post[rs_off(d, r, n) + s];
    double acc = 0.;
    if (MASK & CM_LT0) acc += ps * d.qt[rn * 2] * LT[0];
    if (MASK & CM_LT1) acc += ps * d.qt[rn * 2 + 1] * LT[1];
    if (MASK & CM_LA0) { const double qa0 = d.qa[rn * 2]; acc += ps * qa0 * d.qs[rn * 2] * LA[0]; acc += ps * qa0 * d.qs[rn * 2 + 1] * LA[1]; }
    if (MASK & CM_LA1) { const double qa1 = d.qa[rn * 2 + 1]; acc += ps * qa1 * d.qs[rn * 2] * LA[2]; acc += ps * qa1 * d.qs[rn * 2 + 1] * LA[3]; }
    return acc;
}
// grid (blk0[nreq], grid_stage ? G : 1), block 256.  partial [Gz][blk0[nreq]].
__global__ __launch_bounds__(256) void k_search_round(Dev d, NmArgs na, const int32_t *samples, const int32_t *counts, NmLayout lay, double *partial, const NmState *state) {
    __shared__ int pre[NM_MAX_SAMPLE + 1];
    __shared__ double k0s[NM_MAX_SAMPLE];
    __shared__ double scratch[8];
    const int gz = blockIdx.y, tid = threadIdx.x;
    int req = 0;
    while (req + 1 < na.nreq && (int)blockIdx.x >= na.blk0[req + 1]) req++;
    const NmState &st = state[req];
    if (!na.grid_stage && st.done) return;
    const int r = na.rlist[req], sl = na.slot[req], mask = na.maskbit[sl];
    const int cnt = counts[sl * d.R + r];
    const int T = lay.pre[(size_t)req * (NM_MAX_SAMPLE + 1) + cnt];
    const int c0 = ((int)blockIdx.x - na.blk0[req]) * 256;
    if (c0 >= T) return;
    const int32_t *smp = samples + ((size_t)sl * d.R + r) * d.N;
    const double v = na.grid_stage ? na.gv[sl][gz] : st.v, lv = na.grid_stage ? na.glv[sl][gz] : st.lv;
    for (int i = tid; i <= cnt; i += 256) pre[i] = lay.pre[(size_t)req * (NM_MAX_SAMPLE + 1) + i];
    __syncthreads();
    auto segment_of = [&](int c) {      // the last segment i with pre[i] <= c (empty segments are skipped: pre[i + 1] > c)
        int lo_ = 0, hi_ = cnt;
        while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (pre[mid] <= c) lo_ = mid; else hi_ = mid; }
        return lo_;
    };
    const int i_first = segment_of(c0), i_last = segment_of(min(c0 + 255, T - 1));
    // the candidate's constant of the segments this block's cells belong to: seg_const_value with its candidate-free part from the setup
    const bool nb = (mask & (CM_LT0 | CM_LT1)) != 0;
    for (int k = tid; k <= i_last - i_first; k += 256) {
        const int i = i_first + k, n = smp[i];
        const double fx = lay.fix[(size_t)req * NM_MAX_SAMPLE + i];
        if (nb) k0s[k] = lgamma_pos(d.x[n] + v) - fx - lgamma_pos(v);
        else { const double ys = d.y[2 * (size_t)n] + d.y[2 * (size_t)n + 1]; k0s[k] = fx - lgamma_pos(ys + v) + lgamma_pos(v); }
    }
    __syncthreads();
    double acc = 0.;
    unsigned err = 0;
    const int c = c0 + tid;
    if (c < T) {
        const int i = segment_of(c), jj = c - pre[i], n = smp[i];
        const size_t rn = (size_t)r * d.N + n;
        const int s = d.sig_cnt[rn] == 255 ? jj : (int)d.sig_idx[rn * RMX_SIGK + jj];
        const double k0 = k0s[i - i_first], k1 = lay.k1[(size_t)req * NM_MAX_SAMPLE + i];
        switch (mask) {
        case CM_LT0: acc = nm_cell<CM_LT0>(d, r, n, s, v, lv, k0, k1, err); break;
        case CM_LT1: acc = nm_cell<CM_LT1>(d, r, n, s, v, lv, k0, k1, err); break;
        case CM_LA0: acc = nm_cell<CM_LA0>(d, r, n, s, v, lv, k0, k1, err); break;
        default:     acc = nm_cell<CM_LA1>(d, r, n, s, v, lv, k0, k1, err); break;
        }
    }
    if (err) atomicOr(&d.err[r], err);
    acc = block_sum<256>(acc, scratch);
    if (tid == 0) partial[(size_t)gz * gridDim.x + blockIdx.x] = acc;
}
// grid (nreq), block 256.  out[q] = {xopt, last point evaluated}; done_out[q] = 1 + the restart's error word
__global__ __launch_bounds__(256) void k_search_advance(Dev d, NmArgs na, const double *partial, NmState *state, double *out, uint32_t *done_out) {
    __shared__ double scratch[8];
    const int req = blockIdx.x, tid = threadIdx.x;
    NmState &st = state[req];
    if (!na.grid_stage && st.done) return;
    const int Gz = na.grid_stage ? na.G : 1;
    const int r = na.rlist[req], sl = na.slot[req];
    const int b0 = na.blk0[req], nbq = na.blk0[req + 1] - b0, TB = na.blk0[na.nreq];
    double f = 0., x0 = 0., best = INFINITY;
    for (int g = 0; g < Gz; g++) {
        double a = 0.;
        for (int k = tid; k < nbq; k += 256) a += partial[(size_t)g * TB + b0 + k];
        a = block_sum<256>(a, scratch);
        if (tid == 0) {
            if (na.grid_stage) { const double J = -a; if (g == 0 || J < best) { best = J; x0 = na.gv[sl][g]; } }      // np.argmin: first minimum
            else f = -a;
        }
    }
    if (tid != 0) return;
    rmxh::Nm1 nm;                      // (a copy in registers: advancing it in place is a chain of dependent global accesses)
    double last;
    if (na.grid_stage) { last = na.gv[sl][na.G - 1]; st.x0 = x0; }
    else { nm = st.nm; x0 = st.x0; last = st.last; }
    const double lo = na.lo[sl], hi = na.hi[sl];
    bool go = false;
    while (nm.advance(x0, f)) {
        const double v = nm.req;
        if (v < lo || v > hi) { f = INFINITY; continue; }
        st.v = v; st.lv = log(v); last = v; go = true;
        break;
    }
    st.nm = nm; st.last = last; st.done = go ? 0 : 1;
    if (!go) {
        out[2 * req] = nm.xopt(); out[2 * req + 1] = last;
        done_out[req] = 1u + d.err[r];
    }
}
// ---- the same searches as ONE launch (round 5, late; search_mode 7) ---------------------------------------------------------------
// k_search_round / k_search_advance are a kernel pair per round, ~53 rounds per M-step; next to the other restart group's sweeps a pair takes
// 69 us for 18 us of kernels (every launch waits for its predecessor to drain and for a slot).  Here a request's blocks stay resident for
// the whole search: a block evaluates its 256 cells for the request's current point, publishes its partial sum (a relaxed agent-scope
// store, sc1: no fence, no L2 write-back -- those cost the neighbours 5 %), then every block of the request fetches ALL the
// request's partial sums of the round -- polling each slot until it is no longer the all-ones word the host filled the buffer with --
// adds them in k_search_advance's order and advances ITS OWN copy of the request's optimiser: the copies see the same doubles in the
// same order, so they stay identical and nothing but the partial sums crosses between blocks.  Same cells, same partition into blocks,
// same sums as the kernel pair: the results are bit-identical to search_mode 5 (tests/test_hip_parity.py).  Blocks depend only on the
// blocks of their own request (consecutive block indices), so the launch makes progress wherever one request's blocks are resident.
// partial: [G + Nm1::maxfun + 2][gridDim.x], filled with 0xff bytes by the host.  out[2 q] = xopt, out[2 q + 1] = the last point.
__device__ __forceinline__ double nm_poll(const double *p, bool &dead) {      // (bounded like the lattice clusters' waits: RMX_CLUSTER_SPINS polls, then the block gives up)
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(p);
    unsigned long long u;
    unsigned spins = 0;
    while ((u = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > RMX_CLUSTER_SPINS) { dead = true; return 0.; }
    }
    return __longlong_as_double((long long)u);
}
__global__ __launch_bounds__(256) void k_search_persist(Dev d, NmArgs na, const int32_t *samples, const int32_t *counts, NmLayout lay, double *partial, double *out) {
    __shared__ int pre[NM_MAX_SAMPLE + 1];
    __shared__ double k0s[NM_MAX_SAMPLE];
    __shared__ double scratch[8];
    __shared__ double bc_v, bc_lv, bc_x0, bc_last;
    __shared__ int bc_go;
    __shared__ __align__(8) unsigned char nm_raw[sizeof(rmxh::Nm1)];      // the block's copy of the request's optimiser (thread 0 advances it; in LDS: the kernel's register budget decides how many blocks are resident)
    rmxh::Nm1 &nm = *reinterpret_cast<rmxh::Nm1 *>(nm_raw);
    const int tid = threadIdx.x, blk = blockIdx.x, TB = gridDim.x;
    int req = 0;
    while (req + 1 < na.nreq && blk >= na.blk0[req + 1]) req++;
    const int r = na.rlist[req], sl = na.slot[req], mask = na.maskbit[sl];
    const int cnt = counts[sl * d.R + r];
    const int T = lay.pre[(size_t)req * (NM_MAX_SAMPLE + 1) + cnt];
    const int b0 = na.blk0[req], nbq = na.blk0[req + 1] - b0;
    const int c0 = (blk - b0) * 256;
    const int32_t *smp = samples + ((size_t)sl * d.R + r) * d.N;
    for (int i = tid; i <= cnt; i += 256) pre[i] = lay.pre[(size_t)req * (NM_MAX_SAMPLE + 1) + i];
    __syncthreads();
    auto segment_of = [&](int c) {
        int lo_ = 0, hi_ = cnt;
        while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if (pre[mid] <= c) lo_ = mid; else hi_ = mid; }
        return lo_;
    };
    const bool any = c0 < T;      // (the host's grid holds exactly the blocks that have cells)
    const int i_first = any ? segment_of(c0) : 0, i_last = any ? segment_of(min(c0 + 255, T - 1)) : -1;
    const bool nb = (mask & (CM_LT0 | CM_LT1)) != 0;
    // this thread's cell: the same for every point of the search
    const int c = c0 + tid;
    const bool has = c < T;
    int ci = 0, n = 0, s = 0; double k1 = 0.;
    if (has) {
        ci = segment_of(c); n = smp[ci];
        const int jj = c - pre[ci];
        const size_t rn = (size_t)r * d.N + n;
        s = d.sig_cnt[rn] == 255 ? jj : (int)d.sig_idx[rn * RMX_SIGK + jj];
        k1 = lay.k1[(size_t)req * NM_MAX_SAMPLE + ci];
    }
    const double lo = na.lo[sl], hi = na.hi[sl];
    const int G = na.G;
    if (tid == 0) { nm = rmxh::Nm1(); bc_x0 = 0.; bc_last = 0.; }
    double v = 0., lv = 0.;
    unsigned err = 0;
    for (int it = 0;; it++) {
        if (it < G) { v = na.gv[sl][it]; lv = na.glv[sl][it]; }
        // the candidate's constant of the segments this block's cells belong to, then the cells
        __syncthreads();
        for (int k = tid; k <= i_last - i_first; k += 256) {
            const int i = i_first + k, n_ = smp[i];
            const double fx = lay.fix[(size_t)req * NM_MAX_SAMPLE + i];
            if (nb) k0s[k] = lgamma_pos(d.x[n_] + v) - fx - lgamma_pos(v);
            else { const double ys = d.y[2 * (size_t)n_] + d.y[2 * (size_t)n_ + 1]; k0s[k] = fx - lgamma_pos(ys + v) + lgamma_pos(v); }
        }
        __syncthreads();
        double acc = 0.;
        if (has) {
            const double k0 = k0s[ci - i_first];
            // (the cell's operands are re-read every round: kept across the rounds they take the kernel to 142 registers, and the number of resident
            //  blocks next to the other restart group's forward-backward workgroups is what the launch's length depends on)
            int n_ = n, s_ = s;
            asm volatile("" : "+v"(n_), "+v"(s_));
            switch (mask) {
            case CM_LT0: acc = nm_cell<CM_LT0>(d, r, n_, s_, v, lv, k0, k1, err); break;
            case CM_LT1: acc = nm_cell<CM_LT1>(d, r, n_, s_, v, lv, k0, k1, err); break;
            case CM_LA0: acc = nm_cell<CM_LA0>(d, r, n_, s_, v, lv, k0, k1, err); break;
            default:     acc = nm_cell<CM_LA1>(d, r, n_, s_, v, lv, k0, k1, err); break;
            }
        }
        acc = block_sum<256>(acc, scratch);
        if (tid == 0) {
            if (acc != acc) acc = __longlong_as_double(0x7ff8000000000000ll);      // (never the all-ones word)
            __hip_atomic_store(partial + (size_t)it * TB + blk, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (it < G - 1) continue;
        double f = 0.;
        bool dead = false;
        if (it == G - 1) {
            // the grid stage: np.argmin over the candidates' sums (first minimum), k_search_advance's order
            double best = INFINITY;
            for (int g = 0; g < G; g++) {
                double a = 0.;
                for (int k = tid; k < nbq; k += 256) a += nm_poll(partial + (size_t)g * TB + b0 + k, dead);
                a = block_sum<256>(a, scratch);
                if (tid == 0) { const double J = -a; if (g == 0 || J < best) { best = J; bc_x0 = na.gv[sl][g]; } }
            }
            if (tid == 0) bc_last = na.gv[sl][G - 1];
        } else {
            double a = 0.;
            for (int k = tid; k < nbq; k += 256) a += nm_poll(partial + (size_t)it * TB + b0 + k, dead);
            a = block_sum<256>(a, scratch);
            f = -a;
        }
        if (__syncthreads_or(dead ? 1 : 0)) {      // a partial sum never came (a block of the request not resident, or gone): the request fails instead of hanging
            if (tid == 0) atomicOr(&d.err[r], RMX_ERR_WAIT);
            return;
        }
        if (tid == 0) {
            bool go = false;
            const double x0 = bc_x0;
            while (nm.advance(x0, f)) {
                const double vv = nm.req;
                if (vv < lo || vv > hi) { f = INFINITY; continue; }
                bc_v = vv; bc_lv = log(vv); bc_last = vv; go = true;
                break;
            }
            bc_go = go ? 1 : 0;
        }
        __syncthreads();
        if (!bc_go) break;
        v = bc_v; lv = bc_lv;
    }
    if (err) atomicOr(&d.err[r], err);
    if (tid == 0 && blk == b0) { out[2 * req] = nm.xopt(); out[2 * req + 1] = bc_last; }
}
// done_out[q] = 1 + the restart's error word, after k_search_persist (a kernel boundary: every block's flags are in)
__global__ void k_search_flags(Dev d, NmArgs na, uint32_t *done_out) {
    const int q = threadIdx.x;
    if (q < na.nreq) done_out[q] = 1u + d.err[na.rlist[q]];
}
// grid (nreq * Gz): the sum of k_ell_final_batch over the partials of (request, candidate)
__global__ void k_ell_search_final(Dev d, SearchVals sv, const int32_t *counts, const double *partial, int maxcnt, double *out, uint32_t *err_out) {
    __shared__ double scratch[8];
    const int req = blockIdx.x / sv.Gz;
    const int r = sv.rlist[req];
    if (err_out && threadIdx.x == 0 && blockIdx.x % sv.Gz == 0) err_out[req] = d.err[r];
    double a = 0.;
    for (int i = threadIdx.x; i < counts[r]; i += 256) a += partial[(size_t)blockIdx.x * maxcnt + i];
    a = block_sum<256>(a, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}
// full-data E[ll] from (A, B) for a restart range: grid (ELBO_BLOCKS, nr) -> partial[(r-r0)][blk]
__global__ void k_ell_full_batch(Dev d, int r0, double *partial) {
    __shared__ double scratch[8];
    const int r = r0 + blockIdx.y;
    double acc = 0.;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < d.N; n += gridDim.x * blockDim.x) {
        const size_t rn = (size_t)r * d.N + n;
        const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
        const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
        acc += qt0 * d.A[rn * 2] + qt1 * d.A[rn * 2 + 1];
        acc += qa0 * qs0 * d.Bv[rn * 4] + qa0 * qs1 * d.Bv[rn * 4 + 1] + qa1 * qs0 * d.Bv[rn * 4 + 2] + qa1 * qs1 * d.Bv[rn * 4 + 3];
    }
    acc = block_sum<256>(acc, scratch);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}
// the same sum split into the four likelihood components (NB total u = 0 / 1, BB allele v = 0 / 1: what negbin_r_0/1 and
// betabin_M_0/1 move): partial[((r-r0) * 4 + c)][blk]
__global__ void k_ell_comp_batch(Dev d, int r0, double *partial) {
    __shared__ double scratch[8];
    const int r = r0 + blockIdx.y;
    double c0 = 0., c1 = 0., c2 = 0., c3 = 0.;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < d.N; n += gridDim.x * blockDim.x) {
        const size_t rn = (size_t)r * d.N + n;
        const double qt0 = d.qt[rn * 2], qt1 = d.qt[rn * 2 + 1], qa0 = d.qa[rn * 2], qa1 = d.qa[rn * 2 + 1];
        const double qs0 = d.qs[rn * 2], qs1 = d.qs[rn * 2 + 1];
        c0 += qt0 * d.A[rn * 2]; c1 += qt1 * d.A[rn * 2 + 1];
        c2 += qa0 * qs0 * d.Bv[rn * 4] + qa0 * qs1 * d.Bv[rn * 4 + 1];
        c3 += qa1 * qs0 * d.Bv[rn * 4 + 2] + qa1 * qs1 * d.Bv[rn * 4 + 3];
    }
    c0 = block_sum<256>(c0, scratch); c1 = block_sum<256>(c1, scratch); c2 = block_sum<256>(c2, scratch); c3 = block_sum<256>(c3, scratch);
    if (threadIdx.x == 0) {
        double *pp = partial + (size_t)blockIdx.y * 4 * gridDim.x + blockIdx.x;
        pp[0] = c0; pp[gridDim.x] = c1; pp[2 * gridDim.x] = c2; pp[3 * gridDim.x] = c3;
    }
}
__global__ void k_sum_partials(const double *partial, int n, double *out) {   // grid (nr)
    __shared__ double scratch[8];
    double a = 0.;
    for (int i = threadIdx.x; i < n; i += 256) a += partial[(size_t)blockIdx.x * n + i];
    a = block_sum<256>(a, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}

// hmm_log_norm_const = sum of the per-row shares (bpmodel.pyx:946).  grid 1, block 256
__global__ void k_logz(Dev d, int r, double *out) {
    __shared__ double scratch[8];
    double z = 0.;
    for (int n = threadIdx.x; n < d.N; n += 256) z += d.rowZ[(size_t)r * d.N + n];
    z = block_sum<256>(z, scratch);
    if (threadIdx.x == 0) *out = z;
}

// single cell, for tests
__global__ void k_cell_probe(Dev d, int r, int n, int s, double *out6) {
    const RestartParams &rp = d.rp[r];
    SegCtx sc; load_seg(d, r, n, sc);
    unsigned err = 0; double LT[2], LA[4];
    cell_ll(d, rp, sc, r, d.seg_class[n], s, LT, LA, err);
    out6[0] = LT[0]; out6[1] = LT[1]; out6[2] = LA[0]; out6[3] = LA[1]; out6[4] = LA[2]; out6[5] = LA[3];
    if (err) atomicOr(&d.err[r], err);
}

// The other per-cell cpdef methods of RemixtModel for one (segment n, state s), reference accumulation order:
//   out[0]            calculate_expected_total_reads                     (bpmodel.pyx:686-698)
//   out[1 .. 1+M)     calculate_expected_total_reads_partial_h           (:700-708)
//   out[5]            calculate_expected_allele_ratio                    (:710-725)  [error: total_depth <= 0]
//   out[6 .. 6+M)     calculate_expected_allele_ratio_partial_h          (:727-745)
//   out[10]           calculate_log_prior_cn                             (:747-750)
//   out[11 .. 11+M)   calculate_log_likelihood_total_partial_h (u)       (:778-807)
//   out[15 .. 15+M)   calculate_log_likelihood_allele_partial_h (v, w)   (:855-896)
// `want` selects which groups are evaluated (bit 0: total reads, 1: allele ratio, 2: prior, 3: d ll_total, 4: d ll_allele),
// so that a query raises only the errors the reference's method would raise.
__global__ void k_cell_probe_h(Dev d, int r, int n, int s, int u, int v, int w, int want, double *out) {
    const RestartParams &rp = d.rp[r];
    const int cls = d.seg_class[n], M = d.M;
    const int8_t *cn = d.cn + ((size_t)cls * d.S + s) * M * 2;
    const int8_t *tot = d.tot + ((size_t)cls * d.S + s) * M;
    const unsigned sf = d.sflags[(size_t)cls * d.S + s];
    const double l = d.l[n];
    unsigned err = 0;
    for (int i = 0; i < 19; i++) out[i] = 0.;
    double mu = 0.;
    for (int m = 0; m < M; m++) mu += rp.h[m] * (double)tot[m];
    mu *= l;
    if (want & 1) { out[0] = mu; for (int m = 0; m < M; m++) out[1 + m] = l * (double)tot[m]; }
    double minor = 0., total = 0.;
    for (int m = 0; m < M; m++) { minor += rp.h[m] * (double)cn[m * 2]; total += rp.h[m] * (double)tot[m]; }
    if (want & 2) {
        if (total <= 0.) err |= RMX_ERR_TOTAL_DEPTH;
        else {
            out[5] = minor / total;
            for (int m = 0; m < M; m++) out[6 + m] = ((double)cn[m * 2] * total - minor * (double)tot[m]) / (total * total);
        }
    }
    if (want & 4) out[10] = -1.0 * (double)((sf >> 2) & 3) * l * rp.p[RMX_P_DIVERGENCE_WEIGHT];
    if ((want & 8) && d.mask_t[n] && !(!d.nc && (sf & 1u))) {
        const double rr = u == 0 ? rp.p[RMX_P_NEGBIN_R_0] : rp.p[RMX_P_NEGBIN_R_1];
        const double x = d.x[n];
        const double pm = x / mu - (rr + x) / (rr + mu);
        if (pm != pm) err |= RMX_ERR_NAN_GRAD;
        for (int m = 0; m < M; m++) out[11 + m] = (l * (double)tot[m]) * pm;
    }
    if ((want & 16) && d.mask_a[n] && !(!d.nc && (sf & 2u))) {
        if (total <= 0.) err |= RMX_ERR_TOTAL_DEPTH;
        else {
            const double p = minor / total;
            const double Mv = v == 0 ? rp.p[RMX_P_BETABIN_M_0] : rp.p[RMX_P_BETABIN_M_1];
            const double y0 = d.y[2 * (size_t)n], y1 = d.y[2 * (size_t)n + 1], ys = y0 + y1;
            if (ys != 0.) {
                const double k = w == 0 ? y0 : y1;
                if (p <= 0. || (1 - p) <= 0.) err |= RMX_ERR_BAD_P;
                else {
                    const double pp = (Mv * digamma_as103(k + Mv * p, err) + (-Mv) * digamma_as103(ys - k + Mv * (1 - p), err)
                                       - Mv * digamma_as103(Mv * p, err) - (-Mv) * digamma_as103(Mv * (1 - p), err));
                    if (pp != pp) err |= RMX_ERR_NAN_GRAD;
                    for (int m = 0; m < M; m++) out[15 + m] = (((double)cn[m * 2] * total - minor * (double)tot[m]) / (total * total)) * pp;
                }
            }
        }
    }
    if (err) atomicOr(&d.err[r], err);
}

// =============================================================================
// Viterbi (max_product, bpmodel.pyx:1296-1333), bit-exact: the lattice is carried
// through telomeres exactly like the reference (no per-chain re-basing, which
// would change float rounding), additions and comparisons only.
// Forward: one workgroup per restart (grid nr); thread (o,p); back-pointers (first maximum)
// are recorded so the trace-back is pointer chasing; identical to the reference's
// recomputed argmax (:1327-1331) because it is the same expression and tie rule.
// =============================================================================
__global__ __launch_bounds__(1024) void k_viterbi(Dev d, int r0, int P, uint16_t *bp_all /* [nr][N][S] */, double *final_all /* [nr][S] */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    uint16_t *bp = bp_all + (size_t)blockIdx.x * d.N * S;
    double *final_row = final_all + (size_t)blockIdx.x * S;
    double *V = (double *)smem_raw;        // [2][S]
    double *pdl = V + 2 * S;               // [M*D]
    const int o = t / P, p = t % P;
    const bool act = o < S;
    const int QPT = (S + P - 1) / P;
    const double *f = d.f + rs_off(d, r, 0);
    if (t < S) V[t] = f[t];
    __syncthreads();
    for (int n = 1; n < d.N; n++) {
        const int cur = (n - 1) & 1, nxt = n & 1, tn = n - 1;
        const int tc = d.tclass[tn], bs = d.brk_slot[tn];
        const double *pd = nullptr;
        if (tc >= 0 && bs >= 0) {
            const double *pdg = d.pd_lt + ((size_t)r * d.NBE + bs) * M * D;
            for (int i = t; i < M * D; i += NT) pdl[i] = pdg[i];
            __syncthreads();
            pd = pdl;
        }
        double best = -INFINITY; int bi = 0;
        if (act) for (int rr = 0; rr < QPT; rr++) {
            const int i = p * QPT + rr;
            if (i < S) {
                const double T = (tc < 0) ? 0. : trans_value(d, tn, i, o, pd);
                const double v = V[cur * S + i] + T;
                if (v > best) { best = v; bi = i; }
            }
        }
        for (int off = 1; off < P; off <<= 1) {
            const double ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (act && p == 0) {
            V[nxt * S + o] = best + f[(size_t)n * d.SP + o];
            bp[(size_t)n * S + o] = (uint16_t)bi;
        }
        __syncthreads();
    }
    if (t < S) final_row[t] = V[((d.N - 1) & 1) * S + t];
}
// Same lattice, one workgroup per restart (grid nr), with the transition values of class 0 held
// in registers: thread (o, p) keeps T(i, o) for its QPT source states i, which never change along
// the genome while the adjacency is a plain one of class 0 (every adjacency of an experiment
// without masked segments).  A step is then QPT LDS reads of V, adds and compares per thread and
// one barrier; breakend, telomere and other-class adjacencies take the expression of k_viterbi.
// Values, comparison order and tie rule are those of k_viterbi, so the paths are identical.
// Vector-memory traffic of the step loop goes through untracked inline asm (see gload8): the row of
// f and the adjacency's class / breakend slot for step n+1 are requested at the end of step n and
// retired by ONE vmcnt(0) after the compares of step n+1, when they and the back-pointer store of
// step n are a step old.
__device__ __forceinline__ void gload4(int &dst, const int *src) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
__device__ __forceinline__ void gwait_all(double &a, int &b, int &c) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) :: "memory"); }
__device__ __forceinline__ void gstore2(uint16_t *dst, int v) {
    asm volatile("global_store_short %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
// The QMAX LDS reads of a step as an explicit pipeline (hipcc serialises them: read, wait, compare):
// pairs of values by ds_read2_b64 through untracked asm, DEPTH pairs in flight, each retired by a
// counted lgkmcnt wait right before its two compares.  LDS returns in order, and the block issues
// no other LDS traffic between the first read and the last wait.
typedef double vit_d2 __attribute__((ext_vector_type(2)));
template <int K> __device__ __forceinline__ void vit_rd2(vit_d2 &dst, unsigned addr) {
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(2 * K), "n"(2 * K + 1) : "memory");
}
template <int CNT> __device__ __forceinline__ void vit_wait(vit_d2 &x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(CNT) : "memory"); }
template <int G, int NG, int DEPTH> struct vit_pipe {
    // (one running maximum: four independent ones merged at the end -- shorter dependent chains, the same tie rule -- were measured
    // in round 4 and are slower, 164 against 153 ms per 50 000-step lattice: the step is not bound by the compare -> select chain)
    template <int QMAX>
    static __device__ __forceinline__ void step(vit_d2 (&buf)[DEPTH], const double (&T)[QMAX], unsigned addr, double &best, int &bi, int i0) {
        constexpr int younger = (NG - 1 - G) < (DEPTH - 1) ? (NG - 1 - G) : (DEPTH - 1);
        vit_wait<younger>(buf[G % DEPTH]);
        const vit_d2 x = buf[G % DEPTH];
        if constexpr (G + DEPTH < NG) vit_rd2<G + DEPTH>(buf[G % DEPTH], addr);
        const double v0 = x.x + T[2 * G], v1 = x.y + T[2 * G + 1];
        if (v0 > best) { best = v0; bi = i0 + 2 * G; }
        if (v1 > best) { best = v1; bi = i0 + 2 * G + 1; }
        if constexpr (G + 1 < NG) vit_pipe<G + 1, NG, DEPTH>::step(buf, T, addr, best, bi, i0);
    }
    template <int QMAX>
    static __device__ __forceinline__ void run(const double (&T)[QMAX], unsigned addr, double &best, int &bi, int i0) {
        static_assert(G == 0 && 2 * NG == QMAX && DEPTH <= NG && 2 * NG + 1 < 256, "pipeline shape");
        vit_d2 buf[DEPTH];
        fill<0>(buf, addr);
        step(buf, T, addr, best, bi, i0);
    }
    template <int I> static __device__ __forceinline__ void fill(vit_d2 (&buf)[DEPTH], unsigned addr) {
        vit_rd2<I>(buf[I], addr);
        if constexpr (I + 1 < DEPTH) fill<I + 1>(buf, addr);
    }
};
template <int QMAX>
__global__ __launch_bounds__(768) void k_viterbi_reg(Dev d, int r0, int P, uint16_t *bp_all /* [nr][N][S] */, double *final_all /* [nr][S] */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    const int QPT = (S + P - 1) / P;
    const int SV = P * QPT + QMAX;         // padded row: reads past S see 0 and are paired with T = -inf
    double *V = (double *)smem_raw;        // [2][SV]
    double *pdl = V + 2 * SV;              // [M*D]
    uint16_t *bp = bp_all + (size_t)blockIdx.x * d.N * S;
    const int o = t / P, p = t % P;
    const bool act = o < S;
    const int oc = act ? o : S - 1;
    const int i0 = p * QPT;
    const double *f = d.f + rs_off(d, r, 0);
    double Treg[QMAX];
#pragma unroll
    for (int rr = 0; rr < QMAX; rr++) {
        const int i = i0 + rr;
        Treg[rr] = (act && rr < QPT && i < S && d.TC > 0) ? d.Tval[(size_t)i * S + o] : -INFINITY;
    }
    // a use of every value: their loads retire here, not at a vmcnt(0) inside the step loop
#pragma unroll
    for (int rr = 0; rr < QMAX; rr++) asm volatile("" : "+v"(Treg[rr]));
    for (int i = t; i < 2 * SV; i += NT) V[i] = 0.;
    __syncthreads();
    if (t < S) V[t] = f[t];
    __syncthreads();
    if (d.N > 1) {
        double fn; int tcv, bsv;
        gload8(fn, f + (size_t)1 * d.SP + oc); gload4(tcv, d.tclass); gload4(bsv, d.brk_slot);
        gwait_all(fn, tcv, bsv);
        for (int n = 1; n < d.N; n++) {
            const int cur = (n - 1) & 1, nxt = n & 1, tn = n - 1;
            const int tc = __builtin_amdgcn_readfirstlane(tcv), bs = __builtin_amdgcn_readfirstlane(bsv);
            const double fcur = fn;
            {   // requests for step n+1 (the last step re-reads its own row)
                const int nn = n + 1 < d.N ? n + 1 : n;
                gload8(fn, f + (size_t)nn * d.SP + oc); gload4(tcv, d.tclass + (nn - 1)); gload4(bsv, d.brk_slot + (nn - 1));
            }
            double best = -INFINITY; int bi = 0;
            if (tc == 0 && bs < 0) {
                const double *Vc = V + cur * SV + i0;
                vit_pipe<0, QMAX / 2, (QMAX / 2 < 6 ? QMAX / 2 : 6)>::run(Treg, lds_addr(Vc), best, bi, i0);
            } else {
                const double *pd = nullptr;
                if (tc >= 0 && bs >= 0) {
                    const double *pdg = d.pd_lt + ((size_t)r * d.NBE + bs) * M * D;
                    for (int i = t; i < M * D; i += NT) pdl[i] = pdg[i];
                    __syncthreads();
                    pd = pdl;
                }
                if (act) for (int rr = 0; rr < QPT; rr++) {
                    const int i = i0 + rr;
                    if (i < S) {
                        const double T = (tc < 0) ? 0. : trans_value(d, tn, i, o, pd);
                        const double v = V[cur * SV + i] + T;
                        if (v > best) { best = v; bi = i; }
                    }
                }
            }
            for (int off = 1; off < P; off <<= 1) {
                const double ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            // retires the requests made at the top of this step and the store of the previous one;
            // this step's store stays in flight over the next step's compares
            gwait_all(fn, tcv, bsv);
            if (act && p == 0) {
                V[nxt * SV + o] = best + fcur;
                gstore2(bp + (size_t)n * S + o, bi);
            }
            __syncthreads();
        }
    }
    if (t < S) final_all[(size_t)blockIdx.x * S + t] = V[((d.N - 1) & 1) * SV + t];
}
// ---- round 5: the lattice as the REFERENCE runs it -- maxima forward, arg-maxima in the trace-back ------------------------------------
// max_product (bpmodel.pyx:1314-1331) keeps no back-pointers: its induction is viterbi_lattice[n, j] = _max(lattice[n-1, :] +
// log_transmat[n-1, :, j]) + framelogprob[n, j], and the trace-back RECOMPUTES lattice[n, i] + log_transmat[n, i, state[n+1]] and takes its
// first maximum.  k_viterbi_reg above records the arg-maximum of every (n, j) instead -- an add, a compare and three selects per state pair,
// 2 500 cycles of vector ALU per step on the fullest SIMD, and nearly all of those arg-maxima are never looked at.  k_viterbi_max runs the
// reference's induction literally: an add and a v_max_f64 per pair (the maximum of the same values is the same value whatever the order it
// is taken in), the merge of a column's P partial maxima is a maximum too (no tie rule to carry), and the lattice ROW goes to memory
// (8 bytes per state instead of a 2-byte pointer: 66 MB per restart at 50 000 x 165).  k_backtrace_max then does what the reference's
// trace-back does, one arg-maximum over S values per segment, first maximum wins.  Same expressions, same values, same tie rule: bit-exact.
// (the LDS reads as ds_read_b128: 4 LDS cycles per wave instruction, 256 B/clk, against 8 cycles and 128 B/clk for the ds_read2_b64 of
// k_viterbi_reg -- MI355X_MICROARCH.md, LDS table -- and a step is 11 waves x 21 of them)
template <int K> __device__ __forceinline__ void vitm_rd(vit_d2 &dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(16 * K) : "memory");
}
template <int G, int NG, int DEPTH> struct vitm_pipe {
    template <int QMAX>
    static __device__ __forceinline__ void step(vit_d2 (&buf)[DEPTH], const double (&T)[QMAX], unsigned addr, double &b0, double &b1) {
        constexpr int younger = (NG - 1 - G) < (DEPTH - 1) ? (NG - 1 - G) : (DEPTH - 1);
        vit_wait<younger>(buf[G % DEPTH]);
        const vit_d2 x = buf[G % DEPTH];
        if constexpr (G + DEPTH < NG) vitm_rd<G + DEPTH>(buf[G % DEPTH], addr);
        b0 = fmax(b0, x.x + T[2 * G]);                   // (two running maxima: independent chains; a maximum does not depend on the order it is taken in)
        b1 = fmax(b1, x.y + T[2 * G + 1]);
        if constexpr (G + 1 < NG) vitm_pipe<G + 1, NG, DEPTH>::step(buf, T, addr, b0, b1);
    }
    template <int QMAX>
    static __device__ __forceinline__ void run(const double (&T)[QMAX], unsigned addr, double &best) {
        static_assert(G == 0 && 2 * NG == QMAX && DEPTH <= NG, "pipeline shape");
        vit_d2 buf[DEPTH];
        fill<0>(buf, addr);
        double b0 = -INFINITY, b1 = -INFINITY;
        step(buf, T, addr, b0, b1);
        best = fmax(b0, b1);
    }
    template <int I> static __device__ __forceinline__ void fill(vit_d2 (&buf)[DEPTH], unsigned addr) {
        vitm_rd<I>(buf[I], addr);
        if constexpr (I + 1 < DEPTH) fill<I + 1>(buf, addr);
    }
};
// Vector-memory traffic of the step loop, kept to ONE request and ONE store per wave and step, both by the writer lanes only (p == 0: 16 of a
// wave's 64).  The first version had every thread request its column's framelogprob value and the adjacency's class and breakend slot each
// step -- 33 wave-level loads per step and workgroup -- and the step took 1 500 cycles longer than without them, whether they were issued
// before or after the step's LDS reads (tools/vit_stamps.py, profiles/r05_decode.txt): waves queue for the CU's one address unit.  Now
//   * the steps that are NOT plain class-0 adjacencies (breakends, telomeres, other classes: 4 % of them) come as a sorted list the host made
//     once per dataset (d.tclass / d.brk_slot do not depend on the restart), held in LDS: a step compares its adjacency with the next entry;
//   * a writer lane's framelogprob value is requested VM_K steps ahead into a ring of registers through loads the compiler does not track,
//     and retired by a COUNTED s_waitcnt: memory operations of a wave complete in order, so when step n waits, its own request must have
//     landed while the younger ones -- the requests of steps n + 1 .. n + VM_K - 1 and the row stores of the last VM_K steps -- stay in flight.
#define VM_K 4
template <int CNT> __device__ __forceinline__ void vm_wait(double &a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(CNT) : "memory"); }
// maximum over the P lanes (a power of two, aligned) that hold the partial maxima of one target state; every lane of the group gets it.  The
// lanes are neighbours (thread = o * P + p): quad permutes and row mirrors, single-instruction DPP moves -- __shfl_xor compiles to ds_bpermute,
// an LDS round trip per step and 32-bit half (tools/micro/vit_loop_bench.hip: merge + row write 960 of a step's 2 100 cycles)
__device__ __forceinline__ double group_max_p(double v, int P) {
    if (P >= 2) v = fmax(v, dpp_mov_f64<0xB1>(v));     // quad_perm [1,0,3,2]
    if (P >= 4) v = fmax(v, dpp_mov_f64<0x4E>(v));     // quad_perm [2,3,0,1]
    if (P >= 8) v = fmax(v, dpp_mov_f64<0x141>(v));    // row_half_mirror (on quad-uniform values: xor 4)
    if (P >= 16) v = fmax(v, dpp_mov_f64<0x140>(v));   // row_mirror (on half-row-uniform values: xor 8)
    if (P >= 32) v = fmax(v, __shfl_xor(v, 16, 64));
    if (P >= 64) v = fmax(v, __shfl_xor(v, 32, 64));
    return v;
}
// grid (nr), block 64 * ceil(S * P / 64); lattice rows to vrow_all [nr][N][SR] (SR = S rounded up to 4, pads 0); special [nspecial]: the
// adjacencies (ascending) that are not plain class-0 ones; dynamic LDS: 2 SV + M D doubles, then nspecial ints
template <int QMAX>
__global__ __launch_bounds__(768) void k_viterbi_max(Dev d, int r0, int P, int SR, double *vrow_all, const int32_t *special, int nspecial, int be_tables, int ca0, int cb0,
                                                      unsigned long long *dbg) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    const int QPT = (((S + P - 1) / P) + 1) & ~1;      // source states per thread, even: 16-byte aligned ds_read_b128
    const int SV = P * QPT + QMAX;         // padded row: reads past S see 0 and are paired with T = -inf
    double *V = (double *)smem_raw;        // [2][SV]
    double *pdl = V + 2 * SV;              // [M*D]
    int *spl = (int *)(pdl + ((M * D + 1) & ~1));      // [nspecial]
    // tables of a breakend step inside transition class 0 (be_tables; classes ca0 -> cb0), so that the step's S x S transition values
    // (bpmodel.pyx:659-668, accumulated in the reference's order: trans_value) are formed from LDS instead of 42 chains of dependent global
    // loads per thread: the states' totals, one byte per clone, and the allele-flip term, row = target state
    unsigned *totA = (unsigned *)(spl + nspecial), *totB = totA + S;      // [S] each
    int8_t *abl = (int8_t *)(totB + S);                                    // [S][SR]
    double *vrow = vrow_all + (size_t)blockIdx.x * d.N * SR;
    const int o = t / P, p = t % P;
    const bool act = o < S;
    const int oc = act ? o : S - 1;
    const int i0 = p * QPT;
    const double *f = d.f + rs_off(d, r, 0);
    double Treg[QMAX];
#pragma unroll
    for (int rr = 0; rr < QMAX; rr++) {
        const int i = i0 + rr;
        Treg[rr] = (act && rr < QPT && i < S && d.TC > 0) ? d.Tval[(size_t)i * S + o] : -INFINITY;
    }
#pragma unroll
    for (int rr = 0; rr < QMAX; rr++) asm volatile("" : "+v"(Treg[rr]));      // their loads retire here, not at a vmcnt(0) inside the step loop
    for (int i = t; i < 2 * SV; i += NT) V[i] = 0.;
    for (int i = t; i < nspecial; i += NT) spl[i] = special[i];
    if (be_tables) {
        for (int i = t; i < S; i += NT) {
            unsigned pa = 0, pb = 0;
            for (int c = 0; c < M; c++) { pa |= ((unsigned)d.tot[((size_t)ca0 * S + i) * M + c] & 0xffu) << (8 * c); pb |= ((unsigned)d.tot[((size_t)cb0 * S + i) * M + c] & 0xffu) << (8 * c); }
            totA[i] = pa; totB[i] = pb;
        }
        for (int k = t; k < S * SR; k += NT) { const int oo = k / SR, ii = k - oo * SR; abl[k] = ii < S ? d.ab[(size_t)oo * S + ii] : (int8_t)0; }
    }
    __syncthreads();
    if (t < SR) { const double v0 = t < S ? f[t] : 0.; if (t < S) V[t] = v0; vrow[t] = v0; }
    __syncthreads();
    if (d.N > 1) {
        const bool writer = act && p == 0;
        // does this WAVE issue a request and a row store per step?  (both sit under `if (writer)`: a wave none of whose lanes writes skips the
        // instructions, and has nothing to wait for)
        const bool wave_stores = __builtin_amdgcn_readfirstlane((int)(__ballot(writer) != 0ull)) != 0;
        double fnr[VM_K] = {0., 0., 0., 0.};
#define VM_ISSUE(j_, n_) { if (writer) { const int nn_ = (n_) < d.N ? (n_) : d.N - 1; gload8(fnr[j_], f + (size_t)nn_ * d.SP + o); } }
#pragma unroll
        for (int j = 0; j < VM_K; j++) VM_ISSUE(j, 1 + j)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(fnr[0]), "+v"(fnr[1]), "+v"(fnr[2]), "+v"(fnr[3]) :: "memory");
        static_assert(VM_K == 4, "the ring is written out for four slots");
        int sp_k = 0;
        int next_sp = nspecial > 0 ? __builtin_amdgcn_readfirstlane(spl[0]) : 0x7fffffff;      // the next adjacency that is not a plain class-0 one
#ifdef RMX_VIT_STAMPS
        // diagnostic build (tools/vit_stamps.py): cycles a wave spends per step waiting for its ring slot (0), in the LDS reads + add / max of a
        // plain step (1), in the merge of the P partial maxima (2), from there to its arrival at the barrier (3) and in the barrier (4)
        unsigned long long vst_acc[5] = {0, 0, 0, 0, 0}, vst_last;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(vst_last) :: "memory");
#define VST(i_) { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); vst_acc[i_] += t_ - vst_last; vst_last = t_; }
#else
#define VST(i_)
#endif
        // one step with ring slot J_: behind a writer wave's own request lie VM_K row stores and VM_K - 1 requests
#define VM_STEP(J_)                                                                                                                \
        {                                                                                                                          \
            if (wave_stores) vm_wait<VM_K + (VM_K - 1)>(fnr[J_]);                                                                   \
            VST(0)                                                                                                                 \
            const int cur = (n - 1) & 1, nxt = n & 1, tn = n - 1;                                                                  \
            const double fcur = fnr[J_];                                                                                           \
            VM_ISSUE(J_, n + VM_K)                                                                                                 \
            double best = -INFINITY;                                                                                               \
            if (tn != next_sp) {                                                                                                   \
                const double *Vc = V + cur * SV + i0;                                                                              \
                vitm_pipe<0, QMAX / 2, (QMAX / 2 < 6 ? QMAX / 2 : 6)>::run(Treg, lds_addr(Vc), best);                              \
            } else {                                                                                                               \
                sp_k++;                                                                                                            \
                next_sp = sp_k < nspecial ? __builtin_amdgcn_readfirstlane(spl[sp_k]) : 0x7fffffff;                                \
                const int tc = d.tclass[tn], bs = d.brk_slot[tn];      /* (tracked loads: the compiler's own wait drains the ring too -- harmless, 4 % of the steps) */ \
                const double *pd = nullptr;                                                                                        \
                if (tc >= 0 && bs >= 0) {                                                                                          \
                    const double *pdg = d.pd_lt + ((size_t)r * d.NBE + bs) * M * D;                                                \
                    for (int i = t; i < M * D; i += NT) pdl[i] = pdg[i];                                                           \
                    __syncthreads();                                                                                               \
                    pd = pdl;                                                                                                      \
                }                                                                                                                  \
                if (be_tables && tc == 0 && pd != nullptr) {                                                                       \
                    /* trans_value(d, tn, i, o, pd) from the LDS tables: the same products and sums in the same order */              \
                    if (act) {                                                                                                     \
                        const unsigned tb_ = totB[o];                                                                              \
                        const int8_t *abr_ = abl + (size_t)o * SR;                                                                 \
                        const int off_ = d.cn_max + 1;                                                                             \
                        for (int rr = 0; rr < QPT; rr++) {                                                                         \
                            const int i = i0 + rr;                                                                                 \
                            if (i < S) {                                                                                           \
                                const unsigned ta_ = totA[i];                                                                      \
                                double T = 0.;                                                                                     \
                                for (int c = 0; c < M; c++) {                                                                      \
                                    const int dd = (int)(int8_t)((ta_ >> (8 * c)) & 0xffu) - (int)(int8_t)((tb_ >> (8 * c)) & 0xffu); \
                                    T += -d.pen * pd[c * D + dd + off_];                                                           \
                                }                                                                                                  \
                                T += -d.pen * (double)abr_[i];                                                                     \
                                best = fmax(best, V[cur * SV + i] + T);                                                            \
                            }                                                                                                      \
                        }                                                                                                          \
                    }                                                                                                              \
                } else if (act) for (int rr = 0; rr < QPT; rr++) {                                                                 \
                    const int i = i0 + rr;                                                                                         \
                    if (i < S) {                                                                                                   \
                        const double T = (tc < 0) ? 0. : trans_value(d, tn, i, o, pd);                                             \
                        best = fmax(best, V[cur * SV + i] + T);                                                                    \
                    }                                                                                                              \
                }                                                                                                                  \
            }                                                                                                                      \
            VST(1)                                                                                                                 \
            best = group_max_p(best, P);                                                                                           \
            VST(2)                                                                                                                 \
            if (writer) {                                                                                                          \
                const double vn = best + fcur;                                                                                     \
                V[nxt * SV + o] = vn;                                                                                              \
                gstore8(vrow + (size_t)n * SR + o, vn);                                                                            \
            }                                                                                                                      \
            VST(3)                                                                                                                 \
            __syncthreads();                                                                                                       \
            VST(4)                                                                                                                 \
            n++;                                                                                                                   \
        }
        int n = 1;
        while (n < d.N) {
            VM_STEP(0)
            if (n < d.N) VM_STEP(1)
            if (n < d.N) VM_STEP(2)
            if (n < d.N) VM_STEP(3)
        }
#undef VM_STEP
#undef VM_ISSUE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's last requests: nobody reads them, the registers must not be re-used under them)
#ifdef RMX_VIT_STAMPS
        if (dbg && blockIdx.x == 0 && (t & 63) == 0) {
            const int w_ = t >> 6, nw_ = NT >> 6;
            const int slot = w_ == 0 ? 0 : (w_ == nw_ / 2 ? 1 : (w_ == nw_ - 1 ? 2 : -1));
            if (slot >= 0) for (int i = 0; i < 5; i++) dbg[8 + slot * 5 + i] = vst_acc[i];
            if (w_ == 0) dbg[4] = d.N - 1;
        }
#endif
#undef VST
    }
}
// Trace-back of k_viterbi_max's lattice, one workgroup of 256 threads per restart (bpmodel.pyx:1320-1331): the final row's first maximum,
// then per segment n = N-2 .. 0 the first maximum over i of lattice[n, i] + log_transmat[n, i, state[n+1]].  Rows of the lattice, the
// adjacency classes and the breakend slots are staged through LDS a chunk at a time by the whole workgroup; the chain itself -- 50 000
// dependent steps -- runs on wave 0, lane l owning source states 256 g + 4 l .. + 3, and is kept short:
//   * a plain class-0 adjacency takes its transition values from the 8-bit code table of the class (row = target state, a 32-bit LDS read
//     whose address is the only thing that depends on the previous step); the value is -pen * code where the host has verified that form
//     bit for bit (mulpen != 0: one conversion and one multiplication), else a second LDS lookup in valtab -- the SAME doubles as d.Tval;
//   * the lattice row of step n - 1 is read while step n's reduction runs (its address does not depend on the state);
//   * the arg-maximum over the wave: the maximum by DPP row rotations and four v_readlane, then the LOWEST lane holding it (ballot + find-first)
//     -- lanes ascend with the state index and a lane keeps the first of its own maxima, so this is the reference's first maximum.
// Everything else (telomeres: log_transmat == 0; breakend and other-class adjacencies: trans_value) takes the plain expression.
// dynamic LDS: ROWS * SR doubles, 256 doubles, 2 * ROWS ints, S * SR code bytes (0 if codeT == nullptr: every step through trans_value)
__device__ __forceinline__ double wave_max_f64(double v) {      // maximum over the 64 lanes (wave-uniform result); any finite / -inf values
    v = fmax(v, dpp_mov_f64<0x121>(v));   // row_ror:1
    v = fmax(v, dpp_mov_f64<0x122>(v));   // row_ror:2
    v = fmax(v, dpp_mov_f64<0x124>(v));   // row_ror:4
    v = fmax(v, dpp_mov_f64<0x128>(v));   // row_ror:8  -> every lane holds its row's maximum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return fmax(fmax(r0, r1), fmax(r2, r3));
}
// MUL: transition value = mulpen * code (verified by the host), else valtab[code]; TWO: a second group of 256 states (256 < S <= 512)
template <bool MUL, bool TWO>
__global__ __launch_bounds__(256) void k_backtrace_max(Dev d, int r0, int SR, const double *vrow_all, const uint8_t *codeT, const double *valtab, double mulpen,
                                                        int64_t *path_all, double *logprob_all, int ROWS, int be_tables, int ca0, int cb0) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int cur_state;
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    double *Vc = (double *)smem_raw;                  // [ROWS][SR]  (pads: -inf)
    double *val = Vc + (size_t)ROWS * SR;             // [256]
    int *tcl = (int *)(val + 256);                    // [ROWS]  1 = plain class-0 adjacency (the coded fast path), 0 = anything else
    uint8_t *cl = (uint8_t *)(tcl + ROWS);            // [S][SR] codes (coded), then the breakend-step tables (be_tables): pd [M*D], totals [2][S], allele-flip bytes [S][SR]
    const double *vrow = vrow_all + (size_t)blockIdx.x * d.N * SR;
    int64_t *path = path_all + (size_t)blockIdx.x * d.N;
    const bool coded = codeT != nullptr;
    double *pdl = (double *)(cl + (coded ? (((size_t)S * SR + 7) & ~(size_t)7) : 0));
    unsigned *totA = (unsigned *)(pdl + ((M * D + 1) & ~1)), *totB = totA + S;
    int8_t *abl = (int8_t *)(totB + S);
    if (be_tables) {
        for (int i = t; i < S; i += NT) {
            unsigned pa = 0, pb = 0;
            for (int c = 0; c < M; c++) { pa |= ((unsigned)d.tot[((size_t)ca0 * S + i) * M + c] & 0xffu) << (8 * c); pb |= ((unsigned)d.tot[((size_t)cb0 * S + i) * M + c] & 0xffu) << (8 * c); }
            totA[i] = pa; totB[i] = pb;
        }
        for (int k = t; k < S * SR; k += NT) { const int oo = k / SR, ii = k - oo * SR; abl[k] = ii < S ? d.ab[(size_t)oo * S + ii] : (int8_t)0; }
    }
    if (coded) {
        for (int k = t; k < S * SR; k += NT) { const int oo = k / SR, ii = k - oo * SR; cl[k] = ii < S ? codeT[(size_t)oo * S + ii] : (uint8_t)0; }      // (pad codes: any finite value; the lattice pads are -inf)
        for (int i = t; i < 256; i += NT) val[i] = i < 255 ? valtab[i] : 0.;
    }
    const int lane = t & 63;
    const bool in0 = 4 * lane < S, in1 = TWO && 256 + 4 * lane < S;
    // first maximum of the candidates (a: states 4 lane .. + 3, b: 256 + 4 lane .. + 3): the state index, wave-uniform; mx = the maximum
#define BT_FIRST_MAX(a0_, a1_, a2_, a3_, b0_, b1_, b2_, b3_, mx_, out_)                                                            \
    {                                                                                                                              \
        double best_ = a0_; int bi_ = 4 * lane;                                                                                    \
        if (a1_ > best_) { best_ = a1_; bi_ = 4 * lane + 1; }                                                                      \
        if (a2_ > best_) { best_ = a2_; bi_ = 4 * lane + 2; }                                                                      \
        if (a3_ > best_) { best_ = a3_; bi_ = 4 * lane + 3; }                                                                      \
        double m_ = wave_max_f64(best_);                                                                                           \
        double best2_ = -INFINITY; int bi2_ = 0;                                                                                   \
        if (TWO) {                                                                                                                 \
            best2_ = b0_; bi2_ = 256 + 4 * lane;                                                                                   \
            if (b1_ > best2_) { best2_ = b1_; bi2_ = 256 + 4 * lane + 1; }                                                         \
            if (b2_ > best2_) { best2_ = b2_; bi2_ = 256 + 4 * lane + 2; }                                                         \
            if (b3_ > best2_) { best2_ = b3_; bi2_ = 256 + 4 * lane + 3; }                                                         \
            m_ = fmax(m_, wave_max_f64(best2_));                                                                                   \
        }                                                                                                                          \
        mx_ = m_;                                                                                                                  \
        /* the lowest state holding the maximum: the first group's lanes (states 0 .. 255 in lane order), then the second group's */ \
        const unsigned long long k1_ = __ballot(best_ == m_);                                                                      \
        if (!TWO || k1_) out_ = __builtin_amdgcn_readlane(bi_, (int)__builtin_ctzll(k1_));                                         \
        else { const unsigned long long k2_ = __ballot(best2_ == m_); out_ = __builtin_amdgcn_readlane(bi2_, (int)__builtin_ctzll(k2_)); } \
    }
    // the last row: np.argmax(viterbi_lattice[-1, :])
    if (t < 64) {
        double a[4], b2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            a[k] = (4 * lane + k < S) ? vrow[(size_t)(d.N - 1) * SR + 4 * lane + k] : -INFINITY;
            b2[k] = (TWO && 256 + 4 * lane + k < S) ? vrow[(size_t)(d.N - 1) * SR + 256 + 4 * lane + k] : -INFINITY;
        }
        double mx; int bi;
        BT_FIRST_MAX(a[0], a[1], a[2], a[3], b2[0], b2[1], b2[2], b2[3], mx, bi)
        if (t == 0) { cur_state = bi; path[d.N - 1] = bi; logprob_all[blockIdx.x] = mx; }
    }
    __syncthreads();
    const double NEG = -INFINITY;
    for (int hi = d.N - 2; hi >= 0; hi -= ROWS) {
        const int lo = hi - ROWS + 1 > 0 ? hi - ROWS + 1 : 0;   // lattice rows lo .. hi, adjacencies lo .. hi
        const int nrow = hi - lo + 1;
        for (int i = t; i < nrow * SR; i += NT) { const int col = i % SR; Vc[i] = col < S ? vrow[(size_t)lo * SR + i] : NEG; }
        for (int i = t; i < nrow; i += NT) tcl[i] = (coded && d.tclass[lo + i] == 0 && d.brk_slot[lo + i] < 0) ? 1 : 0;
        __syncthreads();
        if (t < 64) {
            int s = cur_state;
            // the lattice row and the kind of the step are read a step ahead (neither depends on the state)
            double2 va0 = make_double2(NEG, NEG), vb0 = va0, va1 = va0, vb1 = va0;
            int kind;
#define BT_LOAD_ROW(n_)                                                                                                            \
            {                                                                                                                      \
                const double *Vn_ = Vc + (size_t)((n_) - lo) * SR + 4 * lane;                                                      \
                if (in0) { va0 = *reinterpret_cast<const double2 *>(Vn_); vb0 = *reinterpret_cast<const double2 *>(Vn_ + 2); }      \
                if (in1) { va1 = *reinterpret_cast<const double2 *>(Vn_ + 256); vb1 = *reinterpret_cast<const double2 *>(Vn_ + 258); } \
                kind = tcl[(n_) - lo];                                                                                             \
            }
            BT_LOAD_ROW(hi)
            for (int n = hi; n >= lo; n--) {
                double a0 = va0.x, a1 = va0.y, a2 = vb0.x, a3 = vb0.y, b0 = va1.x, b1 = va1.y, b2 = vb1.x, b3 = vb1.y;
                const int knd = __builtin_amdgcn_readfirstlane(kind);
                if (n > lo) BT_LOAD_ROW(n - 1)
                if (knd) {
                    // plain class-0 adjacency: lattice + T(i, s), T from the code row of target state s (pads: -inf + finite)
                    const unsigned c0 = in0 ? *reinterpret_cast<const unsigned *>(cl + (size_t)s * SR + 4 * lane) : 0u;
                    if (MUL) {      // (mulpen * code is exact -- verified by the host -- so one fused multiply-add rounds like the reference's single addition)
                        a0 = fma(mulpen, (double)(c0 & 255u), a0); a1 = fma(mulpen, (double)((c0 >> 8) & 255u), a1);
                        a2 = fma(mulpen, (double)((c0 >> 16) & 255u), a2); a3 = fma(mulpen, (double)(c0 >> 24), a3);
                    } else { a0 += val[c0 & 255u]; a1 += val[(c0 >> 8) & 255u]; a2 += val[(c0 >> 16) & 255u]; a3 += val[c0 >> 24]; }
                    if (TWO) {
                        const unsigned c1 = in1 ? *reinterpret_cast<const unsigned *>(cl + (size_t)s * SR + 256 + 4 * lane) : 0u;
                        if (MUL) {
                            b0 = fma(mulpen, (double)(c1 & 255u), b0); b1 = fma(mulpen, (double)((c1 >> 8) & 255u), b1);
                            b2 = fma(mulpen, (double)((c1 >> 16) & 255u), b2); b3 = fma(mulpen, (double)(c1 >> 24), b3);
                        } else { b0 += val[c1 & 255u]; b1 += val[(c1 >> 8) & 255u]; b2 += val[(c1 >> 16) & 255u]; b3 += val[c1 >> 24]; }
                    }
                } else {
                    // telomere (log_transmat == 0), breakend or other-class adjacency: the plain expression
                    const int tc = d.tclass[n], bs = d.brk_slot[n];
                    const double *pd = (tc >= 0 && bs >= 0) ? d.pd_lt + ((size_t)r * d.NBE + bs) * M * D : nullptr;
                    double *vv[8] = {&a0, &a1, &a2, &a3, &b0, &b1, &b2, &b3};
                    if (be_tables && tc == 0 && pd != nullptr) {
                        // a breakend adjacency inside class 0: trans_value's products and sums in its order, the operands from LDS (the breakend's distance
                        // table first: M * D doubles, one round trip)
                        for (int i = lane; i < M * D; i += 64) pdl[i] = pd[i];
                        const unsigned tb_ = totB[s];
                        const int off_ = d.cn_max + 1;
#pragma unroll
                        for (int q = 0; q < (TWO ? 8 : 4); q++) {
                            const int i = (q < 4 ? 0 : 256) + 4 * lane + (q & 3);
                            if (i < S) {
                                const unsigned ta_ = totA[i];
                                double T = 0.;
                                for (int c = 0; c < M; c++) {
                                    const int dd = (int)(int8_t)((ta_ >> (8 * c)) & 0xffu) - (int)(int8_t)((tb_ >> (8 * c)) & 0xffu);
                                    T += -d.pen * pdl[c * D + dd + off_];
                                }
                                T += -d.pen * (double)abl[(size_t)s * SR + i];
                                *vv[q] = *vv[q] + T;
                            }
                        }
                    } else {
#pragma unroll
                    for (int q = 0; q < (TWO ? 8 : 4); q++) {
                        const int i = (q < 4 ? 0 : 256) + 4 * lane + (q & 3);
                        if (i < S) *vv[q] = *vv[q] + (tc < 0 ? 0. : trans_value(d, n, i, s, pd));
                    }
                    }
                }
                double mx;
                BT_FIRST_MAX(a0, a1, a2, a3, b0, b1, b2, b3, mx, s)
                if (lane == 0) path[n] = s;
            }
            if (lane == 0) cur_state = s;
#undef BT_LOAD_ROW
        }
        __syncthreads();
    }
#undef BT_FIRST_MAX
}
// The same for grids whose S x S transition values do not fit the register file (176 < S <= ~380): the
// plain-adjacency table of class 0 has few distinct values (sums of -pen x small integers), so the
// workgroup keeps 8-bit CODES of all S x S values in LDS (126 KB at S = 355), row o = target state padded
// to a multiple of 4, and looks the doubles up in a 256-entry LDS table (entry 255 = -inf pads the tiles).
// Thread (o, p) walks its QPT codes a 32-bit word at a time.  Same values, comparison order and tie rule.
__global__ __launch_bounds__(768) void k_viterbi_code(Dev d, int r0, int P, int QPT /* multiple of 4 */, const uint8_t *codeT /* [S][S]: (o, i) */,
                                                      const double *valtab /* [256] */, uint16_t *bp_all, double *final_all) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    const int SV = P * QPT + 4;
    const int SC = P * QPT;                // code row stride (bytes), multiple of 4
    double *V = (double *)smem_raw;        // [2][SV]
    double *pdl = V + 2 * SV;              // [M*D]
    double *val = pdl + ((M * D + 1) & ~1);   // [256]
    uint8_t *cl = (uint8_t *)(val + 256);  // [S][SC]
    uint16_t *bp = bp_all + (size_t)blockIdx.x * d.N * S;
    const int o = t / P, p = t % P;
    const bool act = o < S;
    const int oc = act ? o : S - 1;
    const int i0 = p * QPT;
    const double *f = d.f + rs_off(d, r, 0);
    for (int k = t; k < S * SC; k += NT) { const int oo = k / SC, ii = k - oo * SC; cl[k] = (ii < S && d.TC > 0) ? codeT[(size_t)oo * S + ii] : (uint8_t)255; }
    for (int i = t; i < 2 * SV; i += NT) V[i] = 0.;
    for (int i = t; i < 256; i += NT) val[i] = valtab[i];
    __syncthreads();
    if (t < S) V[t] = f[t];
    __syncthreads();
    const unsigned *cw = (const unsigned *)(cl + (size_t)oc * SC + i0);
    const int NW = QPT / 4;
    if (d.N > 1) {
        double fn; int tcv, bsv;
        gload8(fn, f + (size_t)1 * d.SP + oc); gload4(tcv, d.tclass); gload4(bsv, d.brk_slot);
        gwait_all(fn, tcv, bsv);
        for (int n = 1; n < d.N; n++) {
            const int cur = (n - 1) & 1, nxt = n & 1, tn = n - 1;
            const int tc = __builtin_amdgcn_readfirstlane(tcv), bs = __builtin_amdgcn_readfirstlane(bsv);
            const double fcur = fn;
            {
                const int nn = n + 1 < d.N ? n + 1 : n;
                gload8(fn, f + (size_t)nn * d.SP + oc); gload4(tcv, d.tclass + (nn - 1)); gload4(bsv, d.brk_slot + (nn - 1));
            }
            double best = -INFINITY; int bi = 0;
            if (tc == 0 && bs < 0) {
                if (act) {
                    const double *Vc = V + cur * SV + i0;
#pragma unroll 2
                    for (int w = 0; w < NW; w++) {
                        const unsigned c = cw[w];
                        const double v0 = Vc[4 * w] + val[c & 255u], v1 = Vc[4 * w + 1] + val[(c >> 8) & 255u];
                        const double v2 = Vc[4 * w + 2] + val[(c >> 16) & 255u], v3 = Vc[4 * w + 3] + val[c >> 24];
                        if (v0 > best) { best = v0; bi = i0 + 4 * w; }
                        if (v1 > best) { best = v1; bi = i0 + 4 * w + 1; }
                        if (v2 > best) { best = v2; bi = i0 + 4 * w + 2; }
                        if (v3 > best) { best = v3; bi = i0 + 4 * w + 3; }
                    }
                }
            } else {
                const double *pd = nullptr;
                if (tc >= 0 && bs >= 0) {
                    const double *pdg = d.pd_lt + ((size_t)r * d.NBE + bs) * M * D;
                    for (int i = t; i < M * D; i += NT) pdl[i] = pdg[i];
                    __syncthreads();
                    pd = pdl;
                }
                if (act) for (int rr = 0; rr < QPT; rr++) {
                    const int i = i0 + rr;
                    if (i < S) {
                        const double T = (tc < 0) ? 0. : trans_value(d, tn, i, o, pd);
                        const double v = V[cur * SV + i] + T;
                        if (v > best) { best = v; bi = i; }
                    }
                }
            }
            for (int off = 1; off < P; off <<= 1) {
                const double ob = __shfl_xor(best, off, 64); const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            gwait_all(fn, tcv, bsv);
            if (act && p == 0) {
                V[nxt * SV + o] = best + fcur;
                gstore2(bp + (size_t)n * S + o, bi);
            }
            __syncthreads();
        }
    }
    if (t < S) final_all[(size_t)blockIdx.x * S + t] = V[((d.N - 1) & 1) * SV + t];
}
// k_viterbi_code with the reference's induction (maxima forward, the lattice rows to memory; see k_viterbi_max): 176 < S <= ~380
// MUL: the transition value of a code is mulpen * code (verified by the host bit for bit against the table: then one conversion and one fused
// multiply-add replace the second, bank-conflicted LDS lookup of every pair; pad codes 255 are paired with lattice pads of -inf)
template <bool MUL>
__global__ __launch_bounds__(768) void k_viterbi_code_max(Dev d, int r0, int P, int QPT /* multiple of 4 */, const uint8_t *codeT /* [S][S]: (o, i) */,
                                                          const double *valtab /* [256] */, int SR, double *vrow_all, double mulpen) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    const int SV = P * QPT + 4;
    const int SC = P * QPT;                // code row stride (bytes), multiple of 4
    double *V = (double *)smem_raw;        // [2][SV]
    double *pdl = V + 2 * SV;              // [M*D]
    double *val = pdl + ((M * D + 1) & ~1);   // [256]
    uint8_t *cl = (uint8_t *)(val + 256);  // [S][SC]
    double *vrow = vrow_all + (size_t)blockIdx.x * d.N * SR;
    const int o = t / P, p = t % P;
    const bool act = o < S;
    const int oc = act ? o : S - 1;
    const int i0 = p * QPT;
    const double *f = d.f + rs_off(d, r, 0);
    for (int k = t; k < S * SC; k += NT) { const int oo = k / SC, ii = k - oo * SC; cl[k] = (ii < S && d.TC > 0) ? codeT[(size_t)oo * S + ii] : (uint8_t)255; }
    for (int i = t; i < 2 * SV; i += NT) V[i] = MUL ? -INFINITY : 0.;
    for (int i = t; i < 256; i += NT) val[i] = valtab[i];
    __syncthreads();
    if (t < SR) { const double v0 = t < S ? f[t] : 0.; if (t < S) V[t] = v0; vrow[t] = v0; }
    __syncthreads();
    const unsigned *cw = (const unsigned *)(cl + (size_t)oc * SC + i0);
    const int NW = QPT / 4;
    if (d.N > 1) {
        double fn; int tcv, bsv;
        gload8(fn, f + (size_t)1 * d.SP + oc); gload4(tcv, d.tclass); gload4(bsv, d.brk_slot);
        gwait_all(fn, tcv, bsv);
        for (int n = 1; n < d.N; n++) {
            const int cur = (n - 1) & 1, nxt = n & 1, tn = n - 1;
            const int tc = __builtin_amdgcn_readfirstlane(tcv), bs = __builtin_amdgcn_readfirstlane(bsv);
            const double fcur = fn;
            {
                const int nn = n + 1 < d.N ? n + 1 : n;
                gload8(fn, f + (size_t)nn * d.SP + oc); gload4(tcv, d.tclass + (nn - 1)); gload4(bsv, d.brk_slot + (nn - 1));
            }
            double best = -INFINITY;
            if (tc == 0 && bs < 0) {
                if (act) {
                    const double *Vc = V + cur * SV + i0;
#pragma unroll 2
                    for (int w = 0; w < NW; w++) {
                        const unsigned c = cw[w];
                        double v0, v1, v2, v3;
                        if (MUL) {
                            // (mulpen * code is exact, so the fused multiply-add rounds once, like the reference's addition of the tabulated value; the
                            //  code becomes a double through the 2^52 mantissa form: one full-rate subtraction instead of a quarter-rate conversion; a pad
                            //  code stands next to a lattice pad of -inf)
                            const double M52 = 4503599627370496.0;
                            const double x0 = __hiloint2double(0x43300000, (int)(c & 255u)) - M52, x1 = __hiloint2double(0x43300000, (int)((c >> 8) & 255u)) - M52;
                            const double x2 = __hiloint2double(0x43300000, (int)((c >> 16) & 255u)) - M52, x3 = __hiloint2double(0x43300000, (int)(c >> 24)) - M52;
                            v0 = fma(mulpen, x0, Vc[4 * w]); v1 = fma(mulpen, x1, Vc[4 * w + 1]); v2 = fma(mulpen, x2, Vc[4 * w + 2]); v3 = fma(mulpen, x3, Vc[4 * w + 3]);
                        } else {
                            v0 = Vc[4 * w] + val[c & 255u]; v1 = Vc[4 * w + 1] + val[(c >> 8) & 255u];
                            v2 = Vc[4 * w + 2] + val[(c >> 16) & 255u]; v3 = Vc[4 * w + 3] + val[c >> 24];
                        }
                        best = fmax(fmax(best, v0), fmax(v1, fmax(v2, v3)));
                    }
                }
            } else {
                const double *pd = nullptr;
                if (tc >= 0 && bs >= 0) {
                    const double *pdg = d.pd_lt + ((size_t)r * d.NBE + bs) * M * D;
                    for (int i = t; i < M * D; i += NT) pdl[i] = pdg[i];
                    __syncthreads();
                    pd = pdl;
                }
                if (act) for (int rr = 0; rr < QPT; rr++) {
                    const int i = i0 + rr;
                    if (i < S) {
                        const double T = (tc < 0) ? 0. : trans_value(d, tn, i, o, pd);
                        best = fmax(best, V[cur * SV + i] + T);
                    }
                }
            }
            for (int off = 1; off < P; off <<= 1) best = fmax(best, __shfl_xor(best, off, 64));
            gwait_all(fn, tcv, bsv);
            if (act && p == 0) {
                const double vn = best + fcur;
                V[nxt * SV + o] = vn;
                gstore8(vrow + (size_t)n * SR + o, vn);
            }
            __syncthreads();
        }
    }
}
// ---- grids whose codes do not fit the LDS either (380 < S <= 1 024 states; round 5) ----------------------------------------------------
// The transition value of a plain class-0 adjacency is -pen * k with k = min(SAD(cn_q, cn_o), SAD(cn_q, swap_alleles(cn_o))) over the
// byte-packed allele copies of the tumour clones (k_fbk's closed form; the host has verified it against log_transmat's table bit for bit:
// fbk_ok), so nothing S x S is kept: thread (o, p) holds its target state's packed copies in registers, the source states' copies and the
// lattice row are wave-uniform LDS reads (128 bits: four states / two values), a pair is two v_sad_u8, a minimum, the 2^52 conversion, one
// fused multiply-add (mulpen * k is exact: one rounding, as the reference's addition of the tabulated value) and v_max_f64.  Induction,
// maxima, rows to memory, special steps as k_viterbi_code_max.  P slices of the source states per target (threads o + p * SO), merged
// through LDS (a maximum: order-free).
// dynamic LDS: 2 * SV doubles, P * SO doubles (P > 1), M * D doubles, (M4 ? 2 : 1) * SV words
// CL: W workgroups per restart (grid (W, restarts)), each owning OW target states; a step ends with the exchange of the new row through memory --
// every element stored and loaded as a relaxed agent-scope atomic (sc1: written through to / read at the level all XCDs share, no L2-wide write-back
// or invalidate).  No flag and no counter: the host fills the rows with all-ones words before the launch, a consumer polls the element itself until
// it is something else (8-byte stores land whole; a NaN result is stored as the canonical quiet NaN).  All W * restarts workgroups must be resident
// together: the host keeps their number below the CU count.
template <bool M4, bool CL>
__global__ __launch_bounds__(1024) void k_viterbi_sad_max(Dev d, int r0, int P, int SO, int OW, int SR, double *vrow_all, const uint32_t *cnpack, const uint32_t *cnpack2,
                                                          double mulpen, int cls0, int Wcl, int nrst, unsigned *fail_flag, int stall_test) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x;
    // CL: a 1-D grid of 8 * W * ceil(restarts / 8) workgroups, W passed in OW's neighbour argument: workgroup L works on restart 8 * (L / (8 W)) + L % 8 as
    // member (L / 8) % W -- the members of a restart have the same L mod 8, which is the XCD a round-robin dispatch gives them (a placement hint:
    // nothing depends on it)
    const int W = CL ? Wcl : 1;
    const int L = (int)blockIdx.x;
    const int wg = CL ? (L >> 3) % W : 0, rb = CL ? 8 * (L / (8 * W)) + (L & 7) : L;
    if (CL && rb >= nrst) return;
    const int r = r0 + rb;
    const int SQ = ((S + 4 * P - 1) / (4 * P)) * 4;   // source states per slice (a multiple of 4); SO: threads per slice (a multiple of 64, >= OW)
    const int SV = P * SQ;
    double *V = (double *)smem_raw;              // [2][SV]  (pads: -inf)
    double *part = V + 2 * SV;                   // [P][SO]
    double *pdl = part + (P > 1 ? P * SO : 0);   // [M*D]
    uint32_t *cnl = (uint32_t *)(pdl + ((M * D + 1) & ~1));   // [SV]
    uint32_t *cnl2 = cnl + SV;                   // [SV] (four clones: the third tumour clone)
    double *vrow = vrow_all + (size_t)rb * d.N * SR;
    const int p = t / SO, ol = t - p * SO;
    const int o = wg * OW + ol;                  // this thread's target state
    const bool act = p < P && ol < OW && o < S;
    const int oc = act ? o : S - 1;
    const int i0 = (p < P ? p : 0) * SQ;
    const double *f = d.f + rs_off(d, r, 0);
    auto swap_alleles = [](uint32_t x) { return ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu); };
    for (int i = t; i < SV; i += NT) { cnl[i] = i < S ? cnpack[(size_t)cls0 * S + i] : 0u; if (M4) cnl2[i] = i < S ? cnpack2[(size_t)cls0 * S + i] : 0u; }
    for (int i = t; i < 2 * SV; i += NT) V[i] = -INFINITY;
    const uint32_t co = cnpack[(size_t)cls0 * S + oc], cos = swap_alleles(co);
    const uint32_t cb = M4 ? cnpack2[(size_t)cls0 * S + oc] : 0u, cbs = swap_alleles(cb);
    __syncthreads();
    for (int i = t; i < SR; i += NT) { const double v0 = i < S ? f[i] : 0.; if (i < S) V[i] = v0; if (wg == 0) vrow[i] = v0; }      // (row 0 is every workgroup's own copy of f[0])
    __syncthreads();
    const int NW = SQ / 4;
    const double M52 = 4503599627370496.0;
    unsigned budget = RMX_CLUSTER_SPINS;      // polls this thread may still spend waiting for partners' rows (CL)
    bool flagged = false;
    if (d.N > 1) {
        double fn; int tcv, bsv;
        gload8(fn, f + (size_t)1 * d.SP + oc); gload4(tcv, d.tclass); gload4(bsv, d.brk_slot);
        gwait_all(fn, tcv, bsv);
        for (int n = 1; n < d.N; n++) {
            const int cur = (n - 1) & 1, nxt = n & 1, tn = n - 1;
            const int tc = __builtin_amdgcn_readfirstlane(tcv), bs = __builtin_amdgcn_readfirstlane(bsv);
            const double fcur = fn;
            {
                const int nn = n + 1 < d.N ? n + 1 : n;
                gload8(fn, f + (size_t)nn * d.SP + oc); gload4(tcv, d.tclass + (nn - 1)); gload4(bsv, d.brk_slot + (nn - 1));
            }
            double best = -INFINITY;
            if (tc == 0 && bs < 0) {
                if (act) {
                    const double *Vc = V + cur * SV + i0;
                    const uint32_t *cq = cnl + i0, *cq2 = cnl2 + i0;
#pragma unroll 2
                    for (int w = 0; w < NW; w++) {
                        const uint4 c4 = *reinterpret_cast<const uint4 *>(cq + 4 * w);
                        uint4 e4 = make_uint4(0u, 0u, 0u, 0u);
                        if (M4) e4 = *reinterpret_cast<const uint4 *>(cq2 + 4 * w);
                        const double2 va = *reinterpret_cast<const double2 *>(Vc + 4 * w), vb = *reinterpret_cast<const double2 *>(Vc + 4 * w + 2);
#define VS_K(c_, e_) min(__builtin_amdgcn_sad_u8(c_, co, M4 ? __builtin_amdgcn_sad_u8(e_, cb, 0u) : 0u), __builtin_amdgcn_sad_u8(c_, cos, M4 ? __builtin_amdgcn_sad_u8(e_, cbs, 0u) : 0u))
                        const unsigned k0 = VS_K(c4.x, e4.x), k1 = VS_K(c4.y, e4.y), k2 = VS_K(c4.z, e4.z), k3 = VS_K(c4.w, e4.w);
#undef VS_K
                        const double x0 = __hiloint2double(0x43300000, (int)k0) - M52, x1 = __hiloint2double(0x43300000, (int)k1) - M52;
                        const double x2 = __hiloint2double(0x43300000, (int)k2) - M52, x3 = __hiloint2double(0x43300000, (int)k3) - M52;
                        const double v0 = fma(mulpen, x0, va.x), v1 = fma(mulpen, x1, va.y), v2 = fma(mulpen, x2, vb.x), v3 = fma(mulpen, x3, vb.y);
                        best = fmax(fmax(best, v0), fmax(v1, fmax(v2, v3)));
                    }
                }
            } else {
                const double *pd = nullptr;
                if (tc >= 0 && bs >= 0) {
                    const double *pdg = d.pd_lt + ((size_t)r * d.NBE + bs) * M * D;
                    for (int i = t; i < M * D; i += NT) pdl[i] = pdg[i];
                    __syncthreads();
                    pd = pdl;
                }
                if (act) for (int rr = 0; rr < SQ; rr++) {
                    const int i = i0 + rr;
                    if (i < S) {
                        const double T = (tc < 0) ? 0. : trans_value(d, tn, i, o, pd);
                        best = fmax(best, V[cur * SV + i] + T);
                    }
                }
            }
            if (P > 1) {
                if (p < P) part[p * SO + ol] = best;
                __syncthreads();
                if (p == 0) for (int q = 1; q < P; q++) best = fmax(best, part[q * SO + ol]);
            }
            gwait_all(fn, tcv, bsv);
            if (!CL) {
                if (act && p == 0) {
                    const double vn = best + fcur;
                    V[nxt * SV + o] = vn;
                    gstore8(vrow + (size_t)n * SR + o, vn);
                }
                __syncthreads();
            } else {
                if (act && p == 0 && !(stall_test && wg == 0 && rb == 0 && n == 1)) {      // (stall_test: a member that never publishes a row -- the watchdog's test)
                    double vn = best + fcur;
                    if (vn != vn) vn = __longlong_as_double(0x7ff8000000000000ll);      // (never the all-ones word the fetch below waits on)
                    __hip_atomic_store(vrow + (size_t)n * SR + o, vn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // every thread fetches its share of the new row as the owners' stores arrive: the host filled the rows with all-ones words, which no stored
                // value is, and an 8-byte store lands whole.  The waits are bounded: a member that is not resident (a device shared with another process's
                // resident kernels) or has gone would otherwise hold its partners for ever.  A thread has RMX_CLUSTER_SPINS polls (seconds) for one wait;
                // when a wait uses them up it raises the launch's flag and from then on takes what it finds (-inf for a missing element) --
                // the workgroup runs to its end without waiting, its partners get rows and are not held up, and the host, seeing the flag, decodes again with one
                // workgroup per restart.
                for (int i = t; i < S; i += NT) {
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(vrow + (size_t)n * SR + i);
                    unsigned long long u;
                    while ((u = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == ~0ull && budget) { __builtin_amdgcn_s_sleep(1); --budget; }
                    if (u == ~0ull) {
                        if (!flagged) { __hip_atomic_store(fail_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); flagged = true; }
                        V[nxt * SV + i] = -INFINITY;
                    } else { V[nxt * SV + i] = __longlong_as_double((long long)u); if (!flagged) budget = RMX_CLUSTER_SPINS; }      // (a wait that ends refills the budget)
                }
                __syncthreads();
            }
        }
    }
}
// Trace-back of k_viterbi_sad_max's lattice (k_backtrace_max's scheme: chunks of lattice rows through LDS, the chain on wave 0, first maximum by
// wave maximum + ballot); lane l owns source states 256 g + 4 l .. + 3 of NG groups, a plain class-0 adjacency forms its transition values from
// the packed copies (one dependent LDS read: the target state's word).
// dynamic LDS: ROWS * SR doubles, ROWS ints, (M4 ? 2 : 1) * SR words
template <int NG, bool M4>
__global__ __launch_bounds__(256) void k_backtrace_sad(Dev d, int r0, int SR, const double *vrow_all, const uint32_t *cnpack, const uint32_t *cnpack2, double mulpen, int cls0,
                                                        int64_t *path_all, double *logprob_all, int ROWS) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int cur_state;
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, r = r0 + blockIdx.x;
    double *Vc = (double *)smem_raw;                  // [ROWS][SR]  (pads: -inf)
    int *tcl = (int *)(Vc + (size_t)ROWS * SR);       // [ROWS]  1 = plain class-0 adjacency
    uint32_t *cnl = (uint32_t *)(tcl + ROWS);         // [SR]
    uint32_t *cnl2 = cnl + SR;                        // [SR] (M4)
    const double *vrow = vrow_all + (size_t)blockIdx.x * d.N * SR;
    int64_t *path = path_all + (size_t)blockIdx.x * d.N;
    const int lane = t & 63;
    const double NEG = -INFINITY, M52 = 4503599627370496.0;
    auto swap_alleles = [](uint32_t x) { return ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu); };
    for (int i = t; i < SR; i += NT) { cnl[i] = i < S ? cnpack[(size_t)cls0 * S + i] : 0u; if (M4) cnl2[i] = i < S ? cnpack2[(size_t)cls0 * S + i] : 0u; }
    // this lane's source states' packed copies never change
    uint32_t cq[NG][4], eq[NG][4];
#pragma unroll
    for (int g = 0; g < NG; g++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = 256 * g + 4 * lane + k;
            cq[g][k] = i < S ? cnpack[(size_t)cls0 * S + i] : 0u; eq[g][k] = (M4 && i < S) ? cnpack2[(size_t)cls0 * S + i] : 0u;
        }
    // first maximum over the candidates a[g][k] (state 256 g + 4 lane + k): state index (wave-uniform) and the maximum
#define BS_FIRST_MAX(a_, mx_, out_)                                                                                                \
    {                                                                                                                              \
        double bg_[NG]; int big_[NG];                                                                                              \
        double m_ = -INFINITY;                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                           \
            double b_ = a_[g][0]; int bi_ = 256 * g + 4 * lane;                                                                    \
            if (a_[g][1] > b_) { b_ = a_[g][1]; bi_ = 256 * g + 4 * lane + 1; }                                                    \
            if (a_[g][2] > b_) { b_ = a_[g][2]; bi_ = 256 * g + 4 * lane + 2; }                                                    \
            if (a_[g][3] > b_) { b_ = a_[g][3]; bi_ = 256 * g + 4 * lane + 3; }                                                    \
            bg_[g] = b_; big_[g] = bi_;                                                                                            \
            m_ = fmax(m_, wave_max_f64(b_));                                                                                       \
        }                                                                                                                          \
        mx_ = m_;                                                                                                                  \
        bool found_ = false;                                                                                                       \
        _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                           \
            const unsigned long long kk_ = __ballot(bg_[g] == m_);                                                                 \
            if (!found_ && kk_) { out_ = __builtin_amdgcn_readlane(big_[g], (int)__builtin_ctzll(kk_)); found_ = true; }           \
        }                                                                                                                          \
    }
    if (t < 64) {
        double a[NG][4];
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int k = 0; k < 4; k++) a[g][k] = (256 * g + 4 * lane + k < S) ? vrow[(size_t)(d.N - 1) * SR + 256 * g + 4 * lane + k] : NEG;
        double mx; int bi = 0;
        BS_FIRST_MAX(a, mx, bi)
        if (t == 0) { cur_state = bi; path[d.N - 1] = bi; logprob_all[blockIdx.x] = mx; }
    }
    __syncthreads();
    for (int hi = d.N - 2; hi >= 0; hi -= ROWS) {
        const int lo = hi - ROWS + 1 > 0 ? hi - ROWS + 1 : 0;   // lattice rows lo .. hi, adjacencies lo .. hi
        const int nrow = hi - lo + 1;
        for (int i = t; i < nrow * SR; i += NT) { const int col = i % SR; Vc[i] = col < S ? vrow[(size_t)lo * SR + i] : NEG; }
        for (int i = t; i < nrow; i += NT) tcl[i] = (d.tclass[lo + i] == 0 && d.brk_slot[lo + i] < 0) ? 1 : 0;
        __syncthreads();
        if (t < 64) {
            int s = cur_state;
            double2 va[NG], vb[NG];
            int kind;
#define BS_LOAD_ROW(n_)                                                                                                            \
            {                                                                                                                      \
                const double *Vn_ = Vc + (size_t)((n_) - lo) * SR + 4 * lane;                                                      \
                _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                                   \
                    if (256 * g + 4 * lane < SR) { va[g] = *reinterpret_cast<const double2 *>(Vn_ + 256 * g); vb[g] = *reinterpret_cast<const double2 *>(Vn_ + 256 * g + 2); } \
                    else { va[g] = make_double2(NEG, NEG); vb[g] = va[g]; }                                                        \
                }                                                                                                                  \
                kind = tcl[(n_) - lo];                                                                                             \
            }
            BS_LOAD_ROW(hi)
            for (int n = hi; n >= lo; n--) {
                double a[NG][4];
#pragma unroll
                for (int g = 0; g < NG; g++) { a[g][0] = va[g].x; a[g][1] = va[g].y; a[g][2] = vb[g].x; a[g][3] = vb[g].y; }
                const int knd = __builtin_amdgcn_readfirstlane(kind);
                if (n > lo) BS_LOAD_ROW(n - 1)
                if (knd) {
                    const uint32_t cs = cnl[s], css = swap_alleles(cs);
                    const uint32_t es = M4 ? cnl2[s] : 0u, ess = swap_alleles(es);
#pragma unroll
                    for (int g = 0; g < NG; g++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const unsigned kk = min(__builtin_amdgcn_sad_u8(cq[g][k], cs, M4 ? __builtin_amdgcn_sad_u8(eq[g][k], es, 0u) : 0u),
                                                    __builtin_amdgcn_sad_u8(cq[g][k], css, M4 ? __builtin_amdgcn_sad_u8(eq[g][k], ess, 0u) : 0u));
                            a[g][k] = fma(mulpen, __hiloint2double(0x43300000, (int)kk) - M52, a[g][k]);
                        }
                } else {
                    // telomere (log_transmat == 0), breakend or other-class adjacency: the plain expression
                    const int tc = d.tclass[n], bs = d.brk_slot[n];
                    const double *pd = (tc >= 0 && bs >= 0) ? d.pd_lt + ((size_t)r * d.NBE + bs) * M * D : nullptr;
#pragma unroll
                    for (int g = 0; g < NG; g++)
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int i = 256 * g + 4 * lane + k;
                            if (i < S) a[g][k] = a[g][k] + (tc < 0 ? 0. : trans_value(d, n, i, s, pd));
                        }
                }
                double mx;
                BS_FIRST_MAX(a, mx, s)
                if (lane == 0) path[n] = s;
            }
            if (lane == 0) cur_state = s;
#undef BS_LOAD_ROW
        }
        __syncthreads();
    }
#undef BS_FIRST_MAX
}
// ---- the trace-back in parallel (round 5, late) ------------------------------------------------------------------------------------
// The reference walks N - 1 dependent steps, each the first maximum over i of lattice[n, i] + log_transmat[n, i, state[n + 1]]
// (bpmodel.pyx:1327-1331); on one wave that is 0.6-1.3 us per step, a third of the decode.  The same arg-maxima for EVERY target state of
// every row are independent of each other: k_bp_all computes them on the whole chip (the forward lattice's pair arithmetic once more --
// -pen * k from the packed copies, one fused multiply-add -- with the first maximum kept: a strictly greater candidate replaces, within a
// group of four the lowest equal one), then the walk itself is a composition of maps: k_chase_compose composes the B maps of a block of
// rows for all S end states at once (LDS lookups, every block of every restart in parallel), k_chase_ends walks the N / B composed maps
// from the final row's first maximum, k_chase_fill walks inside the blocks from their known end states.  Same path, bit for bit: the
// candidates are the very doubles the sequential trace-back forms.
// k_bp_all: grid (row blocks, restarts); thread (o, p): target state o, rows p, p + P, ... of the block (a wave has one p: its source-state
// reads are broadcasts).  dynamic LDS: NB * SR doubles, (M4 ? 2 : 1) * SR words
template <bool M4>
__global__ __launch_bounds__(1024) void k_bp_all(Dev d, int r0, int P, int SO, int NB, int SR, const double *vrow_all, const uint32_t *cnpack, const uint32_t *cnpack2,
                                                 double mulpen, int cls0, uint16_t *bp_all) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int S = d.S, M = d.M, D = d.D, t = threadIdx.x, NT = blockDim.x, rb = blockIdx.y, r = r0 + rb;
    double *Vc = (double *)smem_raw;                       // [NB][SR]: lattice rows n0 - 1 .. (pads: -inf)
    uint32_t *cnl = (uint32_t *)(Vc + (size_t)NB * SR);    // [SR]
    uint32_t *cnl2 = cnl + SR;                             // [SR] (M4)
    const double *vrow = vrow_all + (size_t)rb * d.N * SR;
    uint16_t *bp = bp_all + (size_t)rb * d.N * S;
    const int n0 = 1 + blockIdx.x * NB;                    // target rows n0 .. n0 + nrow - 1
    const int nrow = min(NB, d.N - n0);
    const double NEG = -INFINITY, M52 = 4503599627370496.0;
    auto swap_alleles = [](uint32_t x) { return ((x & 0x00ff00ffu) << 8) | ((x >> 8) & 0x00ff00ffu); };
    for (int i = t; i < nrow * SR; i += NT) { const int col = i % SR; Vc[i] = col < S ? vrow[(size_t)(n0 - 1) * SR + i] : NEG; }
    for (int i = t; i < SR; i += NT) { cnl[i] = i < S ? cnpack[(size_t)cls0 * S + i] : 0u; if (M4) cnl2[i] = i < S ? cnpack2[(size_t)cls0 * S + i] : 0u; }
    const int p = t / SO, o = t - p * SO;
    const bool act = o < S;
    const int oc = act ? o : S - 1;
    const uint32_t co = cnpack[(size_t)cls0 * S + oc], cos = swap_alleles(co);
    const uint32_t cb = M4 ? cnpack2[(size_t)cls0 * S + oc] : 0u, cbs = swap_alleles(cb);
    __syncthreads();
    const int NW = SR / 4;
    for (int rr = p; rr < nrow; rr += P) {                 // (wave-uniform: SO is a multiple of 64)
        const int n = n0 + rr, tn = n - 1;
        const int tc = d.tclass[tn], bs = d.brk_slot[tn];
        if (!act) continue;
        const double *Vr = Vc + (size_t)rr * SR;
        double best = NEG; int bi = 0;
        if (tc == 0 && bs < 0) {
#pragma unroll 2
            for (int w = 0; w < NW; w++) {
                const uint4 c4 = *reinterpret_cast<const uint4 *>(cnl + 4 * w);
                uint4 e4 = make_uint4(0u, 0u, 0u, 0u);
                if (M4) e4 = *reinterpret_cast<const uint4 *>(cnl2 + 4 * w);
                const double2 va = *reinterpret_cast<const double2 *>(Vr + 4 * w), vb = *reinterpret_cast<const double2 *>(Vr + 4 * w + 2);
#define VS_K(c_, e_) min(__builtin_amdgcn_sad_u8(c_, co, M4 ? __builtin_amdgcn_sad_u8(e_, cb, 0u) : 0u), __builtin_amdgcn_sad_u8(c_, cos, M4 ? __builtin_amdgcn_sad_u8(e_, cbs, 0u) : 0u))
                const unsigned k0 = VS_K(c4.x, e4.x), k1 = VS_K(c4.y, e4.y), k2 = VS_K(c4.z, e4.z), k3 = VS_K(c4.w, e4.w);
#undef VS_K
                const double v0 = fma(mulpen, __hiloint2double(0x43300000, (int)k0) - M52, va.x), v1 = fma(mulpen, __hiloint2double(0x43300000, (int)k1) - M52, va.y);
                const double v2 = fma(mulpen, __hiloint2double(0x43300000, (int)k2) - M52, vb.x), v3 = fma(mulpen, __hiloint2double(0x43300000, (int)k3) - M52, vb.y);
                const double m4 = fmax(fmax(v0, v1), fmax(v2, v3));
                if (m4 > best) { best = m4; bi = 4 * w + (v0 == m4 ? 0 : (v1 == m4 ? 1 : (v2 == m4 ? 2 : 3))); }
            }
        } else {
            // telomere (log_transmat == 0), breakend or other-class adjacency: the plain expression
            const double *pd = (tc >= 0 && bs >= 0) ? d.pd_lt + ((size_t)r * d.NBE + bs) * M * D : nullptr;
            for (int i = 0; i < S; i++) {
                const double v = Vr[i] + (tc < 0 ? 0. : trans_value(d, tn, i, o, pd));
                if (v > best) { best = v; bi = i; }
            }
        }
        bp[(size_t)n * S + o] = (uint16_t)bi;
    }
}
// composed maps: block j covers rows j B + 1 .. min((j + 1) B, N - 1); comp[j][o] = the state at row j B on the path that is in state o at the block's last row
__global__ __launch_bounds__(256) void k_chase_compose(int N, int S, int B, const uint16_t *bp_all, uint16_t *comp_all) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint16_t *chunk = (uint16_t *)smem_raw;
    const int j = blockIdx.x, rb = blockIdx.y, NBLK = gridDim.x, t = threadIdx.x, NT = blockDim.x;
    const int lo = j * B, hi = min(lo + B, N - 1);
    const uint16_t *bp = bp_all + (size_t)rb * N * S + (size_t)(lo + 1) * S;
    const int cnt = (hi - lo) * S;
    for (int i = t; i < cnt; i += NT) chunk[i] = bp[i];
    __syncthreads();
    for (int o = t; o < S; o += NT) {
        int s = o;
        for (int n = hi; n > lo; n--) s = chunk[(n - lo - 1) * S + s];
        comp_all[((size_t)rb * NBLK + j) * S + o] = (uint16_t)s;
    }
}
// the final row's first maximum, then the state at every block's last row (ends[j]) through the composed maps; one wave per restart
__global__ __launch_bounds__(64) void k_chase_ends(int N, int S, int SR, int NBLK, const double *vrow_all, const uint16_t *comp_all, int32_t *ends_all, double *logprob_all) {
    const int rb = blockIdx.x, lane = threadIdx.x;
    const double *last = vrow_all + ((size_t)rb * N + (N - 1)) * SR;
    double best = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < S; i += 64) { const double v = last[i]; if (v > best || bi == 0x7fffffff) { best = v; bi = i; } }      // (a lane's states ascend: its first maximum)
    const double m = wave_max_f64(best);
    int cand = (best == m && bi != 0x7fffffff) ? bi : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off, 64));
    if (lane == 0) {
        int s = cand == 0x7fffffff ? 0 : cand;
        logprob_all[rb] = m;
        const uint16_t *comp = comp_all + (size_t)rb * NBLK * S;
        for (int j = NBLK - 1; j >= 0; j--) { ends_all[(size_t)rb * NBLK + j] = s; s = comp[(size_t)j * S + s]; }
    }
}
// the path inside each block from its known end state
__global__ __launch_bounds__(256) void k_chase_fill(int N, int S, int B, const uint16_t *bp_all, const int32_t *ends_all, int64_t *path_all) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint16_t *chunk = (uint16_t *)smem_raw;
    const int j = blockIdx.x, rb = blockIdx.y, NBLK = gridDim.x, t = threadIdx.x, NT = blockDim.x;
    const int lo = j * B, hi = min(lo + B, N - 1);
    const uint16_t *bp = bp_all + (size_t)rb * N * S + (size_t)(lo + 1) * S;
    const int cnt = (hi - lo) * S;
    for (int i = t; i < cnt; i += NT) chunk[i] = bp[i];
    __syncthreads();
    if (t == 0) {
        int64_t *path = path_all + (size_t)rb * N;
        int s = ends_all[(size_t)rb * NBLK + j];
        path[hi] = s;
        for (int n = hi; n > lo; n--) { s = chunk[(n - lo - 1) * S + s]; path[n - 1] = s; }
    }
}
// trace-back: one workgroup per restart; chunks of back-pointer rows staged through LDS
__global__ void k_backtrace(Dev d, const uint16_t *bp_all, const double *final_all, int64_t *path_all, double *logprob_all, int ROWS) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint16_t *chunk = (uint16_t *)smem_raw;
    __shared__ int cur_state;
    const int S = d.S, t = threadIdx.x, NT = blockDim.x;
    const uint16_t *bp = bp_all + (size_t)blockIdx.x * d.N * S;
    const double *final_row = final_all + (size_t)blockIdx.x * S;
    int64_t *path = path_all + (size_t)blockIdx.x * d.N;
    double *logprob = logprob_all + blockIdx.x;
    if (t == 0) {
        int mp = 0; double vm = final_row[0];
        for (int i = 1; i < S; i++) if (final_row[i] > vm) { vm = final_row[i]; mp = i; }
        cur_state = mp; path[d.N - 1] = mp; *logprob = vm;
    }
    __syncthreads();
    for (int hi = d.N - 1; hi >= 1; hi -= ROWS) {
        const int lo = hi - ROWS + 1 > 1 ? hi - ROWS + 1 : 1;   // rows lo..hi of bp
        const int cnt = (hi - lo + 1) * S;
        for (int i = t; i < cnt; i += NT) chunk[i] = bp[(size_t)lo * S + i];
        __syncthreads();
        if (t == 0) {
            int s = cur_state;
            for (int n = hi; n >= lo; n--) { s = chunk[(n - lo) * S + s]; path[n - 1] = s; }
            cur_state = s;
        }
        __syncthreads();
    }
}

// =============================================================================
// module-level sum_product / max_product on caller-supplied dense (f, T):
// literal log-domain evaluation in the reference's operation order.
// One workgroup; thread j owns column j (S <= 1024).
// =============================================================================
__global__ __launch_bounds__(1024) void k_sum_product_dense(const double *f, const double *T, double *alphas, double *betas, int N, int S) {
    const int j = threadIdx.x;
    extern __shared__ double prev[];   // [S]
    if (j < S) { alphas[j] = f[j]; prev[j] = f[j]; }
    __syncthreads();
    for (int n = 1; n < N; n++) {
        double out = 0.;
        if (j < S) {
            const double *Tn = T + (size_t)(n - 1) * S * S;
            double vmax = -INFINITY;
            for (int i = 0; i < S; i++) { const double v = prev[i] + Tn[(size_t)i * S + j]; if (v > vmax) vmax = v; }
            double ps = 0.;
            for (int i = 0; i < S; i++) ps += exp(prev[i] + Tn[(size_t)i * S + j] - vmax);
            out = log(ps) + vmax + f[(size_t)n * S + j];
        }
        __syncthreads();
        if (j < S) { prev[j] = out; alphas[(size_t)n * S + j] = out; }
        __syncthreads();
    }
    if (j < S) { betas[(size_t)(N - 1) * S + j] = 0.0; prev[j] = 0.0; }
    __syncthreads();
    for (int n = N - 2; n >= 0; n--) {
        double out = 0.;
        if (j < S) {
            const double *Tn = T + (size_t)n * S * S + (size_t)j * S;   // row i = j
            const double *fn = f + (size_t)(n + 1) * S;
            double vmax = -INFINITY;
            for (int c = 0; c < S; c++) { const double v = (Tn[c] + fn[c]) + prev[c]; if (v > vmax) vmax = v; }
            double ps = 0.;
            for (int c = 0; c < S; c++) ps += exp(((Tn[c] + fn[c]) + prev[c]) - vmax);
            out = log(ps) + vmax;
        }
        __syncthreads();
        if (j < S) { betas[(size_t)n * S + j] = out; prev[j] = out; }
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_max_product_dense(const double *f, const double *T, uint16_t *bp, double *final_row, int N, int S) {
    const int j = threadIdx.x;
    extern __shared__ double prev[];
    if (j < S) prev[j] = f[j];
    __syncthreads();
    for (int n = 1; n < N; n++) {
        double out = 0.; int bi = 0;
        if (j < S) {
            const double *Tn = T + (size_t)(n - 1) * S * S;
            double best = -INFINITY;
            for (int i = 0; i < S; i++) { const double v = prev[i] + Tn[(size_t)i * S + j]; if (v > best) { best = v; bi = i; } }
            out = best + f[(size_t)n * S + j];
        }
        __syncthreads();
        if (j < S) { prev[j] = out; bp[(size_t)n * S + j] = (uint16_t)bi; }
        __syncthreads();
    }
    if (j < S) final_row[j] = prev[j];
}

// =============================================================================
// dense materialisation on request (tests / attribute read-back only)
// =============================================================================
// which: 0 log_transmat (lt snapshot), 1 cached_log_transmat.  grid (N-1), block 256
__global__ void k_materialize_T(Dev d, int r, int which, int zero_all, double *out) {
    const int n = blockIdx.x, S = d.S;
    const int bs = d.brk_slot[n];
    const double *pd = nullptr;
    if (bs >= 0) pd = (which == 0 ? d.pd_lt : d.pd_cached) + ((size_t)r * d.NBE + bs) * d.M * d.D;
    for (int idx = threadIdx.x; idx < S * S; idx += blockDim.x) {
        const int i = idx / S, j = idx - i * S;
        out[(size_t)n * S * S + idx] = zero_all ? 0. : trans_value(d, n, i, j, pd);
    }
}
// joint posterior marginals (bpmodel.pyx:954-960).  grid (N-1), block 256
__global__ void k_materialize_joint(Dev d, int r, int uniform, double *out) {
    __shared__ double scratch[8];
    __shared__ double zsh;
    const int n = blockIdx.x, S = d.S, t = threadIdx.x;
    double *o = out + (size_t)n * S * S;
    if (uniform) { for (int idx = t; idx < S * S; idx += 256) o[idx] = 1.0 / (double)((size_t)S * S); return; }
    const int bs = d.brk_slot[n];
    const double *pd = (bs >= 0) ? d.pd_lt + ((size_t)r * d.NBE + bs) * d.M * d.D : nullptr;
    const double *fa = d.fa + rs_off(d, r, n), *fb = d.fb + rs_off(d, r, n + 1), *fe = d.fe + rs_off(d, r, n + 1);
    double z = 0.;
    for (int idx = t; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx - i * S;
        const double J = fa[i] * exp(trans_value(d, n, i, j, pd)) * (fe[j] * fb[j]);
        o[idx] = J; z += J;
    }
    z = block_sum<256>(z, scratch);
    if (t == 0) zsh = z;
    __syncthreads();
    for (int idx = t; idx < S * S; idx += 256) o[idx] = o[idx] / zsh;
}
// cn_states_total / num_alleles_subclonal / is_hdel / is_loh expansions are done on the host.
