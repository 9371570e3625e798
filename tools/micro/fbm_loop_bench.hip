// Where do the cycles of a k_fbm step go?  The step loop of the forward-backward kernel reduced to its skeleton, with
// the parts switchable: A operand from LDS or a register, the 2 KB block exchange, the barrier, wave priorities.
//   hipcc --offload-arch=gfx950 -O3 -o fbm_loop_bench fbm_loop_bench.hip && ./fbm_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define KJ 11
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));
    return v;
}
// RES: 0 none, 1 + scale / row maximum / ds_max, 2 + global emission load and result store
template <bool LDSA, bool EXCH, bool BAR, bool PRIO, int NG, int RES = 0, bool RTNG = false>
__global__ __launch_bounds__(768) void loopk(const double *W, double *out, int steps, unsigned long long *cyc, double *big = nullptr, int ngrt = 4) {
    __shared__ unsigned red32[16];
    __shared__ double vec[2 * 64 * KJ];
    __shared__ double xch[12 * 256];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double w[KJ][4];
#pragma unroll
    for (int j = 0; j < KJ; j++)
#pragma unroll
        for (int g = 0; g < 4; g++) w[j][g] = W[((wave * KJ + j) * 4 + g) * 64 + lane];
    for (int i = t; i < 2 * 64 * KJ; i += 768) vec[i] = 1e-3 * (i % 7);
    __syncthreads();
    if (PRIO) { if (wave < 4) __builtin_amdgcn_s_setprio(2); else if (wave < 8) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    const int id = lane >> 4, bq = (lane >> 2) & 3, nq = lane & 3;
    double areg = 1e-3 * lane;
    if (t < 16) red32[t] = 0x3ff00000u;
    double *gp = big ? big + (size_t)blockIdx.x * 768 * 8 + t : nullptr;
    const int ng = RTNG ? (wave < 6 ? ngrt : ngrt - 1) : NG;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 1; k <= steps; k++) {
        const double *apc = vec + ((k - 1) & 1) * 64 * KJ + lane;
        double e = 1.0;
        if (RES >= 2) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(e) : "v"(gp + (size_t)(k & 1023) * 768 * 64) : "memory");
        const unsigned hi_prev = RES >= 1 ? red32[((k - 1) % 3) * 4 + id] : 0u;
        double acc0 = 0., acc1 = 0., acc2 = 0., acc3 = 0.;
        double a0 = LDSA ? apc[0] : areg, a1 = LDSA ? apc[64] : areg;
#pragma unroll
        for (int j = 0; j < KJ; j++) {
            const double an = (LDSA && j + 2 < KJ) ? apc[(j + 2) * 64] : areg;
            acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, w[j][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, w[j][1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, w[j][2], acc2, 0, 0, 0);
            if (RTNG ? ng > 3 : NG > 3) acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, w[j][3], acc3, 0, 0, 0);
            a0 = a1; a1 = an;
        }
        double sum;
        if (EXCH) {
            double *xw = xch + wave * 256 + (id * 16 + nq) * 4 + bq;
            xw[0] = acc0; xw[16] = acc1; xw[32] = acc2; xw[48] = acc3;
            const double2 p01 = *reinterpret_cast<const double2 *>(xch + wave * 256 + lane * 4);
            const double2 p23 = *reinterpret_cast<const double2 *>(xch + wave * 256 + lane * 4 + 2);
            sum = ((p01.x + p01.y) + p23.x) + p23.y;
        } else sum = (acc0 + acc1) + (acc2 + acc3);
        sum *= 1e-30;
        if (RES >= 1) {
            const unsigned ef = (hi_prev >> 20) & 0x7ffu;
            const double inv = (ef == 0x7ffu || ef == 0u) ? 0. : __hiloint2double((int)((2046u - ef) << 20), 0);
            if (RES >= 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(e) :: "memory");
            const double val = sum * inv, vecv = val * e;
            if (RES >= 2) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(gp + (size_t)((k & 1023) * 768 * 64 + 768 * 32)), "v"(vecv) : "memory");
            sum = vecv;
            const unsigned rm = row_max_u32((unsigned)__double2hiint(vecv) | 0x3ff00000u);
            if ((lane & 15) == 0 && rm) { const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&red32[(k % 3) * 4 + id]; asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(rm) : "memory"); }
            if (wave == 11 && lane < 4) red32[((k + 1) % 3) * 4 + lane] = 0u;
        }
        if (LDSA) vec[(k & 1) * 64 * KJ + (wave * 64 + lane) % (64 * KJ)] = sum; else areg = sum + 1e-3 * lane;
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 768 + t] = LDSA ? vec[t] : areg;
    if (t == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// v3 shape: wave owns 16 columns, KB = 42 k-blocks, one replicated (broadcast) A read per k-block (b64) or per pair (b128).
// FEAT bits: 1 global emission load + wait + result store, 2 scale from the previous row's maximum, 4 per-lane ds_max,
// 8 service wave (slot clear + scale store), 16 wave priorities, 32 predication of stores (live lanes only)
typedef double d2v __attribute__((ext_vector_type(2)));
template <int WIDE, bool BAR, int FEAT = 0>
__global__ __launch_bounds__(768) void loopv3(const double *W, double *out, int steps, unsigned long long *cyc, double *big = nullptr, int ngrt = 4) {
    __shared__ double vec[2 * 176 * 4];
    __shared__ unsigned red32[16];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int KB = 42;
    double w[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) w[kb] = W[(wave * KB + kb) * 64 + lane];
    for (int i = t; i < 2 * 176 * 4; i += 768) vec[i] = 1e-3 * (i % 7);
    if (t < 16) red32[t] = 0x3ff00000u;
    __syncthreads();
    const int kq = lane >> 4, ib = lane & 3, id = lane >> 4;
    if (FEAT & 16) { if (wave < 4) __builtin_amdgcn_s_setprio(2); else if (wave < 8) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    double *gp = big ? big + (size_t)blockIdx.x * 768 * 8 + t : nullptr;
    const bool live = !(FEAT & 32) || ((wave * 16 + (lane & 15)) < 165 && id < ngrt);
    int s_prev = 0, s_cur = 1, s_next = 2;
    for (int k = 1; k <= steps; k++) {
        double acc[4] = {0., 0., 0., 0.};
        if (wave < 11) {
            double e = 1.0;
            if (FEAT & 1) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(e) : "v"(gp + (size_t)(k & 1023) * 768 * 64) : "memory");
            const unsigned hi_prev = (FEAT & 2) ? red32[s_prev * 4 + id] : 0x3ff00000u;
            if (WIDE == 1) {
                const double *apc = vec + ((k - 1) & 1) * 176 * 4 + kq * 4 + ib;
#pragma unroll
                for (int kb = 0; kb < KB; kb++) acc[kb & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(apc[kb * 16], w[kb], acc[kb & 3], 0, 0, 0);
            } else {
                const d2v *apc = reinterpret_cast<const d2v *>(vec + ((k - 1) & 1) * 176 * 4 + (kq * 4 + ib) * 2);
#pragma unroll
                for (int p = 0; p < KB / 2; p++) {
                    const d2v av = apc[p * 16];
                    acc[(2 * p) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av.x, w[2 * p], acc[(2 * p) & 3], 0, 0, 0);
                    acc[(2 * p + 1) & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av.y, w[2 * p + 1], acc[(2 * p + 1) & 3], 0, 0, 0);
                }
            }
            double sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * 1e-30;
            const unsigned ef = (hi_prev >> 20) & 0x7ffu;
            const double inv = (ef == 0x7ffu || ef == 0u) ? 0. : __hiloint2double((int)((2046u - ef) << 20), 0);
            if (FEAT & 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(e) :: "memory");
            const double val = sum * inv, vecv = val * e + 1.0;
            if (live) {
                if (FEAT & 1) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(gp + (size_t)((k & 1023) * 768 * 64 + 768 * 32)), "v"(vecv) : "memory");
                if (FEAT & 4) { const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)&red32[s_cur * 4 + id]; asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"((unsigned)__double2hiint(vecv)) : "memory"); }
            }
            vec[(k & 1) * 176 * 4 + (wave * 64 + lane) % (176 * 4)] = live ? vecv : 0.;
        } else if ((FEAT & 8) && lane < 4) {
            const unsigned ef = (red32[s_prev * 4 + lane] >> 20) & 0x7ffu;
            const double m = __hiloint2double((int)(ef << 20), 0);
            if (gp) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(gp + (size_t)((k & 1023) * 768 * 64 + 768 * 48)), "v"(m) : "memory");
            red32[s_next * 4 + lane] = (FEAT & 4) ? 0u : 0x3ff00000u;
        }
        { const int t_ = s_prev; s_prev = s_cur; s_cur = s_next; s_next = t_; }
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    out[blockIdx.x * 768 + t] = vec[t];
}
template <typename K> void run(const char *name, K kern, const double *W, double *out, unsigned long long *cyc, int nmfma, double *big = nullptr, int ngrt = 4) {
    const int steps = 4000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(768), 0, 0, W, out, steps, cyc, big, ngrt); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1), dim3(768), 0, 0, W, out, steps, cyc, big, ngrt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f ns/step  = %6.0f cycles at 2.4 GHz  (%d MFMA per wave-step: %.1f cycles per MFMA per SIMD)\n", name, ms * 1e6 / steps, ms * 1e6 / steps * 2.4, nmfma,
           ms * 1e6 / steps * 2.4 / (3 * nmfma));
}
int main() {
    double *W, *out; unsigned long long *cyc;
    hipMalloc(&W, 12 * KJ * 4 * 64 * 8); hipMalloc(&out, 768 * 8 * 4); hipMalloc(&cyc, 64);
    hipMemset(W, 0, 12 * KJ * 4 * 64 * 8);
    run("regA  noexch nobar noprio 4 groups", loopk<false, false, false, false, 4>, W, out, cyc, 44);
    run("ldsA  noexch nobar noprio 4 groups", loopk<true, false, false, false, 4>, W, out, cyc, 44);
    run("ldsA  noexch bar   noprio 4 groups", loopk<true, false, true, false, 4>, W, out, cyc, 44);
    run("ldsA  exch   bar   noprio 4 groups", loopk<true, true, true, false, 4>, W, out, cyc, 44);
    run("ldsA  exch   bar   prio   4 groups", loopk<true, true, true, true, 4>, W, out, cyc, 44);
    run("ldsA  exch   bar   prio   3 groups", loopk<true, true, true, true, 3>, W, out, cyc, 33);
    run("regA  noexch bar   noprio 4 groups", loopk<false, false, true, false, 4>, W, out, cyc, 44);
    run("v3: 16-col tiles, b64 read per MFMA, bar", loopv3<1, true>, W, out, cyc, 42);
    run("v3: 16-col tiles, b128 read per 2 MFMA, bar", loopv3<2, true>, W, out, cyc, 42);
    run("v3: 16-col tiles, b128 read per 2 MFMA, nobar", loopv3<2, false>, W, out, cyc, 42);
    double *big; hipMalloc(&big, (size_t)1024 * 768 * 64 * 8 + 768 * 64 * 8);      // 1024 slots x 768 threads x 64 doubles stride
    run("v3 b128 bar + global e/store (1)", loopv3<2, true, 1>, W, out, cyc, 42, big);
    run("v3 b128 bar + scale (2)", loopv3<2, true, 2>, W, out, cyc, 42, big);
    run("v3 b128 bar + scale + ds_max (6)", loopv3<2, true, 6>, W, out, cyc, 42, big);
    run("v3 b128 bar + scale + ds_max + service (14)", loopv3<2, true, 14>, W, out, cyc, 42, big);
    run("v3 b128 bar + all but prio (47)", loopv3<2, true, 47>, W, out, cyc, 42, big);
    run("v3 b128 bar + all (63)", loopv3<2, true, 63>, W, out, cyc, 42, big);
    run("ldsA exch bar prio 4 groups + scale/max", loopk<true, true, true, true, 4, 1>, W, out, cyc, 44);
    run("ldsA exch bar prio 4 groups + scale/max + global", loopk<true, true, true, true, 4, 2>, W, out, cyc, 44, big);
    run("ldsA exch bar prio runtime 4/3 groups", loopk<true, true, true, true, 4, 0, true>, W, out, cyc, 44);
    run("ldsA exch bar prio runtime 4/3 + scale/max + global", loopk<true, true, true, true, 4, 2, true>, W, out, cyc, 44, big);
    run("ldsA exch bar NOprio runtime 4/3 + scale/max + global", loopk<true, true, true, false, 4, 2, true>, W, out, cyc, 44, big);
    return 0;
}
