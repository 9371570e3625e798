// Where do the cycles of a Viterbi lattice step go (k_viterbi_max, 165 states: 11 waves, thread (o, p) with 42 source states)?  The step's
// skeleton with its parts switchable: the 22 LDS reads (ds_read_b128 pipeline), the 42 add + 42 max, the merge, the LDS write, the barrier --
// and the alternative product forms: V broadcast through the FMA's DPP operand (v_mov_b64 + v_fmac_f64_dpp + v_max_f64, 3 LDS reads per lane).
//   hipcc --offload-arch=gfx950 -O3 -o vit_loop_bench vit_loop_bench.hip && ./vit_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int K> __device__ __forceinline__ void rd128(d2 &dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(16 * K) : "memory"); }
template <int CNT> __device__ __forceinline__ void lwait(d2 &x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(CNT) : "memory"); }
template <int G, int NG, int DEPTH, bool LDS, bool VALU> struct pipe {
    static __device__ __forceinline__ void step(d2 (&buf)[DEPTH], const double (&T)[2 * NG], unsigned addr, double &b0, double &b1) {
        constexpr int younger = (NG - 1 - G) < (DEPTH - 1) ? (NG - 1 - G) : (DEPTH - 1);
        if (LDS) lwait<younger>(buf[G % DEPTH]);
        const d2 x = buf[G % DEPTH];
        if (LDS) if constexpr (G + DEPTH < NG) rd128<G + DEPTH>(buf[G % DEPTH], addr);
        if (VALU) { b0 = fmax(b0, x.x + T[2 * G]); b1 = fmax(b1, x.y + T[2 * G + 1]); }
        if constexpr (G + 1 < NG) pipe<G + 1, NG, DEPTH, LDS, VALU>::step(buf, T, addr, b0, b1);
    }
    template <int I> static __device__ __forceinline__ void fill(d2 (&buf)[DEPTH], unsigned addr) { rd128<I>(buf[I], addr); if constexpr (I + 1 < DEPTH) fill<I + 1>(buf, addr); }
};
template <int J> __device__ __forceinline__ void vfma_dpp(double &acc, const double a, const double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(x), "n"(J));
}
template <int KBI, int KB> struct dppchain {
    static __device__ __forceinline__ void run(const double (&av)[3], const double (&T)[KB], double &b0, double &b1) {
        double tmp = T[KBI];
        asm volatile("" : "+v"(tmp));
        vfma_dpp<KBI % 16>(tmp, av[KBI / 16], 1.0);
        if (KBI & 1) b1 = fmax(b1, tmp); else b0 = fmax(b0, tmp);
        if constexpr (KBI + 1 < KB) dppchain<KBI + 1, KB>::run(av, T, b0, b1);
    }
};
// MODE bits: 1 LDS reads, 2 add/max, 4 merge (2 shuffles), 8 LDS write, 16 barrier; FORM 0 = LDS pipeline (thread (o, p)), 1 = DPP broadcast (lane (kq, c))
template <int MODE, int FORM>
__global__ __launch_bounds__(768) void loopk(double *out, int steps, unsigned long long *cyc) {
    __shared__ __align__(16) double V[2 * 256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double T[44];
#pragma unroll
    for (int i = 0; i < 44; i++) T[i] = -10. * ((t * 7 + i * 3) % 9);
    for (int i = t; i < 512; i += blockDim.x) V[i] = -1e3 * (i % 13);
    __syncthreads();
    const int o = t / 4, p = t % 4;
    double keep = 0.;
    unsigned long long t0 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int k = 1; k <= steps; k++) {
        const int cur = (k - 1) & 1, nxt = k & 1;
        double b0 = -INFINITY, b1 = -INFINITY;
        if (FORM == 0) {
            const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(V + cur * 256 + p * 42);
            d2 buf[6];
            if (MODE & 1) pipe<0, 22, 6, true, true>::template fill<0>(buf, addr);
            else { for (int i = 0; i < 6; i++) { buf[i].x = keep + i; buf[i].y = keep - i; } }
            if ((MODE & 1) && (MODE & 2)) pipe<0, 22, 6, true, true>::step(buf, T, addr, b0, b1);
            else if (MODE & 1) pipe<0, 22, 6, true, false>::step(buf, T, addr, b0, b1);
            else if (MODE & 2) pipe<0, 22, 6, false, true>::step(buf, T, addr, b0, b1);
        } else {
            double av[3];
            const double *avp = V + cur * 256 + 16 * (lane >> 4) + (lane & 15);
            if (MODE & 1) { av[0] = avp[0]; av[1] = avp[64]; av[2] = avp[128]; } else { av[0] = keep; av[1] = keep + 1; av[2] = keep + 2; }
            if (MODE & 2) dppchain<0, 42>::run(av, reinterpret_cast<const double (&)[42]>(T), b0, b1);
        }
        double best = fmax(b0, b1);
        if (MODE & 4) { best = fmax(best, __shfl_xor(best, 1, 64)); best = fmax(best, __shfl_xor(best, 2, 64)); }
        keep = best * 1e-9;
        if (MODE & 8) { if (p == 0 && o < 165) V[nxt * 256 + o] = best + 1.0; }
        if (MODE & 16) __syncthreads();
    }
    unsigned long long t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0) cyc[wave] = t1 - t0;
    out[t] = keep;
}
template <int MODE, int FORM> void run(const char *name, int nt) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 768 * 8); hipMalloc(&cyc, 16 * 8);
    const int steps = 20000;
    hipLaunchKernelGGL((loopk<MODE, FORM>), dim3(1), dim3(nt), 0, 0, out, 200, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((loopk<MODE, FORM>), dim3(1), dim3(nt), 0, 0, out, steps, cyc);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-62s %2d waves: %6.0f cycles per step (wave 0), %6.0f (last wave)\n", name, nt / 64, (double)h[0] / steps, (double)h[nt / 64 - 1] / steps);
    hipFree(out); hipFree(cyc);
}
int main() {
    printf("# thread (o, p) form: 22 ds_read_b128 + 42 add + 42 max per thread and step\n");
    run<2, 0>("add / max only (operands in registers)", 704);
    run<1, 0>("LDS reads only", 704);
    run<3, 0>("LDS reads + add / max", 704);
    run<3 | 16, 0>("... + barrier", 704);
    run<3 | 4 | 8 | 16, 0>("... + merge + LDS write (the whole step without global memory)", 704);
    run<2, 0>("add / max only, ONE wave", 64);
    run<3, 0>("LDS reads + add / max, ONE wave", 64);
    run<3, 0>("LDS reads + add / max, 4 waves (one per SIMD)", 256);
    run<3, 0>("LDS reads + add / max, 8 waves", 512);
    printf("# lane (kq, c) form: 3 LDS reads, 42 x (v_mov_b64, v_fmac_f64_dpp row_newbcast, v_max_f64) per thread and step\n");
    run<2, 1>("mov / fmac_dpp / max only", 704);
    run<3, 1>("3 LDS reads + mov / fmac_dpp / max", 704);
    run<3 | 16, 1>("... + barrier", 704);
    run<3, 1>("3 LDS reads + mov / fmac_dpp / max, ONE wave", 64);
    return 0;
}
