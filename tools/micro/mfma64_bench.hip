// Microbenchmark + operand-layout probe of the FP64 MFMA instructions of gfx950 (design input for the forward-backward kernel).
//   hipcc --offload-arch=gfx950 -O3 -o mfma64_bench mfma64_bench.hip && ./mfma64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// ---- layout probe: D for one-hot A (lane la) and one-hot B (lane lb) ----
__global__ void probe4(int la, int lb, double *out) {
    const int l = threadIdx.x;
    double a = l == la ? 1. : 0., b = l == lb ? 1. : 0.;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
    out[l] = d;
}
__global__ void probe16(int la, int lb, double *out) {
    const int l = threadIdx.x;
    double a = l == la ? 1. : 0., b = l == lb ? 1. : 0.;
    d4 c = {0., 0., 0., 0.};
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) out[l * 4 + i] = d[i];
}

// ---- throughput: NACC independent accumulators, ITER rounds, per wave; blockDim = 64 * waves ----
template <int NACC> __global__ void bench4(double *out, int iters, unsigned long long *cyc) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> __global__ void bench16(double *out, int iters, unsigned long long *cyc) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (d4){0., 0., 0., 0.};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// vector FMA reference: same shape
template <int NACC> __global__ void benchv(double *out, int iters, unsigned long long *cyc) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K> double run(K kern, int waves, int nacc, int iters, double *dout, unsigned long long *dcyc, int grid = 1) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, 0, dout, iters, dcyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, 0, dout, iters, dcyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    // s_memtime-style counter runs at 100 MHz: report wall time per instruction per SIMD instead
    const double per_simd = (double)iters * nacc * ((waves + 3) / 4);     // instructions issued by the busiest SIMD
    return ms * 1e6 / per_simd;   // ns per instruction on one SIMD
}

int main() {
    double *dout; unsigned long long *dcyc;
    hipMalloc(&dout, (size_t)1024 * 768 * 8 * 2); hipMalloc(&dcyc, 64);   // largest launch: 1024 blocks x 768 threads, one double each
    const bool layout = false;
    if (layout) {
    std::vector<double> h(256);
    int amap[64][64];
    printf("== v_mfma_f64_4x4x4f64: for each (A lane, B lane) pair that multiplies: D lane ==\n");

    for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
        hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, la, lb, dout);
        hipMemcpy(h.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
        amap[la][lb] = -1;
        for (int l = 0; l < 64; l++) if (h[l] != 0.) amap[la][lb] = l;
    }
    for (int la = 0; la < 64; la += 1) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; lb++) if (amap[la][lb] >= 0) printf(" (B%d->D%d)", lb, amap[la][lb]);
        printf("\n");
    }
    printf("== v_mfma_f64_16x16x4f64: (A lane, B lane) -> D (lane, reg) for A lanes 0,1,16,17,32 ==\n");
    for (int la : {0, 1, 16, 17, 32, 63}) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; lb++) {
            hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, la, lb, dout);
            hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; i++) if (h[i] != 0.) printf(" (B%d->D%d.%d)", lb, i / 4, i % 4);
        }
        printf("\n");
    }
    }
    // ---------------- throughput (one CU: grid 1) ----------------
    const int iters = 20000;
    printf("== ns per instruction on one SIMD (2.4 GHz: 1 ns = 2.4 cycles) ==\n");
    for (int waves : {1, 4, 8, 12}) {
        printf("waves/CU %2d | 4x4x4: nacc1 %.2f nacc2 %.2f nacc4 %.2f nacc8 %.2f | 16x16x4: nacc1 %.2f nacc2 %.2f nacc4 %.2f | v_fmac_f64: nacc4 %.2f nacc8 %.2f\n", waves,
               run(bench4<1>, waves, 1, iters, dout, dcyc), run(bench4<2>, waves, 2, iters, dout, dcyc), run(bench4<4>, waves, 4, iters, dout, dcyc), run(bench4<8>, waves, 8, iters, dout, dcyc),
               run(bench16<1>, waves, 1, iters, dout, dcyc), run(bench16<2>, waves, 2, iters, dout, dcyc), run(bench16<4>, waves, 4, iters, dout, dcyc),
               run(benchv<4>, waves, 4, iters, dout, dcyc), run(benchv<8>, waves, 8, iters, dout, dcyc));
    }
    printf("== all CUs busy (grid 1024, 12 waves per block) ==\n");
    printf("4x4x4 nacc4 %.2f | 16x16x4 nacc2 %.2f | v_fmac nacc8 %.2f (ns per instruction per SIMD, 4 blocks/CU-round)\n",
           run(bench4<4>, 12, 4, iters, dout, dcyc, 1024) / 4, run(bench16<2>, 12, 2, iters, dout, dcyc, 1024) / 4, run(benchv<8>, 12, 8, iters, dout, dcyc, 1024) / 4);
    return 0;
}
