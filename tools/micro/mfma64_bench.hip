// Microbenchmark + operand-layout probe of the FP64 MFMA instructions of gfx950 (design input for the forward-backward kernel).
//   hipcc --offload-arch=gfx950 -O3 -o mfma64_bench mfma64_bench.hip && ./mfma64_bench [layout] [throughput]
//   (no argument = both phases; profiles/r03_mfma64_probe.txt is the output the lane maps of DESIGN.md 4.4 were read from)
//
// History of a GPU memory-access fault (round 2, gpurun_out/mfma64.txt, 09:05): the first version of this probe allocated
// `dout` for the layout phase and the one-CU throughput launches only; the last launch of the throughput phase (grid 1024 x
// 768 threads, every thread storing out[blockIdx.x * blockDim.x + threadIdx.x]) wrote 786 432 doubles = 6.3 MB past it.
// The log seems to stop inside the layout phase because stdout was a pipe (block-buffered, cut at a 4 096-byte boundary): the
// fault is the out-of-bounds store of that last launch, not the lane-map kernels (which write 64 / 256 doubles).  The rerun one
// minute later (mfma64b.txt, clean) had the larger allocation -- and the layout phase switched off by a constant, which is why
// the lane maps could not be regenerated from the committed file.  Now: every launch goes through launch_checked(), which
// refuses a launch whose grid x block exceeds the allocation; HIP return codes are checked; stdout is line-buffered; the
// phases are argv switches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
// FP64 MFMA with NVALU cheap integer vector instructions (v_bfe_u32: the operand-address extraction of k_fbq) and NLDS
// ds_read_b64 (per-lane bank pair: conflict-free) per MFMA, 4 accumulators: does the matrix pipe run beside the vector ALU?
template <int NVALU, int NLDS> __global__ void bench4mix(double *out, int iters, unsigned long long *cyc) {
    __shared__ double tab[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) tab[i] = 1.0 + i * 1e-9;
    __syncthreads();
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double acc[4] = {0., 0., 0., 0.};
    unsigned c = threadIdx.x * 2654435761u, x = 0;
    double l[4] = {b, b, b, b};
    const unsigned lane_off = (threadIdx.x & 31) * 8;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, l[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NVALU; v++) asm volatile("v_bfe_u32 %0, %1, %2, 9" : "=v"(x) : "v"(c), "n"((i * 3 + v) % 23));
            if (NLDS) { unsigned ad = ((c >> (i + 3)) & 0x3f00u) | lane_off; asm volatile("ds_read_b64 %0, %1" : "=v"(l[i]) : "v"(ad) : "memory"); }
        }
        if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]) :: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]) + (double)x;
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

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(1); } } while (0)
static size_t g_out_doubles = 0;      // capacity of dout: no launch may have more threads than this
static void check_fits(int grid, int block, int per_thread = 1) {
    if ((size_t)grid * block * per_thread > g_out_doubles) { fprintf(stderr, "launch %d x %d x %d exceeds the %zu-double output buffer\n", grid, block, per_thread, g_out_doubles); exit(1); }
}
template <typename K> double run(K kern, int waves, int nacc, int iters, double *dout, unsigned long long *dcyc, int grid = 1) {
    check_fits(grid, 64 * waves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, 0, dout, iters, dcyc);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, 0, dout, iters, dcyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
    // s_memtime-style counter runs at 100 MHz: report wall time per instruction per SIMD instead
    const double per_simd = (double)iters * nacc * ((waves + 3) / 4);     // instructions issued by the busiest SIMD
    return ms * 1e6 / per_simd;   // ns per instruction on one SIMD
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    bool layout = argc < 2, throughput = argc < 2;
    for (int i = 1; i < argc; i++) { layout |= !strcmp(argv[i], "layout"); throughput |= !strcmp(argv[i], "throughput"); }
    double *dout; unsigned long long *dcyc;
    const int kMaxGrid = 1024, kMaxBlock = 768;                            // the largest launch below: one double per thread
    g_out_doubles = (size_t)kMaxGrid * kMaxBlock;
    CK(hipMalloc(&dout, g_out_doubles * sizeof(double))); CK(hipMalloc(&dcyc, 64));
    if (layout) {
    check_fits(1, 64, 4);
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
    if (!throughput) return 0;
    // ---------------- throughput (one CU: grid 1) ----------------
    const int iters = 20000;
    printf("== ns per instruction on one SIMD (2.4 GHz: 1 ns = 2.4 cycles) ==\n");
    for (int waves : {1, 4, 8, 12}) {
        printf("waves/CU %2d | 4x4x4: nacc1 %.2f nacc2 %.2f nacc4 %.2f nacc8 %.2f | 16x16x4: nacc1 %.2f nacc2 %.2f nacc4 %.2f | v_fmac_f64: nacc4 %.2f nacc8 %.2f\n", waves,
               run(bench4<1>, waves, 1, iters, dout, dcyc), run(bench4<2>, waves, 2, iters, dout, dcyc), run(bench4<4>, waves, 4, iters, dout, dcyc), run(bench4<8>, waves, 8, iters, dout, dcyc),
               run(bench16<1>, waves, 1, iters, dout, dcyc), run(bench16<2>, waves, 2, iters, dout, dcyc), run(bench16<4>, waves, 4, iters, dout, dcyc),
               run(benchv<4>, waves, 4, iters, dout, dcyc), run(benchv<8>, waves, 8, iters, dout, dcyc));
    }
    printf("== v_mfma_f64_4x4x4 beside other work, 12 waves per CU (3 per SIMD), ns per MFMA per SIMD: does integer VALU / LDS issue cost matrix-pipe time? ==\n");
    printf("valu0 %.2f | valu1 %.2f | valu2 %.2f | valu4 %.2f | lds1 (address from the data path: 2 extra VALU) %.2f | valu1+lds1 %.2f\n",
           run(bench4mix<0, 0>, 12, 4, iters, dout, dcyc), run(bench4mix<1, 0>, 12, 4, iters, dout, dcyc), run(bench4mix<2, 0>, 12, 4, iters, dout, dcyc),
           run(bench4mix<4, 0>, 12, 4, iters, dout, dcyc), run(bench4mix<0, 1>, 12, 4, iters, dout, dcyc), run(bench4mix<1, 1>, 12, 4, iters, dout, dcyc));
    printf("== all CUs busy (grid 1024, 12 waves per block) ==\n");
    printf("4x4x4 nacc4 %.2f | 16x16x4 nacc2 %.2f | v_fmac nacc8 %.2f (ns per instruction per SIMD, 4 blocks/CU-round)\n",
           run(bench4<4>, 12, 4, iters, dout, dcyc, kMaxGrid) / 4, run(bench16<2>, 12, 2, iters, dout, dcyc, kMaxGrid) / 4, run(benchv<8>, 12, 8, iters, dout, dcyc, kMaxGrid) / 4);
    return 0;
}
