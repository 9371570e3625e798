// What does the memory system give the access pattern of the fused marginal pass?  A wave owns a strip of 8 consecutive rows of 165
// doubles (pitch 168); per row it reads 8 streams (forward, backward, six cached planes: separate arrays) and writes one row.
// Variants: 8- or 16-byte loads per lane; occupancy capped at 4 waves per SIMD by 36 KB of LDS per block (as the pass) or free;
// with and without the output row.
// hipcc --offload-arch=gfx950 -O3 -o stream_rows stream_rows.hip && ./stream_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int S = 165, SP = 168, RPW = 8, NSTREAM = 8;
template <int VEC, bool CAP, bool DEEP>
__global__ __launch_bounds__(256) void k_rows(const double *in, double *out, size_t nrows, size_t pad) {
    __shared__ double cap[CAP ? 36 * 128 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nbeg = ((size_t)blockIdx.x * 4 + wave) * RPW;
    if (nbeg >= nrows) return;
    const size_t plane = nrows * SP + (pad > 16 ? pad : 0);
    if (CAP && threadIdx.x == 1000) cap[0] = 1.;      // (keeps the array)
    if (VEC == 1) {
        for (int n = 0; n < RPW; n++) {
            const size_t ro = (nbeg + n) * SP;
            double acc[3] = {0., 0., 0.};
            double v[NSTREAM][3];
#pragma unroll
            for (int q = 0; q < NSTREAM; q++)
#pragma unroll
                for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; v[q][k] = s < S ? in[q * plane + ro + s] : 0.; }
#pragma unroll
            for (int q = 0; q < NSTREAM; q++)
#pragma unroll
                for (int k = 0; k < 3; k++) acc[k] += v[q][k];
            if (DEEP) { if (acc[0] + acc[1] + acc[2] == 123.456) out[ro] = 1.; }      // (DEEP: reads only)
            else if (pad == 7) {      // (pad 7: non-temporal stores)
#pragma unroll
            for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; if (s < SP) __builtin_nontemporal_store(acc[k], out + ro + s); }
            } else if (pad == 9) {    // (pad 9: output rows at a pitch of 176 doubles = 11 cache lines)
#pragma unroll
            for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; if (s < 176) out[(nbeg + n) * 176 + s] = acc[k]; }
            } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; if (s < SP) out[ro + s] = acc[k]; }
            }
        }
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        for (int n = 0; n < RPW; n++) {
            const size_t ro = (nbeg + n) * SP;
            d2 acc[2] = {{0., 0.}, {0., 0.}};
            d2 v[NSTREAM][2];
#pragma unroll
            for (int q = 0; q < NSTREAM; q++)
#pragma unroll
                for (int k = 0; k < 2; k++) { const int e = lane + 64 * k; v[q][k] = e * 2 < SP ? *(const d2 *)(in + q * plane + ro + e * 2) : d2{0., 0.}; }
#pragma unroll
            for (int q = 0; q < NSTREAM; q++)
#pragma unroll
                for (int k = 0; k < 2; k++) acc[k] += v[q][k];
#pragma unroll
            for (int k = 0; k < 2; k++) { const int e = lane + 64 * k; if (e * 2 < SP) *(d2 *)(out + ro + e * 2) = acc[k]; }
        }
    }
    if (CAP && cap[0] == 123.) out[0] = 0.;
}
// write-only rows; and rows with the strip's eight output rows kept in registers and written at the end of the strip
__global__ __launch_bounds__(256) void k_rows_wo(double *out, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nbeg = ((size_t)blockIdx.x * 4 + wave) * RPW;
    if (nbeg >= nrows) return;
    for (int n = 0; n < RPW; n++) {
        const size_t ro = (nbeg + n) * SP;
#pragma unroll
        for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; if (s < SP) out[ro + s] = (double)s; }
    }
}
__global__ __launch_bounds__(256) void k_rows_late(const double *in, double *out, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nbeg = ((size_t)blockIdx.x * 4 + wave) * RPW;
    if (nbeg >= nrows) return;
    const size_t plane = nrows * SP;
    double res[RPW][3];
#pragma unroll
    for (int n = 0; n < RPW; n++) {
        const size_t ro = (nbeg + n) * SP;
        double acc[3] = {0., 0., 0.};
#pragma unroll
        for (int q = 0; q < NSTREAM; q++)
#pragma unroll
            for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; acc[k] += s < S ? in[q * plane + ro + s] : 0.; }
#pragma unroll
        for (int k = 0; k < 3; k++) res[n][k] = acc[k];
    }
#pragma unroll
    for (int n = 0; n < RPW; n++)
#pragma unroll
        for (int k = 0; k < 3; k++) { const int s = lane + 64 * k; if (s < SP) out[(nbeg + n) * SP + s] = res[n][k]; }
}
static int run_extra(const double *in, double *out, size_t nrows) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((nrows + 4 * RPW - 1) / (4 * RPW));
    for (int which = 0; which < 2; which++) {
        for (int it = 0; it < 7; it++) {
            if (it == 2) CHECK(hipEventRecord(a));
            if (which == 0) hipLaunchKernelGGL(k_rows_wo, dim3(grid), dim3(256), 0, 0, out, nrows);
            else hipLaunchKernelGGL(k_rows_late, dim3(grid), dim3(256), 0, 0, in, out, nrows);
        }
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        const double bytes = (double)nrows * S * 8 * (which == 0 ? 1 : NSTREAM + 1);
        printf("%-40s %.3f ms  %.2f TB/s (algorithmic %.2f GB)\n", which == 0 ? "rows, the write only" : "rows, the strip's output written last", ms, bytes / ms * 1e-9, bytes * 1e-9);
    }
    return 0;
}
template <int VEC, bool CAP, bool DEEP> static int run(const char *name, const double *in, double *out, size_t nrows, size_t pad = 0) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((nrows + 4 * RPW - 1) / (4 * RPW));
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_rows<VEC, CAP, DEEP>), dim3(grid), dim3(256), 0, 0, in, out, nrows, pad);
    CHECK(hipEventRecord(a));
    const int reps = 5;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL((k_rows<VEC, CAP, DEEP>), dim3(grid), dim3(256), 0, 0, in, out, nrows, pad);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double bytes = (double)nrows * S * 8 * (NSTREAM + (DEEP ? 0 : 1));
    printf("%-40s %.3f ms  %.2f TB/s (algorithmic %.2f GB)\n", name, ms, bytes / ms * 1e-9, bytes * 1e-9);
    return 0;
}
// plain streams for comparison: every lane 16 bytes, consecutive lanes consecutive, a block walks a contiguous chunk
__global__ __launch_bounds__(256) void k_plain_read(const double *in, double *out, size_t n16, int per_block) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *p = (const d2 *)in;
    d2 acc = {0., 0.};
    const size_t base = (size_t)blockIdx.x * per_block * 256;
    for (int i = 0; i < per_block; i++) { const size_t e = base + (size_t)i * 256 + threadIdx.x; if (e < n16) acc += p[e]; }
    if (acc.x + acc.y == 123.456) out[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_plain_copy(const double *in, double *out, size_t n16, int per_block) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *p = (const d2 *)in; d2 *q = (d2 *)out;
    const size_t base = (size_t)blockIdx.x * per_block * 256;
    for (int i = 0; i < per_block; i++) { const size_t e = base + (size_t)i * 256 + threadIdx.x; if (e < n16) q[e] = p[e]; }
}
static int run_plain(const double *in, double *out, size_t bytes_in, bool copy) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t n16 = bytes_in / 16; const int per_block = 16;
    const unsigned grid = (unsigned)((n16 + (size_t)per_block * 256 - 1) / ((size_t)per_block * 256));
    for (int it = 0; it < 7; it++) {
        if (it == 2) CHECK(hipEventRecord(a));
        if (copy) hipLaunchKernelGGL(k_plain_copy, dim3(grid), dim3(256), 0, 0, in, out, n16, per_block);
        else hipLaunchKernelGGL(k_plain_read, dim3(grid), dim3(256), 0, 0, in, out, n16, per_block);
    }
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = (double)bytes_in * (copy ? 2 : 1);
    printf("%-40s %.3f ms  %.2f TB/s (%.2f GB)\n", copy ? "plain copy (read + write)" : "plain read", ms, bytes / ms * 1e-9, bytes * 1e-9);
    return 0;
}
int main() {
    const size_t nrows = 400000;
    double *in, *out;
    CHECK(hipMalloc(&in, (nrows * SP + 65536) * 8 * NSTREAM)); CHECK(hipMalloc(&out, nrows * 176 * 8));
    CHECK(hipMemset(in, 0, (nrows * SP + 65536) * 8 * NSTREAM));
    if (run_plain(in, out, nrows * SP * 8 * NSTREAM, false)) return 1;
    if (run_plain(in, out, nrows * SP * 8, true)) return 1;
    if (run<1, true, false>("8-byte loads, 4 waves per SIMD", in, out, nrows)) return 1;
    if (run<1, false, false>("8-byte loads, occupancy free", in, out, nrows)) return 1;
    if (run<2, true, false>("16-byte loads, 4 waves per SIMD", in, out, nrows)) return 1;
    if (run<2, false, false>("16-byte loads, occupancy free", in, out, nrows)) return 1;
    if (run<1, true, false>("8-byte loads, planes skewed by 4160 B", in, out, nrows, 520)) return 1;
    if (run<1, true, true>("8-byte loads, the eight reads only", in, out, nrows)) return 1;
    if (run_extra(in, out, nrows)) return 1;
    if (run<1, true, false>("8-byte loads, non-temporal stores", in, out, nrows, 7)) return 1;
    if (run<1, true, false>("8-byte loads, output pitch 176", in, out, nrows, 9)) return 1;
    return 0;
}
