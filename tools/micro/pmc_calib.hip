// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels of KNOWN bytes, in the access widths this library's streaming
// passes use (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  VERDICT r3 item 4: tools/pmc_summary.py
// doubles FETCH_SIZE for every kernel, and the fused marginal pass reads 8 bytes per lane.
//   read8 / read16 : every lane streams `bytes` of a buffer far larger than the Infinity Cache, 8 / 16 bytes per load, a wave a contiguous
//                    512 / 1024 bytes (the row reads of k_cells are the 8-byte form), and folds them into one value per workgroup
//   write8 / write16: the same for stores
//   rowsum165     : rows of 165 doubles at a pitch of 168 read as k_cells reads them (lane l: elements l, l + 64, l + 128 of a row)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/pmc_calib tools/micro/pmc_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE -d out_f -o c --output-format csv -- tools/micro/pmc_calib ;  the same with WRITE_SIZE;
//        python3 tools/pmc_calib_summary.py out_f out_w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read8(const double *src, size_t n, double *out) {
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 12345.678) out[blockIdx.x] = acc;      // (never true: keeps the loads)
}
__global__ void read16(const double2 *src, size_t n, double *out) {
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = src[i]; acc += v.x + v.y; }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}
__global__ void write8(double *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)i;
}
__global__ void write16(double2 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_double2((double)i, 1.);
}
// a wave per row of 165 doubles (pitch 168), lanes 0..63 read elements l, l + 64, l + 128 (the last load with 37 active lanes)
__global__ void rowsum165(const double *src, size_t rows, double *out) {
    const int lane = threadIdx.x & 63;
    double acc = 0.;
    for (size_t r = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * (blockDim.x >> 6)) {
        const double *row = src + r * 168;
        acc += row[lane] + row[lane + 64] + (lane + 128 < 165 ? row[lane + 128] : 0.);
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;      // 2 GiB per pass: eight times the Infinity Cache
    double *buf, *out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(buf, 0, bytes));
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read8, dim3(grid), dim3(256), 0, 0, (const double *)buf, bytes / 8, out);
        hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const double2 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(rowsum165, dim3(grid), dim3(256), 0, 0, (const double *)buf, bytes / (168 * 8), out);
        hipLaunchKernelGGL(write8, dim3(grid), dim3(256), 0, 0, buf, bytes / 8);
        hipLaunchKernelGGL(write16, dim3(grid), dim3(256), 0, 0, (double2 *)buf, bytes / 16);
        CHECK(hipGetLastError());
    }
    CHECK(hipDeviceSynchronize());
    printf("known bytes per launch: read8 %zu read16 %zu rowsum165 %zu (rows x 165 x 8; lines touched: rows x 168 x 8 = %zu) write8 %zu write16 %zu\n",
           bytes, bytes, bytes / (168 * 8) * 165 * 8, bytes / (168 * 8) * 168 * 8, bytes, bytes);
    return 0;
}
