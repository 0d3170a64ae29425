// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access pattern of the lane-per-problem kernels (ilqr_batch.hpp): one
// double per lane, consecutive lanes on consecutive addresses (512 B per wavefront access), buffers far past the 256 MiB
// Infinity Cache.  MI355X_MICROARCH.md calibrates FETCH_SIZE for 16 B / lane streams only (x2) and leaves other widths open.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream8.hip -o tools/ubench/stream8
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench/stream8      (WRITE_SIZE: a second pass)
// Kernels: read8 (reads `bytes`, writes one double per workgroup), write8 (writes `bytes`), copy8 (reads and writes `bytes`).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double* __restrict__ a, double* __restrict__ out, size_t n) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 12345.678) out[blockIdx.x] = s;              // (keeps the loads alive; never true for the data below)
}
__global__ void write8(double* __restrict__ a, size_t n, double v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)2048) << 20;      // MiB
  const size_t n = bytes / 8;
  double *a, *b, *o;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 1 << 20) != hipSuccess) return 1;
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    float ms[3];
    hipEventRecord(e0); hipLaunchKernelGGL(read8, grid, block, 0, 0, a, o, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL(write8, grid, block, 0, 0, b, n, 1.0); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
    hipEventRecord(e0); hipLaunchKernelGGL(copy8, grid, block, 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
    std::printf("bytes %zu  read8 %.3f ms (%.0f GB/s)  write8 %.3f ms (%.0f GB/s)  copy8 %.3f ms (%.0f GB/s moved)\n", bytes,
                ms[0], bytes / ms[0] * 1e-6, ms[1], bytes / ms[1] * 1e-6, ms[2], 2.0 * bytes / ms[2] * 1e-6);
  }
  return 0;
}
