// Shader clock under load and the dead time of a launch: 1024 one-wave workgroups (the C2 shape, 36 KB of LDS
// each) run a dependent fp64 FMA chain; every wave records s_memtime (shader clock) and s_memrealtime (100 MHz)
// at its first and last instruction.  Prints the clock the waves actually ran at, how long after the first wave
// the last one started, and the kernel time the HIP events report around the same launch.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clk.hip -o tools/ubench/clk && tools/ubench/clk
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(64) spin(long long* out, double* sink, int iters) {
  extern __shared__ double lds[];
  const long long c0 = __builtin_readcyclecounter();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  double x = threadIdx.x * 1e-3, y = 1.0000001;
  for (int i = 0; i < iters; ++i) { x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); }
  const long long c1 = __builtin_readcyclecounter();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (x == 12345.678) sink[0] = x + lds[0];
  if (threadIdx.x == 0) { long long* o = out + 4 * blockIdx.x; o[0] = c0; o[1] = c1; o[2] = r0; o[3] = r1; }
}
int main() {
  const int B = 1024;
  long long* d; double* sink;
  hipMalloc(&d, B * 4 * sizeof(long long)); hipMalloc(&sink, 8);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t st; hipStreamCreate(&st);
  for (int iters : {2000, 16000, 16000, 16000}) {
    void* args[] = {&d, &sink, (void*)&iters};
    for (int rep = 0; rep < 3; ++rep) hipExtLaunchKernel((const void*)spin, dim3(B), dim3(64), args, 36 * 1024, st, e0, e1, 0);
    hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(B * 4);
    hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost);
    long long r_first = h[2], r_last_start = h[2], r_end = h[3]; double fsum = 0;
    for (int b = 0; b < B; ++b) {
      r_first = std::min(r_first, h[4 * b + 2]); r_last_start = std::max(r_last_start, h[4 * b + 2]); r_end = std::max(r_end, h[4 * b + 3]);
      fsum += double(h[4 * b + 1] - h[4 * b]) / double(h[4 * b + 3] - h[4 * b + 2]) * 100.0;
    }
    printf("iters %6d  kernel(events) %8.2f us  first wave -> last wave end %8.2f us  last wave started %6.2f us after the first  mean clock %7.1f MHz  cycles per fma %.2f\n",
           iters, ms * 1e3, (r_end - r_first) / 100.0, (r_last_start - r_first) / 100.0, fsum / B, double(h[1] - h[0]) / (4.0 * iters));
  }
  return 0;
}
