// Gap between dependent dispatches on one stream as a function of the launch shape (gfx950):
// 20 back-to-back launches of a ~150 us spin kernel, wall time per launch minus the kernel's own duration.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gap.hip -o tools/ubench/gap && tools/ubench/gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void __launch_bounds__(64) spin(double* sink, int iters, int stores) {
  extern __shared__ double lds[];
  double x = threadIdx.x * 1e-3, y = 1.0000001;
  for (int i = 0; i < iters; ++i) { x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); }
  for (int s = 0; s < stores; ++s) sink[((size_t)blockIdx.x * stores + s) * 64 + threadIdx.x] = x + s;
  if (x == 12345.678) sink[0] = lds[0];
}
__global__ void tiny(double* sink) { if (threadIdx.x == 9999) sink[0] = 1.0; }
int main() {
  double* sink; hipMalloc(&sink, (size_t)1024 * 256 * 64 * 8);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t st; hipStreamCreate(&st);
  struct Case { const char* name; int grid; int lds; int stores; int with_tiny; int events; };
  const Case cases[] = {{"grid 1024, 36 KB LDS, no stores", 1024, 36 * 1024, 0, 0, 1}, {"grid 1024, no LDS", 1024, 0, 0, 0, 1},
                        {"grid 256, 36 KB LDS", 256, 36 * 1024, 0, 0, 1}, {"grid 1024, 36 KB LDS, 11 MB of result stores", 1024, 36 * 1024, 21, 0, 1},
                        {"grid 1024, 36 KB LDS + a tiny kernel after each", 1024, 36 * 1024, 0, 1, 1},
                        {"grid 1024, 36 KB LDS, plain launch without events", 1024, 36 * 1024, 0, 0, 0}};
  int iters = 7600;
  for (const Case& c : cases) {
    void* args[] = {&sink, &iters, (void*)&c.stores};
    auto go = [&](int n) {
      for (int i = 0; i < n; ++i) {
        if (c.events) hipExtLaunchKernel((const void*)spin, dim3(c.grid), dim3(64), args, c.lds, st, e0, e1, 0);
        else hipLaunchKernelGGL(spin, dim3(c.grid), dim3(64), c.lds, st, sink, iters, c.stores);
        if (c.with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, sink);
      }
      hipStreamSynchronize(st);
    };
    go(5);
    auto t0 = std::chrono::steady_clock::now();
    go(20);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 20.0;
    float ms = 0; if (c.events) hipEventElapsedTime(&ms, e0, e1);
    printf("%-55s per launch %8.2f us   kernel (events) %8.2f us   difference %6.2f us\n", c.name, us, ms * 1e3, us - ms * 1e3);
  }
  return 0;
}
