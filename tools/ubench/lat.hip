// Micro-benchmark: fp64 issue interval and dependent latency for ONE wave on a SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int CHAINS>
__global__ void fma_chain(double* out, long long* cyc, double a, double b) {
  double x[CHAINS];
  for (int i = 0; i < CHAINS; ++i) x[i] = a + i + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) x[i] = fma(x[i], b, a);
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void mix_chain(double* out, long long* cyc, double a, double b) {
  // one dependent fp64 chain with independent SALU + int VALU filler between links
  double x = a + threadIdx.x; int k = threadIdx.x; 
  long long t0 = clock64();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) { x = fma(x, b, a); k = k * 3 + 1; }
  }
  long long t1 = clock64();
  out[threadIdx.x] = x + k;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void add_chain(double* out, long long* cyc, double a) {
  double x = a + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) x = x + a;
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void f32_chain(float* out, long long* cyc, float a, float b) {
  float x = a + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) x = fmaf(x, b, a);
  }
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void lds_chain(double* out, long long* cyc) {
  __shared__ int idx[256];
  for (int i = threadIdx.x; i < 256; i += 64) idx[i] = (i + 1) & 255;
  __syncthreads();
  int k = 0;
  long long t0 = clock64();
  for (int it = 0; it < 64 * 16; ++it) k = idx[k];   // uniform address, dependent
  long long t1 = clock64();
  out[threadIdx.x] = k;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  long long h;
  auto rep = [&](const char* name, double per) { hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.2f cycles/op\n", name, (double)h / per); };
  for (int w = 0; w < 2; ++w) {
    fma_chain<1><<<1, 64>>>(out, cyc, 1.0, 0.999); hipDeviceSynchronize(); rep("fma_f64 dependent (1 chain)", 64.0 * REP);
    fma_chain<2><<<1, 64>>>(out, cyc, 1.0, 0.999); hipDeviceSynchronize(); rep("fma_f64 2 independent chains", 64.0 * REP);
    fma_chain<4><<<1, 64>>>(out, cyc, 1.0, 0.999); hipDeviceSynchronize(); rep("fma_f64 4 independent chains", 64.0 * REP);
    fma_chain<8><<<1, 64>>>(out, cyc, 1.0, 0.999); hipDeviceSynchronize(); rep("fma_f64 8 independent chains", 64.0 * REP);
    add_chain<<<1, 64>>>(out, cyc, 1.0); hipDeviceSynchronize(); rep("add_f64 dependent", 64.0 * REP);
    mix_chain<<<1, 64>>>(out, cyc, 1.0, 0.999); hipDeviceSynchronize(); rep("fma_f64 dep + int filler (per pair)", 64.0 * REP);
    f32_chain<<<1, 64>>>((float*)out, cyc, 1.0f, 0.999f); hipDeviceSynchronize(); rep("fma_f32 dependent", 64.0 * REP);
    lds_chain<<<1, 64>>>(out, cyc); hipDeviceSynchronize(); rep("ds_read_b32 dependent (uniform addr)", 64.0 * 16);
  }
  return 0;
}
