// Lane layout and rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 (four independent 4x4x4 products per instruction).
// Probe: A = 1 in one lane, B = 1 in one lane -> which output lanes become 1.  Prints, per output lane, the
// (A lane, B lane) pairs that feed it, and cycles per instruction (dependent chain / 4 independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {          // out[la * 64 + lb] = bitmask-free: first output lane hit (or -1), count in +4096
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (l == 0) { out[la * 64 + lb] = m ? __ffsll((long long)m) - 1 : -1; out[4096 + la * 64 + lb] = __popcll(m); }
    }
}
__global__ void rate(long long* cyc, double* sink) {
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5;
  double x = 0.0;
  long long t0 = clock64();
  for (int i = 0; i < 256; ++i) x = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x, 0, 0, 0);
  long long t1 = clock64();
  double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
  for (int i = 0; i < 64; ++i) {
    y0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, y0, 0, 0, 0);
    y1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, y1, 0, 0, 0);
    y2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, y2, 0, 0, 0);
    y3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, y3, 0, 0, 0);
  }
  long long t2 = clock64();
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  sink[threadIdx.x] = x + y0 + y1 + y2 + y3;
}
int main() {
  int* d; hipMalloc(&d, 8192 * 4);
  probe<<<1, 64>>>(d); hipDeviceSynchronize();
  static int h[8192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int ld = 0; ld < 64; ++ld) {
    printf("D lane %2d <-", ld);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] == ld) printf(" (A%d,B%d)%s", la, lb, h[4096 + la * 64 + lb] > 1 ? "*" : "");
    printf("\n");
  }
  long long* dc; double* ds; long long hc[2]; hipMalloc(&dc, 16); hipMalloc(&ds, 512);
  rate<<<1, 64>>>(dc, ds); hipDeviceSynchronize(); hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
  printf("v_mfma_f64_4x4x4_4b: dependent chain %.1f cycles/instr ; 4 independent accumulators %.1f cycles/instr\n", hc[0] / 256.0, hc[1] / 256.0);
  return 0;
}
