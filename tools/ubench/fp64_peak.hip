// The fp64 peak of THIS chip, measured: what bench.py divides by (profiles/rNN_fp64_peak.json), instead of a datasheet figure.
// (a) VALU: every SIMD of every CU runs W waves of 8 independent v_fma_f64 chains; (b) matrix cores: the same with
// v_mfma_f64_16x16x4_f64 (2048 flops per wave instruction).  Rate = flops / kernel time from HIP events; the shader clock the
// waves ran at comes from s_memtime / s_memrealtime (100 MHz) inside the same launch.
// (-amdgpu-mfma-vgpr-form=1, the library's own flag: without it this hipcc carries the four accumulators through 64 v_accvgpr moves
//  per loop trip and the matrix row reads 49 TF instead of the pipe's rate - the first r06 file had that)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/fp64_peak.hip -o tools/ubench/fp64_peak && tools/ubench/fp64_peak [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) valu(double* sink, long long* clk, int iters, double y) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    x0 = __builtin_fma(x0, y, 1e-9); x1 = __builtin_fma(x1, y, 1e-9); x2 = __builtin_fma(x2, y, 1e-9); x3 = __builtin_fma(x3, y, 1e-9);
    x4 = __builtin_fma(x4, y, 1e-9); x5 = __builtin_fma(x5, y, 1e-9); x6 = __builtin_fma(x6, y, 1e-9); x7 = __builtin_fma(x7, y, 1e-9);
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  const double s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void __launch_bounds__(256) mfma(double* sink, long long* clk, int iters, double a, double b) {
  d4 y0 = {a, b, a, b}, y1 = y0, y2 = y0, y3 = y0;
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y0, 0, 0, 0);
    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y1, 0, 0, 0);
    y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y2, 0, 0, 0);
    y3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, y3, 0, 0, 0);
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  const double s = y0[0] + y1[1] + y2[2] + y3[3];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* sink; long long* clk;
  hipMalloc(&sink, 8); hipMalloc(&clk, 2 * 8 * cus * 8 * sizeof(long long));   // (room for 8 workgroups per CU)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best_valu = 0, best_mfma = 0, mhz_valu = 0, mhz_mfma = 0; int wv = 0, wm = 0;
  for (int kind = 0; kind < 2; ++kind)
    for (int wgs_per_cu : {1, 2, 4, 8}) {              // 256 threads = one wave per SIMD per workgroup
      const int blocks = cus * wgs_per_cu, iters = kind == 0 ? 40000 : 6000;
      float ms = 0;
      for (int rep = 0; rep < 4; ++rep) {           // the last repetition counts: the clock has ramped by then
        hipEventRecord(e0);
        if (kind == 0) valu<<<blocks, 256>>>(sink, clk, iters, 1.0000001);
        else mfma<<<blocks, 256>>>(sink, clk, iters, 1e-3, 1e-3);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<long long> h(2 * blocks);
      hipMemcpy(h.data(), clk, 2 * blocks * sizeof(long long), hipMemcpyDeviceToHost);
      double mhz = 0; for (int b = 0; b < blocks; ++b) mhz += double(h[2 * b]) / double(h[2 * b + 1]) * 100.0; mhz /= blocks;
      const double flops = kind == 0 ? double(blocks) * 256 * iters * 8 * 2 : double(blocks) * 4 * iters * 4 * 2048.0;
      const double tf = flops / (ms * 1e-3) / 1e12;
      printf("%s  %d wave(s) per SIMD  %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz  -> %.2f flop / clk / SIMD\n", kind == 0 ? "v_fma_f64          " : "v_mfma_f64_16x16x4",
             wgs_per_cu, ms, tf, mhz, tf * 1e12 / (mhz * 1e6) / (cus * 4));
      if (kind == 0 && tf > best_valu) { best_valu = tf; mhz_valu = mhz; wv = wgs_per_cu; }
      if (kind == 1 && tf > best_mfma) { best_mfma = tf; mhz_mfma = mhz; wm = wgs_per_cu; }
    }
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    fprintf(f, "{\"source\": \"tools/ubench/fp64_peak.hip on %s (%d CUs)\", \"valu_fp64_TFLOPs\": %.3f, \"valu_waves_per_simd\": %d, \"valu_clock_MHz\": %.0f, "
               "\"mfma_fp64_TFLOPs\": %.3f, \"mfma_waves_per_simd\": %d, \"mfma_clock_MHz\": %.0f, \"fp64_peak_TFLOPs\": %.3f, "
               "\"note\": \"measured dense fp64 rate, all SIMDs busy, kernel time from HIP events; fp64_peak = max(valu, mfma)\"}\n",
            p.gcnArchName, cus, best_valu, wv, mhz_valu, best_mfma, wm, mhz_mfma, std::max(best_valu, best_mfma));
    fclose(f);
  }
  return 0;
}
