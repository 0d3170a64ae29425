// Does a 64-bit-encoded instruction that starts on an odd dword cost one wave an extra cycle in MIXED code too?
// Loop bodies of 64 groups, first instruction 256-byte aligned; B4 = v_add_f32_e32 (4 bytes), B8 = the same in its e64 encoding,
// A8 = v_fma_f64 (8 bytes), N4 = s_nop 0.  Cycles per group from s_memtime (the back edge, ~30 cycles per trip, included).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/instr_align.hip -o tools/ubench/instr_align && tools/ubench/instr_align
#include <hip/hip_runtime.h>
#include <cstdio>

#define A8(i) "v_fma_f64 %[x" #i "], %[x" #i "], %[y], %[x" #i "]\n"
#define B4(i) "v_add_f32_e32 %[f" #i "], %[g], %[f" #i "]\n"
#define B8(i) "v_add_f32_e64 %[f" #i "], %[g], %[f" #i "]\n"
#define N4 "s_nop 0\n"

#define KERNEL(NAME, BODY)                                                                                                        \
  __global__ void __launch_bounds__(64) NAME(double* sink, long long* clk, int iters, double y, float g) {                       \
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;                                                       \
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                                                               \
    long long c0, c1;                                                                                                             \
    asm volatile("s_memtime %[c0]\n s_waitcnt lgkmcnt(0)\n s_branch 1f\n .p2align 8\n 1:\n .rept 16\n" BODY ".endr\n"            \
                 "s_sub_u32 %[it], %[it], 1\n s_cmp_lg_u32 %[it], 0\n s_cbranch_scc1 1b\n s_memtime %[c1]\n s_waitcnt lgkmcnt(0)\n" \
                 : [c0] "=&s"(c0), [c1] "=&s"(c1), [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [f0] "+v"(f0),     \
                   [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [it] "+s"(iters)                                                  \
                 : [y] "v"(y), [g] "v"(g)                                                                                         \
                 : "scc", "memory");                                                                                              \
    const double s = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3;                                                                       \
    if (s == 12345.678) sink[0] = s;                                                                                              \
    if (threadIdx.x == 0) clk[0] = c1 - c0;                                                                                       \
  }

// every BODY holds four groups (x0..x3 / f0..f3), repeated 16 times: 64 groups per trip
KERNEL(k_a8, A8(0) A8(1) A8(2) A8(3))                                                       // 8-byte only, aligned
KERNEL(k_n_a8, N4 A8(0) A8(1) A8(2) A8(3) N4)                                               // the same, every A8 on an odd dword (+ 2 s_nop per 4 groups)
KERNEL(k_b4b4a8, B4(0) B4(0) A8(0) B4(1) B4(1) A8(1) B4(2) B4(2) A8(2) B4(3) B4(3) A8(3))   // A8 always aligned
KERNEL(k_b4a8b4, B4(0) A8(0) B4(0) B4(1) A8(1) B4(1) B4(2) A8(2) B4(2) B4(3) A8(3) B4(3))   // A8 always on an odd dword
KERNEL(k_b4a8, B4(0) A8(0) B4(1) A8(1) B4(2) A8(2) B4(3) A8(3))                             // A8 alternately odd / even
KERNEL(k_b8a8, B8(0) A8(0) B8(1) A8(1) B8(2) A8(2) B8(3) A8(3))                             // the 4-byte instruction widened: all aligned
// runs of k 8-byte instructions between two 4-byte ones: odd (B4 A8^k B4) against even (B4 B4 A8^k) start, period 2 + 2k dwords
KERNEL(k_r2o, B4(0) A8(0) A8(1) B4(1) B4(2) A8(2) A8(3) B4(3))
KERNEL(k_r2e, B4(0) B4(1) A8(0) A8(1) B4(2) B4(3) A8(2) A8(3))
KERNEL(k_r3o, B4(0) A8(0) A8(1) A8(2) B4(1) B4(2) A8(3) A8(0) A8(1) B4(3))
KERNEL(k_r3e, B4(0) B4(1) A8(0) A8(1) A8(2) B4(2) B4(3) A8(3) A8(0) A8(1))
KERNEL(k_r4o, B4(0) A8(0) A8(1) A8(2) A8(3) B4(1) B4(2) A8(0) A8(1) A8(2) A8(3) B4(3))
KERNEL(k_r4e, B4(0) B4(1) A8(0) A8(1) A8(2) A8(3) B4(2) B4(3) A8(0) A8(1) A8(2) A8(3))
KERNEL(k_r8o, B4(0) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3) B4(1))
KERNEL(k_r8e, B4(0) B4(1) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3))
KERNEL(k_r16o, B4(0) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3) B4(1))
KERNEL(k_r16e, B4(0) B4(1) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3) A8(0) A8(1) A8(2) A8(3))
KERNEL(k_b4, B4(0) B4(1) B4(2) B4(3))
KERNEL(k_b8, B8(0) B8(1) B8(2) B8(3))

template <class K>
void run(const char* what, K kern, int per_group, double* sink, long long* clk) {
  const int iters = 20000;
  long long h = 0;
  for (int rep = 0; rep < 2; ++rep) { kern<<<1, 64>>>(sink, clk, iters, 1e-9, 1e-9f); hipDeviceSynchronize(); }
  hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%-64s %7.2f cycles per %s of %d instruction(s)\n", what, double(h) / iters / (per_group > 3 ? 16 : 64), per_group > 3 ? "body" : "group", per_group);
}

int main() {
  double* sink; long long* clk;
  hipMalloc(&sink, 8); hipMalloc(&clk, 8);
  run("A8                    (fp64 fma, aligned)", k_a8, 1, sink, clk);
  run("A8 on odd dwords      (+ 2 s_nop per 4)", k_n_a8, 1, sink, clk);
  run("B4                    (v_add_f32_e32)", k_b4, 1, sink, clk);
  run("B8                    (v_add_f32_e64)", k_b8, 1, sink, clk);
  run("B4 B4 A8              (A8 aligned)", k_b4b4a8, 3, sink, clk);
  run("B4 A8 B4              (A8 on an odd dword)", k_b4a8b4, 3, sink, clk);
  run("B4 A8                 (A8 alternately odd / even)", k_b4a8, 2, sink, clk);
  run("B8 A8                 (widened: A8 aligned)", k_b8a8, 2, sink, clk);
  printf("runs of k 8-byte instructions between 4-byte ones, cycles per body (x 16 per trip): starting on an odd dword / on an even one\n");
  run("k = 2 (two runs + 4 B4 per body)   odd", k_r2o, 8, sink, clk);  run("k = 2                              even", k_r2e, 8, sink, clk);
  run("k = 3 (two runs + 4 B4 per body)   odd", k_r3o, 10, sink, clk); run("k = 3                              even", k_r3e, 10, sink, clk);
  run("k = 4 (two runs + 4 B4 per body)   odd", k_r4o, 12, sink, clk); run("k = 4                              even", k_r4e, 12, sink, clk);
  run("k = 8 (one run + 2 B4 per body)    odd", k_r8o, 10, sink, clk); run("k = 8                              even", k_r8e, 10, sink, clk);
  run("k = 16 (one run + 2 B4 per body)   odd", k_r16o, 18, sink, clk); run("k = 16                             even", k_r16e, 18, sink, clk);
  return 0;
}
