// Does a loop body of S bytes of straight-line code cost more per iteration than its instructions?
// (instruction cache / instruction TLB reach of one wave on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ void k(long long* cyc, int iters) {
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    // KB * 256 four-byte instructions
    asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(KB * 256));
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KB>
void run(long long* cyc, int blocks) {
  long long h;
  k<KB><<<blocks, 64>>>(cyc, 4); hipDeviceSynchronize();
  k<KB><<<blocks, 64>>>(cyc, 64); hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("blocks %4d body %3d KB: %.0f cycles/iteration = %.2f cycles per instruction\n", blocks, KB, (double)h / 64, (double)h / 64 / (KB * 256));
}
int main() {
  long long* cyc; hipMalloc(&cyc, 8);
  for (int blocks : {1, 256}) {
    run<1>(cyc, blocks); run<2>(cyc, blocks); run<4>(cyc, blocks); run<8>(cyc, blocks); run<12>(cyc, blocks); run<16>(cyc, blocks); run<24>(cyc, blocks);
    run<32>(cyc, blocks); run<48>(cyc, blocks); run<64>(cyc, blocks); 
  }
  return 0;
}
