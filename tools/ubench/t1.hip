// Micro-benchmark of the large-path T1 = Vxx F phase (256 threads, LDS operands).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int n = 36, nm = 48, TS = 52, RG1 = 9;
template <int VAR>
__global__ void __launch_bounds__(256) k(double* out, long long* cyc) {
  __shared__ double Vxx[n * n], F[n * nm], T1[n * TS];
  const int tid = threadIdx.x;
  for (int e = tid; e < n * n; e += 256) Vxx[e] = 1.0 / (1 + e);
  for (int e = tid; e < n * nm; e += 256) F[e] = 0.5 + e * 1e-3;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < 20; ++it) {
    if (tid < 4 * nm) {
      const int ig = tid / nm, j = tid - ig * nm;
      double col[n];
#pragma unroll
      for (int kk = 0; kk < n; ++kk) col[kk] = F[kk * nm + j];
      if (VAR == 0) {
#pragma unroll 3
        for (int r = 0; r < RG1; ++r) {
          const int i = ig * RG1 + r;
          const double2* vr = reinterpret_cast<const double2*>(Vxx + i * n);
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < n / 2; ++kk) { const double2 q = vr[kk]; s0 += q.x * col[2 * kk]; s1 += q.y * col[2 * kk + 1]; }
          T1[i * TS + j] = s0 + s1;
        }
      } else if (VAR == 1) {
        for (int r = 0; r < RG1; ++r) {
          const int i = ig * RG1 + r;
          const double2* vr = reinterpret_cast<const double2*>(Vxx + i * n);
          double2 q[n / 2];
#pragma unroll
          for (int kk = 0; kk < n / 2; ++kk) q[kk] = vr[kk];
          __builtin_amdgcn_sched_barrier(0);
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int kk = 0; kk < n / 2; ++kk) { s0 += q[kk].x * col[2 * kk]; s1 += q[kk].y * col[2 * kk + 1]; }
          T1[i * TS + j] = s0 + s1;
        }
      } else {
        // all 9 rows' accumulators live; k outer: each F-independent Vxx element pair read once per row
        double acc[RG1];
#pragma unroll
        for (int r = 0; r < RG1; ++r) acc[r] = 0.0;
#pragma unroll
        for (int kk = 0; kk < n / 2; ++kk) {
#pragma unroll
          for (int r = 0; r < RG1; ++r) {
            const double2 q = reinterpret_cast<const double2*>(Vxx + (ig * RG1 + r) * n)[kk];
            acc[r] += q.x * col[2 * kk];
            acc[r] += q.y * col[2 * kk + 1];
          }
        }
#pragma unroll
        for (int r = 0; r < RG1; ++r) T1[(ig * RG1 + r) * TS + j] = acc[r];
      }
    }
    __syncthreads();
  }
  long long t1 = clock64();
  if (tid == 0) cyc[0] = (t1 - t0) / 20;
  out[tid] = T1[tid];
}
int main() {
  double* out; long long* cyc, h; hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 8);
  k<0><<<1, 256>>>(out, cyc); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("variant 0 (as in kernel): %lld cycles\n", h);
  k<1><<<1, 256>>>(out, cyc); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("variant 1 (batched row loads): %lld cycles\n", h);
  k<2><<<1, 256>>>(out, cyc); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("variant 2 (k-outer, 9 accumulators): %lld cycles\n", h);
  return 0;
}
