// Which hand-off forms between two workgroups of ONE launch deliver fresh data on gfx950 - the question behind the cluster
// hand-shake of csrc/ilqr_large.hpp (round-5 review, item 2a).  A consumer workgroup keeps a 4 KB block L1-WARM (it re-reads
// it with plain loads every round), a producer workgroup on (a) the same XCD, (b) another XCD rewrites the block and raises a
// flag; the consumer then reads the block again and counts the words that still hold the previous round's value.
//   producer store : plain | sc1 (what __hip_atomic_store(relaxed, agent) lowers to)
//   producer publish: s_waitcnt vmcnt(0) only | + agent-scope release fence (buffer_wbl2 sc1)
//   consumer acquire: none | buffer_inv sc0 | agent-scope acquire fence (buffer_inv sc1)
//   consumer load  : plain | sc1 (what __hip_atomic_load(relaxed, agent) lowers to)
// Placement is verified from HW_REG_XCC_ID / HW_REG_HW_ID, not assumed.  Every load / store of the payload is inline assembly so that
// the cache bits are exactly the ones named.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l1_probe.hip -o tools/ubench/l1_probe && tools/ubench/l1_probe [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>

constexpr int kWordsPerLane = 8, kLanes = 64, kWords = kWordsPerLane * kLanes;   // 4 KB
constexpr int kRounds = 200;

__device__ __forceinline__ unsigned long long ld_plain(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_plain(unsigned long long* p, unsigned long long v) {
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

// mode bits: [0] store sc1, [1] producer release fence, [3:2] consumer acquire (0 none, 1 buffer_inv sc0, 2 agent acquire fence), [4] load sc1
__global__ void __launch_bounds__(64) probe(unsigned long long* data, unsigned long long* flags, unsigned* where, unsigned long long* stale_out,
                                            int consumer_block, int producer_block, int mode) {
  extern __shared__ double occupy[];     // 100 KB per workgroup: one workgroup per CU, so producer and consumer never share an L1
  if (mode == 12345) occupy[threadIdx.x] = 1.0;
  const int lane = threadIdx.x;
  const bool is_c = (int)blockIdx.x == consumer_block, is_p = (int)blockIdx.x == producer_block;
  if (!is_c && !is_p) return;
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (lane == 0) { where[is_c ? 0 : 2] = xcc & 15u; where[is_c ? 1 : 3] = (hwid >> 8) & 0xffu; }   // cu_id[11:8] sh[12] se[15:13]
  unsigned long long* fa = flags;        // consumer -> producer: "round r may be written"
  unsigned long long* fb = flags + 16;   // producer -> consumer: "round r is written"
  unsigned long long stale = 0;
  for (int r = 1; r <= kRounds; ++r) {
    if (is_c) {
      // keep the block L1-warm: plain loads of every word (values of round r - 1)
      unsigned long long s = 0;
      for (int w = 0; w < kWordsPerLane; ++w) s += ld_plain(data + w * kLanes + lane);
      if (s == 0x123456789ull) stale_out[4] = s;
      __builtin_amdgcn_s_barrier();
      if (lane == 0) {
        __hip_atomic_store(fa, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (long long sp = 0; sp < (1ll << 22) && __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r; ++sp) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_s_barrier();
      const int acq = (mode >> 2) & 3;
      if (acq == 1) asm volatile("buffer_inv sc0" ::: "memory");
      else if (acq == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int w = 0; w < kWordsPerLane; ++w) {
        const unsigned long long* p = data + w * kLanes + lane;
        const unsigned long long v = (mode & 16) ? ld_sc1(p) : ld_plain(p);
        stale += (v != (unsigned long long)r) ? 1 : 0;
      }
    } else {
      if (lane == 0) for (long long sp = 0; sp < (1ll << 22) && __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)r; ++sp) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_s_barrier();
      for (int w = 0; w < kWordsPerLane; ++w) {
        unsigned long long* p = data + w * kLanes + lane;
        if (mode & 1) st_sc1(p, (unsigned long long)r); else st_plain(p, (unsigned long long)r);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (mode & 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
      if (lane == 0) __hip_atomic_store(fb, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (is_c) {
    for (int o = 32; o > 0; o >>= 1) stale += __shfl_xor(stale, o);
    if (lane == 0) stale_out[0] = stale;
  }
}

int main(int argc, char** argv) {
  unsigned long long *data, *flags, *stale; unsigned* where;
  hipMalloc(&data, kWords * 8); hipMalloc(&flags, 32 * 8); hipMalloc(&stale, 64); hipMalloc(&where, 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  std::string json = "{\"source\": \"tools/ubench/l1_probe.hip\", \"block_bytes\": 4096, \"rounds\": " + std::to_string(kRounds) + ", \"cases\": [";
  bool first = true;
  const char* st_name[2] = {"plain", "sc1"};
  const char* rel_name[2] = {"vmcnt(0)", "vmcnt(0) + agent release fence"};
  const char* acq_name[3] = {"none", "buffer_inv sc0", "agent acquire fence (buffer_inv sc1)"};
  const char* ld_name[2] = {"plain", "sc1"};
  for (int place = 0; place < 2; ++place) {
    const int cblk = 0, pblk = place == 0 ? 8 : 1;        // the dispatcher deals blocks to XCDs round-robin: 0 and 8 share one, 0 and 1 do not (checked below)
    for (int mode = 0; mode < 32; ++mode) {
      if (((mode >> 2) & 3) == 3) continue;
      hipMemset(data, 0, kWords * 8); hipMemset(flags, 0, 32 * 8); hipMemset(stale, 0, 64); hipMemset(where, 0xff, 16);
      probe<<<16, 64, 100 * 1024>>>(data, flags, where, stale, cblk, pblk, mode);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      unsigned long long h = 0; unsigned w[4];
      hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost); hipMemcpy(w, where, 16, hipMemcpyDeviceToHost);
      const double frac = double(h) / double(kRounds) / double(kWords);
      printf("%s  store %-5s  publish %-31s  acquire %-36s  load %-5s : stale %.4f   (consumer xcc %u cu %02x, producer xcc %u cu %02x)\n",
             w[0] == w[2] ? "same XCD " : "other XCD", st_name[mode & 1], rel_name[(mode >> 1) & 1], acq_name[(mode >> 2) & 3], ld_name[(mode >> 4) & 1], frac, w[0], w[1], w[2], w[3]);
      char buf[512];
      snprintf(buf, sizeof buf, "%s{\"same_xcd\": %s, \"same_cu\": %s, \"store\": \"%s\", \"publish\": \"%s\", \"acquire\": \"%s\", \"load\": \"%s\", \"stale_fraction\": %.6f}",
               first ? "" : ", ", w[0] == w[2] ? "true" : "false", (w[0] == w[2] && w[1] == w[3]) ? "true" : "false", st_name[mode & 1], rel_name[(mode >> 1) & 1], acq_name[(mode >> 2) & 3],
               ld_name[(mode >> 4) & 1], frac);
      json += buf; first = false;
    }
  }
  json += "]}\n";
  if (argc > 1) { FILE* f = fopen(argv[1], "w"); fputs(json.c_str(), f); fclose(f); }
  return 0;
}
