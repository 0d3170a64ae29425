// Launch templates of the workgroup-per-problem kernels (ilqr_large.hpp); included by the k_<model>.hip units.
#pragma once
#include "host.hpp"
#include "ilqr_large.hpp"

namespace mi_host {
template <class M, int JAC, int MODE, bool PIV = false>
int launch_one_large(mi_ilqr* h, const KArgs& a) {
  auto kern = ilqr_large_kernel<M, JAC, MODE, PIV>;
  static bool lds_ok[kMaxDevices] = {};
  { const int rc = allow_max_lds(kern, lds_ok, h->d.device_id); if (rc != MI_ILQR_OK) return rc; }
  const int cluster = (MODE == MODE_SOLVE || MODE == MODE_MPC) ? (a.cluster & 0xff) : 1;
  if (MODE == MODE_SOLVE || MODE == MODE_MPC) h->last_clustered = cluster > 1;
  if (cluster > 1) HIPCHK(hipMemsetAsync(h->cluster_sync, 0, (size_t)h->B * kSyncWords * sizeof(unsigned long long), h->stream));
  // clusters: 8 G ceil(B / 8) workgroups, a cluster's members congruent mod 8 = on one XCD (ilqr_large_kernel: XCD-aware placement)
  const unsigned grid = cluster > 1 ? (((a.cluster >> 8) & 3) ? 8u * (unsigned)cluster * (unsigned)((h->B + 7) / 8) : (unsigned)(h->B * cluster)) : (unsigned)h->B;
  return launch_timed(h, kern, dim3(grid), dim3(kLargeThreads), h->lds, a);
}

// Which models get BOTH forms of the kernels with a backward pass (ilqr_large.hpp: PIV) - with and without the pivoted-inverse cold
// path: those that declare kPivSplit (models.hpp: Synth36, Quad3D, PlanarQuad).  Same-box A/B, cycles per iteration of the bench's
// MPC loops without | with the ~150 cold instructions in the kernel: 36-state chain 429 k | 451 k (its LINE SEARCH 74 k | 85 k - the
// register allocation of the phases around the backward pass, not the pass), 3-D quadruped 682 k | 695 k, planar quadruped + 2.5 %
// it/s without.  The default (on_indefinite = 0, symmetric costs) is what the benchmarks run, so these get the lean form for it.
// Every other model keeps the ONE form that carries the path: the arm + ball is no faster without it (130 k | 134 k cycles per
// trial the other way round), plugin models would pay twice the kernels per build - the (34, 12) chain's lean form does not even
// compile with this hipcc ("Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base").
// The planar quadruped's lean MPC kernel was MISCOMPILED when it was first built (round 5, first session: another x_1 from bitwise
// the same x_0, u_0, non-deterministically; eight tests of the GPU suite caught it) and left out; the round's last session found
// the mechanism of this hipcc's miscompiles of these kernels (a VGPR spill copy placed in a zero-EXEC block prologue: DESIGN
// section 8, tools/check_exec_spill.py), rebuilt the form on the current sources - the checker finds no such site in it, the
// 50 planar-quadruped tests of the GPU suite pass on it - and adopted it.
template <class M, class = void>
struct HasPivSplit { static constexpr bool value = false; };
template <class M>
struct HasPivSplit<M, decltype((void)M::kPivSplit)> { static constexpr bool value = M::kPivSplit; };
template <class M>
constexpr bool kPivSplit = HasPivSplit<M>::value;

template <class M, int JAC>
int launch_mode_large(mi_ilqr* h, int mode, const KArgs& a) {
  if (a.pd_continue || a.cost_asym || !kPivSplit<M>) {
    switch (mode) {
      case MODE_SOLVE: return launch_one_large<M, JAC, MODE_SOLVE, true>(h, a);
      case MODE_BACKWARD: return launch_one_large<M, JAC, MODE_BACKWARD, true>(h, a);
      case MODE_MPC: return launch_one_large<M, JAC, MODE_MPC, true>(h, a);
      default: break;
    }
  }
  if constexpr (kPivSplit<M>) {
    switch (mode) {
      case MODE_SOLVE: return launch_one_large<M, JAC, MODE_SOLVE>(h, a);
      case MODE_BACKWARD: return launch_one_large<M, JAC, MODE_BACKWARD>(h, a);
      case MODE_MPC: return launch_one_large<M, JAC, MODE_MPC>(h, a);
      default: break;
    }
  }
  switch (mode) {
    case MODE_ROLLOUT: return launch_one_large<M, JAC, MODE_ROLLOUT>(h, a);
    case MODE_FORWARD: return launch_one_large<M, JAC, MODE_FORWARD>(h, a);
    case MODE_LINEARIZE: return launch_one_large<M, JAC, MODE_LINEARIZE>(h, a);
    default: break;
  }
  return MI_ILQR_E_BAD_ARG;
}

template <class M>
int launch_jac_large(mi_ilqr* h, int mode, const KArgs& a) {
  if (h->d.jacobian_mode == MI_JAC_AUTODIFF) return launch_mode_large<M, MI_JAC_AUTODIFF>(h, mode, a);
  return launch_mode_large<M, MI_JAC_FD_CENTRAL>(h, mode, a);
}

}  // namespace mi_host
