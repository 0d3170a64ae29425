// Launch templates of the wave-per-problem kernels (ilqr_small.hpp); included by the k_<model>.hip units.
#pragma once
#include "host.hpp"
#include "ilqr_small.hpp"

namespace mi_host {
template <class M, int JAC, int MODE>
int launch_one(mi_ilqr* h, const KArgs& a) {
  auto kern = ilqr_small_kernel<M, JAC, MODE>;
  static bool lds_ok[kMaxDevices] = {};
  { const int rc = allow_max_lds(kern, lds_ok, h->d.device_id); if (rc != MI_ILQR_OK) return rc; }
  const int waves = (a.helpers > 0 && (MODE == MODE_SOLVE || MODE == MODE_MPC)) ? 1 + a.helpers : 1;
  return launch_timed(h, kern, dim3(h->B), dim3(64 * waves), h->lds, a);
}

template <class M, int JAC>
int launch_mode(mi_ilqr* h, int mode, const KArgs& a) {
  switch (mode) {
    case MODE_SOLVE: return launch_one<M, JAC, MODE_SOLVE>(h, a);
    case MODE_ROLLOUT: return launch_one<M, JAC, MODE_ROLLOUT>(h, a);
    case MODE_FORWARD: return launch_one<M, JAC, MODE_FORWARD>(h, a);
    case MODE_LINEARIZE: return launch_one<M, JAC, MODE_LINEARIZE>(h, a);
    case MODE_BACKWARD: return launch_one<M, JAC, MODE_BACKWARD>(h, a);
    case MODE_MPC: return launch_one<M, JAC, MODE_MPC>(h, a);
  }
  return MI_ILQR_E_BAD_ARG;
}

template <class M>
int launch_jac(mi_ilqr* h, int mode, const KArgs& a) {
  if constexpr (M::n >= 3 && M::n <= 4 && M::m == 1) {
    // cost matrices outside the symmetric-PSD class: the kernels whose backward pass is the reference's recursion
    if (h->exact_backward && (mode == MODE_SOLVE || mode == MODE_MPC || mode == MODE_BACKWARD)) {
      using E = ExactCost<M>;
      const bool ad = h->d.jacobian_mode == MI_JAC_AUTODIFF;
      switch (mode) {
        case MODE_SOLVE: return ad ? launch_one<E, MI_JAC_AUTODIFF, MODE_SOLVE>(h, a) : launch_one<E, MI_JAC_FD_CENTRAL, MODE_SOLVE>(h, a);
        case MODE_MPC: return ad ? launch_one<E, MI_JAC_AUTODIFF, MODE_MPC>(h, a) : launch_one<E, MI_JAC_FD_CENTRAL, MODE_MPC>(h, a);
        default: return launch_one<E, MI_JAC_FD_CENTRAL, MODE_BACKWARD>(h, a);
      }
    }
    // two or more steps per lane: the kernels whose backward pass is the time-parallel scan (ilqr_small.hpp:
    // LongHorizon) - only the modes that run a backward pass have such an instantiation
    if (h->N > 128 && (mode == MODE_SOLVE || mode == MODE_MPC || mode == MODE_BACKWARD)) {
      using L = LongHorizon<M>;
      const bool ad = h->d.jacobian_mode == MI_JAC_AUTODIFF;
      switch (mode) {
        case MODE_SOLVE: return ad ? launch_one<L, MI_JAC_AUTODIFF, MODE_SOLVE>(h, a) : launch_one<L, MI_JAC_FD_CENTRAL, MODE_SOLVE>(h, a);
        case MODE_MPC: return ad ? launch_one<L, MI_JAC_AUTODIFF, MODE_MPC>(h, a) : launch_one<L, MI_JAC_FD_CENTRAL, MODE_MPC>(h, a);
        default: return launch_one<L, MI_JAC_FD_CENTRAL, MODE_BACKWARD>(h, a);
      }
    }
  }
  if (h->d.jacobian_mode == MI_JAC_AUTODIFF) return launch_mode<M, MI_JAC_AUTODIFF>(h, mode, a);
  return launch_mode<M, MI_JAC_FD_CENTRAL>(h, mode, a);
}

}  // namespace mi_host
