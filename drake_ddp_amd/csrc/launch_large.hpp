// Launch templates of the workgroup-per-problem kernels (ilqr_large.hpp); included by the k_<model>.hip units.
#pragma once
#include "host.hpp"
#include "ilqr_large.hpp"

namespace mi_host {
template <class M, int JAC, int MODE>
int launch_one_large(mi_ilqr* h, const KArgs& a) {
  auto kern = ilqr_large_kernel<M, JAC, MODE>;
  static bool lds_ok[kMaxDevices] = {};
  { const int rc = allow_max_lds(kern, lds_ok, h->d.device_id); if (rc != MI_ILQR_OK) return rc; }
  const int cluster = (MODE == MODE_SOLVE || MODE == MODE_MPC) ? a.cluster : 1;
  if (cluster > 1) HIPCHK(hipMemsetAsync(h->cluster_sync, 0, (size_t)h->B * 4 * sizeof(unsigned long long), h->stream));
  return launch_timed(h, kern, dim3(h->B * cluster), dim3(kLargeThreads), h->lds, a);
}

template <class M, int JAC>
int launch_mode_large(mi_ilqr* h, int mode, const KArgs& a) {
  switch (mode) {
    case MODE_SOLVE: return launch_one_large<M, JAC, MODE_SOLVE>(h, a);
    case MODE_ROLLOUT: return launch_one_large<M, JAC, MODE_ROLLOUT>(h, a);
    case MODE_FORWARD: return launch_one_large<M, JAC, MODE_FORWARD>(h, a);
    case MODE_LINEARIZE: return launch_one_large<M, JAC, MODE_LINEARIZE>(h, a);
    case MODE_BACKWARD: return launch_one_large<M, JAC, MODE_BACKWARD>(h, a);
    case MODE_MPC: return launch_one_large<M, JAC, MODE_MPC>(h, a);
  }
  return MI_ILQR_E_BAD_ARG;
}

template <class M>
int launch_jac_large(mi_ilqr* h, int mode, const KArgs& a) {
  if (h->d.jacobian_mode == MI_JAC_AUTODIFF) return launch_mode_large<M, MI_JAC_AUTODIFF>(h, mode, a);
  return launch_mode_large<M, MI_JAC_FD_CENTRAL>(h, mode, a);
}

}  // namespace mi_host
