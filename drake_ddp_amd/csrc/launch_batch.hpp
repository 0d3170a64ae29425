// Launch templates of the lane-per-problem ("throughput") kernels (ilqr_batch.hpp); included by k_batch.hip (built-in
// small-state models) and by the family-0 plugin units (drake_ddp_amd/plugin.py).
#pragma once
#include "host.hpp"
#include "ilqr_batch.hpp"

namespace mi_host {
template <class M, int JAC>
int launch_batch_one(mi_ilqr* h, const KArgs& a) {
  if (a.bm_scratch != nullptr) {                          // key-point configurations other than setInterval / 1
    auto kern = ilqr_batch_kernel<M, JAC, true>;
    return launch_timed(h, kern, dim3((h->B + 63) / 64), dim3(64), 0, a);
  }
  auto kern = ilqr_batch_kernel<M, JAC, false>;
  return launch_timed(h, kern, dim3((h->B + 63) / 64), dim3(64), 0, a);
}

template <class M>
int launch_batch(mi_ilqr* h, int mode, const KArgs& a) {
  if (mode != MODE_SOLVE) return MI_ILQR_E_UNSUPPORTED;    // stage-level entries: latency kernels only
  if (h->d.jacobian_mode == MI_JAC_AUTODIFF) return launch_batch_one<M, MI_JAC_AUTODIFF>(h, a);
  return launch_batch_one<M, MI_JAC_FD_CENTRAL>(h, a);
}

}  // namespace mi_host
