// Lane-per-problem ("throughput") kernels of every small-state model: one translation unit.
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

MI_INTERNAL int launch_batch_minor(mi_ilqr* h, int mode, const mi::KArgs& a) {
  using namespace mi_host;
  switch (h->d.model_id) {
    case MI_MODEL_PENDULUM: return launch_batch<Pendulum>(h, mode, a);
    case MI_MODEL_ACROBOT: return launch_batch<Acrobot>(h, mode, a);
    case MI_MODEL_CARTPOLE: return launch_batch<CartPole>(h, mode, a);
    case MI_MODEL_CARTPOLE_WALL: return launch_batch<CartPoleWall>(h, mode, a);
    default: return MI_ILQR_E_UNSUPPORTED;
  }
}
