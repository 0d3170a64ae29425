// Lane-per-problem ("throughput") kernels of every small-state model: one translation unit.
#include "launch_batch.hpp"

MI_INTERNAL int launch_batch_minor(mi_ilqr* h, int mode, const mi::KArgs& a) {
  using namespace mi_host;
  switch (h->d.model_id) {
    case MI_MODEL_PENDULUM: return launch_batch<Pendulum>(h, mode, a);
    case MI_MODEL_ACROBOT: return launch_batch<Acrobot>(h, mode, a);
    case MI_MODEL_CARTPOLE: return launch_batch<CartPole>(h, mode, a);
    case MI_MODEL_CARTPOLE_WALL: return launch_batch<CartPoleWall>(h, mode, a);
    default: return MI_ILQR_E_UNSUPPORTED;                 // (plugin models: through the plugin's own launch entry, mi_ilqr.hip)
  }
}
