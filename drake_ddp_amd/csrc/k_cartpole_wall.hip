// Wave-per-problem kernels of the CartPoleWall model: every (Jacobian mode, kernel mode) instantiation.
#include "launch_small.hpp"

MI_INTERNAL int launch_cartpole_wall(mi_ilqr* h, int mode, const mi::KArgs& a) { return mi_host::launch_jac<mi::CartPoleWall>(h, mode, a); }
