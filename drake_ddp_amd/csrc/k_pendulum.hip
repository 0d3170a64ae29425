// Wave-per-problem kernels of the Pendulum model: every (Jacobian mode, kernel mode) instantiation.
#include "launch_small.hpp"

MI_INTERNAL int launch_pendulum(mi_ilqr* h, int mode, const mi::KArgs& a) { return mi_host::launch_jac<mi::Pendulum>(h, mode, a); }
