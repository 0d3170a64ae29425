// Workgroup-per-problem kernels of the Arm27C model (the arm + ball with coupled rigid-body joint dynamics, n = 27, m = 7: the mid-size
// family of ilqr_large.hpp): every (Jacobian mode, kernel mode) instantiation.
#include "launch_large.hpp"

MI_INTERNAL int launch_arm27c(mi_ilqr* h, int mode, const mi::KArgs& a) { return mi_host::launch_jac_large<mi::Arm27C>(h, mode, a); }
