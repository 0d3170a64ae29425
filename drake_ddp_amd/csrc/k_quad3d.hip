// Workgroup-per-problem kernels of the Quad3D model (n = 37: the split tile layout of ilqr_large.hpp): every (Jacobian mode, kernel mode) instantiation.
#include "launch_large.hpp"

MI_INTERNAL int launch_quad3d(mi_ilqr* h, int mode, const mi::KArgs& a) { return mi_host::launch_jac_large<mi::Quad3D>(h, mode, a); }
