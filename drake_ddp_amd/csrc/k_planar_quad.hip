// Workgroup-per-problem kernels of the PlanarQuad model: every (Jacobian mode, kernel mode) instantiation.
#include "launch_large.hpp"

MI_INTERNAL int launch_planar_quad(mi_ilqr* h, int mode, const mi::KArgs& a) { return mi_host::launch_jac_large<mi::PlanarQuad>(h, mode, a); }
