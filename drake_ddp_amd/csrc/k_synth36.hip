// Workgroup-per-problem kernels of the Synth36 model: every (Jacobian mode, kernel mode) instantiation.
#include "launch_large.hpp"

MI_INTERNAL int launch_synth36(mi_ilqr* h, int mode, const mi::KArgs& a) { return mi_host::launch_jac_large<mi::Synth36>(h, mode, a); }
