// g-SDDMM kernels instantiated for f64 feature tensors (see sddmm.hip.h).
#include "sddmm.hip.h"
namespace dgla {
int launch_sddmm_f64(const SddmmLaunch& L) { return launch_sddmm_typed<double>(L); }
}  // namespace dgla
