// g-SDDMM kernels instantiated for f32 feature tensors (see sddmm.hip.h).
#include "sddmm.hip.h"
namespace dgla {
int launch_sddmm_f32(const SddmmLaunch& L) { return launch_sddmm_typed<float>(L); }
}  // namespace dgla
