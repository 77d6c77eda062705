// g-SDDMM kernels instantiated for f16 feature tensors (see sddmm.hip.h).
#include "sddmm.hip.h"
namespace dgla {
int launch_sddmm_f16(const SddmmLaunch& L) { return launch_sddmm_typed<f16_t>(L); }
}  // namespace dgla
