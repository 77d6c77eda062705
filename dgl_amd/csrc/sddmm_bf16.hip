// g-SDDMM kernels instantiated for bf16 feature tensors (see sddmm.hip.h).
#include "sddmm.hip.h"
namespace dgla {
int launch_sddmm_bf16(const SddmmLaunch& L) { return launch_sddmm_typed<bf16_t>(L); }
}  // namespace dgla
