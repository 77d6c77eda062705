// g-SDDMM kernels instantiated for f16 feature tensors (see sddmm.cuh).
#include "sddmm.cuh"
namespace dgla {
int launch_sddmm_f16(const SddmmLaunch& L) { return launch_sddmm_typed<f16_t>(L); }
}  // namespace dgla
