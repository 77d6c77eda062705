// g-SDDMM kernels instantiated for f32 feature tensors (see sddmm.cuh).
#include "sddmm.cuh"
namespace dgla {
int launch_sddmm_f32(const SddmmLaunch& L) { return launch_sddmm_typed<float>(L); }
}  // namespace dgla
