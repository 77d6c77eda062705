// g-SDDMM kernels instantiated for f64 feature tensors (see sddmm.cuh).
#include "sddmm.cuh"
namespace dgla {
int launch_sddmm_f64(const SddmmLaunch& L) { return launch_sddmm_typed<double>(L); }
}  // namespace dgla
