// g-SpMM CSR kernels instantiated for f32 feature tensors (see spmm_csr.hip.h).
#include "spmm_csr.hip.h"
namespace dgla {
int launch_spmm_csr_f32(const SpmmLaunch& L) { return launch_spmm_csr_typed<float>(L); }
size_t spmm_csr_workspace_f32(const SpmmLaunch& L) { return spmm_csr_workspace_typed<float>(L); }
}  // namespace dgla
