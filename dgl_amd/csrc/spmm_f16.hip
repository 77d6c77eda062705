// g-SpMM CSR kernels instantiated for f16 feature tensors (see spmm_csr.hip.h).
#include "spmm_csr.hip.h"
namespace dgla {
int launch_spmm_csr_f16(const SpmmLaunch& L) { return launch_spmm_csr_typed<f16_t>(L); }
size_t spmm_csr_workspace_f16(const SpmmLaunch& L) { return spmm_csr_workspace_typed<f16_t>(L); }
}  // namespace dgla
