// g-SpMM CSR kernels instantiated for bf16 feature tensors (see spmm_csr.hip.h).
#include "spmm_csr.hip.h"
namespace dgla {
int launch_spmm_csr_bf16(const SpmmLaunch& L) { return launch_spmm_csr_typed<bf16_t>(L); }
size_t spmm_csr_workspace_bf16(const SpmmLaunch& L) { return spmm_csr_workspace_typed<bf16_t>(L); }
}  // namespace dgla
