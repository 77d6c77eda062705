// g-SpMM CSR kernels instantiated for f64 feature tensors (see spmm_csr.hip.h).
#include "spmm_csr.hip.h"
namespace dgla {
int launch_spmm_csr_f64(const SpmmLaunch& L) { return launch_spmm_csr_typed<double>(L); }
size_t spmm_csr_workspace_f64(const SpmmLaunch& L) { return spmm_csr_workspace_typed<double>(L); }
}  // namespace dgla
