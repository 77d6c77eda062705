// COO -> CSR / CSC materialisation on the device.
//
// Reference: aten::COOToCSR<kDGLCUDA> (src/array/cuda/coo2csr.cu:28-110): if the COO is not
// row-sorted, COOSort by row (src/array/cuda/coo_sort.cu, a radix sort of encoded keys carrying
// the edge permutation), then cusparseXcoo2csr compresses the sorted row array; the CSR's
// `data` is the permutation, i.e. the original edge id of every CSR position.  UnitGraph builds
// the in-edge CSR the same way from the transposed COO (src/graph/unit_graph.cc:1418-1450).
//
// Here: the library's own stable bucket sort over exactly the bits the row ids need (sort.hip.h:
// most-significant-digit first, one 64-bit word [remaining row bits | column | position] per edge,
// exact LDS match-mask ranking).  Its LAST level's bins are the rows: the scan of that level writes
// `indptr`, its scatter unpacks the words into `indices` / `eids` — no vendor sort, no separate
// histogram / compress pass.  Stable: edges keep their COO order inside a row, which fixes the CSR
// position order and with it arg-max / arg-min tie-breaking.
#include "../../include/dgl_amd.h"

#include <cstring>

#include "common.h"
#include "sort.hip.h"

namespace dgla {
namespace {

int cfail(const std::string& m) {
  last_error() = m;
  return -1;
}

template <typename Idx>
__global__ void zero_indptr_kernel(Idx* indptr, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) indptr[i] = 0;
}

// bits of the column ids: from the caller's bound when there is one, else everything the id type can hold
int col_bits(int idtype_bits, int64_t num_minor) {
  if (num_minor > 0) return msd::bits_for(num_minor);
  return idtype_bits == 32 ? 31 : 63;
}

msd::Plan plan_for(int idtype_bits, int64_t num_rows, int64_t num_minor, int64_t nnz) {
  return msd::make_plan(nnz, msd::bits_for(num_rows), col_bits(idtype_bits, num_minor), msd::bits_for(nnz));
}

template <typename Idx>
int run(const msd::Plan& p, int64_t num_rows, int64_t nnz, const void* row, const void* col, const void* eids, void* indptr,
        void* indices, void* eids_out, char* ws, hipStream_t s) {
  if (nnz == 0) {
    hipLaunchKernelGGL(zero_indptr_kernel<Idx>, dim3(static_cast<unsigned>((num_rows + 256) / 256)), dim3(256),
                       0, s, static_cast<Idx*>(indptr), num_rows + 1);
    DGLA_CHECK_HIP(hipGetLastError());
    return 0;
  }
  msd::LevelArgs<Idx> a{};
  a.row = static_cast<const Idx*>(row);
  a.col = static_cast<const Idx*>(col);
  a.eids_in = static_cast<const Idx*>(eids);
  a.indices = static_cast<Idx*>(indices);
  a.eids_out = static_cast<Idx*>(eids_out);
  a.indptr = static_cast<Idx*>(indptr);
  a.num_rows = num_rows;
  return p.wide ? msd::run_levels<true, Idx>(p, a, ws, s) : msd::run_levels<false, Idx>(p, a, ws, s);
}

}  // namespace
}  // namespace dgla

using namespace dgla;

extern "C" {

size_t dgla_coo_to_csr_workspace_bytes(int idtype_bits, int64_t num_rows, int64_t nnz) {
  if (nnz <= 0 || num_rows <= 0) return 0;
  return plan_for(idtype_bits, num_rows, 0, nnz).bytes;   // (no column bound: the widest layout the call can take)
}

static int coo_to_csr_impl(int idtype_bits, int64_t num_rows, int64_t num_minor, int64_t nnz, const void* row,
                           const void* col, const void* eids, void* indptr, void* indices, void* eids_out,
                           void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return cfail("idtype must be int32 or int64");
  if (num_rows < 0 || nnz < 0) return cfail("negative size");
  if (!indptr) return cfail("indptr is null");
  if (nnz > 0 && (!row || !col || !indices || !eids_out)) return cfail("coo / output arrays are null");
  if (idtype_bits == 32 && (nnz > 0x7fffffffLL || num_rows > 0x7fffffffLL))
    return cfail("int32 ids cannot address this many edges / rows");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, indptr);
  msd::Plan p;
  if (nnz > 0 && num_rows > 0) p = plan_for(idtype_bits, num_rows, num_minor, nnz);
  void* owned = nullptr;
  if (p.bytes && (!workspace || workspace_bytes < p.bytes)) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, p.bytes, s));
    workspace = owned;
  }
  const int rc = idtype_bits == 32
                     ? run<int32_t>(p, num_rows, nnz, row, col, eids, indptr, indices, eids_out,
                                    static_cast<char*>(workspace), s)
                     : run<int64_t>(p, num_rows, nnz, row, col, eids, indptr, indices, eids_out,
                                    static_cast<char*>(workspace), s);
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

int dgla_coo_to_csr(int idtype_bits, int64_t num_rows, int64_t nnz, const void* row, const void* col,
                    const void* eids, void* indptr, void* indices, void* eids_out, void* workspace,
                    size_t workspace_bytes, void* hip_stream) {
  return coo_to_csr_impl(idtype_bits, num_rows, 0, nnz, row, col, eids, indptr, indices, eids_out, workspace,
                         workspace_bytes, hip_stream);
}

int dgla_coo_to_csr_bounded(int idtype_bits, int64_t num_rows, int64_t num_minor, int64_t nnz, const void* row,
                            const void* col, const void* eids, void* indptr, void* indices, void* eids_out,
                            void* workspace, size_t workspace_bytes, void* hip_stream) {
  return coo_to_csr_impl(idtype_bits, num_rows, num_minor, nnz, row, col, eids, indptr, indices, eids_out,
                         workspace, workspace_bytes, hip_stream);
}

}  // extern "C"
