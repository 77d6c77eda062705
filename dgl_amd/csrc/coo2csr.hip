// COO -> CSR / CSC materialisation on the device.
//
// Reference: aten::COOToCSR<kDGLCUDA> (src/array/cuda/coo2csr.cu:28-110): if the COO is not
// row-sorted, COOSort by row (src/array/cuda/coo_sort.cu, a radix sort of encoded keys carrying
// the edge permutation), then cusparseXcoo2csr compresses the sorted row array; the CSR's
// `data` is the permutation, i.e. the original edge id of every CSR position.  UnitGraph builds
// the in-edge CSR the same way from the transposed COO (src/graph/unit_graph.cc:1418-1450).
//
// Here: one stable radix sort of (row, position) pairs over exactly the bits the row ids
// need (rocPRIM's device radix sort: the sort is not the product, the kernels around it
// are), then ONE fused kernel that gathers `col` / `eids` through the permutation and writes
// indptr from the run boundaries of the sorted rows — no separate histogram, scan or
// compression pass.  Stable: edges keep their COO order inside a row, which fixes the CSR
// position order and with it arg-max / arg-min tie-breaking.
#include "../../include/dgl_amd.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

namespace dgla {
namespace {

int cfail(const std::string& m) {
  last_error() = m;
  return -1;
}

template <typename Idx>
__global__ __launch_bounds__(256) void iota_kernel(Idx* p, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride)
    p[i] = static_cast<Idx>(i);
}

// sorted_row / perm: the sort's outputs.  Position i of the CSR takes edge perm[i]; indptr[r] is
// the first position whose row is >= r, written by the thread that sees the run boundary.
template <typename Idx>
__global__ __launch_bounds__(256) void compress_kernel(const Idx* __restrict__ sorted_row,
                                                       const Idx* __restrict__ perm,
                                                       const Idx* __restrict__ col,
                                                       const Idx* __restrict__ eids,
                                                       Idx* __restrict__ indptr,
                                                       Idx* __restrict__ indices,
                                                       Idx* __restrict__ eids_out, int64_t nnz,
                                                       int64_t num_rows) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < nnz; i += stride) {
    const Idx e = perm[i];
    indices[i] = col[e];
    eids_out[i] = eids ? eids[e] : e;
    const int64_t r = static_cast<int64_t>(sorted_row[i]);
    const int64_t rp = i > 0 ? static_cast<int64_t>(sorted_row[i - 1]) : -1;
    for (int64_t q = rp + 1; q <= r; ++q) indptr[q] = static_cast<Idx>(i);
    if (i == nnz - 1)
      for (int64_t q = r + 1; q <= num_rows; ++q) indptr[q] = static_cast<Idx>(nnz);
  }
}

template <typename Idx>
__global__ void zero_indptr_kernel(Idx* indptr, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) indptr[i] = 0;
}

// int32 ids: the column id travels WITH the sort as the low half of a 64-bit value (position in the
// high half), so that the compress pass reads everything in order — gathering col[perm[i]] afterwards
// was one fabric request per edge (62 M of them: 1.1 of the 2.8 ms at C2 size) against 4 more bytes
// per element and pass in the sort.
struct PackColPos {
  const int32_t* col;
  __host__ __device__ uint64_t operator()(int32_t i) const {
    return static_cast<uint64_t>(static_cast<uint32_t>(col[i])) | (static_cast<uint64_t>(static_cast<uint32_t>(i)) << 32);
  }
};

__global__ __launch_bounds__(256) void compress_packed_kernel(const int32_t* __restrict__ sorted_row,
                                                              const uint64_t* __restrict__ packed,
                                                              const int32_t* __restrict__ eids,
                                                              int32_t* __restrict__ indptr,
                                                              int32_t* __restrict__ indices,
                                                              int32_t* __restrict__ eids_out, int64_t nnz,
                                                              int64_t num_rows) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < nnz; i += stride) {
    const uint64_t v = __builtin_nontemporal_load(packed + i);
    const int32_t pos = static_cast<int32_t>(v >> 32);
    indices[i] = static_cast<int32_t>(v & 0xffffffffu);
    eids_out[i] = eids ? eids[pos] : pos;
    const int64_t r = static_cast<int64_t>(sorted_row[i]);
    const int64_t rp = i > 0 ? static_cast<int64_t>(sorted_row[i - 1]) : -1;
    for (int64_t q = rp + 1; q <= r; ++q) indptr[q] = static_cast<int32_t>(i);
    if (i == nnz - 1)
      for (int64_t q = r + 1; q <= num_rows; ++q) indptr[q] = static_cast<int32_t>(nnz);
  }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

using PackIt = rocprim::transform_iterator<rocprim::counting_iterator<int32_t>, PackColPos, uint64_t>;

size_t sort_packed_temp_bytes(int64_t nnz, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                                  PackIt(rocprim::counting_iterator<int32_t>(0), PackColPos{nullptr}),
                                  static_cast<uint64_t*>(nullptr), static_cast<size_t>(nnz), 0, end_bit, nullptr);
  return bytes;
}

int bits_for(int64_t num_rows) {
  int b = 1;
  while ((int64_t(1) << b) < num_rows) ++b;
  return b;
}

template <typename Idx>
size_t sort_temp_bytes(int64_t nnz, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, const Idx*, Idx*, const Idx*, Idx*>(
      nullptr, bytes, nullptr, nullptr, nullptr, nullptr, static_cast<size_t>(nnz), 0, end_bit, nullptr);
  return bytes;
}

template <typename Idx>
size_t workspace_typed(int64_t nnz, int64_t num_rows) {
  const size_t plain = align256(sizeof(Idx) * nnz) * 3 + align256(sort_temp_bytes<Idx>(nnz, bits_for(num_rows)));
  if (sizeof(Idx) == 4) {  // packed form: sorted rows + 64-bit (column, position) values
    const size_t packed = align256(4 * nnz) + align256(8 * nnz) + align256(sort_packed_temp_bytes(nnz, bits_for(num_rows)));
    return plain > packed ? plain : packed;
  }
  return plain;
}

template <typename Idx>
int run(int64_t num_rows, int64_t nnz, const void* row, const void* col, const void* eids, void* indptr,
        void* indices, void* eids_out, char* ws, hipStream_t s) {
  if (nnz == 0) {
    hipLaunchKernelGGL(zero_indptr_kernel<Idx>, dim3(static_cast<unsigned>((num_rows + 256) / 256)), dim3(256),
                       0, s, static_cast<Idx*>(indptr), num_rows + 1);
    DGLA_CHECK_HIP(hipGetLastError());
    return 0;
  }
  int64_t blocks = (nnz + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if constexpr (sizeof(Idx) == 4) {
    int32_t* sorted_row = reinterpret_cast<int32_t*>(ws);
    uint64_t* packed = reinterpret_cast<uint64_t*>(ws + align256(4 * nnz));
    void* temp = ws + align256(4 * nnz) + align256(8 * nnz);
    const int end_bit = bits_for(num_rows);
    size_t temp_bytes = sort_packed_temp_bytes(nnz, end_bit);
    DGLA_CHECK_HIP(rocprim::radix_sort_pairs(
        temp, temp_bytes, static_cast<const int32_t*>(row), sorted_row,
        PackIt(rocprim::counting_iterator<int32_t>(0), PackColPos{static_cast<const int32_t*>(col)}), packed,
        static_cast<size_t>(nnz), 0, end_bit, s));
    hipLaunchKernelGGL(compress_packed_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, sorted_row,
                       packed, static_cast<const int32_t*>(eids), static_cast<int32_t*>(indptr),
                       static_cast<int32_t*>(indices), static_cast<int32_t*>(eids_out), nnz, num_rows);
    DGLA_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const size_t arr = align256(sizeof(Idx) * nnz);
  Idx* pos = reinterpret_cast<Idx*>(ws);
  Idx* sorted_row = reinterpret_cast<Idx*>(ws + arr);
  Idx* perm = reinterpret_cast<Idx*>(ws + 2 * arr);
  void* temp = ws + 3 * arr;
  const int end_bit = bits_for(num_rows);
  size_t temp_bytes = sort_temp_bytes<Idx>(nnz, end_bit);
  hipLaunchKernelGGL(iota_kernel<Idx>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, pos, nnz);
  DGLA_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, static_cast<const Idx*>(row), sorted_row,
                                           static_cast<const Idx*>(pos), perm, static_cast<size_t>(nnz), 0,
                                           end_bit, s));
  hipLaunchKernelGGL(compress_kernel<Idx>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, sorted_row,
                     perm, static_cast<const Idx*>(col), static_cast<const Idx*>(eids),
                     static_cast<Idx*>(indptr), static_cast<Idx*>(indices), static_cast<Idx*>(eids_out), nnz,
                     num_rows);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace
}  // namespace dgla

using namespace dgla;

extern "C" {

size_t dgla_coo_to_csr_workspace_bytes(int idtype_bits, int64_t num_rows, int64_t nnz) {
  if (nnz <= 0 || num_rows <= 0) return 0;
  return idtype_bits == 32 ? workspace_typed<int32_t>(nnz, num_rows) : workspace_typed<int64_t>(nnz, num_rows);
}

int dgla_coo_to_csr(int idtype_bits, int64_t num_rows, int64_t nnz, const void* row, const void* col,
                    const void* eids, void* indptr, void* indices, void* eids_out, void* workspace,
                    size_t workspace_bytes, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return cfail("idtype must be int32 or int64");
  if (num_rows < 0 || nnz < 0) return cfail("negative size");
  if (!indptr) return cfail("indptr is null");
  if (nnz > 0 && (!row || !col || !indices || !eids_out)) return cfail("coo / output arrays are null");
  if (idtype_bits == 32 && (nnz > 0x7fffffffLL || num_rows > 0x7fffffffLL))
    return cfail("int32 ids cannot address this many edges / rows");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, indptr);
  const size_t need = dgla_coo_to_csr_workspace_bytes(idtype_bits, num_rows, nnz);
  void* owned = nullptr;
  if (need && (!workspace || workspace_bytes < need)) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, need, s));
    workspace = owned;
  }
  const int rc = idtype_bits == 32
                     ? run<int32_t>(num_rows, nnz, row, col, eids, indptr, indices, eids_out,
                                    static_cast<char*>(workspace), s)
                     : run<int64_t>(num_rows, nnz, row, col, eids, indptr, indices, eids_out,
                                    static_cast<char*>(workspace), s);
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

}  // extern "C"
