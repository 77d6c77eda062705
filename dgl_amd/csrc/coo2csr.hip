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

// Ids below 2^31 (int32 graphs always; int64 graphs whose node and edge counts fit, i.e. nearly all):
// the column id travels WITH the sort as the low half of a 64-bit value (position in the high half)
// and the key is the 32-bit row id, both built by transform iterators, so that the compress pass
// reads everything in order — gathering col[perm[i]] afterwards was one fabric request per edge
// (62 M of them: 1.1 of the 2.8 ms at C2 size) against 4 more bytes per element and pass in the sort.
template <typename Idx>
struct PackColPos {
  const Idx* col;
  __host__ __device__ uint64_t operator()(int32_t i) const {
    return static_cast<uint64_t>(static_cast<uint32_t>(col[i])) | (static_cast<uint64_t>(static_cast<uint32_t>(i)) << 32);
  }
};
template <typename Idx>
struct Key32 {
  const Idx* row;
  __host__ __device__ int32_t operator()(int32_t i) const { return static_cast<int32_t>(row[i]); }
};

template <typename Idx>
__global__ __launch_bounds__(256) void compress_packed_kernel(const int32_t* __restrict__ sorted_row,
                                                              const uint64_t* __restrict__ packed,
                                                              const Idx* __restrict__ eids,
                                                              Idx* __restrict__ indptr,
                                                              Idx* __restrict__ indices,
                                                              Idx* __restrict__ eids_out, int64_t nnz,
                                                              int64_t num_rows) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < nnz; i += stride) {
    const uint64_t v = __builtin_nontemporal_load(packed + i);
    const int64_t pos = static_cast<int64_t>(v >> 32);
    indices[i] = static_cast<Idx>(v & 0xffffffffu);
    eids_out[i] = eids ? eids[pos] : static_cast<Idx>(pos);
    const int64_t r = static_cast<int64_t>(sorted_row[i]);
    const int64_t rp = i > 0 ? static_cast<int64_t>(sorted_row[i - 1]) : -1;
    for (int64_t q = rp + 1; q <= r; ++q) indptr[q] = static_cast<Idx>(i);
    if (i == nnz - 1)
      for (int64_t q = r + 1; q <= num_rows; ++q) indptr[q] = static_cast<Idx>(nnz);
  }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

template <typename Idx>
using PackIt = rocprim::transform_iterator<rocprim::counting_iterator<int32_t>, PackColPos<Idx>, uint64_t>;
template <typename Idx>
using KeyIt = rocprim::transform_iterator<rocprim::counting_iterator<int32_t>, Key32<Idx>, int32_t>;

template <typename Idx>
size_t sort_packed_temp_bytes(int64_t nnz, int end_bit) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, KeyIt<Idx>(rocprim::counting_iterator<int32_t>(0), Key32<Idx>{nullptr}),
                                  static_cast<int32_t*>(nullptr),
                                  PackIt<Idx>(rocprim::counting_iterator<int32_t>(0), PackColPos<Idx>{nullptr}),
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
  // packed form (ids below 2^31): sorted 32-bit rows + 64-bit (column, position) values
  const size_t packed =
      align256(4 * nnz) + align256(8 * nnz) + align256(sort_packed_temp_bytes<Idx>(nnz, bits_for(num_rows)));
  return plain > packed ? plain : packed;
}

template <typename Idx>
int run(int64_t num_rows, int64_t nnz, const void* row, const void* col, const void* eids, void* indptr,
        void* indices, void* eids_out, char* ws, hipStream_t s, bool cols_fit32) {
  if (nnz == 0) {
    hipLaunchKernelGGL(zero_indptr_kernel<Idx>, dim3(static_cast<unsigned>((num_rows + 256) / 256)), dim3(256),
                       0, s, static_cast<Idx*>(indptr), num_rows + 1);
    DGLA_CHECK_HIP(hipGetLastError());
    return 0;
  }
  int64_t blocks = (nnz + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  const bool packable = cols_fit32 && nnz < (int64_t(1) << 31) && num_rows < (int64_t(1) << 31);
  if (packable) {
    int32_t* sorted_row = reinterpret_cast<int32_t*>(ws);
    uint64_t* packed = reinterpret_cast<uint64_t*>(ws + align256(4 * nnz));
    void* temp = ws + align256(4 * nnz) + align256(8 * nnz);
    const int end_bit = bits_for(num_rows);
    size_t temp_bytes = sort_packed_temp_bytes<Idx>(nnz, end_bit);
    const rocprim::counting_iterator<int32_t> zero(0);
    DGLA_CHECK_HIP(rocprim::radix_sort_pairs(
        temp, temp_bytes, KeyIt<Idx>(zero, Key32<Idx>{static_cast<const Idx*>(row)}), sorted_row,
        PackIt<Idx>(zero, PackColPos<Idx>{static_cast<const Idx*>(col)}), packed, static_cast<size_t>(nnz), 0,
        end_bit, s));
    hipLaunchKernelGGL(compress_packed_kernel<Idx>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, sorted_row,
                       packed, static_cast<const Idx*>(eids), static_cast<Idx*>(indptr), static_cast<Idx*>(indices),
                       static_cast<Idx*>(eids_out), nnz, num_rows);
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
  const size_t need = dgla_coo_to_csr_workspace_bytes(idtype_bits, num_rows, nnz);
  void* owned = nullptr;
  if (need && (!workspace || workspace_bytes < need)) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, need, s));
    workspace = owned;
  }
  // int64 ids: the packed form needs the minor ids to fit 32 bits, which only the caller can promise
  const bool minor_fit32 = num_minor > 0 && num_minor < (int64_t(1) << 31);
  const int rc = idtype_bits == 32
                     ? run<int32_t>(num_rows, nnz, row, col, eids, indptr, indices, eids_out,
                                    static_cast<char*>(workspace), s, true)
                     : run<int64_t>(num_rows, nnz, row, col, eids, indptr, indices, eids_out,
                                    static_cast<char*>(workspace), s, minor_fit32);
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
