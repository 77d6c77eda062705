// Segment reduce / scatter add / backward of segment max-min for gfx950.
//
// Reference: src/array/cuda/segment_reduce.cuh:30-113 (SegmentReduceKernel, ScatterAddKernel,
// BackwardSegmentCmpKernel), src/array/cpu/segment_reduce.h:27-187, registered at
// src/array/kernel.cc:658-708.
//
//  * segment reduce IS a g-SpMM: offsets are a CSR indptr whose "edges" are the rows of
//    `feat` in place (copy_rhs, edge id == position), so it runs on the merge-path kernel of
//    spmm_csr.hip.h — one wavefront per 512 items whatever the segment lengths — instead of
//    the reference's one-block-per-segment loop.  Only difference from g-SpMM: arg of an
//    element nothing won is -1 (segment_reduce.cuh:39, cpu/segment_reduce.h:66), not 0.
//  * scatter add: out[idx[i], :] += feat[i, :].  Small inputs: hardware float atomics, 16-byte
//    row pieces per lane (the reference: one scalar atomic per thread).  Large inputs: no
//    atomics — rows are grouped by target with one radix sort and summed by the merge-path
//    kernel reading through the permutation (deterministic, several times faster).
//  * backward of segment max/min: out[arg[i, k], k] = feat[i, k] where arg >= 0; every
//    (row, k) is written at most once, so plain stores.
#include "../../include/dgl_amd.h"

#include <cstring>

#include "common.h"

namespace dgla {

bool narrow_reduce_eligible(const SpmmLaunch&);          // narrow_reduce.hip: 1 ... 8 fp32 columns per row
size_t narrow_reduce_workspace_bytes(const SpmmLaunch&);
int launch_narrow_reduce(const SpmmLaunch&, void* ws);
int launch_spmm_csr_f32(const SpmmLaunch&);
int launch_spmm_csr_f64(const SpmmLaunch&);
int launch_spmm_csr_f16(const SpmmLaunch&);
int launch_spmm_csr_bf16(const SpmmLaunch&);
size_t spmm_csr_workspace_f32(const SpmmLaunch&);
size_t spmm_csr_workspace_f64(const SpmmLaunch&);
size_t spmm_csr_workspace_f16(const SpmmLaunch&);
size_t spmm_csr_workspace_bf16(const SpmmLaunch&);

namespace {

int sfail(const std::string& m) {
  last_error() = m;
  return -1;
}

int64_t row_len(const dgla_tensor* t) {
  int64_t n = 1;
  for (int i = 1; i < t->ndim; ++i) n *= t->shape[i];
  return n;
}

// ---- scatter add ----------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void atomic_add_elem(T* p, T v) {
  atomicAdd(p, v);  // fp32 / fp64: hardware atomics (-munsafe-fp-atomics)
}
// 16-bit storage: compare-and-swap on the enclosing aligned 32-bit word
template <typename T>
__device__ __forceinline__ void atomic_add_16(T* p, float v) {
  // (an explicit GLOBAL pointer: through uintptr_t the address space is lost and the compiler emits flat loads and
  // flat atomics, which count in both vmcnt and lgkmcnt — tools/isa_audit.py listed them)
  typedef __attribute__((address_space(1))) uint32_t gword_t;
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  gword_t* word = (gword_t*)(a & ~uintptr_t(3));
  const int shift = (a & 2) ? 16 : 0;
  uint32_t old = *word;
  while (true) {
    T cur;
    const uint16_t bits = static_cast<uint16_t>(old >> shift);
    __builtin_memcpy(&cur, &bits, 2);
    const T nv = from_acc<T>(to_acc<T>(cur) + v);
    uint16_t nb;
    __builtin_memcpy(&nb, &nv, 2);
    const uint32_t want = (old & ~(0xffffu << shift)) | (static_cast<uint32_t>(nb) << shift);
    // (on failure `old` is replaced by the value seen)
    if (__hip_atomic_compare_exchange_strong(word, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      break;
  }
}
template <>
__device__ __forceinline__ void atomic_add_elem<f16_t>(f16_t* p, f16_t v) {
  atomic_add_16<f16_t>(p, to_acc<f16_t>(v));
}
template <>
__device__ __forceinline__ void atomic_add_elem<bf16_t>(bf16_t* p, bf16_t v) {
  atomic_add_16<bf16_t>(p, to_acc<bf16_t>(v));
}

// One lane per VEC consecutive features of one input row; rows of a workgroup are
// consecutive so the feat reads are fully coalesced.
template <typename Idx, typename DT, int VEC>
__global__ __launch_bounds__(256) void scatter_add_kernel(const DT* __restrict__ feat,
                                                          const Idx* __restrict__ idx,
                                                          DT* __restrict__ out, int64_t n,
                                                          int dim, int lanes_per_row) {
  const int64_t total = n * lanes_per_row;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < total;
       t += stride) {
    const int64_t row = t / lanes_per_row;
    const int k = static_cast<int>(t - row * lanes_per_row) * VEC;
    const int64_t dst = static_cast<int64_t>(idx[row]);
    const VecT<DT, VEC> v = *reinterpret_cast<const VecT<DT, VEC>*>(feat + row * dim + k);
#pragma unroll
    for (int j = 0; j < VEC; ++j) atomic_add_elem<DT>(out + dst * dim + k + j, v.v[j]);
  }
}

template <typename Idx, typename DT>
int run_scatter_add(const void* feat, const void* idx, void* out, int64_t n, int64_t dim,
                    hipStream_t s) {
  constexpr int full = 16 / sizeof(DT);
  const bool vec_ok = dim % full == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0;
  const int vec = vec_ok ? full : 1;
  const int lanes = static_cast<int>(dim / vec);
  int64_t blocks = (n * lanes + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (blocks < 1) blocks = 1;
  if (vec_ok)
    hipLaunchKernelGGL((scatter_add_kernel<Idx, DT, full>), dim3(static_cast<unsigned>(blocks)),
                       dim3(256), 0, s, static_cast<const DT*>(feat), static_cast<const Idx*>(idx),
                       static_cast<DT*>(out), n, static_cast<int>(dim), lanes);
  else
    hipLaunchKernelGGL((scatter_add_kernel<Idx, DT, 1>), dim3(static_cast<unsigned>(blocks)),
                       dim3(256), 0, s, static_cast<const DT*>(feat), static_cast<const Idx*>(idx),
                       static_cast<DT*>(out), n, static_cast<int>(dim), lanes);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- backward of segment max / min ------------------------------------------------------
template <typename Idx, typename DT>
__global__ __launch_bounds__(256) void bwd_segment_cmp_kernel(const DT* __restrict__ feat,
                                                              const Idx* __restrict__ arg,
                                                              DT* __restrict__ out,
                                                              int64_t total, int dim) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < total;
       t += stride) {
    const int64_t w = static_cast<int64_t>(arg[t]);
    if (w >= 0) out[w * dim + (t % dim)] = feat[t];
  }
}

template <typename Idx, typename DT>
int run_bwd_segment_cmp(const void* feat, const void* arg, void* out, int64_t n, int64_t dim,
                        hipStream_t s) {
  const int64_t total = n * dim;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((bwd_segment_cmp_kernel<Idx, DT>), dim3(static_cast<unsigned>(blocks)),
                     dim3(256), 0, s, static_cast<const DT*>(feat), static_cast<const Idx*>(arg),
                     static_cast<DT*>(out), total, static_cast<int>(dim));
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// UpdateGradMinMaxHeteroKernel (src/array/cuda/segment_reduce.cuh:73-92): element (row, col) of
// the incoming gradient goes to the node / edge that won the forward max / min, but only through
// the relation type that winner belongs to: out[idx[row, col], col] += feat[row, col] where
// idx_type[row, col] == type.  Several destinations can share a winner -> atomics, as there.
template <typename Idx, typename DT>
__global__ __launch_bounds__(256) void update_grad_minmax_kernel(
    const DT* __restrict__ feat, const Idx* __restrict__ idx, const Idx* __restrict__ idx_type,
    DT* __restrict__ out, int64_t n, int64_t dim, Idx type, int64_t out_rows) {
  const int64_t total = n * dim;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total; i += stride) {
    if (idx_type[i] != type) continue;
    const int64_t r = static_cast<int64_t>(idx[i]);
    if (r < 0 || r >= out_rows) continue;
    atomic_add_elem<DT>(out + r * dim + (i % dim), feat[i]);
  }
}

template <typename Idx, typename DT>
int run_update_grad_minmax(const void* feat, const void* idx, const void* idx_type, void* out,
                           int64_t n, int64_t dim, int64_t type, int64_t out_rows, hipStream_t s) {
  const int64_t total = n * dim;
  const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 256 * 64));
  hipLaunchKernelGGL((update_grad_minmax_kernel<Idx, DT>), dim3(blocks), dim3(256), 0, s,
                     static_cast<const DT*>(feat), static_cast<const Idx*>(idx),
                     static_cast<const Idx*>(idx_type), static_cast<DT*>(out), n, dim,
                     static_cast<Idx>(type), out_rows);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// Backward of g-SpMM max / min (python/dgl/backend/pytorch/sparse.py:217-244): the gradient of output
// element (i, k) goes to the operand row that WON it — arg[i, k], read in the graph's own idtype — times,
// for `mul`, the other operand's winning row (other[arg_other[i, k], (k / group) % other_len]: the
// broadcast forms "leading dims then ones" and "ones then trailing dims").  One pass over dz where the
// reference makes four (two .long() casts, a gather, a scatter_add_).
//   ATOMIC = false: the target is the EDGE operand — an edge has one destination, so (arg[i, k], k) is
//                   written at most once: plain stores, deterministic.  The ONE ambiguous target is row 0:
//                   the forward records arg = 0 for every destination without in-edges, so edge 0 can be
//                   named by many rows; row 0 therefore ADDS (out is zero-filled), exactly the reference's
//                   scatter_add_ — the empty rows contribute their dz (0 after update_all's inf -> 0);
//   ATOMIC = true:  the target is the NODE operand — a source node can win at many destinations, the
//                   sum over them is a hardware float atomic (as the reference's scatter_add_ / its
//                   UpdateGradMinMaxHeteroKernel, segment_reduce.cuh:73-92).
template <typename Idx, typename DT, bool MUL, bool ATOMIC>
__global__ __launch_bounds__(256) void spmm_cmp_backward_kernel(
    const DT* __restrict__ dz, const Idx* __restrict__ arg, const DT* __restrict__ other,
    const Idx* __restrict__ arg_other, DT* __restrict__ out, int64_t total, int dim, int other_len,
    int group, int64_t out_rows) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t t = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; t < total; t += stride) {
    const int64_t w = static_cast<int64_t>(arg[t]);
    if (w < 0 || w >= out_rows) continue;
    const int k = static_cast<int>(t % dim);
    DT g = dz[t];
    if constexpr (MUL) {
      const int64_t wo = static_cast<int64_t>(arg_other[t]);
      const DT o = other[wo * other_len + (k / group) % other_len];
      g = from_acc<DT>(to_acc<DT>(g) * to_acc<DT>(o));
    }
    if (ATOMIC || w == 0)
      atomic_add_elem<DT>(out + w * dim + k, g);
    else
      out[w * dim + k] = g;
  }
}

template <typename Idx, typename DT>
int run_spmm_cmp_backward(const void* dz, const void* arg, const void* other, const void* arg_other,
                          void* out, int64_t n, int64_t dim, int64_t other_len, int64_t group,
                          int64_t out_rows, bool atomic, hipStream_t s) {
  const int64_t total = n * dim;
  const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 256 * 64));
#define DGLA_CMPB(M, A)                                                                          \
  hipLaunchKernelGGL((spmm_cmp_backward_kernel<Idx, DT, M, A>), dim3(blocks), dim3(256), 0, s,    \
                     static_cast<const DT*>(dz), static_cast<const Idx*>(arg),                   \
                     static_cast<const DT*>(other), static_cast<const Idx*>(arg_other),          \
                     static_cast<DT*>(out), total, static_cast<int>(dim),                        \
                     static_cast<int>(other_len), static_cast<int>(group), out_rows)
  if (other) {
    if (atomic) DGLA_CMPB(true, true); else DGLA_CMPB(true, false);
  } else {
    if (atomic) DGLA_CMPB(false, true); else DGLA_CMPB(false, false);
  }
#undef DGLA_CMPB
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

#define DGLA_IDX_DTYPE_SWITCH(idbits, dtype, FN, ...)                                   \
  do {                                                                                  \
    if ((idbits) == 32) {                                                               \
      switch (dtype) {                                                                  \
        case DGLA_F32: return FN<int32_t, float>(__VA_ARGS__);                          \
        case DGLA_F64: return FN<int32_t, double>(__VA_ARGS__);                         \
        case DGLA_F16: return FN<int32_t, f16_t>(__VA_ARGS__);                          \
        case DGLA_BF16: return FN<int32_t, bf16_t>(__VA_ARGS__);                        \
      }                                                                                 \
    } else {                                                                            \
      switch (dtype) {                                                                  \
        case DGLA_F32: return FN<int64_t, float>(__VA_ARGS__);                          \
        case DGLA_F64: return FN<int64_t, double>(__VA_ARGS__);                         \
        case DGLA_F16: return FN<int64_t, f16_t>(__VA_ARGS__);                          \
        case DGLA_BF16: return FN<int64_t, bf16_t>(__VA_ARGS__);                        \
      }                                                                                 \
    }                                                                                   \
  } while (0)

int build_segment_launch(const char* reduce, int idbits, dgla_dtype dtype,
                         const dgla_tensor* feat, const void* offsets, int64_t num_segments,
                         const dgla_tensor* out, SpmmLaunch* L) {
  int red = -1;
  if (reduce && !strcmp(reduce, "sum")) red = kSum;
  if (reduce && !strcmp(reduce, "max")) red = kMax;
  if (reduce && !strcmp(reduce, "min")) red = kMin;
  if (red < 0)
    return sfail(std::string("Unsupported reduce function ") + (reduce ? reduce : "(null)"));
  if (idbits != 32 && idbits != 64) return sfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return sfail("unsupported feature dtype");
  if (!feat || !out || feat->ndim < 1 || out->ndim < 1 || !feat->shape || !out->shape)
    return sfail("feat / out is null");
  if (feat->ndim != out->ndim) return sfail("feat and out must have the same number of dimensions");
  if (out->shape[0] != num_segments) return sfail("out has a different number of rows than segments");
  if (row_len(feat) != row_len(out)) return sfail("feat and out have different feature shapes");
  if (num_segments > 0 && !offsets) return sfail("offsets is null");
  if (row_len(out) > 0x7fffffffLL / 4) return sfail("feature length too large");
  L->csr.num_rows = num_segments;
  L->csr.num_cols = 0;
  L->csr.nnz = feat->shape[0];
  L->csr.idbits = idbits;
  L->csr.indptr = offsets;
  L->csr.indices = nullptr;
  L->csr.eids = nullptr;
  L->op = kCopyRhs;
  L->red = red;
  L->dtype = dtype;
  L->ufeat = nullptr;
  L->efeat = feat->data;
  L->out = out->data;
  L->out_len = L->lhs_len = L->rhs_len = row_len(out);
  L->bcast = kBcNone;
  L->rhs_group = 1;
  L->arg_empty = -1;
  L->tune = tuning_flags() & ~static_cast<uint32_t>(kTuneSplit);
  return 0;
}

// Large scatter adds do not use atomics at all: grouping the rows by target (one stable radix
// sort of (idx, position) + the fused compress kernel of coo2csr.hip) turns the operation into
// a segment sum whose "edges" are read through the permutation — the merge-path g-SpMM kernel
// with copy_rhs / sum / accumulate.  15.5 M rows x 100 fp32 into 612 k targets: 23.2 ms with
// hardware atomics (67 G atomics/s), a few ms sorted; and the result no longer depends on
// the order in which atomics land (rows of one target are added in input order).
constexpr int64_t kScatterSortMinElems = int64_t(1) << 20;

size_t align_256(size_t x) { return (x + 255) / 256 * 256; }

int scatter_add_sorted(int idbits, dgla_dtype dtype, const dgla_tensor* feat, const void* idx,
                       const dgla_tensor* out, hipStream_t s) {
  const int64_t n = feat->shape[0], rows = out->shape[0];
  const size_t ib = idbits / 8;
  SpmmLaunch L{};
  L.csr.num_rows = rows;
  L.csr.num_cols = 0;
  L.csr.nnz = n;
  L.csr.idbits = idbits;
  L.op = kCopyRhs;
  L.red = kSum;
  L.dtype = dtype;
  L.efeat = feat->data;
  L.out = out->data;
  L.out_len = L.lhs_len = L.rhs_len = row_len(out);
  L.bcast = kBcNone;
  L.rhs_group = 1;
  L.accumulate = true;  // out += ...: scatter add keeps what out already holds
  L.tune = tuning_flags() & ~static_cast<uint32_t>(kTuneSplit);
  L.stream = s;
  size_t spmm_ws = 0;
  switch (dtype) {
    case DGLA_F32: spmm_ws = spmm_csr_workspace_f32(L); break;
    case DGLA_F64: spmm_ws = spmm_csr_workspace_f64(L); break;
    case DGLA_F16: spmm_ws = spmm_csr_workspace_f16(L); break;
    case DGLA_BF16: spmm_ws = spmm_csr_workspace_bf16(L); break;
  }
  const size_t sort_ws = dgla_coo_to_csr_workspace_bytes(idbits, rows, n);
  const size_t off_indptr = 0, off_junk = align_256(ib * (rows + 1)), off_eids = off_junk + align_256(ib * n),
               off_sort = off_eids + align_256(ib * n), off_spmm = off_sort + align_256(sort_ws),
               total = off_spmm + align_256(spmm_ws);
  char* ws = nullptr;
  DGLA_CHECK_HIP(hipMallocAsync(reinterpret_cast<void**>(&ws), total, s));
  // rows = targets, "columns" are not needed (written to scratch), eids_out = input rows in target order
  int rc = dgla_coo_to_csr(idbits, rows, n, idx, idx, nullptr, ws + off_indptr, ws + off_junk, ws + off_eids,
                           ws + off_sort, sort_ws, s);
  if (rc == 0) {
    L.csr.indptr = ws + off_indptr;
    L.csr.indices = nullptr;
    L.csr.eids = ws + off_eids;
    L.workspace = ws + off_spmm;
    L.workspace_bytes = spmm_ws;
    switch (dtype) {
      case DGLA_F32: rc = launch_spmm_csr_f32(L); break;
      case DGLA_F64: rc = launch_spmm_csr_f64(L); break;
      case DGLA_F16: rc = launch_spmm_csr_f16(L); break;
      case DGLA_BF16: rc = launch_spmm_csr_bf16(L); break;
    }
  }
  (void)hipFreeAsync(ws, s);
  return rc;
}

}  // namespace
}  // namespace dgla

using namespace dgla;

// ---- max / min backward of the NODE operand without atomics: winner masks (dgla_spmm_cmp_mask) ------------------------
// The reference adds dZ[v][k] to dX[arg_u[v][k]][k] with scatter_add_ (python/dgl/backend/pytorch/sparse.py:216-224:
// float atomics, 25 G/s on this chip whatever the footprint, a different sum order on every run).  Turned around it is a
// g-SpMM over the REVERSE graph: dX[u][k] = sum over out-edges e = (u -> v) of [e delivered the winner of (v, k)] dZ[v][k].
// This kernel writes the bracket: one BIT per (edge, output column), in the edge's position order of the forward CSR,
// packed into words of the feature type's width; dgla_spmm_csr_masked then runs the merge-path kernel over the reverse CSR
// with those bits as its edge operand (416 instead of 800 bytes per edge of a compare against arg_u rows; deterministic).
//   by_edge == 0: `arg` = arg_u; the FIRST edge of row v whose source is arg_u[v][k] gets the bit (parallel edges carry
//                 equal values: any one of them is the winner, and all of them lead to the same dX row);
//   by_edge != 0: `arg` = arg_e; the edge whose id it names gets the bit.
// An element no edge of its row claims (no in-edges, or nothing beat the identity: arg = 0) keeps the reference's
// behaviour — dZ goes to row arg = 0 of dX all the same — summed in a fixed order (per wave in row order, then over the
// waves' slots by spmm_cmp_leak_kernel); dX must be zeroed BEFORE this kernel and the masked SpMM must ACCUMULATE into it.
// One wavefront per (row, 64 columns): lane = column; 64 edges at a time are loaded with one instruction and their keys
// broadcast from registers; every edge's 64-bit ballot is parked in lane j of the batch and the batch's words leave with
// one store.
__device__ __forceinline__ int32_t bcast_lane(int32_t v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ int64_t bcast_lane(int64_t v, int j) {
  return (static_cast<int64_t>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), j)) << 32) |
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), j));
}

// CPL columns per lane (round 6): a wave covers 64 CPL columns of a row — F = 100 is ONE wave per row with two columns per lane
// instead of two waves (the second with 36 live lanes): the per-row overhead and the key broadcast are paid once
// (mask kernel 1.68 -> see profiles/r6/cmp_backward_gated.jsonl).
// MODE 0: bits, the partial sums of unclaimed elements, and the atomic add of an unclaimed element with a non-zero target (a
//         hand-made arg), all in one launch: dx must be zeroed before and the masked g-SpMM must accumulate into it.
// MODE 1 (DGLA_CMP_MASK_DEFER): bits and partial sums only — dx is NOT touched, so the masked g-SpMM can STORE its rows (no
//         zero fill, no read of dx); an unclaimed element with a non-zero target only raises `rare`.
// MODE 2 (DGLA_CMP_MASK_FINISH, launched behind the g-SpMM): exits on its first load unless `rare` is up; else scans again and
//         does those atomic adds.  (The sums for dX[0] are added by spmm_cmp_leak_kernel behind it.)
template <typename Idx, typename W, typename DT, int CPL, int MODE>
__global__ __launch_bounds__(256) void spmm_cmp_mask_kernel(
    const Idx* __restrict__ indptr, const Idx* __restrict__ indices, const Idx* __restrict__ eids,
    const Idx* __restrict__ arg, int by_edge, int64_t rows, int F, int words, W* __restrict__ mask,
    const DT* __restrict__ dz, DT* __restrict__ dx, int64_t dx_rows, typename Acc<DT>::type* __restrict__ leak,
    uint32_t* __restrict__ rare) {
  using A = typename Acc<DT>::type;
  if constexpr (MODE == 2) {
    if (*rare == 0u) return;
  }
  constexpr int BITS = 8 * static_cast<int>(sizeof(W));
  constexpr int WPC = 64 / BITS;  // words per 64 columns
  const int chunks = (F + 64 * CPL - 1) / (64 * CPL);
  const int lane = threadIdx.x & 63;
  // wave -> (fixed chunk c, rows slot, slot + slots, ...): the grid is a whole number of chunk groups, so a wave's
  // unclaimed elements all belong to the same columns and can be summed in registers in ROW order
  const int64_t nw = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  const int64_t wid = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t slots = nw / chunks;
  const int c = static_cast<int>(wid % chunks);
  const int64_t slot = wid / chunks;
  if (slot >= slots) return;
  int f[CPL];
  bool live[CPL];
  A leaked[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    f[k] = (c * CPL + k) * 64 + lane;
    live[k] = f[k] < F;
    leaked[k] = A(0);
  }
  __shared__ uint64_t s_bits[4][CPL][64];
  uint64_t(*const wb)[64] = s_bits[threadIdx.x >> 6];  // this wave's words: edge j of the batch -> bits of columns 64 k ..
  for (int64_t row = slot; row < rows; row += slots) {
    const int64_t b = static_cast<int64_t>(indptr[row]), e = static_cast<int64_t>(indptr[row + 1]);
    Idx a[CPL];
    bool found[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      a[k] = live[k] ? arg[row * F + f[k]] : static_cast<Idx>(-1);
      found[k] = !live[k];
    }
    for (int64_t p0 = b; p0 < e; p0 += 64) {
      const int nb = static_cast<int>(e - p0 < 64 ? e - p0 : 64);
      Idx key = static_cast<Idx>(-2);  // (never equal to a column's winner: ids are >= 0, a dead lane asks for -1)
      if (lane < nb) {
        if (by_edge)
          key = eids ? eids[p0 + lane] : static_cast<Idx>(p0 + lane);
        else
          key = indices[p0 + lane];
      }
      // lane = column(s): scan the batch's keys from the LAST edge to the first, so the first match stays — one
      // broadcast per edge and a compare + select per column; the per-edge ballot + park-in-lane-j form of the
      // first version cost ten instructions and made this kernel 5 ms at 62 M edges
      int pos[CPL];
#pragma unroll
      for (int k = 0; k < CPL; ++k) pos[k] = -1;
#pragma unroll
      for (int j8 = 56; j8 >= 0; j8 -= 8) {
        if (j8 < nb) {  // (uniform) lanes >= nb hold the never-matching key: whole groups of eight, lane numbers constant
#pragma unroll
          for (int j = 7; j >= 0; --j) {
            const Idx kj = bcast_lane(key, j8 + j);
#pragma unroll
            for (int k = 0; k < CPL; ++k) pos[k] = (a[k] == kj) ? j8 + j : pos[k];
          }
        }
      }
      // column -> edge: every winning column ORs its bit into its edge's word (one LDS atomic per lane), then lane j
      // picks up the word(s) of edge j.  One wave's LDS operations execute in order; the wave barriers keep the
      // compiler from re-ordering them.
      if constexpr (MODE == 2) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) found[k] |= pos[k] >= 0;
      } else {
#pragma unroll
        for (int k = 0; k < CPL; ++k) wb[k][lane] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          if (!found[k] && pos[k] >= 0) atomicOr(reinterpret_cast<unsigned long long*>(wb[k] + pos[k]), 1ull << lane);
          found[k] |= pos[k] >= 0;
        }
        __builtin_amdgcn_wave_barrier();
        uint64_t m[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) m[k] = wb[k][lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < nb) {
          W* dst = mask + (p0 + lane) * words + c * CPL * WPC;
#pragma unroll
          for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int w = 0; w < WPC; ++w)
              if ((c * CPL + k) * WPC + w < words) dst[k * WPC + w] = static_cast<W>(m[k] >> (w * BITS));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      if (live[k] && !found[k]) {
        // nothing of the row claimed this element.  From the library's own forward that means arg = 0 (the value an
        // element without a winner gets): summed here in row order, added to dX[0] by spmm_cmp_leak_kernel in slot
        // order — the same bits on every run.  Any other target (a hand-made arg): one atomic add, like the scatter.
        const int64_t a64 = static_cast<int64_t>(a[k]);
        if (a64 == 0) {
          if constexpr (MODE != 2) leaked[k] += to_acc<DT>(dz[row * F + f[k]]);
        } else if (a64 > 0 && a64 < dx_rows) {
          if constexpr (MODE == 1)
            atomicOr(rare, 1u);
          else
            atomic_add_elem<DT>(dx + a64 * F + f[k], dz[row * F + f[k]]);
        }
      }
    }
  }
  if constexpr (MODE != 2) {
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (live[k]) leak[slot * F + f[k]] = leaked[k];
  }
}

// dX[0][k] += sum over the mask kernel's slots in a FIXED order: one workgroup per column, thread t sums slots
// t, t + 256, ... in order, then a fixed tree over the 256 partial sums.
template <typename DT>
__global__ __launch_bounds__(256) void spmm_cmp_leak_kernel(const typename Acc<DT>::type* __restrict__ leak,
                                                            int64_t slots, int F, DT* __restrict__ dx) {
  using A = typename Acc<DT>::type;
  __shared__ A part[256];
  const int k = blockIdx.x;
  A s = A(0);
  for (int64_t i = threadIdx.x; i < slots; i += 256) s += leak[i * F + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) dx[k] = from_acc<DT>(to_acc<DT>(dx[k]) + part[0]);
}

// Launch shape of the mask kernel: every wave keeps ONE chunk of 64 CPL columns (two columns per lane above 64 columns, four above 128);
// at most 8192 waves.
struct CmpMaskShape {
  int64_t chunks, slots;
  unsigned blocks;
  int cpl;
};
inline CmpMaskShape cmp_mask_shape(int64_t rows, int64_t F) {
  CmpMaskShape m;
  m.cpl = F > 128 ? 4 : (F > 64 ? 2 : 1);
  m.chunks = (F + 64 * m.cpl - 1) / (64 * m.cpl);
  int64_t slots = std::min<int64_t>(rows, std::max<int64_t>(1, 8192 / m.chunks));
  m.blocks = static_cast<unsigned>((slots * m.chunks + 3) / 4);
  m.slots = static_cast<int64_t>(m.blocks) * 4 / m.chunks;  // what the kernel derives from its grid
  return m;
}
inline size_t cmp_mask_align(size_t x) { return (x + 255) / 256 * 256; }

template <typename Idx, typename DT>
int run_spmm_cmp_mask(const void* indptr, const void* indices, const void* eids, const void* arg, int by_edge,
                      int64_t rows, int64_t nnz, int64_t F, void* mask, const void* dz, void* dx, int64_t dx_rows,
                      hipStream_t s) {
  typedef typename std::conditional<sizeof(DT) == 2, uint16_t, typename std::conditional<sizeof(DT) == 4, uint32_t, uint64_t>::type>::type W;
  using A = typename Acc<DT>::type;
  const int bits = 8 * static_cast<int>(sizeof(W));
  const int words = static_cast<int>((F + bits - 1) / bits);
  const CmpMaskShape m = cmp_mask_shape(rows, F);
  // the per-slot partial sums of unclaimed elements live BEHIND the mask words in the caller's buffer
  // (dgla_spmm_cmp_mask_bytes): no allocation in here, so the call can be captured in a hipGraph
  A* leak = reinterpret_cast<A*>(static_cast<char*>(mask) + cmp_mask_align(sizeof(W) * static_cast<size_t>(nnz) * words));
  uint32_t* rare = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(leak) + cmp_mask_align(sizeof(A) * static_cast<size_t>(m.slots) * F));
  const int mode = (by_edge & DGLA_CMP_MASK_FINISH) ? 2 : ((by_edge & DGLA_CMP_MASK_DEFER) ? 1 : 0);
  by_edge &= 1;
  if (mode == 1) DGLA_CHECK_HIP(hipMemsetAsync(rare, 0, sizeof(uint32_t), s));
#define DGLA_CMP_MASK_LAUNCH(CPLV, MODEV)                                                                                       \
  hipLaunchKernelGGL((spmm_cmp_mask_kernel<Idx, W, DT, CPLV, MODEV>), dim3(m.blocks), dim3(256), 0, s,                          \
                     static_cast<const Idx*>(indptr), static_cast<const Idx*>(indices), static_cast<const Idx*>(eids),          \
                     static_cast<const Idx*>(arg), by_edge, rows, static_cast<int>(F), words, static_cast<W*>(mask),            \
                     static_cast<const DT*>(dz), static_cast<DT*>(dx), dx_rows, leak, rare)
#define DGLA_CMP_MASK_CPL(MODEV)                      \
  if (m.cpl == 4) {                                   \
    DGLA_CMP_MASK_LAUNCH(4, MODEV);                   \
  } else if (m.cpl == 2) {                            \
    DGLA_CMP_MASK_LAUNCH(2, MODEV);                   \
  } else {                                            \
    DGLA_CMP_MASK_LAUNCH(1, MODEV);                   \
  }
  if (mode == 0) {
    DGLA_CMP_MASK_CPL(0)
  } else if (mode == 1) {
    DGLA_CMP_MASK_CPL(1)
  } else {
    DGLA_CMP_MASK_CPL(2)
  }
#undef DGLA_CMP_MASK_CPL
#undef DGLA_CMP_MASK_LAUNCH
  if (mode != 1)   // (deferred: the sums wait for the FINISH call behind the g-SpMM)
    hipLaunchKernelGGL((spmm_cmp_leak_kernel<DT>), dim3(static_cast<unsigned>(F)), dim3(256), 0, s, leak,
                       std::min<int64_t>(m.slots, rows), static_cast<int>(F), static_cast<DT*>(dx));
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" {

size_t dgla_segment_reduce_workspace_bytes(const char* reduce, int idtype_bits, dgla_dtype dtype,
                                           const dgla_tensor* feat, int64_t num_segments,
                                           const dgla_tensor* out) {
  SpmmLaunch L{};
  static const int64_t dummy = 0;
  if (build_segment_launch(reduce, idtype_bits, dtype, feat, &dummy, num_segments, out, &L)) return 0;
  if (narrow_reduce_eligible(L))   // (behind the merge kernel's share: the plan at the front of the workspace stays valid)
    return ((spmm_csr_workspace_f32(L) + 255) & ~static_cast<size_t>(255)) + narrow_reduce_workspace_bytes(L);
  switch (dtype) {
    case DGLA_F32: return spmm_csr_workspace_f32(L);
    case DGLA_F64: return spmm_csr_workspace_f64(L);
    case DGLA_F16: return spmm_csr_workspace_f16(L);
    case DGLA_BF16: return spmm_csr_workspace_bf16(L);
  }
  return 0;
}

int dgla_segment_reduce(const char* reduce, int idtype_bits, dgla_dtype dtype,
                        const dgla_tensor* feat, const void* offsets, int64_t num_segments,
                        const dgla_tensor* out, void* arg, void* workspace,
                        size_t workspace_bytes, uint32_t flags, void* hip_stream) {
  SpmmLaunch L{};
  if (build_segment_launch(reduce, idtype_bits, dtype, feat, offsets, num_segments, out, &L))
    return -1;
  if (L.red != kSum && !arg) return sfail("arg is required for max/min");
  if (num_segments == 0 || L.out_len == 0) return 0;
  L.arg_u = nullptr;
  L.arg_e = arg;
  L.accumulate = false;
  L.plan_valid = (flags & DGLA_PLAN_VALID) != 0;
  L.stream = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(L.stream, L.out);
  // scratch: the caller's, or a stream-ordered allocation released after the launches
  void* owned = nullptr;
  if (!workspace) {
    const size_t need = dgla_segment_reduce_workspace_bytes(reduce, idtype_bits, dtype, feat,
                                                            num_segments, out);
    DGLA_CHECK_HIP(hipMallocAsync(&owned, need ? need : 256, L.stream));
    workspace = owned;
    workspace_bytes = need;
    L.plan_valid = false;
  }
  L.workspace = workspace;
  L.workspace_bytes = workspace_bytes;
  int rc = -1;
  bool narrow = false;
  if (narrow_reduce_eligible(L)) {
    const size_t front = (spmm_csr_workspace_f32(L) + 255) & ~static_cast<size_t>(255);
    if (workspace_bytes >= front + narrow_reduce_workspace_bytes(L)) {
      narrow = true;
      rc = 0;
      if (!L.plan_valid) {   // (the caller may say DGLA_PLAN_VALID next time: leave the merge plan behind as well)
        SpmmLaunch P = L;
        P.prepare_only = true;
        rc = launch_spmm_csr_f32(P);
      }
      if (rc == 0) rc = launch_narrow_reduce(L, static_cast<char*>(workspace) + front);
    }
  }
  if (!narrow) switch (dtype) {
    case DGLA_F32: rc = launch_spmm_csr_f32(L); break;
    case DGLA_F64: rc = launch_spmm_csr_f64(L); break;
    case DGLA_F16: rc = launch_spmm_csr_f16(L); break;
    case DGLA_BF16: rc = launch_spmm_csr_bf16(L); break;
  }
  if (owned) {
    const hipError_t e = hipFreeAsync(owned, L.stream);
    if (e != hipSuccess && rc == 0) return sfail(std::string("hipFreeAsync: ") + hipGetErrorString(e));
  }
  return rc;
}

int dgla_scatter_add(int idtype_bits, dgla_dtype dtype, const dgla_tensor* feat, const void* idx,
                     const dgla_tensor* out, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return sfail("unsupported feature dtype");
  if (!feat || !out || feat->ndim < 1 || out->ndim < 1 || !feat->shape || !out->shape)
    return sfail("feat / out is null");
  if (row_len(feat) != row_len(out)) return sfail("feat and out have different feature shapes");
  const int64_t n = feat->shape[0], dim = row_len(out);
  if (n == 0 || dim == 0) return 0;
  if (!feat->data || !out->data || !idx) return sfail("feat / idx / out data is null");
  if (dim > 0x7fffffffLL / 4) return sfail("feature length too large");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out->data);
  const int64_t out_rows = out->shape[0];
  if (n * dim >= kScatterSortMinElems && out_rows > 0) return scatter_add_sorted(idtype_bits, dtype, feat, idx, out, s);
  DGLA_IDX_DTYPE_SWITCH(idtype_bits, dtype, run_scatter_add, feat->data, idx, out->data, n, dim, s);
  return sfail("unsupported feature dtype");
}

int dgla_update_grad_minmax(int idtype_bits, dgla_dtype dtype, const dgla_tensor* feat,
                            const void* idx, const void* idx_type, int64_t type,
                            const dgla_tensor* out, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return sfail("unsupported feature dtype");
  if (!feat || !out || feat->ndim < 1 || out->ndim < 1 || !feat->shape || !out->shape)
    return sfail("feat / out is null");
  if (row_len(feat) != row_len(out)) return sfail("feat and out have different feature shapes");
  const int64_t n = feat->shape[0], dim = row_len(out);
  if (n == 0 || dim == 0 || out->shape[0] == 0) return 0;
  if (!feat->data || !out->data || !idx || !idx_type) return sfail("feat / idx / idx_type / out data is null");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out->data);
  DGLA_IDX_DTYPE_SWITCH(idtype_bits, dtype, run_update_grad_minmax, feat->data, idx, idx_type, out->data,
                        n, dim, type, out->shape[0], s);
  return sfail("unsupported feature dtype");
}

int dgla_spmm_cmp_backward(int idtype_bits, dgla_dtype dtype, const dgla_tensor* dz, const void* arg,
                           const dgla_tensor* other, const void* arg_other, int64_t other_group,
                           const dgla_tensor* out, int atomic, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return sfail("unsupported feature dtype");
  if (!dz || !out || dz->ndim < 1 || out->ndim < 1 || !dz->shape || !out->shape) return sfail("dz / out is null");
  if (row_len(dz) != row_len(out)) return sfail("dz and out have different feature shapes");
  const int64_t n = dz->shape[0], dim = row_len(out);
  if (n == 0 || dim == 0 || out->shape[0] == 0) return 0;
  if (!dz->data || !out->data || !arg) return sfail("dz / arg / out data is null");
  int64_t other_len = 1;
  if (other && other->data) {
    if (!arg_other) return sfail("arg_other is required with `other`");
    if (other->ndim < 1 || !other->shape) return sfail("other has no shape");
    other_len = row_len(other);
    if (other_group < 1 || other_group > dim || other_len < 1) return sfail("other_group out of range");
  }
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out->data);
  const void* od = (other && other->data) ? other->data : nullptr;
  DGLA_IDX_DTYPE_SWITCH(idtype_bits, dtype, run_spmm_cmp_backward, dz->data, arg, od, arg_other, out->data, n, dim,
                        other_len, other_group < 1 ? 1 : other_group, out->shape[0], atomic != 0, s);
  return sfail("unsupported feature dtype");
}

int64_t dgla_spmm_cmp_mask_words(dgla_dtype dtype, int64_t feat_len) {
  const int64_t bits = dtype == DGLA_F64 ? 64 : (dtype == DGLA_F32 ? 32 : 16);
  return (feat_len + bits - 1) / bits;
}

size_t dgla_spmm_cmp_mask_bytes(dgla_dtype dtype, int64_t num_rows, int64_t nnz, int64_t feat_len) {
  if (dtype < DGLA_F32 || dtype > DGLA_BF16 || num_rows < 0 || nnz < 0 || feat_len <= 0) return 0;
  const size_t wsize = dtype == DGLA_F64 ? 8 : (dtype == DGLA_F32 ? 4 : 2);
  const size_t asize = dtype == DGLA_F64 ? 8 : 4;  // accumulator type of the partial sums
  const CmpMaskShape m = cmp_mask_shape(num_rows > 0 ? num_rows : 1, feat_len);
  return cmp_mask_align(wsize * static_cast<size_t>(nnz) * dgla_spmm_cmp_mask_words(dtype, feat_len)) +
         cmp_mask_align(asize * static_cast<size_t>(m.slots) * feat_len) + 256;   // (+ the word of the deferred mode)
}

int dgla_spmm_cmp_mask(const dgla_csr* csr, dgla_dtype dtype, const void* arg, int by_edge, const dgla_tensor* dz,
                       void* mask, const dgla_tensor* dx, void* hip_stream) {
  if (!csr) return sfail("csr is null");
  if (csr->idtype_bits != 32 && csr->idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return sfail("unsupported feature dtype");
  if (!dz || !dx || dz->ndim < 1 || dx->ndim < 1 || !dz->shape || !dx->shape) return sfail("dz / dx is null");
  if (row_len(dz) != row_len(dx)) return sfail("dz and dx have different feature shapes");
  if (dz->shape[0] != csr->num_rows) return sfail("dz must have one row per row of the CSR");
  const int64_t F = row_len(dz);
  if (csr->num_rows == 0 || F == 0 || dx->shape[0] == 0) return 0;
  if (F > (1 << 20)) return sfail("feature rows too long");
  if (!csr->indptr || (csr->nnz && !csr->indices) || !arg || !dz->data || !dx->data || !mask)
    return sfail("csr / arg / dz / dx / mask data is null");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, dx->data);
  DGLA_IDX_DTYPE_SWITCH(csr->idtype_bits, dtype, run_spmm_cmp_mask, csr->indptr, csr->indices, csr->data, arg, by_edge,
                        csr->num_rows, csr->nnz, F, mask, dz->data, dx->data, dx->shape[0], s);
  return sfail("unsupported feature dtype");
}

int dgla_backward_segment_cmp(int idtype_bits, dgla_dtype dtype, const dgla_tensor* feat,
                              const void* arg, const dgla_tensor* out, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return sfail("unsupported feature dtype");
  if (!feat || !out || feat->ndim < 1 || out->ndim < 1 || !feat->shape || !out->shape)
    return sfail("feat / out is null");
  if (row_len(feat) != row_len(out)) return sfail("feat and out have different feature shapes");
  const int64_t n = feat->shape[0], dim = row_len(out);
  if (n == 0 || dim == 0) return 0;
  if (!feat->data || !out->data || !arg) return sfail("feat / arg / out data is null");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out->data);
  DGLA_IDX_DTYPE_SWITCH(idtype_bits, dtype, run_bwd_segment_cmp, feat->data, arg, out->data, n, dim, s);
  return sfail("unsupported feature dtype");
}

}  // extern "C"
