// Device side of the multi-GPU exchange step: row pack kernels for the halo
// all-to-all and the NDArrayPartition index maps.
//
// Reference: python/dgl/cuda/nccl.py:7-183 (sparse_all_to_all_push / _pull: `value[perm]`,
// `value[resp_idx]`, `return_value[perm] = req_value` are torch index kernels there) and
// src/partition/cuda/partition_op.cu (MapToLocal / MapToGlobal / GeneratePermutation for the
// remainder and range partitions; src/partition/ndarray_partition.cc:30-230).
//
// gather_rows: dst[i, :] = src[idx[i], :] with 16-byte lane accesses and K rows-pieces in
// flight per lane — the pack step in front of the all-to-all (and, read the other way round,
// the un-permute after it).  The scatter-add direction (gradient push) is dgla_scatter_add
// (segment.hip).  Index maps: one thread per index, closed-form for the remainder partition,
// a binary search over the (<= a few hundred) range boundaries otherwise.
#include "../../include/dgl_amd.h"

#include <cstring>

#include "common.h"

namespace dgla {
namespace {

int xfail(const std::string& m) {
  last_error() = m;
  return -1;
}

template <typename Idx, typename Piece, int K>
__global__ __launch_bounds__(256) void gather_rows_kernel(const Piece* __restrict__ src,
                                                          const Idx* __restrict__ idx,
                                                          Piece* __restrict__ dst, int64_t n,
                                                          int pieces, unsigned magic) {
  const int64_t total = n * pieces;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * (256 * K); base < total;
       base += static_cast<int64_t>(gridDim.x) * (256 * K)) {
    const int64_t r0 = base / pieces;  // block-uniform
    const unsigned j0 = static_cast<unsigned>(base - r0 * pieces);
    Piece v[K];
    int64_t at[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int64_t i = base + k * 256 + threadIdx.x;
      if (i >= total) i = total - 1;
      // row of piece i: block-local piece number / pieces by multiply-high (exact: the numbers
      // stay below 2^16, magic = ceil(2^32 / pieces))
      const unsigned loc = j0 + static_cast<unsigned>(i - base);
      const unsigned dr = pieces == 1 ? loc : __umulhi(loc, magic);  // magic wraps to 0 for pieces == 1
      const unsigned j = loc - dr * static_cast<unsigned>(pieces);
      at[k] = static_cast<int64_t>(idx[r0 + dr]) * pieces + j;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = src[at[k]];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      if (i < total) {
        if constexpr (K > 4)
          __builtin_nontemporal_store(v[k], dst + i);  // short rows: the output is a pure stream, keep L2 for the reads
        else
          dst[i] = v[k];
      }
    }
  }
}

template <typename Idx, typename Piece>
int run_gather(const void* src, const void* idx, void* dst, int64_t n, int64_t row_bytes,
               hipStream_t s) {
  const int64_t pieces = row_bytes / static_cast<int64_t>(sizeof(Piece));
  if (pieces > 0xffff) return xfail("gather_rows: rows longer than 65535 pieces are not supported");
  const int64_t total = n * pieces;
  const unsigned magic = 0xFFFFFFFFu / static_cast<unsigned>(pieces) + 1u;
  if (pieces <= 4 && n >= (int64_t(1) << 20)) {
    // a permutation of SHORT rows (edge tensors: 16 - 64 bytes a row, tens of millions of rows) is bound by the
    // number of independent scattered reads in flight, not by bytes: twice the loads per thread (round 5: the edge
    // softmax hands edge-id order out through this kernel)
    constexpr int K = 8;
    const unsigned blocks =
        static_cast<unsigned>(std::min<int64_t>((total + 256 * K - 1) / (256 * K), int64_t(1) << 20));
    hipLaunchKernelGGL((gather_rows_kernel<Idx, Piece, K>), dim3(blocks), dim3(256), 0, s,
                       static_cast<const Piece*>(src), static_cast<const Idx*>(idx),
                       static_cast<Piece*>(dst), n, static_cast<int>(pieces), magic);
  } else {
    constexpr int K = 4;
    const unsigned blocks =
        static_cast<unsigned>(std::min<int64_t>((total + 256 * K - 1) / (256 * K), int64_t(1) << 20));
    hipLaunchKernelGGL((gather_rows_kernel<Idx, Piece, K>), dim3(blocks), dim3(256), 0, s,
                       static_cast<const Piece*>(src), static_cast<const Idx*>(idx),
                       static_cast<Piece*>(dst), n, static_cast<int>(pieces), magic);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// dst[idx[i], :] = src[i, :] — the un-permute after the all-to-all of a pull
// (`return_value[perm] = req_value`, python/dgl/cuda/nccl.py:180-181).  The source rows and the
// index stream are read once (non-temporal); blocks are renumbered so that each XCD walks ONE
// contiguous eighth of the source (block b runs on XCD b % 8), which keeps the partially written
// destination lines of neighbouring source rows in one L2.
template <typename Idx, typename Piece, int K>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const Piece* __restrict__ src,
                                                           const Idx* __restrict__ idx,
                                                           Piece* __restrict__ dst, int64_t n,
                                                           int pieces, unsigned magic) {
  const int64_t total = n * pieces;
  const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u;
  const unsigned x = blockIdx.x & 7u, i8 = blockIdx.x >> 3;
  const unsigned blk = x * q + (x < r ? x : r) + i8;
  const int64_t base = static_cast<int64_t>(blk) * (256 * K);
  if (base >= total) return;
  const int64_t r0 = base / pieces;  // block-uniform
  const unsigned j0 = static_cast<unsigned>(base - r0 * pieces);
  Piece v[K];
  int64_t at[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    int64_t i = base + k * 256 + threadIdx.x;
    if (i >= total) i = total - 1;
    v[k] = __builtin_nontemporal_load(src + i);
    const unsigned loc = j0 + static_cast<unsigned>(i - base);
    const unsigned dr = pieces == 1 ? loc : __umulhi(loc, magic);  // magic wraps to 0 for pieces == 1
    const unsigned j = loc - dr * static_cast<unsigned>(pieces);
    at[k] = static_cast<int64_t>(__builtin_nontemporal_load(idx + (r0 + dr))) * pieces + j;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < total) dst[at[k]] = v[k];
  }
}

template <typename Idx, typename Piece>
int run_scatter(const void* src, const void* idx, void* dst, int64_t n, int64_t row_bytes,
                hipStream_t s) {
  constexpr int K = 4;
  const int64_t pieces = row_bytes / static_cast<int64_t>(sizeof(Piece));
  if (pieces > 0xffff) return xfail("scatter_rows: rows longer than 65535 pieces are not supported");
  const int64_t total = n * pieces;
  const int64_t blocks = (total + 256 * K - 1) / (256 * K);
  if (blocks > 0x7fffffffLL) return xfail("scatter_rows: too many pieces for one launch");
  const unsigned magic = 0xFFFFFFFFu / static_cast<unsigned>(pieces) + 1u;
  hipLaunchKernelGGL((scatter_rows_kernel<Idx, Piece, K>), dim3(static_cast<unsigned>(blocks)), dim3(256),
                     0, s, static_cast<const Piece*>(src), static_cast<const Idx*>(idx),
                     static_cast<Piece*>(dst), n, static_cast<int>(pieces), magic);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// mode 0: remainder (part = id % k, local = id / k); mode 1: range (part = the range holding id,
// local = id - range[part]).  Either output may be NULL.
template <typename Idx>
__global__ __launch_bounds__(256) void partition_map_kernel(int mode, int num_parts,
                                                            const Idx* __restrict__ range,
                                                            const Idx* __restrict__ idx, int64_t n,
                                                            Idx* __restrict__ part_out,
                                                            Idx* __restrict__ local_out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
    const Idx g = idx[i];
    Idx p, l;
    if (mode == 0) {
      p = g % static_cast<Idx>(num_parts);
      l = g / static_cast<Idx>(num_parts);
    } else {
      int lo = 0, hi = num_parts - 1;  // largest p with range[p] <= g
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (range[mid] <= g)
          lo = mid;
        else
          hi = mid - 1;
      }
      p = static_cast<Idx>(lo);
      l = g - range[lo];
    }
    if (part_out) part_out[i] = p;
    if (local_out) local_out[i] = l;
  }
}

template <typename Idx>
__global__ __launch_bounds__(256) void partition_to_global_kernel(int mode, int num_parts,
                                                                  const Idx* __restrict__ range,
                                                                  const Idx* __restrict__ local,
                                                                  int64_t n, int part_id,
                                                                  Idx* __restrict__ out) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const Idx base = mode == 0 ? Idx(0) : range[part_id];
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride)
    out[i] = mode == 0 ? local[i] * static_cast<Idx>(num_parts) + static_cast<Idx>(part_id)
                       : local[i] + base;
}

unsigned grid_for(int64_t n) {
  return static_cast<unsigned>(std::min<int64_t>((n + 255) / 256, 256 * 32));
}


// ---- peer-mapped halo exchange (round 4; VERDICT r3 Next #2a) ---------------------------------------------
// The reference pulls halo rows with three NCCL all-to-alls per call (python/dgl/cuda/nccl.py:98-183); round 3
// here packed rows into a send buffer and called all_to_all_single.  With every GPU of the node mapped into
// every process (hipIpcOpenMemHandle over xGMI) the pack kernel can write the rows where the CONSUMER reads
// them: no send buffer, no collective, no host-side wait.
//   * segment = (destination rank, chunk): rows serve_rows[begin, end) of the local features go to
//     dst + (i - begin) * row_bytes inside that rank's halo buffer;
//   * ONE launch covers all segments (blocks of 64 rows; an empty segment still gets a block so that its flag
//     moves); a block's last thread publishes with a system-scope release, the LAST block of a segment
//     (device-scope arrival counter) stores the step number into the segment's flag in the consumer's memory;
//   * the consumer runs dgla_peer_wait — one wavefront polling its flags with system-scope loads and a
//     bounded spin (a stuck peer turns into an error code, not a hung GPU) — in front of the launch that reads
//     the halo rows; the kernel boundary behind it carries the system-scope acquire.
//   Flags are monotone step counters: no reset, no race between "set" and "clear".  Two halo buffers per rank
//   (step parity) cover the write-after-read hazard: a peer starts writing buffer s % 2 for step s + 2 only
//   after it has consumed this rank's step-(s + 1) rows, which this rank sent after reading buffer s % 2.
struct PeerSegment {
  int64_t row_begin, row_end;  // range of serve_rows
  char* dst;                   // where row_begin lands (peer memory)
  uint64_t* flag;              // the consumer's flag for this (owner, chunk)
  int64_t blk_begin;           // first block of this segment in the launch
};

template <typename Piece>
__global__ __launch_bounds__(256) void peer_push_kernel(const char* __restrict__ x, const int64_t* __restrict__ serve_rows,
                                                        const PeerSegment* __restrict__ segs, int nseg,
                                                        int64_t row_bytes, uint64_t epoch, unsigned* __restrict__ arrive) {
  // which segment does this block belong to?  (<= a few dozen segments: linear scan by one thread)
  __shared__ int s_seg;
  if (threadIdx.x == 0) {
    int sg = 0;
    while (sg + 1 < nseg && static_cast<int64_t>(blockIdx.x) >= segs[sg + 1].blk_begin) ++sg;
    s_seg = sg;
  }
  __syncthreads();
  const PeerSegment sg = segs[s_seg];
  const int64_t nblk = (s_seg + 1 < nseg ? segs[s_seg + 1].blk_begin : static_cast<int64_t>(gridDim.x)) - sg.blk_begin;
  const int pieces = static_cast<int>(row_bytes / sizeof(Piece));
  const int64_t r0 = sg.row_begin + (static_cast<int64_t>(blockIdx.x) - sg.blk_begin) * 64;
  const int64_t r1 = r0 + 64 < sg.row_end ? r0 + 64 : sg.row_end;
  const int64_t total = (r1 > r0 ? r1 - r0 : 0) * pieces;
  constexpr int K = 4;
  for (int64_t base = 0; base < total; base += 256 * K) {
    Piece v[K];
    int64_t at[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int64_t i = base + k * 256 + threadIdx.x;
      if (i >= total) i = total - 1;
      const int64_t r = i / pieces;
      at[k] = r * pieces + (i - r * pieces);
      v[k] = reinterpret_cast<const Piece*>(x + serve_rows[r0 + r] * row_bytes)[i - r * pieces];
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (base + k * 256 + threadIdx.x < total)
        reinterpret_cast<Piece*>(sg.dst + (r0 - sg.row_begin) * row_bytes)[at[k]] = v[k];
  }
  __threadfence_system();  // this thread's rows are visible to the peer before anything that follows
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(arrive + s_seg, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == static_cast<unsigned>(nblk)) {  // every block of the segment has published
      __hip_atomic_store(arrive + s_seg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next step
      __threadfence_system();
      __hip_atomic_store(sg.flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(64) void peer_wait_kernel(const uint64_t* __restrict__ flags, int n, uint64_t epoch,
                                                       int* __restrict__ status, int64_t max_spins) {
  for (int i = threadIdx.x; i < n; i += 64) {
    int64_t spins = 0;
    while (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > max_spins) {
        atomicExch(status, 1);  // a peer never wrote: reported by dgla_peer_status, the GPU moves on
        break;
      }
    }
  }
  __threadfence_system();
}

}  // namespace
}  // namespace dgla

using namespace dgla;

extern "C" {

int dgla_gather_rows(int idtype_bits, const void* src, const void* idx, int64_t n,
                     int64_t row_bytes, void* dst, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return xfail("idtype must be int32 or int64");
  if (n < 0 || row_bytes < 0) return xfail("negative size");
  if (n == 0 || row_bytes == 0) return 0;
  if (!src || !idx || !dst) return xfail("gather_rows: null pointer");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, dst);
  const uintptr_t both = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
  const bool v16 = row_bytes % 16 == 0 && both % 16 == 0;
  const bool v4 = row_bytes % 4 == 0 && both % 4 == 0;
#define DGLA_GR(IDX)                                                              \
  if (v16) return run_gather<IDX, u32x4>(src, idx, dst, n, row_bytes, s);          \
  if (v4) return run_gather<IDX, uint32_t>(src, idx, dst, n, row_bytes, s);        \
  return run_gather<IDX, unsigned char>(src, idx, dst, n, row_bytes, s)
  if (idtype_bits == 32) {
    DGLA_GR(int32_t);
  }
  DGLA_GR(int64_t);
#undef DGLA_GR
}

int dgla_scatter_rows(int idtype_bits, const void* src, const void* idx, int64_t n,
                      int64_t row_bytes, void* dst, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return xfail("idtype must be int32 or int64");
  if (n < 0 || row_bytes < 0) return xfail("negative size");
  if (n == 0 || row_bytes == 0) return 0;
  if (!src || !idx || !dst) return xfail("scatter_rows: null pointer");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, dst);
  const uintptr_t both = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
  const bool v16 = row_bytes % 16 == 0 && both % 16 == 0;
  const bool v4 = row_bytes % 4 == 0 && both % 4 == 0;
#define DGLA_SR(IDX)                                                               \
  if (v16) return run_scatter<IDX, u32x4>(src, idx, dst, n, row_bytes, s);          \
  if (v4) return run_scatter<IDX, uint32_t>(src, idx, dst, n, row_bytes, s);        \
  return run_scatter<IDX, unsigned char>(src, idx, dst, n, row_bytes, s)
  if (idtype_bits == 32) {
    DGLA_SR(int32_t);
  }
  DGLA_SR(int64_t);
#undef DGLA_SR
}

int dgla_partition_map(int idtype_bits, int mode, int num_parts, const void* range, const void* idx,
                       int64_t n, void* part_out, void* local_out, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return xfail("idtype must be int32 or int64");
  if (mode != 0 && mode != 1) return xfail("partition mode must be 0 (remainder) or 1 (range)");
  if (num_parts < 1) return xfail("num_parts must be positive");
  if (mode == 1 && !range) return xfail("range partition needs the range array");
  if (n == 0) return 0;
  if (n < 0 || !idx) return xfail("partition_map: bad index array");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, idx);
  if (idtype_bits == 32)
    hipLaunchKernelGGL(partition_map_kernel<int32_t>, dim3(grid_for(n)), dim3(256), 0, s, mode,
                       num_parts, static_cast<const int32_t*>(range),
                       static_cast<const int32_t*>(idx), n, static_cast<int32_t*>(part_out),
                       static_cast<int32_t*>(local_out));
  else
    hipLaunchKernelGGL(partition_map_kernel<int64_t>, dim3(grid_for(n)), dim3(256), 0, s, mode,
                       num_parts, static_cast<const int64_t*>(range),
                       static_cast<const int64_t*>(idx), n, static_cast<int64_t*>(part_out),
                       static_cast<int64_t*>(local_out));
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int dgla_partition_to_global(int idtype_bits, int mode, int num_parts, const void* range,
                             const void* local_idx, int64_t n, int part_id, void* out,
                             void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return xfail("idtype must be int32 or int64");
  if (mode != 0 && mode != 1) return xfail("partition mode must be 0 (remainder) or 1 (range)");
  if (num_parts < 1 || part_id < 0 || part_id >= num_parts) return xfail("invalid part id");
  if (mode == 1 && !range) return xfail("range partition needs the range array");
  if (n == 0) return 0;
  if (n < 0 || !local_idx || !out) return xfail("partition_to_global: bad arrays");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out);
  if (idtype_bits == 32)
    hipLaunchKernelGGL(partition_to_global_kernel<int32_t>, dim3(grid_for(n)), dim3(256), 0, s,
                       mode, num_parts, static_cast<const int32_t*>(range),
                       static_cast<const int32_t*>(local_idx), n, part_id,
                       static_cast<int32_t*>(out));
  else
    hipLaunchKernelGGL(partition_to_global_kernel<int64_t>, dim3(grid_for(n)), dim3(256), 0, s,
                       mode, num_parts, static_cast<const int64_t*>(range),
                       static_cast<const int64_t*>(local_idx), n, part_id,
                       static_cast<int64_t*>(out));
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

/* ---- peer-mapped halo exchange (see the comment above peer_push_kernel) ---- */
int dgla_peer_alloc(size_t bytes, int kind, void** out) {
  if (!out) return xfail("dgla_peer_alloc: null result");
  *out = nullptr;
  if (bytes == 0) bytes = 256;
  // kind 1 / 2: memory a REMOTE GPU's stores become visible in while a kernel here polls it (fine-grained / uncached).
  // There is NO fall-back to plain hipMalloc: coarse-grained memory is not coherent for a peer's writes
  // during a kernel — a wait kernel polling it sees stale flags, a halo row read from it may be an old one.  A failed
  // allocation is an error the caller agrees on over the process group (every rank then takes the all-to-all path).
  // kind 0 (plain hipMalloc) exists for ranks that share ONE device (the single-GPU tests); the Python side refuses it
  // when the ranks sit on different devices.
  hipError_t e = hipErrorInvalidValue;
  if (kind == 0) e = hipMalloc(out, bytes);
  if (kind == 1) e = hipExtMallocWithFlags(out, bytes, hipDeviceMallocFinegrained);
  if (kind == 2) e = hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return xfail(std::string("dgla_peer_alloc(kind ") + std::to_string(kind) + "): " + hipGetErrorString(e) +
                 " (no fall-back to coarse-grained memory: it is not coherent for peer writes)");
  }
  DGLA_CHECK_HIP(hipMemset(*out, 0, bytes));
  DGLA_CHECK_HIP(hipDeviceSynchronize());
  return 0;
}

int dgla_peer_free(void* ptr) {
  if (ptr) DGLA_CHECK_HIP(hipFree(ptr));
  return 0;
}

int dgla_ipc_export(void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handles travel as 64 bytes");
  if (!ptr || !handle64) return xfail("dgla_ipc_export: null argument");
  hipIpcMemHandle_t h;
  DGLA_CHECK_HIP(hipIpcGetMemHandle(&h, ptr));
  memcpy(handle64, &h, 64);
  return 0;
}

int dgla_ipc_import(const void* handle64, void** out) {
  if (!handle64 || !out) return xfail("dgla_ipc_import: null argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  DGLA_CHECK_HIP(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}

int dgla_ipc_release(void* ptr) {
  if (ptr) DGLA_CHECK_HIP(hipIpcCloseMemHandle(ptr));
  return 0;
}

int dgla_peer_push(const void* x_local, int64_t row_bytes, const int64_t* serve_rows, const void* segments,
                   int num_segments, int64_t num_blocks, uint64_t epoch, void* arrive, void* hip_stream) {
  if (num_segments <= 0 || num_blocks <= 0) return 0;
  if (!x_local || !segments || !arrive) return xfail("dgla_peer_push: null argument");
  if (row_bytes <= 0 || row_bytes % 4) return xfail("dgla_peer_push: rows must be a whole number of 4-byte words");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, x_local);
  const dim3 grid(static_cast<unsigned>(num_blocks));
  const bool wide = row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(x_local) & 15) == 0;
  if (wide)
    hipLaunchKernelGGL(peer_push_kernel<u32x4>, grid, dim3(256), 0, s, static_cast<const char*>(x_local), serve_rows,
                       static_cast<const PeerSegment*>(segments), num_segments, row_bytes, epoch,
                       static_cast<unsigned*>(arrive));
  else
    hipLaunchKernelGGL(peer_push_kernel<uint32_t>, grid, dim3(256), 0, s, static_cast<const char*>(x_local), serve_rows,
                       static_cast<const PeerSegment*>(segments), num_segments, row_bytes, epoch,
                       static_cast<unsigned*>(arrive));
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int dgla_peer_wait(const void* flags, int num_flags, uint64_t epoch, void* status, int64_t max_spins,
                   void* hip_stream) {
  if (num_flags <= 0) return 0;
  if (!flags || !status) return xfail("dgla_peer_wait: null argument");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, flags);
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, static_cast<const uint64_t*>(flags), num_flags, epoch,
                     static_cast<int*>(status), max_spins > 0 ? max_spins : (int64_t(1) << 24));
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
