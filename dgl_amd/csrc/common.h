// Shared definitions for the gfx950 g-SpMM / g-SDDMM kernels and their C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace dgla {

// ---- enums shared by host dispatch and kernels -------------------------------------
// Binary operators of the reference (src/array/cuda/functor.cuh:21-145).
enum Op : int { kAdd = 0, kSub = 1, kMul = 2, kDiv = 3, kCopyLhs = 4, kCopyRhs = 5, kDot = 6 };
// Reducers (functor.cuh:148-448).
enum Red : int { kSum = 0, kMax = 1, kMin = 2 };
// Element types of feature tensors (codes of include/dgl_amd.h).
enum DType : int { kF32 = 0, kF64 = 1, kF16 = 2, kBF16 = 3 };
// How the per-output-element operand offsets are formed (replaces the reference's
// per-call H2D copy of offset tables, src/array/cuda/macro.cuh:15-51).
enum Bcast : int {
  kBcNone = 0,      // lhs_off = rhs_off = k
  kBcRhsGroup = 1,  // lhs_off = k, rhs_off = k / rhs_group  (rhs is (.., H, 1) against (.., H, D))
  kBcGeneral = 2    // arbitrary numpy-style broadcast, offsets from dims/strides
};

constexpr int kMaxBcastDims = 6;

// Closed-form replacement for BcastOff::lhs_offset / rhs_offset (include/dgl/bcast.h).
// Output feature index k is decomposed over `dims` (last axis fastest); the operand
// offset is the dot product with that operand's strides (0 on broadcast axes).
struct BcastDims {
  int ndim;
  int32_t dims[kMaxBcastDims];
  int32_t lstride[kMaxBcastDims];
  int32_t rstride[kMaxBcastDims];
};

__host__ __device__ inline void bcast_offsets(const BcastDims& b, int k, int* lo, int* ro) {
  int l = 0, r = 0;
  for (int d = b.ndim - 1; d >= 0; --d) {
    const int idx = k % b.dims[d];
    k /= b.dims[d];
    l += idx * b.lstride[d];
    r += idx * b.rstride[d];
  }
  *lo = l;
  *ro = r;
}

constexpr __host__ __device__ bool op_uses_lhs(int op) { return op != kCopyRhs; }
constexpr __host__ __device__ bool op_uses_rhs(int op) { return op != kCopyLhs; }

// ---- storage types ------------------------------------------------------------------
struct bf16_t {
  uint16_t bits;
};
using f16_t = _Float16;

// Accumulator type: fp32 for fp16/bf16/fp32, fp64 for fp64
// (reference: src/runtime/cuda/cuda_common.h:193-218 accum_dtype).
template <typename T>
struct Acc {
  using type = float;
};
template <>
struct Acc<double> {
  using type = double;
};

template <typename T>
__device__ __forceinline__ typename Acc<T>::type to_acc(T v) {
  return static_cast<typename Acc<T>::type>(v);
}
template <>
__device__ __forceinline__ float to_acc<bf16_t>(bf16_t v) {
  return __uint_as_float(static_cast<uint32_t>(v.bits) << 16);
}

template <typename T>
__device__ __forceinline__ T from_acc(typename Acc<T>::type v) {
  return static_cast<T>(v);
}
template <>
__device__ __forceinline__ bf16_t from_acc<bf16_t>(float f) {
  // round-to-nearest-even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (one instruction)
  const __bf16 h = static_cast<__bf16>(f);
  bf16_t r;
  __builtin_memcpy(&r.bits, &h, 2);
  return r;
}

// Identity ("zero()") of the max / min reducers in the storage type.  fp16 uses the
// largest finite value like the reference (functor.cuh:287-289,392-394); the others +-inf.
template <typename T>
__device__ __forceinline__ typename Acc<T>::type red_identity(int red) {
  using A = typename Acc<T>::type;
  if (red == kSum) return A(0);
  const A inf = static_cast<A>(__builtin_huge_valf());
  return red == kMax ? -inf : inf;
}
template <>
__device__ __forceinline__ float red_identity<f16_t>(int red) {
  if (red == kSum) return 0.f;
  return red == kMax ? -65504.f : 65504.f;
}

// Aligned vector of VEC storage elements (one 16-byte — or smaller — global access).
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) VecT {
  T v[VEC];
};

// ---- launch description handed from the C-ABI layer to per-dtype launchers ----------
struct CsrView {
  int64_t num_rows, num_cols, nnz;
  int idbits;  // 32 or 64
  const void* indptr;
  const void* indices;
  const void* eids;  // nullptr: edge id == position
};

struct CooView {
  int64_t num_rows, num_cols, nnz;  // num_rows = #src nodes, num_cols = #dst nodes
  int idbits;
  const void* row;  // source ids
  const void* col;  // destination ids
  const void* eids;
};

struct SpmmLaunch {
  CsrView csr;
  int op, red, dtype;
  const void* ufeat;
  const void* efeat;
  void* out;
  void* arg_u;
  void* arg_e;
  int64_t out_len, lhs_len, rhs_len;
  int bcast;      // Bcast
  int rhs_group;  // kBcRhsGroup: consecutive outputs sharing one rhs element
  BcastDims bdims;
  // Stacked multi-relation form (dgla_spmm_csr_stacked): the CSR is the row-wise
  // concatenation of several relations; rel[j] names the relation of edge j and
  // ufeat_tab / efeat_tab (device arrays of device pointers) hold each relation's operands.
  const void* rel;               // uint8 [nnz] or nullptr (single relation)
  const void* const* ufeat_tab;  // [num_rel]
  const void* const* efeat_tab;  // [num_rel]
  int num_rel = 0;               // <= 256
  int arg_empty;    // arg value of an output element no edge won: 0 (g-SpMM), -1 (segment reduce)
  uint32_t tune;    // kTune* bits, from dgla_set_tuning()
  bool mean;        // reduce == sum: store sum / max(in_degree, 1)  (the `mean` reducer, fused)
  bool accumulate;  // out += result (reference semantics, spmm.cuh:528-534) vs out = result
  bool plan_valid;  // workspace already holds the merge plan of this CSR
  bool split_valid; // workspace already holds the split-row copy of this ufeat (DGLA_SPLIT_VALID)
  bool split_keep;  // static ufeat: use the split-row layout whatever the probe says (DGLA_SPLIT_KEEP)
  bool prepare_only;  // plan + side copy only (DGLA_PREPARE_ONLY)
  bool rhs_mask = false;  // mul + kBcRhsGroup: efeat holds bit masks (dgla_spmm_csr_masked)
  void* workspace;
  size_t workspace_bytes;
  hipStream_t stream;
};

struct SddmmLaunch {
  // exactly one of csr / coo is used
  bool use_coo;
  CsrView csr;  // rows = source nodes (out-edge CSR), as the reference hands to SDDMMCsr
  CooView coo;
  int op, dtype;
  const void* lhs;
  const void* rhs;
  void* out;
  int lhs_target, rhs_target;  // 0 = u (src), 1 = e, 2 = v (dst)
  int64_t out_len, lhs_len, rhs_len, reduce_size;
  int bcast;
  BcastDims bdims;
  hipStream_t stream;
};

// Tuning bits (dgla_set_tuning / dgla_get_tuning).  The SpMM bits change no result bit.  Round 4 removed
// NT_OUT (2), NT_IDX (4) — both measured neutral —, SPLIT_NT (32), SPLIT_CLASSIC (256), TAIL_PASS (512) and
// NT_STREAM (1024, now a fixed rule): the values of the surviving bits are unchanged.
enum Tune : uint32_t {
  kTuneXcd = 1u,    // XCD-contiguous unit order (one contiguous eighth of the merge path per L2)
  kTuneSplit = 8u,  // side copies of the rows' ragged ends when rows are not a whole number of 128-B lines
  kTuneGlds = 16u,  // segment_mm: LDS-direct (global_load_lds) slab rings instead of register staging
  kTuneSplitForce = 64u,  // split layouts whenever the shape allows, whatever the locality probe says
  kTuneMmF32 = 128u,      // segment_mm fp32: v_mfma_f32_32x32x2_f32 instead of the split-operand kernels
  kTuneNoGate = 4096u,     // masked g-SpMM (max / min backward): gather every row piece, wanted or not (A/B switch of the gated loads)
  kTuneNoStageW = 8192u,   // g-SpMM with scalar edge weights (u_mul_e + sum): read the weight per gather batch from global memory instead of staging the unit's weights in LDS (A/B switch)
  kTuneMmX3 = 2048u,       // segment_mm fp32, weights-stationary kernel: three bf16 terms instead of two scaled fp16 terms
};
constexpr uint32_t kTuneKnown = 1u | 8u | 16u | 64u | 128u | 2048u | 4096u | 8192u;
// Default: XCD-contiguous order (measured on C2: variant L -3 % time, variant U neutral); the
// Split layouts (C2, F = 100 fp32): the edge-layout copy costs 0.10 ms
// and the gather drops 4.86 -> 4.35 ms on variant U; on variant L the locality probe declines it only
// when >= 15/16 of the sampled edges are local.  The LDS-direct segment_mm loop is 23-34 % faster at
// every measured shape (profiles/r1/glds_ab.jsonl) -> on.
constexpr uint32_t kDefaultTuning = 1u | 8u | 16u;
uint32_t& tuning_flags();

// Merge-path geometry of the CSR SpMM (see spmm_csr.hip.h).
constexpr int kWaveItems = 512;     // rows + edges handled by one wavefront
constexpr int kWavesPerBlock = 4;   // 256-thread workgroups

inline int64_t spmm_num_waves(int64_t num_rows, int64_t nnz) {
  return (num_rows + nnz + kWaveItems - 1) / kWaveItems;
}

std::string& last_error();

// Optional HIP events recorded around the dominant (merge) kernel of dgla_spmm_csr, so a
// benchmark can time that kernel alone on the launch stream (dgla_spmm_set_profile_events).
struct ProfileEvents {
  hipEvent_t before = nullptr;
  hipEvent_t after = nullptr;
};
ProfileEvents& profile_events();

// Makes the device that owns the launch current for the lifetime of the guard and restores the
// previous one afterwards (the reference switches device in its DeviceAPI before every launch,
// src/runtime/cuda/cuda_device_api.cc SetDevice).  The owner is taken from the stream when one
// is given; the null stream belongs to whatever device is current, so there the owner is read
// off a device pointer of the call (normally the output).  Failures of the queries (no GPU in
// the process, host pointer) leave the current device alone: the launch reports the real error.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  DeviceGuard(hipStream_t stream, const void* device_ptr) {
    int cur = -1, owner = -1;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (stream != nullptr) {
      hipDevice_t d;
      if (hipStreamGetDevice(stream, &d) == hipSuccess) owner = static_cast<int>(d);
    } else if (device_ptr != nullptr) {
      hipPointerAttribute_t at;
      if (hipPointerGetAttributes(&at, device_ptr) == hipSuccess &&
          at.type == hipMemoryTypeDevice)
        owner = at.device;
      else
        (void)hipGetLastError();  // a host / unregistered pointer is not this guard's business
    }
    if (owner >= 0 && owner != cur && hipSetDevice(owner) == hipSuccess) {
      prev = cur;
      switched = true;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DGLA_CHECK_HIP(expr)                                                       \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::dgla::last_error() = std::string(#expr) + ": " + hipGetErrorString(_e);    \
      return -1;                                                                   \
    }                                                                              \
  } while (0)

}  // namespace dgla
