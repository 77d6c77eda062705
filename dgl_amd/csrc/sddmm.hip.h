// g-SDDMM for gfx950 (MI355X): edge-parallel kernels.
//
// Replaces, from scratch, SDDMMCooKernel / SDDMMCooTreeReduceKernel / SDDMMCsrKernel of the
// reference (src/array/cuda/sddmm.hip.h:97-256, hosts :287-362).
//
// Layout: a wavefront is split into G = 64 / LPE lane groups, LPE = lanes needed to cover one
// output row with 16-byte accesses; consecutive groups take consecutive edges so the COO
// index loads and (for eid == position) the output rows are contiguous across the wave.
// `dot` keeps the whole (H, D) operand row in the group: every lane multiplies its 16-byte
// piece and the D / VEC lanes of one head finish with xor-shuffles inside the 64-wide wave
// (the reference's tree kernel hard-codes 32-lane warps, sddmm.hip.h:88,146-185).
#pragma once
#include "common.h"

namespace dgla {

template <typename Idx>
struct SddmmParams {
  // COO (row = src, col = dst) or CSR (rows = src, indices = dst)
  const Idx* row;
  const Idx* col;
  const Idx* indptr;
  const Idx* eids;
  int64_t nnz, num_rows;
  const void* lhs;
  const void* rhs;
  void* out;
  int lhs_target, rhs_target;
  int out_len, lhs_len, rhs_len, reduce_size;
  int log2_lpe;
  int log2_lph;  // dot fast path: lanes per head = 1 << log2_lph
  BcastDims bd;
};

template <typename Idx>
__device__ __forceinline__ int64_t sddmm_select(int target, int64_t src, int64_t eid, int64_t dst) {
  // src/array/selector.h:30-55 — 0: src, 1: edge, 2: dst
  return target == 0 ? src : (target == 1 ? eid : dst);
}

// Source row of CSR position j: largest r with indptr[r] <= j (the reference's
// BinarySearchSrc, sddmm.hip.h:188-204, restated as an upper-bound search).
template <typename Idx>
__device__ __forceinline__ int64_t csr_row_of(const Idx* __restrict__ indptr, int64_t num_rows,
                                              int64_t j) {
  int64_t lo = 0, hi = num_rows - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (static_cast<int64_t>(indptr[mid]) <= j)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

template <int OP, typename A>
__device__ __forceinline__ A sddmm_apply(A l, A r) {
  if constexpr (OP == kAdd) return l + r;
  if constexpr (OP == kSub) return l - r;
  if constexpr (OP == kMul) return l * r;
  if constexpr (OP == kDiv) return l / r;
  if constexpr (OP == kCopyLhs) return l;
  return r;
}

// Element-wise ops.  BC: kBcNone (vector loads on both sides) or kBcGeneral (VEC == 1).
template <typename Idx, typename DT, int VEC, int OP, int BC, bool CSR>
__global__ __launch_bounds__(256) void sddmm_elementwise_kernel(const SddmmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  constexpr bool UL = op_uses_lhs(OP), UR = op_uses_rhs(OP);
  const int lane = threadIdx.x & 63;
  const int lpe = 1 << p.log2_lpe;
  const int lg = lane & (lpe - 1);
  const int64_t groups_per_block = blockDim.x >> p.log2_lpe;
  const int64_t gid = blockIdx.x * groups_per_block + (threadIdx.x >> p.log2_lpe);
  const int64_t gstride = static_cast<int64_t>(gridDim.x) * groups_per_block;
  const int F = p.out_len;
  const DT* __restrict__ L = static_cast<const DT*>(p.lhs);
  const DT* __restrict__ Rr = static_cast<const DT*>(p.rhs);
  DT* __restrict__ O = static_cast<DT*>(p.out);
  using V = VecT<DT, VEC>;
  for (int64_t i = gid; i < p.nnz; i += gstride) {
    int64_t src, dst;
    if constexpr (CSR) {
      src = csr_row_of<Idx>(p.indptr, p.num_rows, i);
      dst = p.col[i];
    } else {
      src = p.row[i];
      dst = p.col[i];
    }
    const int64_t eid = p.eids ? static_cast<int64_t>(p.eids[i]) : i;
    const int64_t lrow = sddmm_select<Idx>(p.lhs_target, src, eid, dst);
    const int64_t rrow = sddmm_select<Idx>(p.rhs_target, src, eid, dst);
    for (int k0 = lg * VEC; k0 < F; k0 += lpe * VEC) {
      int lo = k0, ro = k0;
      if constexpr (BC == kBcGeneral) bcast_offsets(p.bd, k0, &lo, &ro);
      V lv, rv, ov;
      if constexpr (UL) lv = *reinterpret_cast<const V*>(L + lrow * p.lhs_len + lo);
      if constexpr (UR) rv = *reinterpret_cast<const V*>(Rr + rrow * p.rhs_len + ro);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        A l = A(0), r = A(0);
        if constexpr (UL) l = to_acc<DT>(lv.v[v]);
        if constexpr (UR) r = to_acc<DT>(rv.v[v]);
        ov.v[v] = from_acc<DT>(sddmm_apply<OP, A>(l, r));
      }
      *reinterpret_cast<V*>(O + eid * F + k0) = ov;
    }
  }
}

// dot, fast path: no broadcast.  A lane group covers `heads_per_pass` heads at a time; the
// lph = 2^log2_lph lanes of one head stride over its D elements VEC at a time (D need not be
// a power of two or a multiple of lph * VEC: F = 100 -> 25 of 32 lanes live) and finish with
// xor-shuffles.
template <typename Idx, typename DT, int VEC, bool CSR>
__global__ __launch_bounds__(256) void sddmm_dot_kernel(const SddmmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  const int lane = threadIdx.x & 63;
  const int lpe = 1 << p.log2_lpe;
  const int lph = 1 << p.log2_lph;
  const int lg = lane & (lpe - 1);
  const int64_t groups_per_block = blockDim.x >> p.log2_lpe;
  const int64_t gid = blockIdx.x * groups_per_block + (threadIdx.x >> p.log2_lpe);
  const int64_t gstride = static_cast<int64_t>(gridDim.x) * groups_per_block;
  const int H = p.out_len, D = p.reduce_size;
  const int heads_per_pass = lpe >> p.log2_lph;
  const DT* __restrict__ L = static_cast<const DT*>(p.lhs);
  const DT* __restrict__ Rr = static_cast<const DT*>(p.rhs);
  DT* __restrict__ O = static_cast<DT*>(p.out);
  using V = VecT<DT, VEC>;
  // all lanes of the wave run the same trip count so the shuffles stay convergent
  const int64_t wave_first = gid - (lane >> p.log2_lpe);
  for (int64_t base = wave_first; base < p.nnz; base += gstride) {
    const int64_t i = base + (lane >> p.log2_lpe);
    const bool live = i < p.nnz;
    int64_t src = 0, dst = 0, eid = 0;
    if (live) {
      if constexpr (CSR) {
        src = csr_row_of<Idx>(p.indptr, p.num_rows, i);
      } else {
        src = p.row[i];
      }
      dst = p.col[i];
      eid = p.eids ? static_cast<int64_t>(p.eids[i]) : i;
    }
    const int64_t lrow = sddmm_select<Idx>(p.lhs_target, src, eid, dst);
    const int64_t rrow = sddmm_select<Idx>(p.rhs_target, src, eid, dst);
    for (int h0 = 0; h0 < H; h0 += heads_per_pass) {
      const int h = h0 + (lg >> p.log2_lph);
      A part = A(0);
      if (live && h < H) {
        const DT* lp = L + lrow * p.lhs_len + static_cast<int64_t>(h) * D;
        const DT* rp = Rr + rrow * p.rhs_len + static_cast<int64_t>(h) * D;
        for (int d0 = (lg & (lph - 1)) * VEC; d0 < D; d0 += lph * VEC) {
          const V lv = *reinterpret_cast<const V*>(lp + d0);
          const V rv = *reinterpret_cast<const V*>(rp + d0);
#pragma unroll
          for (int v = 0; v < VEC; ++v) part += to_acc<DT>(lv.v[v]) * to_acc<DT>(rv.v[v]);
        }
      }
      for (int m = lph >> 1; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
      if (live && h < H && (lg & (lph - 1)) == 0) O[eid * H + h] = from_acc<DT>(part);
    }
  }
}

// dot, general path (any D, any broadcast): one lane per (edge, output element), serial
// reduction in operand order like the reference functor (functor.cuh:128-145).
template <typename Idx, typename DT, int BC, bool CSR>
__global__ __launch_bounds__(256) void sddmm_dot_general_kernel(const SddmmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  const int64_t total = p.nnz * p.out_len;
  const DT* __restrict__ L = static_cast<const DT*>(p.lhs);
  const DT* __restrict__ Rr = static_cast<const DT*>(p.rhs);
  DT* __restrict__ O = static_cast<DT*>(p.out);
  for (int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; idx < total;
       idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t i = idx / p.out_len;
    const int k = static_cast<int>(idx - i * p.out_len);
    int64_t src;
    if constexpr (CSR)
      src = csr_row_of<Idx>(p.indptr, p.num_rows, i);
    else
      src = p.row[i];
    const int64_t dst = p.col[i];
    const int64_t eid = p.eids ? static_cast<int64_t>(p.eids[i]) : i;
    int lo = k, ro = k;
    if constexpr (BC == kBcGeneral) bcast_offsets(p.bd, k, &lo, &ro);
    const DT* l = L + sddmm_select<Idx>(p.lhs_target, src, eid, dst) * p.lhs_len +
                  static_cast<int64_t>(lo) * p.reduce_size;
    const DT* r = Rr + sddmm_select<Idx>(p.rhs_target, src, eid, dst) * p.rhs_len +
                  static_cast<int64_t>(ro) * p.reduce_size;
    A acc = A(0);
    for (int d = 0; d < p.reduce_size; ++d) acc += to_acc<DT>(l[d]) * to_acc<DT>(r[d]);
    O[eid * p.out_len + k] = from_acc<DT>(acc);
  }
}

// ---------------------------------------------------------------------------------------
inline int ceil_log2(int64_t x) {
  int l = 0;
  while ((int64_t(1) << l) < x) ++l;
  return l;
}

template <typename Idx>
inline SddmmParams<Idx> make_sddmm_params(const SddmmLaunch& L) {
  SddmmParams<Idx> p;
  if (L.use_coo) {
    p.row = static_cast<const Idx*>(L.coo.row);
    p.col = static_cast<const Idx*>(L.coo.col);
    p.indptr = nullptr;
    p.eids = static_cast<const Idx*>(L.coo.eids);
    p.nnz = L.coo.nnz;
    p.num_rows = L.coo.num_rows;
  } else {
    p.row = nullptr;
    p.col = static_cast<const Idx*>(L.csr.indices);
    p.indptr = static_cast<const Idx*>(L.csr.indptr);
    p.eids = static_cast<const Idx*>(L.csr.eids);
    p.nnz = L.csr.nnz;
    p.num_rows = L.csr.num_rows;
  }
  p.lhs = L.lhs;
  p.rhs = L.rhs;
  p.out = L.out;
  p.lhs_target = L.lhs_target;
  p.rhs_target = L.rhs_target;
  p.out_len = static_cast<int>(L.out_len);
  p.lhs_len = static_cast<int>(L.lhs_len);
  p.rhs_len = static_cast<int>(L.rhs_len);
  p.reduce_size = static_cast<int>(L.reduce_size);
  p.log2_lpe = 0;
  p.log2_lph = 0;
  p.bd = L.bdims;
  return p;
}

inline unsigned sddmm_grid(int64_t work_groups, int64_t groups_per_block) {
  // enough blocks to fill 256 CUs several times over, grid-stride beyond that
  int64_t blocks = (work_groups + groups_per_block - 1) / groups_per_block;
  const int64_t cap = 256 * 32;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

template <typename DT>
inline bool sddmm_vec_ok(const SddmmLaunch& L) {
  constexpr int full = 16 / sizeof(DT);
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (L.bcast != kBcNone) return false;
  if (L.op == kDot) {
    if (L.reduce_size % full) return false;
    if (L.lhs_len % full || L.rhs_len % full) return false;
    return aligned(L.lhs) && aligned(L.rhs);
  }
  if (L.out_len % full || !aligned(L.out)) return false;
  if (op_uses_lhs(L.op) && (L.lhs_len % full || !aligned(L.lhs))) return false;
  if (op_uses_rhs(L.op) && (L.rhs_len % full || !aligned(L.rhs))) return false;
  return true;
}

template <typename Idx, typename DT, int VEC, int BC, bool CSR>
inline int launch_sddmm_ew(const SddmmLaunch& L) {
  SddmmParams<Idx> p = make_sddmm_params<Idx>(L);
  int64_t lanes = (L.out_len + VEC - 1) / VEC;
  if (lanes > 64) lanes = 64;
  p.log2_lpe = ceil_log2(lanes);
  const int64_t gpb = 256 >> p.log2_lpe;
  const unsigned grid = sddmm_grid(p.nnz, gpb);
#define DGLA_EW(OPC)                                                                        \
  case OPC:                                                                                 \
    hipLaunchKernelGGL((sddmm_elementwise_kernel<Idx, DT, VEC, OPC, BC, CSR>), dim3(grid),   \
                       dim3(256), 0, L.stream, p);                                          \
    break;
  switch (L.op) {
    DGLA_EW(kAdd)
    DGLA_EW(kSub)
    DGLA_EW(kMul)
    DGLA_EW(kDiv)
    DGLA_EW(kCopyLhs)
    DGLA_EW(kCopyRhs)
    default:
      last_error() = "unsupported SDDMM binary operator";
      return -1;
  }
#undef DGLA_EW
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx, typename DT, bool CSR>
inline int launch_sddmm_fmt(const SddmmLaunch& L) {
  constexpr int full = 16 / sizeof(DT);
  const bool vec = sddmm_vec_ok<DT>(L);
  if (L.op == kDot) {
    SddmmParams<Idx> p = make_sddmm_params<Idx>(L);
    if (L.bcast == kBcNone) {
      // lanes per head: enough to cover D with one access each, a power of two, <= 64
      const int64_t per = vec ? full : 1;
      int64_t lph = (L.reduce_size + per - 1) / per;
      if (lph > 64) lph = 64;
      p.log2_lph = ceil_log2(lph);
      int64_t lanes = (int64_t(1) << p.log2_lph) * L.out_len;
      if (lanes > 64) lanes = 64;
      p.log2_lpe = ceil_log2(lanes);
      const int64_t gpb = 256 >> p.log2_lpe;
      const dim3 grid(sddmm_grid(p.nnz, gpb));
      if (vec)
        hipLaunchKernelGGL((sddmm_dot_kernel<Idx, DT, full, CSR>), grid, dim3(256), 0, L.stream, p);
      else
        hipLaunchKernelGGL((sddmm_dot_kernel<Idx, DT, 1, CSR>), grid, dim3(256), 0, L.stream, p);
    } else {
      const unsigned grid = sddmm_grid(p.nnz * p.out_len, 256);
      hipLaunchKernelGGL((sddmm_dot_general_kernel<Idx, DT, kBcGeneral, CSR>), dim3(grid),
                         dim3(256), 0, L.stream, p);
    }
    DGLA_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (vec) return launch_sddmm_ew<Idx, DT, full, kBcNone, CSR>(L);
  if (L.bcast == kBcGeneral) return launch_sddmm_ew<Idx, DT, 1, kBcGeneral, CSR>(L);
  return launch_sddmm_ew<Idx, DT, 1, kBcNone, CSR>(L);
}

template <typename DT>
inline int launch_sddmm_typed(const SddmmLaunch& L) {
  const int idbits = L.use_coo ? L.coo.idbits : L.csr.idbits;
  if (idbits == 32)
    return L.use_coo ? launch_sddmm_fmt<int32_t, DT, false>(L) : launch_sddmm_fmt<int32_t, DT, true>(L);
  return L.use_coo ? launch_sddmm_fmt<int64_t, DT, false>(L) : launch_sddmm_fmt<int64_t, DT, true>(L);
}

}  // namespace dgla
