// Fused edge softmax (forward + backward) over the in-edge CSR for gfx950 (MI355X).
//
// Arithmetic of Edge_softmax_csr_forward / _backward (src/array/cpu/spmm.h:484-570), which
// the reference only has on CPU: on GPU it composes five launches (max-SpMM, sub-SDDMM, exp,
// sum-SpMM, div-SDDMM; python/dgl/backend/pytorch/sparse.py:709-713; "TODO" at
// src/array/kernel.cc:313,331).  Here one launch does a row:
//
//   * a destination row is owned by a LANE GROUP of GS = HP * Q lanes of one wavefront:
//     HP lanes across the feature dimension (heads), Q edge slots along the row; the host
//     picks Q from the mean in-degree so that short rows share a wave (64 / GS rows per wave)
//     and long rows get all 64 lanes.
//   * each lane keeps its first R edges' values (and edge ids) in registers, so a row of up
//     to Q * R edges is read from HBM exactly once; longer rows re-read the remainder (L2).
//   * max / sum over the Q edge slots are xor-shuffles inside the 64-wide wave; the trip
//     count of the row loop is wave-uniform so every shuffle is convergent.
//
// HBM-bound: forward moves E*dim*s in + E*dim*s out + index bytes; scores are indexed by
// EDGE ID, so with a permuted edge-id map every edge touches its own 128-byte line.
#include "common.h"

namespace dgla {

constexpr int kEsmRegs = 4;  // R: edges cached per lane

template <typename A>
__device__ __forceinline__ A esm_exp(A x);
template <>
__device__ __forceinline__ float esm_exp<float>(float x) {
  return expf(x);
}
template <>
__device__ __forceinline__ double esm_exp<double>(double x) {
  return exp(x);
}

template <typename A>
__device__ __forceinline__ A group_reduce_max(A v, int lo_mask, int gs) {
  for (int m = lo_mask; m < gs; m <<= 1) {
    const A o = __shfl_xor(v, m, 64);
    v = v > o ? v : o;
  }
  return v;
}
template <typename A>
__device__ __forceinline__ A group_reduce_sum(A v, int lo_mask, int gs) {
  for (int m = lo_mask; m < gs; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// BWD == false:  c[eid] = softmax over the row of a[eid]             (a = score, b unused)
// BWD == true :  c[eid] = b[eid] - a[eid] * sum_row(b)               (a = out, b = sds)
template <typename Idx, typename DT, bool BWD, bool PRECISE>
__global__ __launch_bounds__(256) void edge_softmax_kernel(
    const Idx* __restrict__ indptr, const Idx* __restrict__ eids, const DT* __restrict__ a,
    const DT* __restrict__ b, DT* __restrict__ c, int64_t num_rows, int dim, int log2_hp,
    int log2_q) {
  using A = typename Acc<DT>::type;
  constexpr int R = kEsmRegs;
  const int lane = threadIdx.x & 63;
  const int hp = 1 << log2_hp, q_slots = 1 << log2_q;
  const int gs = hp * q_slots;                 // lanes per row, <= 64
  const int lig = lane & (gs - 1);
  const int h = lig & (hp - 1);                // feature lane
  const int q = lig >> log2_hp;                // edge slot
  const int rows_per_wave = 64 / gs;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t num_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int g_in_wave = lane / gs;

  for (int64_t base = wave * rows_per_wave; base < num_rows; base += num_waves * rows_per_wave) {
    const int64_t row = base + g_in_wave;
    int64_t s = 0, e = 0;
    if (row < num_rows) {
      s = indptr[row];
      e = indptr[row + 1];
    }
    for (int k0 = 0; k0 < dim; k0 += hp) {  // wave-uniform
      const int k = k0 + h;
      const bool kok = k < dim;
      A v[R];
      int64_t ei[R];
      // ---- pass 1: load (first R edges of this slot stay in registers) -----------------
      A red = BWD ? A(0) : -static_cast<A>(__builtin_huge_valf());
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const int64_t j = s + q + static_cast<int64_t>(t) * q_slots;
        v[t] = A(0);
        ei[t] = 0;
        if (j < e && kok) {
          ei[t] = eids ? static_cast<int64_t>(eids[j]) : j;
          v[t] = to_acc<DT>((BWD ? b : a)[ei[t] * dim + k]);
          if constexpr (BWD)
            red += v[t];
          else
            red = red > v[t] ? red : v[t];
        }
      }
      for (int64_t j = s + q + static_cast<int64_t>(R) * q_slots; j < e && kok; j += q_slots) {
        const int64_t eid = eids ? static_cast<int64_t>(eids[j]) : j;
        const A x = to_acc<DT>((BWD ? b : a)[eid * dim + k]);
        if constexpr (BWD)
          red += x;
        else
          red = red > x ? red : x;
      }
      if constexpr (BWD) {
        const A sum = group_reduce_sum<A>(red, hp, gs);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const int64_t j = s + q + static_cast<int64_t>(t) * q_slots;
          if (j < e && kok)
            c[ei[t] * dim + k] = from_acc<DT>(v[t] - sum * to_acc<DT>(a[ei[t] * dim + k]));
        }
        for (int64_t j = s + q + static_cast<int64_t>(R) * q_slots; j < e && kok; j += q_slots) {
          const int64_t eid = eids ? static_cast<int64_t>(eids[j]) : j;
          c[eid * dim + k] = from_acc<DT>(to_acc<DT>(b[eid * dim + k]) -
                                          sum * to_acc<DT>(a[eid * dim + k]));
        }
      } else {
        const A mx = group_reduce_max<A>(red, hp, gs);
        // ---- pass 2: exp + sum ----------------------------------------------------------
        A part = A(0);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const int64_t j = s + q + static_cast<int64_t>(t) * q_slots;
          if (j < e && kok) {
            v[t] = PRECISE ? static_cast<A>(exp(static_cast<double>(v[t] - mx))) : esm_exp<A>(v[t] - mx);
            part += v[t];
          }
        }
        for (int64_t j = s + q + static_cast<int64_t>(R) * q_slots; j < e && kok; j += q_slots) {
          const int64_t eid = eids ? static_cast<int64_t>(eids[j]) : j;
          const A x = to_acc<DT>(a[eid * dim + k]) - mx;
          part += PRECISE ? static_cast<A>(exp(static_cast<double>(x))) : esm_exp<A>(x);
        }
        const A sum = group_reduce_sum<A>(part, hp, gs);
        // ---- pass 3: normalise ----------------------------------------------------------
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const int64_t j = s + q + static_cast<int64_t>(t) * q_slots;
          if (j < e && kok) c[ei[t] * dim + k] = from_acc<DT>(v[t] / sum);
        }
        for (int64_t j = s + q + static_cast<int64_t>(R) * q_slots; j < e && kok; j += q_slots) {
          const int64_t eid = eids ? static_cast<int64_t>(eids[j]) : j;
          const A x = to_acc<DT>(a[eid * dim + k]) - mx;
          const A ex = PRECISE ? static_cast<A>(exp(static_cast<double>(x))) : esm_exp<A>(x);
          c[eid * dim + k] = from_acc<DT>(ex / sum);
        }
      }
    }
  }
}

template <typename Idx, typename DT>
static int edge_softmax_run(const CsrView& csr, const void* a, const void* b, void* c, int dim,
                            bool backward, hipStream_t s) {
  int log2_hp = 0;
  while ((1 << log2_hp) < dim && log2_hp < 6) ++log2_hp;
  // edge slots: about half the mean in-degree (each slot then caches ~2 edges in registers),
  // a power of two, and no more than what is left of the wave
  const int64_t mean_deg = csr.num_rows > 0 ? (csr.nnz + csr.num_rows - 1) / csr.num_rows : 1;
  int log2_q = 0;
  while ((2 << log2_q) <= mean_deg / 2 && log2_hp + log2_q < 6) ++log2_q;
  if (log2_hp + log2_q > 6) log2_q = 6 - log2_hp;
  const int rows_per_wave = 64 >> (log2_hp + log2_q);
  const int64_t rows_per_block = 4 * rows_per_wave;
  int64_t blocks = (csr.num_rows + rows_per_block - 1) / rows_per_block;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  const Idx* indptr = static_cast<const Idx*>(csr.indptr);
  const Idx* eids = static_cast<const Idx*>(csr.eids);
  constexpr bool kPrecise = sizeof(DT) == 8;
  if (!backward)
    hipLaunchKernelGGL((edge_softmax_kernel<Idx, DT, false, kPrecise>), grid, block, 0, s, indptr,
                       eids, static_cast<const DT*>(a), static_cast<const DT*>(nullptr),
                       static_cast<DT*>(c), csr.num_rows, dim, log2_hp, log2_q);
  else
    hipLaunchKernelGGL((edge_softmax_kernel<Idx, DT, true, kPrecise>), grid, block, 0, s, indptr,
                       eids, static_cast<const DT*>(a), static_cast<const DT*>(b),
                       static_cast<DT*>(c), csr.num_rows, dim, log2_hp, log2_q);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_edge_softmax(const CsrView& csr, int dtype, const void* a, const void* b, void* c,
                        int64_t dim, bool backward, hipStream_t s) {
  const int d = static_cast<int>(dim);
#define DGLA_ES(DT)                                                              \
  return csr.idbits == 32 ? edge_softmax_run<int32_t, DT>(csr, a, b, c, d, backward, s) \
                          : edge_softmax_run<int64_t, DT>(csr, a, b, c, d, backward, s)
  switch (dtype) {
    case kF32: DGLA_ES(float);
    case kF64: DGLA_ES(double);
    case kF16: DGLA_ES(f16_t);
    case kBF16: DGLA_ES(bf16_t);
  }
#undef DGLA_ES
  last_error() = "unsupported dtype";
  return -1;
}

}  // namespace dgla
