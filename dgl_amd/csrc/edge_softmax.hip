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
  // exp(x) = 2^(x log2 e) on the hardware's v_exp_f32 (1 ulp).  The product x * log2(e) is kept
  // as hi + lo (one fma recovers its rounding error, a second constant carries the low bits of
  // log2 e), so the error of the argument does not grow with |x|: 2^hi * (1 + lo ln 2).
  // ~7 instructions instead of the ~25 of the library expf; |relative error| < 3e-7 measured
  // against exp() in fp64 over [-88, 0] (tests/test_gpu_softmax_kernels.py).
  // (arguments are <= 0 here; clamped at -200, where 2^hi is 0 already, so that exp(-inf) = 0
  // instead of the NaN that lo would become)
  asm("v_max_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(-200.f));
  const float hi = x * 1.44269504088896341f;
  const float lo = __builtin_fmaf(x, 1.44269504088896341f, -hi) + x * 1.92596299112661746e-8f;
  return __builtin_amdgcn_exp2f(hi) * __builtin_fmaf(lo, 0.693147180559945309f, 1.0f);
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

// =========================================================================================
// Merge-path variant (dim <= 16, caller-provided workspace): degree-balanced.
//
// The lane-group kernel above gives one row to one lane group, so a 10k-edge hub row is
// walked serially by 4-64 lanes while the rest of the chip idles (measured on the
// ogbn-arxiv-shaped graph: 0.53 ms for 80 MB of scores).  Here the CSR is cut by merge path
// into units of at most kEsmItems items (edges + row ends) like the SpMM; one workgroup per unit:
//   1. stage the unit's row ends (and edge ids) in LDS, the row ends also as START BITS of the
//      edges; every lane loads four consecutive edges of the unit with all their features into
//      registers (one HBM round trip for the whole unit);
//   2. per-segment max / sum — a segment is a row, or the piece of a row inside this unit: inside
//      a lane in registers, across lanes by a segmented DPP scan, across waves through LDS; the
//      values never leave the registers;
//   3. rows that lie entirely inside the unit are finished and written (read once, written
//      once); a row cut by a unit boundary leaves (max, sum) per piece in the workspace and its
//      edges unwritten;
//   4. a fix-up kernel merges the pieces of each cut row ((m, s) pairs combine as
//      S = sum_i s_i * exp(m_i - M)) and writes its edges from the scores, one wavefront per
//      boundary, so even a hub row is handled by as many wavefronts as it has units.
// =========================================================================================
// Measured at 62 M edges, H = 8 fp32, forward / backward: 64 threads x 256 items 1.67 / 1.72 ms
// (48 % of the edges in rows cut by a unit boundary -> fix-up traffic), 256 x 1024 1.34 / 1.49 ms,
// 512 x 2048 1.53 / 2.28 ms (one workgroup per CU: the barriers of the cross-wave scan are no longer
// hidden); 256 x 1024 with start bits, row-aligned boundaries and assembly scans **1.00 / 1.09 ms**.
constexpr int kEsmThreads = 256;
constexpr int kEsmWaves = kEsmThreads / 64;
constexpr int kEsmItems = 4 * kEsmThreads;        // items (edges + row ends) a unit holds at most
// Unit boundaries sit kEsmStride items apart on the merge path and are then moved FORWARD to the end
// of the row they cut when that takes at most kEsmSlack edges (+ the row-end item): rows of ordinary
// length are never cut, so the carry / tail / fix-up machinery only runs for rows longer than that.
constexpr int kEsmSlack = 63;
constexpr int kEsmStride = kEsmItems - kEsmSlack - 1;

template <typename Idx>
struct EsmParams {
  const Idx* indptr;
  const Idx* eids;
  int64_t num_rows, nnz, num_units;
  const int64_t* plan;   // [num_units + 1] first row of every unit
  const int64_t* planj;  // [num_units + 1] first edge of every unit
  const void* a;
  const void* b;
  void* c;
  int dim, log2_hp;
  int wave_lds_bytes;
  int vec4;  // fp32, dim % 4 == 0 == padded width, 16-byte aligned operands: rows move as 16-byte pieces
  int xcd;   // units in XCD-contiguous order (kTuneXcd): neighbouring units share an L2
  int region_bytes;  // LDS bytes of the tables / row-transposition slices in front of the rest
  int out_pos;       // DGLA_ESM_OUT_POSITION.  forward: `c` is written in POSITION order while `a` is read through eids;
                     // backward: `a` (the saved softmax) is READ and `c` written in position order, `b` read through eids
  int b_is_grad;     // backward: `b` holds the upstream gradient g, not out * g (DGLA_ESM_B_IS_GRAD): the product is formed here
  int64_t* carry_row;  // [num_units] row continued in the next unit, or -1
  void* carry_stat;    // [num_units, 2 * dim] accumulators: (m | s) forward, (sum | -) backward
  void* tail_stat;     // [num_units, 2 * dim]
};

template <typename Idx>
__global__ void esm_plan_kernel(const Idx* __restrict__ indptr, int64_t num_rows, int64_t nnz,
                                int64_t num_units, int64_t* __restrict__ plan, int64_t* __restrict__ planj) {
  // boundary w: the merge-path point (i, j) on the diagonal w * kEsmStride — i = largest row with
  // indptr[i] + i <= d (see spmm_csr.hip.h) — moved to the end of row i when the row is cut there
  // and has at most kEsmSlack edges left
  const int64_t w = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (w > num_units) return;
  int64_t d = w * kEsmStride;
  const int64_t total = num_rows + nnz;
  if (d > total) d = total;
  int64_t lo = 0, hi = num_rows;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (static_cast<int64_t>(indptr[mid]) + mid <= d)
      lo = mid;
    else
      hi = mid - 1;
  }
  int64_t i = lo, j = d - lo;
  if (i < num_rows) {
    const int64_t begin = static_cast<int64_t>(indptr[i]), end = static_cast<int64_t>(indptr[i + 1]);
    if (j > begin && end - j <= kEsmSlack) {
      j = end;
      i = i + 1;
    }
  }
  plan[w] = i;
  planj[w] = j;
}

struct EsmUnit {
  int64_t i0, j0;
  int R, nE;
};

template <typename Idx>
__device__ __forceinline__ EsmUnit esm_unit(const EsmParams<Idx>& p, int64_t w) {
  EsmUnit u;
  u.i0 = p.plan[w];
  u.j0 = p.planj[w];
  u.R = static_cast<int>(p.plan[w + 1] - u.i0);
  u.nE = static_cast<int>(p.planj[w + 1] - u.j0);
  return u;
}

// 1 / x: the hardware reciprocal for fp32 (1 ulp), a division for fp64
template <typename A>
__device__ __forceinline__ A esm_recip(A x) {
  if constexpr (sizeof(A) == 4)
    return __builtin_amdgcn_rcpf(x);
  else
    return A(1) / x;
}

template <typename A, bool PRECISE>
__device__ __forceinline__ A esm_expx(A x) {
  if constexpr (PRECISE)
    return static_cast<A>(exp(static_cast<double>(x)));
  else
    return esm_exp<A>(x);
}

// Rows of the LDS tables holding the totals of lane-crossing segments: one per lane of the workgroup.
constexpr int kEsmSegCap = kEsmThreads;

// Orders ONE wave's LDS traffic (its own staging slice): LDS operations of a wave execute in
// issue order, so a compiler fence + wave barrier is all it takes.
__device__ __forceinline__ void esm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Cross-lane moves of the segmented scan as DPP modifiers of a VALU move (row_shr:n inside a row
// of 16 lanes, row_bcast:15 / :31 between rows, wave_shr:1 for "the previous lane") instead of
// ds_bpermute: a scan step costs an ALU instruction, not an LDS round trip (the forward pass makes
// two 6-step scans over HP values per unit; with ds_bpermute they were its critical path).
// Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int esm_dpp(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float esm_dpp(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, src), CTRL, ROW_MASK,
                                                               0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double esm_dpp(double old, double src) {
  const uint64_t o = __builtin_bit_cast(uint64_t, old), v = __builtin_bit_cast(uint64_t, src);
  const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(
      static_cast<int>(o), static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
  const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(
      static_cast<int>(o >> 32), static_cast<int>(v >> 32), CTRL, ROW_MASK, 0xf, false));
  return __builtin_bit_cast(double, (static_cast<uint64_t>(hi) << 32) | lo);
}

// ---- fp32 segmented scans over the lanes of a wave, written in assembly ---------------------------
// Step "lane l takes from lane l - d unless a segment starts in (l - d, l]" as ONE instruction per
// value: the receiving lane's flag is kept as a factor nf (1: no start yet, 0: start seen) and the
// shifted operand comes in through the DPP modifier,
//     sum:  x += x[l - d] * nf           v_fmac_f32_dpp
//     max:  t  = x[l - d] + pen          v_add_f32_dpp      (pen = 0 / -inf)
//           x  = max(x, t)               v_max_f32
// (the compiler's version is a DPP move, the operation and a select per value).  Lanes without a
// source lane are not written (bound_ctrl:0): x stays, and a stale t is a value x has already
// absorbed.  The s_nop at the head of every block covers the "VALU write -> DPP read" and
// "EXEC write -> DPP" wait states for whatever the compiler placed in front of it; inside a
// block the producers are the previous step's instructions, HP or more issue slots away.
#define DGLA_ESM_SUM_OP(i, C) "v_fmac_f32_dpp %" #i ", %" #i ", %[nf] " C "\n\t"
#define DGLA_ESM_MAX_OP(i, C) \
  "v_add_f32_dpp %[t" #i "], %" #i ", %[nf] " C "\n\tv_max_f32 %" #i ", %" #i ", %[t" #i "]\n\t"
#define DGLA_ESM_R1(OP, C) OP(0, C)
#define DGLA_ESM_R2(OP, C) DGLA_ESM_R1(OP, C) OP(1, C)
#define DGLA_ESM_R4(OP, C) DGLA_ESM_R2(OP, C) OP(2, C) OP(3, C)
#define DGLA_ESM_R8(OP, C) DGLA_ESM_R4(OP, C) OP(4, C) OP(5, C) OP(6, C) OP(7, C)
#define DGLA_ESM_R16(OP, C) \
  DGLA_ESM_R8(OP, C) OP(8, C) OP(9, C) OP(10, C) OP(11, C) OP(12, C) OP(13, C) OP(14, C) OP(15, C)
#define DGLA_ESM_X1 "+v"(x[0])
#define DGLA_ESM_X2 DGLA_ESM_X1, "+v"(x[1])
#define DGLA_ESM_X4 DGLA_ESM_X2, "+v"(x[2]), "+v"(x[3])
#define DGLA_ESM_X8 DGLA_ESM_X4, "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
#define DGLA_ESM_X16 \
  DGLA_ESM_X8, "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
#define DGLA_ESM_T1 [t0] "+v"(t[0])
#define DGLA_ESM_T2 DGLA_ESM_T1, [t1] "+v"(t[1])
#define DGLA_ESM_T4 DGLA_ESM_T2, [t2] "+v"(t[2]), [t3] "+v"(t[3])
#define DGLA_ESM_T8 DGLA_ESM_T4, [t4] "+v"(t[4]), [t5] "+v"(t[5]), [t6] "+v"(t[6]), [t7] "+v"(t[7])
#define DGLA_ESM_T16                                                                                  \
  DGLA_ESM_T8, [t8] "+v"(t[8]), [t9] "+v"(t[9]), [t10] "+v"(t[10]), [t11] "+v"(t[11]), [t12] "+v"(t[12]), \
      [t13] "+v"(t[13]), [t14] "+v"(t[14]), [t15] "+v"(t[15])
#define DGLA_ESM_SUM_STEP(N, C)                                                             \
  asm volatile("s_nop 4\n\t" DGLA_ESM_R##N(DGLA_ESM_SUM_OP, C) "v_mul_f32_dpp %[nf], %[nf], %[nf] " C \
               : DGLA_ESM_X##N, [nf] "+v"(nf));
#define DGLA_ESM_MAX_STEP(N, C)                                                             \
  asm volatile("s_nop 4\n\t" DGLA_ESM_R##N(DGLA_ESM_MAX_OP, C) "v_add_f32_dpp %[nf], %[nf], %[nf] " C \
               : DGLA_ESM_X##N, DGLA_ESM_T##N, [nf] "+v"(nf));
#define DGLA_ESM_ALL_STEPS(STEP, N)                        \
  STEP(N, "row_shr:1 row_mask:0xf bank_mask:0xf")          \
  STEP(N, "row_shr:2 row_mask:0xf bank_mask:0xf")          \
  STEP(N, "row_shr:4 row_mask:0xf bank_mask:0xf")          \
  STEP(N, "row_shr:8 row_mask:0xf bank_mask:0xf")          \
  STEP(N, "row_bcast:15 row_mask:0xa bank_mask:0xf")       \
  STEP(N, "row_bcast:31 row_mask:0xc bank_mask:0xf")

// x: the lanes' values, f: "a segment starts in this lane"; on return x = segmented inclusive sum
// over the lanes of the wave and f = "a segment starts in this lane or an earlier one"
template <int HP>
__device__ __forceinline__ void esm_scan_sum_f32(float (&x)[HP], int& f) {
  float nf = f ? 0.f : 1.f;
  if constexpr (HP == 1) { DGLA_ESM_ALL_STEPS(DGLA_ESM_SUM_STEP, 1) }
  else if constexpr (HP == 2) { DGLA_ESM_ALL_STEPS(DGLA_ESM_SUM_STEP, 2) }
  else if constexpr (HP == 4) { DGLA_ESM_ALL_STEPS(DGLA_ESM_SUM_STEP, 4) }
  else if constexpr (HP == 8) { DGLA_ESM_ALL_STEPS(DGLA_ESM_SUM_STEP, 8) }
  else { static_assert(HP == 16, "feature widths are padded to 1, 2, 4, 8 or 16"); DGLA_ESM_ALL_STEPS(DGLA_ESM_SUM_STEP, 16) }
  f = nf != 1.f;
}
template <int HP>
__device__ __forceinline__ void esm_scan_max_f32(float (&x)[HP], int& f) {
  const float ninf = -__builtin_huge_valf();
  float nf = f ? ninf : 0.f;  // the penalty added to what comes in from earlier lanes
  float t[HP];
#pragma unroll
  for (int h = 0; h < HP; ++h) t[h] = ninf;
  if constexpr (HP == 1) { DGLA_ESM_ALL_STEPS(DGLA_ESM_MAX_STEP, 1) }
  else if constexpr (HP == 2) { DGLA_ESM_ALL_STEPS(DGLA_ESM_MAX_STEP, 2) }
  else if constexpr (HP == 4) { DGLA_ESM_ALL_STEPS(DGLA_ESM_MAX_STEP, 4) }
  else if constexpr (HP == 8) { DGLA_ESM_ALL_STEPS(DGLA_ESM_MAX_STEP, 8) }
  else { static_assert(HP == 16, "feature widths are padded to 1, 2, 4, 8 or 16"); DGLA_ESM_ALL_STEPS(DGLA_ESM_MAX_STEP, 16) }
  f = nf < 0.f;
}

// x = max(x, x of the DPP source lane); lanes without a source keep x
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int esm_max_dpp(int x) {
  const int o = __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);
  return o > x ? o : x;
}

// 16-byte piece i of a wave's staging slice lives at piece esm_swz(i): the global side moves
// pieces lane-linearly (piece 64 k + lane), the register side wants 4 or 8 CONSECUTIVE pieces per
// lane; XOR-ing the low four bits with bits 4..7 keeps both directions free of bank conflicts
// (every 16-byte bank group is hit by exactly 4 of the 64 lanes, the minimum for ds_*_b128).
__device__ __forceinline__ int esm_swz(int i) { return i ^ ((i >> 4) & 15); }

// One workgroup (kEsmThreads = 256 lanes) per unit of at most kEsmItems = 1024 items.  Every lane
// owns kEsmEpl = 4 CONSECUTIVE edges of the unit and keeps all HP features of them in registers:
// the scores travel HBM -> registers -> HBM (for edge ids = positions a lane's four rows are
// 4 * dim * s contiguous bytes); LDS holds the unit's row ends, the start bits of its edges, its
// edge ids and two 256-row tables.
//
// A segment is a row, or the piece of a row inside this unit; edge e begins one when a row ends
// right before it (bit e of the start-bit mask, set while the row ends are staged; the unit's
// first edge and its end count as bounds too).  A lane's edges are consecutive, so its segments
// are: possibly one that began in an earlier lane (its "head"), segments lying wholly inside the
// lane ("local"), possibly one that goes on into later lanes (its "tail"; head == tail when the
// whole lane sits inside one long row) — all read off the five start bits of the lane's edges
// and the edge after them.  Local segments are reduced in registers with one forward and one
// backward sweep over the four edges.  Crossing segments are reduced ACROSS lanes with a
// segmented inclusive scan over (tail-starts-here flag, tail partial) — 6 DPP steps inside each
// wave, then one hand-over of the four waves' last values through LDS, for any mix of row lengths
// — after which the lane where a crossing segment ENDS holds its total and leaves it in the
// table row of the lane the segment STARTED in (unique: at most one segment crosses out of a
// lane).  Everybody then reads the two table rows it may need (head, tail).
//   forward : max -> exp(x - M) -> sum -> scale by 1 / S; segments cut by the unit boundary only
//             publish (M, S): the fix-up kernel, which sees all pieces of the row, writes their edges
//   backward: sum(sds) -> c = sds - sum * out
// A hub row and forty 5-edge rows cost the same.  (Earlier versions: ds_max / ds_add atomics on
// the tables spent half of their cycles in LDS issue stalls, profiles/r2/softmax_pmc_atomics.txt;
// a binary search through the row ends for every lane's first edge, ds_bpermute shuffles and
// compiler-generated selects made the forward pass VALU-bound, profiles/r2/softmax_pmc_valu.txt.)
constexpr int kEsmEpl = kEsmItems / kEsmThreads;
static_assert(kEsmEpl == 4, "the sweeps below are written out for four edges per lane");

// accumulator value rounded to what a DT store + load would give back (fp32 / fp64: itself)
template <typename DT>
__device__ __forceinline__ typename Acc<DT>::type esm_round(typename Acc<DT>::type v) {
  return to_acc<DT>(from_acc<DT>(v));
}

template <typename Idx, typename DT, bool BWD, bool PRECISE, int HP>
__global__ __launch_bounds__(kEsmThreads, (BWD || HP > 8 || sizeof(DT) == 8) ? 1 : 4) void edge_softmax_merge_kernel(const EsmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  extern __shared__ __align__(16) unsigned char esm_smem[];
  const int wib = threadIdx.x >> 6, wl = threadIdx.x & 63;
  const int lane = threadIdx.x;  // lane of the WORKGROUP: the unit's edges [4 lane, 4 lane + 4)
  unsigned blk = blockIdx.x;
  if (p.xcd) {  // block b runs on XCD b % 8: give every XCD one contiguous eighth of the units
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u;
    const unsigned x = blk & 7u, i = blk >> 3;
    blk = x * q + (x < r ? x : r) + i;
  }
  const int64_t w = blk;  // one unit per workgroup
  const int dim = p.dim;
  // fp32, whole rows of 4 or 8 features, edge ids = positions: rows move between HBM and the
  // registers THROUGH a per-wave LDS slice so that every global instruction is 1 KB contiguous
  // (lane-linear pieces) although a lane owns four consecutive rows.  The slice (256 rows) shares
  // its memory with the segment tables, which are only alive between the two transpositions.
  constexpr int Q = HP / 4;  // 16-byte pieces per row
  const bool tr = std::is_same<DT, float>::value && (HP == 4 || HP == 8) && p.vec4 && p.eids == nullptr;
  // (the slice holds HALF of the wave's rows: lanes 0-31 take theirs in round 0, lanes 32-63 in
  // round 1, so that the slices are no larger than the forward tables they share memory with)
  unsigned char* stage = esm_smem + static_cast<size_t>(wib) * (32 * kEsmEpl * HP * sizeof(float));
  A* tm = reinterpret_cast<A*>(esm_smem);                   // [kEsmThreads * HP]  (forward only)
  A* ts = tm + (BWD ? 0 : kEsmSegCap * HP);                 // [kEsmThreads * HP]
  A* wv = reinterpret_cast<A*>(esm_smem + p.region_bytes);  // [waves * HP] last scanned value of every wave
  int64_t* eid = reinterpret_cast<int64_t*>(wv + kEsmWaves * HP);   // [kEsmItems]
  int* rend = reinterpret_cast<int*>(eid + kEsmItems);      // [kEsmItems + 2]
  int* wf = rend + kEsmItems + 2;                           // [waves] "a segment starts in this wave"
  unsigned* sb = reinterpret_cast<unsigned*>(wf + kEsmWaves);  // [kEsmItems / 32 + 2] start bits of the edges
  const DT* __restrict__ pa = static_cast<const DT*>(p.a);
  const DT* __restrict__ pb = static_cast<const DT*>(p.b);
  DT* __restrict__ pc = static_cast<DT*>(p.c);
  if (w >= p.num_units) return;  // block-uniform
  if (lane < kEsmItems / 32 + 2) sb[lane] = 0u;
  __syncthreads();  // (early: nothing is waiting on memory yet)

  const EsmUnit u = esm_unit<Idx>(p, w);
  const int e0 = lane * kEsmEpl;
  const int n_valid = u.nE - e0 < kEsmEpl ? (u.nE - e0 > 0 ? u.nE - e0 : 0) : kEsmEpl;
  A v[kEsmEpl][HP];
  A v2[BWD ? kEsmEpl : 1][BWD ? HP : 1];  // backward: out values
  int64_t off[kEsmEpl];
  int64_t offa[BWD ? kEsmEpl : 1];  // backward: where `a` (the saved softmax) is read — by edge id like b, or by position

  // this lane's edges with all their features (edges past the unit's end re-load the last one)
#define DGLA_ESM_LOAD()                                                                              \
  if (u.nE > 0) {                                                                                     \
    bool done = false;                                                                                \
    if constexpr (std::is_same<DT, float>::value && HP >= 4) {                                        \
      if (p.vec4) { /* rows are whole 16-byte pieces */                                               \
        typedef float f32x4 __attribute__((ext_vector_type(4)));                                      \
        _Pragma("unroll") for (int j = 0; j < kEsmEpl; ++j)                                           \
          _Pragma("unroll") for (int q = 0; q < HP / 4; ++q) {                                        \
            const f32x4 t = *reinterpret_cast<const f32x4*>((BWD ? pb : pa) + off[j] + 4 * q);        \
            v[j][4 * q] = t.x, v[j][4 * q + 1] = t.y, v[j][4 * q + 2] = t.z, v[j][4 * q + 3] = t.w;   \
            if constexpr (BWD) {                                                                      \
              const f32x4 t2 = *reinterpret_cast<const f32x4*>(pa + offa[j] + 4 * q);                 \
              v2[j][4 * q] = t2.x, v2[j][4 * q + 1] = t2.y, v2[j][4 * q + 2] = t2.z,                  \
              v2[j][4 * q + 3] = t2.w;                                                                \
            }                                                                                         \
          }                                                                                           \
        done = true;                                                                                  \
      }                                                                                               \
    }                                                                                                 \
    if (!done) {                                                                                      \
      _Pragma("unroll") for (int j = 0; j < kEsmEpl; ++j)                                             \
        _Pragma("unroll") for (int h = 0; h < HP; ++h) {                                              \
          const int hh = h < dim ? h : dim - 1; /* padded features load a valid element */           \
          v[j][h] = to_acc<DT>((BWD ? pb : pa)[off[j] + hh]);                                         \
          if constexpr (BWD) v2[j][h] = to_acc<DT>(pa[offa[j] + hh]);                                 \
        }                                                                                             \
    }                                                                                                 \
  }

  const bool direct = p.eids == nullptr;  // edge id == position: addresses need no staging
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  [[maybe_unused]] f32x4 tmp[4 * (Q > 0 ? Q : 1)], tmp2[BWD ? 4 * (Q > 0 ? Q : 1) : 1];
  const int wave_edges = u.nE - wib * (64 * kEsmEpl);  // edges of this wave's slice (may be <= 0)
  // slots past the unit's last edge hold the identity of the first reduction (-inf forward, whose
  // exp is the 0 of the second one; 0 backward) and count as continuing the last edge's segment:
  // the lane's last group is then always the one of slot 3
  [[maybe_unused]] const float pad = BWD ? 0.f : -__builtin_huge_valf();
  if (direct) {
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j) {
      int e = e0 + j;
      if (e >= u.nE) e = u.nE > 0 ? u.nE - 1 : 0;
      off[j] = (u.j0 + e) * dim;
      if constexpr (BWD) offa[j] = off[j];
    }
    if constexpr (std::is_same<DT, float>::value && (HP == 4 || HP == 8)) {
      if (tr) {  // lane-linear pieces of the wave's slice; in flight together with the index loads below
        const int64_t base = (u.j0 + wib * (64 * kEsmEpl)) * dim;
#pragma unroll
        for (int k = 0; k < 4 * Q; ++k) {
          const int i = 64 * k + wl;
          const bool ok = i / Q < wave_edges;
          tmp[k] = ok ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(BWD ? pb : pa) + base + 4 * i)
                      : f32x4{pad, pad, pad, pad};
          if constexpr (BWD)
            tmp2[k] = ok ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(pa) + base + 4 * i)
                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if (!tr) {
      DGLA_ESM_LOAD()  // in flight together with the index loads below: one HBM round trip per unit
    }
  }
  // ---- stage row ends (and edge ids): 4 independent loads per lane issued back to back
  // (addresses clamped, not predicated) ----------------------------------------------------------
  {
    constexpr int KS = kEsmItems / kEsmThreads;
    const int items = u.R + u.nE;
    if (items > 0) {
      int64_t itemv[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        int it = lane + kEsmThreads * k;
        if (it >= items) it = items - 1;
        if (it < u.nE)
          itemv[k] = p.eids ? static_cast<int64_t>(p.eids[u.j0 + it]) : u.j0 + it;
        else
          itemv[k] = static_cast<int64_t>(p.indptr[u.i0 + 1 + (it - u.nE)]) - u.j0;
      }
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int it = lane + kEsmThreads * k;
        if (it < u.nE) {
          if (!direct) eid[it] = itemv[k];
        } else if (it < items) {
          const int r = static_cast<int>(itemv[k]);  // in [0, nE]
          rend[it - u.nE + 1] = r;
          atomicOr(&sb[r >> 5], 1u << (r & 31));
        }
      }
    }
  }
  const int64_t f0 = static_cast<int64_t>(p.indptr[u.i0]) - u.j0;
  const int first = f0 < 0 ? -1 : static_cast<int>(f0);
  if (lane == 0) {
    rend[0] = first;
    // the unit's first edge begins a segment and its end closes one (the pieces of rows cut by
    // the unit boundary are segments of their own; the fix-up kernel joins them)
    atomicOr(&sb[0], 1u);
    atomicOr(&sb[u.nE >> 5], 1u << (u.nE & 31));
  }
  if constexpr (std::is_same<DT, float>::value && (HP == 4 || HP == 8)) {
    if (tr) {  // lane-linear pieces -> this lane's four rows, half of the wave per round
      f32x4* st = reinterpret_cast<f32x4*>(stage);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (r) esm_wave_sync();
#pragma unroll
        for (int k = 0; k < 2 * Q; ++k) st[esm_swz(64 * k + wl)] = tmp[2 * Q * r + k];
        esm_wave_sync();
        if ((wl >> 5) == r) {
#pragma unroll
          for (int m = 0; m < 4 * Q; ++m) {
            const f32x4 t = st[esm_swz(4 * Q * (wl & 31) + m)];
            v[m / Q][4 * (m % Q)] = t.x, v[m / Q][4 * (m % Q) + 1] = t.y, v[m / Q][4 * (m % Q) + 2] = t.z,
            v[m / Q][4 * (m % Q) + 3] = t.w;
          }
        }
      }
      if constexpr (BWD) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          esm_wave_sync();
#pragma unroll
          for (int k = 0; k < 2 * Q; ++k) st[esm_swz(64 * k + wl)] = tmp2[2 * Q * r + k];
          esm_wave_sync();
          if ((wl >> 5) == r) {
#pragma unroll
            for (int m = 0; m < 4 * Q; ++m) {
              const f32x4 t = st[esm_swz(4 * Q * (wl & 31) + m)];
              v2[m / Q][4 * (m % Q)] = t.x, v2[m / Q][4 * (m % Q) + 1] = t.y, v2[m / Q][4 * (m % Q) + 2] = t.z,
              v2[m / Q][4 * (m % Q) + 3] = t.w;
            }
          }
        }
      }
    }
  }
  __syncthreads();  // row ends staged; every wave is done with its staging slice (the tables alias it)
  if (!direct) {
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j) {
      int e = e0 + j;
      if (e >= u.nE) e = u.nE > 0 ? u.nE - 1 : 0;
      off[j] = (u.nE > 0 ? eid[e] : 0) * dim;
      if constexpr (BWD) offa[j] = p.out_pos ? (u.j0 + e) * dim : off[j];
    }
    DGLA_ESM_LOAD()
  }
#undef DGLA_ESM_LOAD
  if (!tr && n_valid < kEsmEpl) {  // (only the unit's last lanes)
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
      for (int h = 0; h < HP; ++h)
        if (j >= n_valid) {
          v[j][h] = BWD ? A(0) : -static_cast<A>(__builtin_huge_valf());
          if constexpr (BWD) v2[j][h] = A(0);
        }
  }

  if constexpr (BWD) {
    if (p.b_is_grad) {
      // b = the upstream gradient: form sds = out * g here (rounded to the storage type, as the separate elementwise
      // kernel of python/dgl/backend/pytorch/sparse.py:709-713 leaves it) instead of reading a product somebody wrote
#pragma unroll
      for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
        for (int h = 0; h < HP; ++h) v[j][h] = esm_round<DT>(v[j][h] * v2[j][h]);
    }
  }

  // segment bounds
  const int tail_end = (first < 0 && u.R > 0) ? rend[1] : 0;  // edges [0, tail_end) belong to a row begun earlier
  int carry_begin = u.nE;                                      // edges [carry_begin, nE) continue in the next unit
  {
    const int cb = rend[u.R] < 0 ? 0 : rend[u.R];
    if (cb < u.nE) carry_begin = cb;
  }
  const bool has_carry = carry_begin < u.nE;
  if (lane == 0) p.carry_row[w] = has_carry ? u.i0 + u.R : int64_t(-1);

  // ---- segments from the START BITS of the unit's edges (bit e: a row ends right before edge e,
  // set by the row-end items while they were staged): the lane reads the bits of its four edges
  // and of the edge after them; no search through the row ends ------------------------------------
  unsigned bits;  // bit j: edge e0 + j begins a segment (j = 0 .. 4)
  {
    const unsigned wlo = sb[e0 >> 5], whi = sb[(e0 >> 5) + 1];
    bits = __builtin_amdgcn_alignbit(whi, wlo, e0 & 31) & 31u;
  }
  const unsigned upto_valid = (2u << n_valid) - 1u;  // positions 0 .. n_valid
  const bool a_starts_here = n_valid && (bits & 1u);
  const bool a_ends_here = n_valid && (bits & upto_valid & ~1u);
  const bool a_local = a_starts_here && a_ends_here;
  // per edge: continues the previous edge's segment / is in the head segment / is in a segment
  // that begins and ends inside the lane
  bool same[kEsmEpl], in_head[kEsmEpl], local[kEsmEpl];
#pragma unroll
  for (int j = 0; j < kEsmEpl; ++j) {
    const bool ok = j < n_valid;
    const unsigned through_j = (2u << j) - 1u;  // positions 0 .. j
    same[j] = j > 0 && (!ok || !((bits >> j) & 1u));
    in_head[j] = ok && !(bits & through_j & ~1u);
    local[j] = ok && (bits & through_j) && (bits & upto_valid & ~through_j);
  }
  // the lane's tail segment (the one of its last valid edge) begins in this lane / also ends in it
  const bool z_starts_here = n_valid && (bits & (upto_valid >> 1));
  const bool z_local = z_starts_here && ((bits >> n_valid) & 1u);
  // table row of the head segment = the lane it begins in: the nearest earlier lane of this wave
  // whose tail segment begins there (a max-scan of lane numbers), else the last start bit in front
  // of the wave's edges (every wave looks through the bit words of the waves before it)
  int slot_a;
  {
    int y = z_starts_here ? lane : -1;
    y = esm_max_dpp<0x111, 0xf>(y);
    y = esm_max_dpp<0x112, 0xf>(y);
    y = esm_max_dpp<0x114, 0xf>(y);
    y = esm_max_dpp<0x118, 0xf>(y);
    y = esm_max_dpp<0x142, 0xa>(y);
    y = esm_max_dpp<0x143, 0xc>(y);
    const int prev = esm_dpp<0x138, 0xf>(-1, y);  // wave_shr:1; lane 0 keeps -1
    int before = -1;  // last start position in front of this wave's first edge
    if (wib > 0) {
      const unsigned word = wl < 8 * wib ? sb[wl] : 0u;
      int t = word ? 32 * wl + 31 - __builtin_clz(word) : -1;
      t = esm_max_dpp<0x111, 0xf>(t);
      t = esm_max_dpp<0x112, 0xf>(t);
      t = esm_max_dpp<0x114, 0xf>(t);
      t = esm_max_dpp<0x118, 0xf>(t);
      t = esm_max_dpp<0x142, 0xa>(t);
      before = __builtin_amdgcn_readlane(t, 31);  // 8 * wib <= 24 words: rows 0 and 1
    }
    const int from_before = before < 0 ? 0 : before >> 2;
    slot_a = a_starts_here ? lane : (prev > from_before ? prev : from_before);
  }
  const int slot_z = z_starts_here ? lane : slot_a;  // a tail that came in from earlier lanes IS the head

  // grp[j][h] = OP over the lane's edges in edge j's segment; ghead[h] / glast[h] = that of the
  // lane's head / tail segment
#define DGLA_ESM_GROUPS(OP, IDENT)                                                  \
  _Pragma("unroll") for (int h = 0; h < HP; ++h) {                                  \
    const A f0 = v[0][h];                                                           \
    const A f1 = OP(v[1][h], same[1] ? f0 : (IDENT));                                \
    const A f2 = OP(v[2][h], same[2] ? f1 : (IDENT));                                \
    const A f3 = OP(v[3][h], same[3] ? f2 : (IDENT));                                \
    const A b3 = f3;                                                                \
    const A b2 = same[3] ? b3 : f2;                                                 \
    const A b1 = same[2] ? b2 : f1;                                                 \
    const A b0 = same[1] ? b1 : f0;                                                 \
    ghead[h] = b0;                                                                  \
    glast[h] = b3;                                                                  \
    grp[0][h] = b0, grp[1][h] = b1, grp[2][h] = b2, grp[3][h] = b3;                  \
  }
  // stat[j][h] = OP over the WHOLE segment of edge j: the lane's own value for a segment inside the
  // lane, else the table row of the head / tail segment (two rows per lane)
#define DGLA_ESM_WHOLE(TABLE)                                                        \
  _Pragma("unroll") for (int h = 0; h < HP; ++h) {                                  \
    const A ta = (TABLE)[slot_a * HP + h], tz = (TABLE)[slot_z * HP + h];            \
    _Pragma("unroll") for (int j = 0; j < kEsmEpl; ++j)                              \
      grp[j][h] = local[j] ? grp[j][h] : (in_head[j] ? ta : tz);                     \
  }
  // Segmented inclusive scan over the lanes of x = the lane's tail partial, f = "the tail segment
  // starts in this lane"; afterwards the previous lane's scanned value = everything of this
  // lane's head segment that lies in earlier lanes, and the lane where a crossing segment ends
  // writes its total to TABLE[start lane].
#define DGLA_ESM_SCAN_STEP(CTRL, ROWS, IDENT, OP)                                                  \
    {                                                                                              \
      const int fp = esm_dpp<CTRL, ROWS>(0, f);                                                     \
      _Pragma("unroll") for (int h = 0; h < HP; ++h) {                                              \
        const A xp = esm_dpp<CTRL, ROWS>(static_cast<A>(IDENT), x[h]);                              \
        if (!f) x[h] = OP(x[h], xp);                                                                \
      }                                                                                             \
      f |= fp;                                                                                      \
    }
#define DGLA_ESM_PUBLISH(TABLE, IDENT, OP, IS_MAX)                                                \
  {                                                                                                \
    A x[HP];                                                                                       \
    _Pragma("unroll") for (int h = 0; h < HP; ++h) x[h] = n_valid ? glast[h] : (IDENT);             \
    int f = (!n_valid || z_starts_here) ? 1 : 0;                                                    \
    /* inside the wave: rows of 16 lanes (row_shr 1, 2, 4, 8), then row 0 -> 1 and 2 -> 3         \
       (row_bcast:15), then rows 0-1 -> 2-3 (row_bcast:31); a lane without a source sees            \
       (IDENT, no start) and stays as it is */                                                      \
    if constexpr (std::is_same<A, float>::value) {                                                  \
      if constexpr (IS_MAX) esm_scan_max_f32<HP>(x, f); else esm_scan_sum_f32<HP>(x, f);            \
    } else {                                                                                        \
      DGLA_ESM_SCAN_STEP(0x111, 0xf, IDENT, OP)                                                     \
      DGLA_ESM_SCAN_STEP(0x112, 0xf, IDENT, OP)                                                     \
      DGLA_ESM_SCAN_STEP(0x114, 0xf, IDENT, OP)                                                     \
      DGLA_ESM_SCAN_STEP(0x118, 0xf, IDENT, OP)                                                     \
      DGLA_ESM_SCAN_STEP(0x142, 0xa, IDENT, OP)                                                     \
      DGLA_ESM_SCAN_STEP(0x143, 0xc, IDENT, OP)                                                     \
    }                                                                                               \
    /* across the four waves: what the earlier waves hold of the segment running into this one */   \
    if (wl == 63) {                                                                                 \
      wf[wib] = f;                                                                                  \
      _Pragma("unroll") for (int h = 0; h < HP; ++h) wv[wib * HP + h] = x[h];                       \
    }                                                                                               \
    __syncthreads();                                                                                \
    A cin[HP];                                                                                      \
    _Pragma("unroll") for (int h = 0; h < HP; ++h) cin[h] = (IDENT);                                \
    for (int q = 0; q < wib; ++q) {                                                                 \
      const bool st = wf[q] != 0;                                                                   \
      _Pragma("unroll") for (int h = 0; h < HP; ++h)                                                \
        cin[h] = OP(st ? (IDENT) : cin[h], wv[q * HP + h]);                                         \
    }                                                                                               \
    _Pragma("unroll") for (int h = 0; h < HP; ++h) x[h] = OP(f ? (IDENT) : cin[h], x[h]);           \
    const bool writes = n_valid && !a_starts_here && a_ends_here;                                   \
    _Pragma("unroll") for (int h = 0; h < HP; ++h) {                                                \
      const A t = esm_dpp<0x138, 0xf>(static_cast<A>(IDENT), x[h]); /* wave_shr:1 */                \
      const A total = OP(wl > 0 ? t : cin[h], ghead[h]);                                            \
      if (writes) (TABLE)[slot_a * HP + h] = total;                                                 \
    }                                                                                               \
  }
  const A neg_inf = -static_cast<A>(__builtin_huge_valf());
  auto f_max = [](A a, A b) {
    if constexpr (sizeof(A) == 4) {
      // one v_max_f32: fmaxf() adds canonicalising v_max of its inputs, a > b ? a : b is a compare
      // + select.  (An asm statement is never if-converted: every use below is unconditional —
      // the in-lane sweeps mask the OPERAND with the identity instead of selecting the result.)
      float r;
      asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
      return r;
    } else {
      return a > b ? a : b;
    }
  };
  auto f_add = [](A a, A b) { return a + b; };
  A ghead[HP], glast[HP];
  const bool pub_tail = first < 0 && u.R > 0 && lane == 0;  // row begun in an earlier unit
  const bool tail_empty = tail_end == 0;                    // ... of which only the END falls here
  const bool pub_carry = has_carry && n_valid && e0 + n_valid == u.nE;  // row going on in the next unit
  A* tail_stat = static_cast<A*>(p.tail_stat) + w * 2 * dim;
  A* carry_stat = static_cast<A*>(p.carry_stat) + w * 2 * dim;
  // statistics of the segments cut by the unit boundary, for the fix-up kernel (OFF = 0: max or the
  // backward sum, OFF = dim: the forward sum)
#define DGLA_ESM_STATS(TABLE, IDENT, OFF)                                                              \
  {                                                                                                     \
    if (pub_tail)                                                                                       \
      _Pragma("unroll") for (int h = 0; h < HP; ++h) if (h < dim)                                       \
        tail_stat[(OFF) + h] = tail_empty ? (IDENT) : (a_local ? ghead[h] : (TABLE)[slot_a * HP + h]);  \
    if (pub_carry)                                                                                      \
      _Pragma("unroll") for (int h = 0; h < HP; ++h) if (h < dim)                                       \
        carry_stat[(OFF) + h] = z_local ? glast[h] : (TABLE)[slot_z * HP + h];                          \
  }

  A grp[kEsmEpl][HP];
  if constexpr (BWD) {
    // sum of sds over the row, then sds - out * sum (parts of rows cut by the unit boundary are
    // rewritten by the fix-up)
    DGLA_ESM_GROUPS(f_add, A(0))
    DGLA_ESM_PUBLISH(ts, A(0), f_add, false)
    __syncthreads();
    DGLA_ESM_STATS(ts, A(0), 0)
    DGLA_ESM_WHOLE(ts)
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
      for (int h = 0; h < HP; ++h) v[j][h] = v[j][h] - grp[j][h] * v2[j][h];
  } else {
    // max, exp(x - M)
    DGLA_ESM_GROUPS(f_max, neg_inf)
    DGLA_ESM_PUBLISH(tm, neg_inf, f_max, true)
    __syncthreads();
    DGLA_ESM_STATS(tm, neg_inf, 0)
    DGLA_ESM_WHOLE(tm)
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
      for (int h = 0; h < HP; ++h) v[j][h] = esm_expx<A, PRECISE>(v[j][h] - grp[j][h]);
    // sum, e / S (pieces of rows cut by the unit boundary are normalised by their partial sum here:
    // they are not stored, the fix-up kernel writes those edges)
    DGLA_ESM_GROUPS(f_add, A(0))
    DGLA_ESM_PUBLISH(ts, A(0), f_add, false)
    __syncthreads();
    DGLA_ESM_STATS(ts, A(0), dim)
    DGLA_ESM_WHOLE(ts)
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
      for (int h = 0; h < HP; ++h) v[j][h] = v[j][h] * esm_recip<A>(grp[j][h]);
  }

  // ---- store: complete rows are final; parts of rows cut by the unit boundary are left to the
  // fix-up kernel --------------------------------------------------------------------------------
  if constexpr (std::is_same<DT, float>::value && (HP == 4 || HP == 8)) {
    if (tr) {
      __syncthreads();  // nobody reads the tables any more: the staging slices may overwrite them
      f32x4* st = reinterpret_cast<f32x4*>(stage);
      const int64_t base = (u.j0 + wib * (64 * kEsmEpl)) * dim;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (r) esm_wave_sync();
        if ((wl >> 5) == r) {
#pragma unroll
          for (int m = 0; m < 4 * Q; ++m) {
            f32x4 t;
            t.x = v[m / Q][4 * (m % Q)], t.y = v[m / Q][4 * (m % Q) + 1], t.z = v[m / Q][4 * (m % Q) + 2],
            t.w = v[m / Q][4 * (m % Q) + 3];
            st[esm_swz(4 * Q * (wl & 31) + m)] = t;
          }
        }
        esm_wave_sync();
#pragma unroll
        for (int k = 0; k < 2 * Q; ++k) {
          const int i = 128 * Q * r + 64 * k + wl;  // piece of the wave's slice
          const int e = wib * (64 * kEsmEpl) + i / Q;
          if (e < u.nE && !(e < tail_end || e >= carry_begin))
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(pc) + base + 4 * i) = st[esm_swz(64 * k + wl)];
        }
      }
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < kEsmEpl; ++j) {
    const int e = e0 + j;
    if (e >= u.nE) continue;
    const bool partial = e < tail_end || e >= carry_begin;
    if (partial) continue;  // written by the fix-up kernel, which knows the whole row's statistics
    // store offset: the edge's id like the loads, or its position (scores gathered in, softmax handed on in
    // position order: one pass instead of a gather pass in front of a map-free softmax)
    const int64_t so = p.out_pos ? (u.j0 + e) * dim : off[j];
    bool done = false;
    if constexpr (std::is_same<DT, float>::value && HP >= 4) {
      if (p.vec4) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < HP / 4; ++q) {
          f32x4 t;
          t.x = v[j][4 * q], t.y = v[j][4 * q + 1], t.z = v[j][4 * q + 2], t.w = v[j][4 * q + 3];
          *reinterpret_cast<f32x4*>(pc + so + 4 * q) = t;
        }
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int h = 0; h < HP; ++h)
        if (h < dim) pc[so + h] = from_acc<DT>(v[j][h]);
    }
  }
}

#undef DGLA_ESM_STATS
#undef DGLA_ESM_GROUPS
#undef DGLA_ESM_WHOLE
#undef DGLA_ESM_PUBLISH
#undef DGLA_ESM_SCAN_STEP

constexpr int kFixU = 8;  // edges in flight per lane of the fix-up kernel
constexpr int kFixB = 8;  // unit boundaries looked at by one wavefront of the fix-up kernel

// The row that unit w carries into unit w + 1 (row >= 0), done by one wavefront: the part of it
// inside unit w plus, when the row ends in unit w + 1, that unit's tail — one contiguous range of
// the row's edges, found from three neighbouring carry_row entries, two plan entries and two row
// pointers (three dependent memory round trips before the edges; walking unit by unit through
// esm_unit() made it ~16 and the kernel latency-bound).
template <typename Idx, typename DT, bool BWD, bool PRECISE>
__device__ __forceinline__ void edge_softmax_fixup_boundary(const EsmParams<Idx>& p, int64_t w, int64_t row) {
  using A = typename Acc<DT>::type;
  const int dim = p.dim, hp = 1 << p.log2_hp;
  const int64_t crm = w > 0 ? p.carry_row[w - 1] : int64_t(-1);
  const int64_t crp = w + 1 < p.num_units ? p.carry_row[w + 1] : int64_t(-1);
  const int64_t j0 = p.planj[w], j1 = p.planj[w + 1];  // first edge of unit w / of unit w + 1
  const A* cs = static_cast<const A*>(p.carry_stat);
  const A* ts = static_cast<const A*>(p.tail_stat);
  const DT* __restrict__ pa = static_cast<const DT*>(p.a);
  const DT* __restrict__ pb = static_cast<const DT*>(p.b);
  DT* __restrict__ pc = static_cast<DT*>(p.c);
  const int lane = threadIdx.x & 63;
  const int h = lane & (hp - 1);
  const int es = 64 >> p.log2_hp;
  const int64_t row_begin = static_cast<int64_t>(p.indptr[row]), row_end = static_cast<int64_t>(p.indptr[row + 1]);
  // units [sa, s2) carry the row, unit s2 holds its tail
  int64_t sa = w, s2 = w + 1;
  if (crm == row) {
    sa = w - 1;
    while (sa > 0 && p.carry_row[sa - 1] == row) --sa;
  }
  if (crp == row) {
    s2 = w + 2;
    while (s2 < p.num_units && p.carry_row[s2] == row) ++s2;
  }
  const int64_t t0 = crm == row ? j0 : row_begin;  // this wave's edges of the row: [t0, t1)
  const int64_t t1 = crp == row ? j1 : row_end;
  if constexpr (std::is_same<DT, float>::value) {
    if (p.vec4) {
      // fp32 rows of whole 16-byte pieces: a lane owns FOUR features of an edge (one 16-byte load
      // and store, one address computation), dim / 4 lanes per edge — the scalar loop below spends
      // ~25 instructions per feature, which for the 7 % of C2's edges in cut rows was most of
      // this kernel's 0.14 ms
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const int Q = dim >> 2, q = lane % Q, slot = lane / Q, per = 64 / Q;  // Q in {1, 2, 4}
      if (slot >= per) return;  // (Q == 3 cannot occur: vec4 needs dim == its power-of-two padding)
      const float* cs4 = static_cast<const float*>(p.carry_stat) + 4 * q;
      const float* ts4 = static_cast<const float*>(p.tail_stat) + 4 * q;
      const float* fa = reinterpret_cast<const float*>(pa);
      const float* fb = reinterpret_cast<const float*>(pb);
      float* fc = reinterpret_cast<float*>(pc);
      auto ld4 = [](const float* ptr) { return *reinterpret_cast<const f32x4*>(ptr); };
      if constexpr (BWD) {
        f32x4 sum = ld4(ts4 + s2 * 2 * dim);
        for (int64_t u = sa; u < s2; ++u) sum += ld4(cs4 + u * 2 * dim);
        for (int64_t tb = t0 + slot; tb < t1; tb += per * kFixU) {
          f32x4 xb[kFixU], xa[kFixU];
          int64_t off[kFixU], so[kFixU];
#pragma unroll
          for (int k = 0; k < kFixU; ++k) {
            const int64_t j = tb + k * per < t1 ? tb + k * per : tb;
            off[k] = (p.eids ? static_cast<int64_t>(p.eids[j]) : j) * dim + 4 * q;
            so[k] = p.out_pos ? j * dim + 4 * q : off[k];   // the saved softmax and the result: by position
          }
#pragma unroll
          for (int k = 0; k < kFixU; ++k) {
            xb[k] = ld4(fb + off[k]), xa[k] = ld4(fa + so[k]);
            if (p.b_is_grad) xb[k] = xb[k] * xa[k];
          }
#pragma unroll
          for (int k = 0; k < kFixU; ++k)
            if (tb + k * per < t1) *reinterpret_cast<f32x4*>(fc + so[k]) = xb[k] - sum * xa[k];
        }
      } else {
        const f32x4 mt = ld4(ts4 + s2 * 2 * dim), st = ld4(ts4 + s2 * 2 * dim + dim);
        f32x4 M = mt;
        for (int64_t u = sa; u < s2; ++u) {
          const f32x4 m = ld4(cs4 + u * 2 * dim);
#pragma unroll
          for (int c = 0; c < 4; ++c) M[c] = M[c] > m[c] ? M[c] : m[c];
        }
        f32x4 S = {0.f, 0.f, 0.f, 0.f};
        for (int64_t u = sa; u < s2; ++u) {
          const f32x4 m = ld4(cs4 + u * 2 * dim), sp = ld4(cs4 + u * 2 * dim + dim);
#pragma unroll
          for (int c = 0; c < 4; ++c) S[c] += sp[c] * esm_expx<float, PRECISE>(m[c] - M[c]);
        }
        f32x4 inv;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (st[c] > 0.f) S[c] += st[c] * esm_expx<float, PRECISE>(mt[c] - M[c]);
          inv[c] = esm_recip<float>(S[c]);
        }
        for (int64_t tb = t0 + slot; tb < t1; tb += per * kFixU) {
          f32x4 xc[kFixU];
          int64_t off[kFixU], so[kFixU];
#pragma unroll
          for (int k = 0; k < kFixU; ++k) {
            const int64_t j = tb + k * per < t1 ? tb + k * per : tb;
            off[k] = (p.eids ? static_cast<int64_t>(p.eids[j]) : j) * dim + 4 * q;
            so[k] = p.out_pos ? j * dim + 4 * q : off[k];
          }
#pragma unroll
          for (int k = 0; k < kFixU; ++k) xc[k] = ld4(fa + off[k]);
#pragma unroll
          for (int k = 0; k < kFixU; ++k)
            if (tb + k * per < t1) {
              f32x4 o;
#pragma unroll
              for (int c = 0; c < 4; ++c) o[c] = esm_expx<float, PRECISE>(xc[k][c] - M[c]) * inv[c];
              *reinterpret_cast<f32x4*>(fc + so[k]) = o;
            }
        }
      }
      return;
    }
  }
  {
    if (h >= dim) return;
    if constexpr (BWD) {
      A sum = A(0);
      for (int64_t q = sa; q < s2; ++q) sum += cs[q * 2 * dim + h];
      sum += ts[s2 * 2 * dim + h];
      // kFixU edges in flight per lane: the loop is a chain of HBM round trips otherwise
      for (int64_t tb = t0 + (lane >> p.log2_hp); tb < t1; tb += es * kFixU) {
        int64_t off[kFixU], so[kFixU];
        A xb[kFixU], xa[kFixU];
#pragma unroll
        for (int k = 0; k < kFixU; ++k) {
          const int64_t j = tb + k * es < t1 ? tb + k * es : tb;
          off[k] = (p.eids ? static_cast<int64_t>(p.eids[j]) : j) * dim + h;
          so[k] = p.out_pos ? j * dim + h : off[k];
        }
#pragma unroll
        for (int k = 0; k < kFixU; ++k) {
          xb[k] = to_acc<DT>(pb[off[k]]);
          xa[k] = to_acc<DT>(pa[so[k]]);
          if (p.b_is_grad) xb[k] = esm_round<DT>(xb[k] * xa[k]);
        }
#pragma unroll
        for (int k = 0; k < kFixU; ++k)
          if (tb + k * es < t1) pc[so[k]] = from_acc<DT>(xb[k] - sum * xa[k]);
      }
    } else {
      A M = ts[s2 * 2 * dim + h];
      for (int64_t q = sa; q < s2; ++q) {
        const A m = cs[q * 2 * dim + h];
        M = M > m ? M : m;
      }
      A S = A(0);
      for (int64_t q = sa; q < s2; ++q)
        S += cs[q * 2 * dim + dim + h] * esm_expx<A, PRECISE>(cs[q * 2 * dim + h] - M);
      {
        const A s_t = ts[s2 * 2 * dim + dim + h];
        if (s_t > A(0)) S += s_t * esm_expx<A, PRECISE>(ts[s2 * 2 * dim + h] - M);
      }
      // straight from the scores (the main kernel wrote nothing for these edges: re-reading x
      // costs what re-reading an un-normalised output would, and the main kernel saves the write)
      const A inv = esm_recip<A>(S);
      for (int64_t tb = t0 + (lane >> p.log2_hp); tb < t1; tb += es * kFixU) {
        int64_t off[kFixU], so[kFixU];
        A xc[kFixU];
#pragma unroll
        for (int k = 0; k < kFixU; ++k) {
          const int64_t j = tb + k * es < t1 ? tb + k * es : tb;
          off[k] = (p.eids ? static_cast<int64_t>(p.eids[j]) : j) * dim + h;
          so[k] = p.out_pos ? j * dim + h : off[k];
        }
#pragma unroll
        for (int k = 0; k < kFixU; ++k) xc[k] = to_acc<DT>(pa[off[k]]);
#pragma unroll
        for (int k = 0; k < kFixU; ++k)
          if (tb + k * es < t1) pc[so[k]] = from_acc<DT>(esm_expx<A, PRECISE>(xc[k] - M) * inv);
      }
    }
  }
}

// kFixB unit boundaries per wavefront, four wavefronts per workgroup: boundaries that cut no row
// (all of them, on a graph without rows longer than kEsmSlack) cost one load for the kFixB of
// them; the others are done one after the other by the whole wavefront.  A wavefront's boundaries
// lie num_waves apart, not next to each other: the 17 consecutive boundaries a 17 k-edge hub row
// cuts then go to 17 wavefronts instead of 3 (the kernel's duration was the longest such chain).
template <typename Idx, typename DT, bool BWD, bool PRECISE>
__global__ __launch_bounds__(256) void edge_softmax_fixup_kernel(const EsmParams<Idx> p) {
  const int64_t num_waves = (p.num_units + kFixB - 1) / kFixB;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (wave >= num_waves) return;
  const int lane = threadIdx.x & 63;
  const int64_t my_w = wave + lane * num_waves;
  const int64_t mine = (lane < kFixB && my_w < p.num_units) ? p.carry_row[my_w] : int64_t(-1);
  uint64_t cut = __ballot(mine >= 0);
  while (cut) {
    const int k = __builtin_ctzll(cut);
    cut &= cut - 1;
    const int64_t row = __shfl(mine, k, 64);
    edge_softmax_fixup_boundary<Idx, DT, BWD, PRECISE>(p, wave + k * num_waves, row);
  }
}

struct EsmGeometry {
  int log2_hp;
  int64_t num_units;
  size_t off_plan, off_planj, off_carry_row, off_carry_stat, off_tail_stat, total;
};

static size_t esm_align(size_t x) { return (x + 255) / 256 * 256; }

static EsmGeometry esm_geometry(int64_t num_rows, int64_t nnz, int dim, size_t acc_bytes) {
  EsmGeometry g;
  g.log2_hp = 0;
  while ((1 << g.log2_hp) < dim) ++g.log2_hp;
  g.num_units = (num_rows + nnz + kEsmStride - 1) / kEsmStride;
  size_t off = 0;
  g.off_plan = off;
  off = esm_align(off + sizeof(int64_t) * (g.num_units + 1));
  g.off_planj = off;
  off = esm_align(off + sizeof(int64_t) * (g.num_units + 1));
  g.off_carry_row = off;
  off = esm_align(off + sizeof(int64_t) * g.num_units);
  g.off_carry_stat = off;
  off = esm_align(off + acc_bytes * g.num_units * 2 * dim);
  g.off_tail_stat = off;
  off = esm_align(off + acc_bytes * g.num_units * 2 * dim);
  g.total = off;
  return g;
}

constexpr int kEsmMaxDim = 16;

template <typename Idx, typename DT>
static int edge_softmax_merge_run(const CsrView& csr, const void* a, const void* b, void* c,
                                  int dim, bool backward, void* ws, bool plan_valid,
                                  hipStream_t s, bool out_pos, bool b_is_grad) {
  using A = typename Acc<DT>::type;
  const EsmGeometry g = esm_geometry(csr.num_rows, csr.nnz, dim, sizeof(A));
  char* wsp = static_cast<char*>(ws);
  EsmParams<Idx> p;
  p.indptr = static_cast<const Idx*>(csr.indptr);
  p.eids = static_cast<const Idx*>(csr.eids);
  p.num_rows = csr.num_rows;
  p.nnz = csr.nnz;
  p.num_units = g.num_units;
  p.plan = reinterpret_cast<const int64_t*>(wsp + g.off_plan);
  p.planj = reinterpret_cast<const int64_t*>(wsp + g.off_planj);
  p.a = a;
  p.b = b;
  p.c = c;
  p.dim = dim;
  p.log2_hp = g.log2_hp;
  {
    const auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    p.vec4 = (std::is_same<DT, float>::value && dim % 4 == 0 && (1 << g.log2_hp) == dim && al(a) && al(c) &&
              (!backward || al(b))) ? 1 : 0;
  }
  p.xcd = (tuning_flags() & kTuneXcd) ? 1 : 0;
  p.out_pos = (out_pos && csr.eids) ? 1 : 0;
  p.b_is_grad = (b_is_grad && backward) ? 1 : 0;
  p.carry_row = reinterpret_cast<int64_t*>(wsp + g.off_carry_row);
  p.carry_stat = wsp + g.off_carry_stat;
  p.tail_stat = wsp + g.off_tail_stat;
  const int hp = 1 << g.log2_hp;
  // per workgroup: the segment tables ((max | sum) forward, sum backward), the waves' scan
  // hand-over, edge ids, row ends
  // (the fp32 row-transposition slices, 4 waves x 128 rows x hp x 4 B, share the tables' memory)
  const size_t tables = sizeof(A) * kEsmSegCap * hp * (backward ? 1 : 2);
  const size_t slices = (sizeof(DT) == 4 && (hp == 4 || hp == 8)) ? size_t(kEsmItems / 2) * hp * 4 : 0;
  p.region_bytes = static_cast<int>(((tables > slices ? tables : slices) + 15) / 16 * 16);
  const size_t per_block = p.region_bytes + sizeof(A) * kEsmWaves * hp +
                           sizeof(int64_t) * kEsmItems + sizeof(int) * (kEsmItems + 2 + kEsmWaves) +
                           sizeof(unsigned) * (kEsmItems / 32 + 2) + 8;
  p.wave_lds_bytes = static_cast<int>((per_block + 15) / 16 * 16);
  if (!plan_valid) {
    const int64_t n = g.num_units + 1;
    hipLaunchKernelGGL((esm_plan_kernel<Idx>), dim3(static_cast<unsigned>((n + 255) / 256)),
                       dim3(256), 0, s, p.indptr, csr.num_rows, csr.nnz, g.num_units,
                       reinterpret_cast<int64_t*>(wsp + g.off_plan),
                       reinterpret_cast<int64_t*>(wsp + g.off_planj));
  }
  const unsigned blocks = static_cast<unsigned>(g.num_units);
  const size_t lds = static_cast<size_t>(p.wave_lds_bytes);
  constexpr bool kPrecise = sizeof(DT) == 8;
#define DGLA_ESM_LAUNCH(HPV)                                                                          \
  do {                                                                                                \
    if (backward) {                                                                                   \
      hipLaunchKernelGGL((edge_softmax_merge_kernel<Idx, DT, true, kPrecise, HPV>), dim3(blocks),      \
                         dim3(kEsmThreads), lds, s, p);                                                  \
      hipLaunchKernelGGL((edge_softmax_fixup_kernel<Idx, DT, true, kPrecise>),                        \
                         dim3(static_cast<unsigned>((g.num_units + 4 * kFixB - 1) / (4 * kFixB))), dim3(256), 0, s, p);     \
    } else {                                                                                          \
      hipLaunchKernelGGL((edge_softmax_merge_kernel<Idx, DT, false, kPrecise, HPV>), dim3(blocks),     \
                         dim3(kEsmThreads), lds, s, p);                                                  \
      hipLaunchKernelGGL((edge_softmax_fixup_kernel<Idx, DT, false, kPrecise>),                       \
                         dim3(static_cast<unsigned>((g.num_units + 4 * kFixB - 1) / (4 * kFixB))), dim3(256), 0, s, p);     \
    }                                                                                                 \
  } while (0)
  switch (hp) {
    case 1: DGLA_ESM_LAUNCH(1); break;
    case 2: DGLA_ESM_LAUNCH(2); break;
    case 4: DGLA_ESM_LAUNCH(4); break;
    case 8: DGLA_ESM_LAUNCH(8); break;
    default: DGLA_ESM_LAUNCH(16); break;
  }
#undef DGLA_ESM_LAUNCH
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t edge_softmax_workspace_bytes(int64_t num_rows, int64_t nnz, int dtype, int64_t dim) {
  // lane-group kernel (no scratch) beyond what one wave's LDS slice holds: 16 features with
  // 4-byte accumulators, 8 with fp64 (2 x 256 x hp x 8 B for the backward pass <= 64 KiB)
  if (dim < 1 || dim > (dtype == kF64 ? kEsmMaxDim / 2 : kEsmMaxDim)) return 0;
  return esm_geometry(num_rows, nnz, static_cast<int>(dim), dtype == kF64 ? 8 : 4).total;
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
                        int64_t dim, bool backward, void* ws, size_t ws_bytes, bool plan_valid,
                        hipStream_t s, bool out_pos, bool b_is_grad) {
  const int d = static_cast<int>(dim);
  const size_t need = edge_softmax_workspace_bytes(csr.num_rows, csr.nnz, dtype, dim);
  const bool merge = need > 0 && ws != nullptr && ws_bytes >= need;
  if (b_is_grad && (!merge || !backward)) {
    last_error() = "DGLA_ESM_B_IS_GRAD: backward only, and only with the merge-path kernels (workspace given, feature length <= 16)";
    return -1;
  }
  if (out_pos && !merge) {
    last_error() = "DGLA_ESM_OUT_POSITION: only with the merge-path kernels (workspace given, feature length <= 16)";
    return -1;
  }
#define DGLA_ES(DT)                                                                          \
  if (merge)                                                                                 \
    return csr.idbits == 32                                                                  \
               ? edge_softmax_merge_run<int32_t, DT>(csr, a, b, c, d, backward, ws, plan_valid, s, out_pos, b_is_grad) \
               : edge_softmax_merge_run<int64_t, DT>(csr, a, b, c, d, backward, ws, plan_valid, s, out_pos, b_is_grad); \
  return csr.idbits == 32 ? edge_softmax_run<int32_t, DT>(csr, a, b, c, d, backward, s)     \
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
