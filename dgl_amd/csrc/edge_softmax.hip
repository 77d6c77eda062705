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
  const float hi = x * 1.44269504088896341f;
  const float lo = __builtin_fmaf(x, 1.44269504088896341f, -hi) + x * 1.92596299112661746e-8f;
  const float r = __builtin_amdgcn_exp2f(hi) * __builtin_fmaf(lo, 0.693147180559945309f, 1.0f);
  return hi < -150.f ? 0.f : r;  // also keeps exp(-inf) = 0 (lo would be NaN there)
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
// into units of kEsmItems items (edges + row ends) exactly like the SpMM; one wavefront per
// unit:
//   1. stage the unit's row ends and edge ids in LDS; every lane loads four consecutive edges of
//      the unit with all their features into registers (one HBM round trip for the whole unit);
//   2. per-segment max / sum through small LDS tables (a segment is a row, or the part of a row
//      inside this unit); the values never leave the registers;
//   3. rows that lie entirely inside the unit are finished and written (read once, written
//      once); a row that straddles units leaves (max, sum) per part in the workspace and its
//      edges un-normalised;
//   4. a fix-up kernel, one 64-lane block per unit, merges the parts of each straddling row
//      ((m, s) pairs combine as S = sum_i s_i * exp(m_i - M)) and rescales its own unit's
//      edges, so even a hub row is handled by as many blocks as it has units.
// =========================================================================================
constexpr int kEsmItems = 256;

template <typename Idx>
struct EsmParams {
  const Idx* indptr;
  const Idx* eids;
  int64_t num_rows, nnz, num_units;
  const int64_t* plan;  // [num_units + 1]
  const void* a;
  const void* b;
  void* c;
  int dim, log2_hp;
  int wave_lds_bytes;
  int vec4;  // fp32, dim % 4 == 0 == padded width, 16-byte aligned operands: rows move as 16-byte pieces
  int xcd;   // units in XCD-contiguous order (kTuneXcd): neighbouring units share an L2
  int64_t* carry_row;  // [num_units] row continued in the next unit, or -1
  void* carry_stat;    // [num_units, 2 * dim] accumulators: (m | s) forward, (sum | -) backward
  void* tail_stat;     // [num_units, 2 * dim]
};

template <typename Idx>
__global__ void esm_plan_kernel(const Idx* __restrict__ indptr, int64_t num_rows, int64_t nnz,
                                int64_t num_units, int64_t* __restrict__ plan) {
  // plan[w] = largest i in [0, N] with indptr[i] + i <= w * kEsmItems (see spmm_csr.cuh)
  const int64_t w = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (w > num_units) return;
  int64_t d = w * kEsmItems;
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
  plan[w] = lo;
}

struct EsmUnit {
  int64_t i0, j0;
  int R, nE;
};

template <typename Idx>
__device__ __forceinline__ EsmUnit esm_unit(const EsmParams<Idx>& p, int64_t w) {
  EsmUnit u;
  const int64_t total = p.num_rows + p.nnz;
  const int64_t d0 = w * kEsmItems;
  int64_t d1 = d0 + kEsmItems;
  if (d1 > total) d1 = total;
  u.i0 = p.plan[w];
  const int64_t i1 = p.plan[w + 1];
  u.j0 = d0 - u.i0;
  u.R = static_cast<int>(i1 - u.i0);
  u.nE = static_cast<int>((d1 - i1) - u.j0);
  return u;
}

template <typename A, bool PRECISE>
__device__ __forceinline__ A esm_expx(A x) {
  if constexpr (PRECISE)
    return static_cast<A>(exp(static_cast<double>(x)));
  else
    return esm_exp<A>(x);
}

// Segments (rows or row parts) whose statistics one round of the balanced reduce keeps in LDS.
constexpr int kEsmSegCap = 64;

// Orders one wave's LDS traffic between the passes of the reduce.  Each wave works on its own
// LDS slice and the number of rounds differs from wave to wave, so a block barrier is neither
// needed nor allowed here; LDS operations of ONE wave execute in issue order.
__device__ __forceinline__ void esm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS atomics (ds_max_f32 / ds_add_f32 and the f64 forms).  Lanes of one wave reach them in
// program order, so the order of the additions is fixed: results are run-to-run identical.
template <typename A>
__device__ __forceinline__ void esm_lds_max(A* addr, A v) {
  (void)__hip_atomic_fetch_max(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename A>
__device__ __forceinline__ void esm_lds_add(A* addr, A v) {
  (void)__hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One wavefront per unit.  Every lane owns kEsmEpl = 4 CONSECUTIVE edges of the unit and keeps
// all HP features of them in registers: the scores travel HBM -> registers -> HBM (for edge ids =
// positions a lane's four rows are 4 * dim * s contiguous bytes: fully coalesced 16-byte loads
// and stores); LDS holds only the unit's row ends, its edge ids and two small per-segment tables.
// A segment is a row, or the part of a row inside this unit.  Lanes that share a segment meet in
// the tables through LDS atomics (ds_max / ds_add; lanes of one wave reach them in program order,
// so the floating-point sums are run-to-run identical):
//   forward : pass 1 max -> tm;  pass 2 ex = exp(x - M) kept in registers, sum -> ts;
//             pass 3 scale by 1 / S; segments cut by the unit boundary stay un-normalised and
//             publish (M, S) for the fix-up kernel
//   backward: pass 1 sum(sds) -> ts;  pass 2 c = sds - sum * out
// A hub row and forty 5-edge rows cost the same (4 edges per lane either way).  The tables hold
// kEsmSegCap segments; a unit with more row ends than that (runs of tiny rows) takes several
// rounds over the same registers.
constexpr int kEsmEpl = kEsmItems / 64;

template <typename Idx, typename DT, bool BWD, bool PRECISE, int HP>
__global__ __launch_bounds__(256) void edge_softmax_merge_kernel(const EsmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  extern __shared__ __align__(16) unsigned char esm_smem[];
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  unsigned blk = blockIdx.x;
  if (p.xcd) {  // block b runs on XCD b % 8: give every XCD one contiguous eighth of the units
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u;
    const unsigned x = blk & 7u, i = blk >> 3;
    blk = x * q + (x < r ? x : r) + i;
  }
  const int64_t w = static_cast<int64_t>(blk) * wpb + wib;
  const int dim = p.dim;
  unsigned char* base = esm_smem + static_cast<size_t>(wib) * p.wave_lds_bytes;
  A* tm = reinterpret_cast<A*>(base);                       // [kEsmSegCap * HP]  (forward only)
  A* ts = tm + (BWD ? 0 : kEsmSegCap * HP);                 // [kEsmSegCap * HP]
  int64_t* eid = reinterpret_cast<int64_t*>(ts + kEsmSegCap * HP);  // [kEsmItems]
  int* rend = reinterpret_cast<int*>(eid + kEsmItems);      // [kEsmItems + 2]
  const DT* __restrict__ pa = static_cast<const DT*>(p.a);
  const DT* __restrict__ pb = static_cast<const DT*>(p.b);
  DT* __restrict__ pc = static_cast<DT*>(p.c);
  if (w >= p.num_units) return;  // no block-wide barrier below: a wave only touches its own slice

  const EsmUnit u = esm_unit<Idx>(p, w);
  // ---- stage row ends and edge ids: kEsmItems / 64 = 4 independent loads per lane issued back to
  // back (addresses clamped, not predicated), one HBM round trip ---------------------------------
  {
    constexpr int KS = kEsmItems / 64;
    const int items = u.R + u.nE;
    if (items > 0) {
      int64_t itemv[KS];
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        int it = lane + 64 * k;
        if (it >= items) it = items - 1;
        if (it < u.nE)
          itemv[k] = p.eids ? static_cast<int64_t>(p.eids[u.j0 + it]) : u.j0 + it;
        else
          itemv[k] = static_cast<int64_t>(p.indptr[u.i0 + 1 + (it - u.nE)]) - u.j0;
      }
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int it = lane + 64 * k;
        if (it < u.nE)
          eid[it] = itemv[k];
        else if (it < items)
          rend[it - u.nE + 1] = static_cast<int>(itemv[k]);
      }
    }
  }
  const int64_t f0 = static_cast<int64_t>(p.indptr[u.i0]) - u.j0;
  const int first = f0 < 0 ? -1 : static_cast<int>(f0);
  if (lane == 0) rend[0] = first;
  esm_wave_sync();

  // segment bounds
  const int tail_end = (first < 0 && u.R > 0) ? rend[1] : 0;  // edges [0, tail_end) belong to a row begun earlier
  int carry_begin = u.nE;                                      // edges [carry_begin, nE) continue in the next unit
  {
    const int cb = rend[u.R] < 0 ? 0 : rend[u.R];
    if (cb < u.nE) carry_begin = cb;
  }
  const bool has_carry = carry_begin < u.nE;
  if (lane == 0) p.carry_row[w] = has_carry ? u.i0 + u.R : int64_t(-1);
  const int nseg = u.R + (has_carry ? 1 : 0);
  auto seg_end = [&](int sg) { return sg < u.R ? rend[sg + 1] : u.nE; };  // carry segment: sg == R

  // ---- load: this lane's edges [e0, e0 + 4) with all their features -----------------------------
  const int e0 = lane * kEsmEpl;
  A v[kEsmEpl][HP];
  A v2[BWD ? kEsmEpl : 1][BWD ? HP : 1];  // backward: out values
  int64_t off[kEsmEpl];
  int seg[kEsmEpl];
#pragma unroll
  for (int j = 0; j < kEsmEpl; ++j) {
    int e = e0 + j;
    if (e >= u.nE) e = u.nE > 0 ? u.nE - 1 : 0;  // clamp: a valid row, never stored
    off[j] = (u.nE > 0 ? eid[e] : 0) * dim;
  }
  if (u.nE > 0) {
    bool done = false;
    if constexpr (std::is_same<DT, float>::value && HP >= 4) {
      if (p.vec4) {  // rows are whole 16-byte pieces
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
          for (int q = 0; q < HP / 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>((BWD ? pb : pa) + off[j] + 4 * q);
            v[j][4 * q] = t.x, v[j][4 * q + 1] = t.y, v[j][4 * q + 2] = t.z, v[j][4 * q + 3] = t.w;
            if constexpr (BWD) {
              const f32x4 t2 = *reinterpret_cast<const f32x4*>(pa + off[j] + 4 * q);
              v2[j][4 * q] = t2.x, v2[j][4 * q + 1] = t2.y, v2[j][4 * q + 2] = t2.z, v2[j][4 * q + 3] = t2.w;
            }
          }
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int j = 0; j < kEsmEpl; ++j)
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          const int hh = h < dim ? h : dim - 1;  // clamp: padded features load a valid element
          v[j][h] = to_acc<DT>((BWD ? pb : pa)[off[j] + hh]);
          if constexpr (BWD) v2[j][h] = to_acc<DT>(pa[off[j] + hh]);
        }
    }
  }
  // segment of every edge: a short search for the first, then a walk
  {
    int lo = 0, hi = nseg > 0 ? nseg - 1 : 0;
    while (lo < hi) {  // smallest sg with seg_end(sg) > e0
      const int mid = (lo + hi) >> 1;
      if (seg_end(mid) > e0)
        hi = mid;
      else
        lo = mid + 1;
    }
    int t = lo;
#pragma unroll
    for (int j = 0; j < kEsmEpl; ++j) {
      const int e = e0 + j;
      if (e < u.nE)
        while (e >= seg_end(t)) ++t;
      seg[j] = e < u.nE ? t : -1;
    }
  }

  const A neg_inf = -static_cast<A>(__builtin_huge_valf());
  for (int c0 = 0; c0 < nseg; c0 += kEsmSegCap) {
    const int c1 = c0 + kEsmSegCap < nseg ? c0 + kEsmSegCap : nseg;
    for (int i = lane; i < (c1 - c0) * HP; i += 64) {
      if constexpr (!BWD) tm[i] = neg_inf;
      ts[i] = A(0);
    }
    esm_wave_sync();
    auto in_round = [&](int j) { return seg[j] >= c0 && seg[j] < c1; };
    if constexpr (BWD) {
      {  // pass 1: per-segment sum of sds
        A acc[HP];
        int cur = -1;
#pragma unroll
        for (int j = 0; j < kEsmEpl; ++j) {
          if (!in_round(j)) continue;
          if (seg[j] != cur) {
            if (cur >= 0)
#pragma unroll
              for (int h = 0; h < HP; ++h) esm_lds_add(&ts[(cur - c0) * HP + h], acc[h]);
            cur = seg[j];
#pragma unroll
            for (int h = 0; h < HP; ++h) acc[h] = A(0);
          }
#pragma unroll
          for (int h = 0; h < HP; ++h) acc[h] += v[j][h];
        }
        if (cur >= 0)
#pragma unroll
          for (int h = 0; h < HP; ++h) esm_lds_add(&ts[(cur - c0) * HP + h], acc[h]);
      }
      esm_wave_sync();
#pragma unroll
      for (int j = 0; j < kEsmEpl; ++j) {  // pass 2 (partial segments are rewritten by the fix-up)
        if (!in_round(j)) continue;
#pragma unroll
        for (int h = 0; h < HP; ++h) v[j][h] = v[j][h] - ts[(seg[j] - c0) * HP + h] * v2[j][h];
      }
    } else {
      {  // pass 1: per-segment max
        A acc[HP];
        int cur = -1;
#pragma unroll
        for (int j = 0; j < kEsmEpl; ++j) {
          if (!in_round(j)) continue;
          if (seg[j] != cur) {
            if (cur >= 0)
#pragma unroll
              for (int h = 0; h < HP; ++h) esm_lds_max(&tm[(cur - c0) * HP + h], acc[h]);
            cur = seg[j];
#pragma unroll
            for (int h = 0; h < HP; ++h) acc[h] = neg_inf;
          }
#pragma unroll
          for (int h = 0; h < HP; ++h) acc[h] = acc[h] > v[j][h] ? acc[h] : v[j][h];
        }
        if (cur >= 0)
#pragma unroll
          for (int h = 0; h < HP; ++h) esm_lds_max(&tm[(cur - c0) * HP + h], acc[h]);
      }
      esm_wave_sync();
      {  // pass 2: ex = exp(x - M), per-segment sum
        A acc[HP];
        int cur = -1;
#pragma unroll
        for (int j = 0; j < kEsmEpl; ++j) {
          if (!in_round(j)) continue;
          if (seg[j] != cur) {
            if (cur >= 0)
#pragma unroll
              for (int h = 0; h < HP; ++h) esm_lds_add(&ts[(cur - c0) * HP + h], acc[h]);
            cur = seg[j];
#pragma unroll
            for (int h = 0; h < HP; ++h) acc[h] = A(0);
          }
#pragma unroll
          for (int h = 0; h < HP; ++h) {
            const A ex = esm_expx<A, PRECISE>(v[j][h] - tm[(cur - c0) * HP + h]);
            v[j][h] = ex;
            acc[h] += ex;
          }
        }
        if (cur >= 0)
#pragma unroll
          for (int h = 0; h < HP; ++h) esm_lds_add(&ts[(cur - c0) * HP + h], acc[h]);
      }
      esm_wave_sync();
#pragma unroll
      for (int j = 0; j < kEsmEpl; ++j) {  // pass 3: normalise whole rows
        if (!in_round(j)) continue;
        const bool partial = (seg[j] == 0 && first < 0) || seg[j] == u.R;
        if (partial) continue;
#pragma unroll
        for (int h = 0; h < HP; ++h) v[j][h] = v[j][h] * (A(1) / ts[(seg[j] - c0) * HP + h]);
      }
    }
    // segments cut by the unit boundary publish their statistics for the fix-up kernel
    if (lane < HP && lane < dim) {
      const int h = lane;
      if (c0 == 0 && first < 0 && u.R > 0) {  // tail of a row begun in an earlier unit
        A* stat = static_cast<A*>(p.tail_stat) + w * 2 * dim;
        if constexpr (BWD) {
          stat[h] = ts[h];
        } else {
          stat[h] = tm[h];
          stat[dim + h] = ts[h];
        }
      }
      if (has_carry && u.R >= c0 && u.R < c1) {
        A* stat = static_cast<A*>(p.carry_stat) + w * 2 * dim;
        const int o = (u.R - c0) * HP + h;
        if constexpr (BWD) {
          stat[h] = ts[o];
        } else {
          stat[h] = tm[o];
          stat[dim + h] = ts[o];
        }
      }
    }
    esm_wave_sync();
  }

  // ---- store: complete rows are final; parts of straddling rows are written un-normalised
  // (forward) or left to the fix-up (backward) ---------------------------------------------------
#pragma unroll
  for (int j = 0; j < kEsmEpl; ++j) {
    const int e = e0 + j;
    if (e >= u.nE) continue;
    const bool partial = e < tail_end || e >= carry_begin;
    if (BWD && partial) continue;
    bool done = false;
    if constexpr (std::is_same<DT, float>::value && HP >= 4) {
      if (p.vec4) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int q = 0; q < HP / 4; ++q) {
          f32x4 t;
          t.x = v[j][4 * q], t.y = v[j][4 * q + 1], t.z = v[j][4 * q + 2], t.w = v[j][4 * q + 3];
          *reinterpret_cast<f32x4*>(pc + off[j] + 4 * q) = t;
        }
        done = true;
      }
    }
    if (!done) {
#pragma unroll
      for (int h = 0; h < HP; ++h)
        if (h < dim) pc[off[j] + h] = from_acc<DT>(v[j][h]);
    }
  }
}

template <typename Idx, typename DT, bool BWD, bool PRECISE>
__global__ __launch_bounds__(64) void edge_softmax_fixup_kernel(const EsmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  const int64_t w = blockIdx.x;
  const int dim = p.dim, hp = 1 << p.log2_hp;
  const EsmUnit u = esm_unit<Idx>(p, w);
  const int64_t f = static_cast<int64_t>(p.indptr[u.i0]) - u.j0;
  const A* cs = static_cast<const A*>(p.carry_stat);
  const A* ts = static_cast<const A*>(p.tail_stat);
  const DT* __restrict__ pa = static_cast<const DT*>(p.a);
  const DT* __restrict__ pb = static_cast<const DT*>(p.b);
  DT* __restrict__ pc = static_cast<DT*>(p.c);
  const int lane = threadIdx.x;
  const int h = lane & (hp - 1);
  const int es = 64 >> p.log2_hp;

  // part: 0 = this unit's carry segment, 1 = this unit's tail segment
  for (int part = 0; part < 2; ++part) {
    int64_t row, sa, s2;  // run of carries [sa, s2) plus the tail held by unit s2
    int t0, t1;
    if (part == 0) {
      row = p.carry_row[w];
      if (row < 0) continue;
      sa = w;
      while (sa > 0 && p.carry_row[sa - 1] == row) --sa;
      s2 = w + 1;
      while (s2 < p.num_units && p.carry_row[s2] == row) ++s2;
      const int64_t cb = static_cast<int64_t>(p.indptr[u.i0 + u.R]) - u.j0;
      t0 = cb < 0 ? 0 : static_cast<int>(cb);
      t1 = u.nE;
    } else {
      if (!(f < 0 && u.R > 0)) continue;
      row = u.i0;
      s2 = w;
      sa = w;
      while (sa > 0 && p.carry_row[sa - 1] == row) --sa;
      t0 = 0;
      t1 = static_cast<int>(static_cast<int64_t>(p.indptr[u.i0 + 1]) - u.j0);
    }
    if (h >= dim) continue;
    if constexpr (BWD) {
      A sum = A(0);
      for (int64_t q = sa; q < s2; ++q) sum += cs[q * 2 * dim + h];
      sum += ts[s2 * 2 * dim + h];
      for (int t = t0 + (lane >> p.log2_hp); t < t1; t += es) {
        const int64_t j = u.j0 + t;
        const int64_t off = (p.eids ? static_cast<int64_t>(p.eids[j]) : j) * dim + h;
        pc[off] = from_acc<DT>(to_acc<DT>(pb[off]) - sum * to_acc<DT>(pa[off]));
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
      const A mine = part == 0 ? cs[w * 2 * dim + h] : ts[w * 2 * dim + h];
      const A scale = esm_expx<A, PRECISE>(mine - M) / S;
      for (int t = t0 + (lane >> p.log2_hp); t < t1; t += es) {
        const int64_t j = u.j0 + t;
        const int64_t off = (p.eids ? static_cast<int64_t>(p.eids[j]) : j) * dim + h;
        pc[off] = from_acc<DT>(to_acc<DT>(pc[off]) * scale);
      }
    }
  }
}

struct EsmGeometry {
  int log2_hp;
  int64_t num_units;
  size_t off_plan, off_carry_row, off_carry_stat, off_tail_stat, total;
};

static size_t esm_align(size_t x) { return (x + 255) / 256 * 256; }

static EsmGeometry esm_geometry(int64_t num_rows, int64_t nnz, int dim, size_t acc_bytes) {
  EsmGeometry g;
  g.log2_hp = 0;
  while ((1 << g.log2_hp) < dim) ++g.log2_hp;
  g.num_units = (num_rows + nnz + kEsmItems - 1) / kEsmItems;
  size_t off = 0;
  g.off_plan = off;
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
                                  hipStream_t s) {
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
  p.carry_row = reinterpret_cast<int64_t*>(wsp + g.off_carry_row);
  p.carry_stat = wsp + g.off_carry_stat;
  p.tail_stat = wsp + g.off_tail_stat;
  const int hp = 1 << g.log2_hp;
  // per wave: the segment tables of the reduce ((max | sum) forward, sum backward), edge ids, row ends
  const size_t per_wave = sizeof(A) * kEsmSegCap * hp * (backward ? 1 : 2) + sizeof(int64_t) * kEsmItems +
                          sizeof(int) * (kEsmItems + 2) + 8;
  p.wave_lds_bytes = static_cast<int>((per_wave + 15) / 16 * 16);
  const int wpb = 4;
  if (!plan_valid) {
    const int64_t n = g.num_units + 1;
    hipLaunchKernelGGL((esm_plan_kernel<Idx>), dim3(static_cast<unsigned>((n + 255) / 256)),
                       dim3(256), 0, s, p.indptr, csr.num_rows, csr.nnz, g.num_units,
                       reinterpret_cast<int64_t*>(wsp + g.off_plan));
  }
  const unsigned blocks = static_cast<unsigned>((g.num_units + wpb - 1) / wpb);
  const size_t lds = static_cast<size_t>(p.wave_lds_bytes) * wpb;
  constexpr bool kPrecise = sizeof(DT) == 8;
#define DGLA_ESM_LAUNCH(HPV)                                                                          \
  do {                                                                                                \
    if (backward) {                                                                                   \
      hipLaunchKernelGGL((edge_softmax_merge_kernel<Idx, DT, true, kPrecise, HPV>), dim3(blocks),      \
                         dim3(64 * wpb), lds, s, p);                                                  \
      hipLaunchKernelGGL((edge_softmax_fixup_kernel<Idx, DT, true, kPrecise>),                        \
                         dim3(static_cast<unsigned>(g.num_units)), dim3(64), 0, s, p);                \
    } else {                                                                                          \
      hipLaunchKernelGGL((edge_softmax_merge_kernel<Idx, DT, false, kPrecise, HPV>), dim3(blocks),     \
                         dim3(64 * wpb), lds, s, p);                                                  \
      hipLaunchKernelGGL((edge_softmax_fixup_kernel<Idx, DT, false, kPrecise>),                       \
                         dim3(static_cast<unsigned>(g.num_units)), dim3(64), 0, s, p);                \
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
                        hipStream_t s) {
  const int d = static_cast<int>(dim);
  const size_t need = edge_softmax_workspace_bytes(csr.num_rows, csr.nnz, dtype, dim);
  const bool merge = need > 0 && ws != nullptr && ws_bytes >= need;
#define DGLA_ES(DT)                                                                          \
  if (merge)                                                                                 \
    return csr.idbits == 32                                                                  \
               ? edge_softmax_merge_run<int32_t, DT>(csr, a, b, c, d, backward, ws, plan_valid, s) \
               : edge_softmax_merge_run<int64_t, DT>(csr, a, b, c, d, backward, ws, plan_valid, s); \
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
