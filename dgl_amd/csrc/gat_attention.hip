// One-pass GAT attention block for gfx950 (MI355X): forward and backward.
//
//   out[v, h, :] = sum_{u -> v} softmax_v( leaky_relu(el[u, h] + er[v, h]) ) * ft[u, h, :]
//
// The reference composes it from four operators (python/dgl/nn/pytorch/conv/gatconv.py:330-347: u_add_v SDDMM,
// leaky_relu, edge_softmax — itself five launches on GPU, python/dgl/backend/pytorch/sparse.py:709-713, "TODO" at
// src/array/kernel.cc:313,331 — and u_mul_e_sum SpMM), writing and re-reading three (E, H) tensors; behind an
// edge-id map every one of those accesses is a scattered 32-byte line.  Here no (E, H) tensor exists:
//
//   forward   per destination row, lane groups gather el[src] and the ft[src] row ONCE per edge, keep an online
//             softmax state (running max m, rescaled sum z, rescaled accumulator) in registers and write
//             out = acc / z plus the row's (m, z) — 2 floats per (node, head) — for the backward.
//   backward  two gather passes that RECOMPUTE the attention weight a = exp(s - m) / z from (m, z):
//             B1 over the in-edge CSR   d_er[v] = sum_e d_pre_e           (gathers ft[u], el[u]; also writes the per-node
//                                       record aux[v, h] = (er, m, 1/z, t = <dout_v, out_v>_h) B2 reads per edge)
//             B2 over the out-edge CSR  d_ft[u] = sum_e a_e dout[v],  d_el[u] = sum_e d_pre_e   (gathers dout[v], aux[v])
//             with d_pre_e = a_e (<dout_v, ft_u>_h - t_v) * leaky_relu'(el_u + er_v).
//
// Work decomposition (all three kernels): the edge array is cut into chunks of kGatChunk edges, one WAVEFRONT per
// chunk, so a 17 k-edge hub row costs its chunks no more than any other 512 edges.  Rows that lie inside a chunk are
// finished by it; the (at most two) rows that cross its ends leave a partial state in scratch, and a fix-up kernel
// merges the partials of each crossing row in chunk order (deterministic: no atomics anywhere).  Inside a wave, a
// group of LPR = next_pow2(H * D / 4) lanes covers one ft row with 16-byte accesses (lane l holds columns 4l .. 4l+3,
// all of one head since D is a power of two >= 4), the 64 / LPR groups take consecutive edges of the row, each with
// U edges in flight, and merge their states by xor-shuffles at the end of the row.
//
// HBM-bound: per edge one ft (or dout) row of H * D * 4 bytes + 4 * H bytes of el (or 16 * H of aux) + one index;
// algorithmic bytes of the forward  E * (H*D*4 + H*4 + i) + N * (H*D*4 + 3*H*4) + (N + 1) * i.
#include "../../include/dgl_amd.h"

#include "common.h"

namespace dgla {
namespace {

constexpr int kGatChunk = 512;  // edges per wavefront

int gfail(const std::string& m) {
  last_error() = m;
  return -1;
}

// exp(x) for x <= 0 on v_exp_f32 with the product x * log2(e) carried as hi + lo (same routine as the fused edge
// softmax, csrc/edge_softmax.hip: |relative error| < 3e-7 over [-88, 0]; exp(-inf) = 0)
__device__ __forceinline__ float gat_exp(float x) {
  asm("v_max_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(-200.f));
  const float hi = x * 1.44269504088896341f;
  const float lo = __builtin_fmaf(x, 1.44269504088896341f, -hi) + x * 1.92596299112661746e-8f;
  return __builtin_amdgcn_exp2f(hi) * __builtin_fmaf(lo, 0.693147180559945309f, 1.0f);
}

struct alignas(16) F4 {
  float x, y, z, w;
};

template <typename Idx>
struct GatArgs {
  const Idx* indptr;
  const Idx* indices;
  int64_t num_rows, nnz, nchunks;
  const int64_t* chunk_row;  // [nchunks] row that holds the chunk's first edge
  int64_t* prow;             // [2 * nchunks] row of the head / tail partial of each chunk, -1: none
  float* pval;               // [2 * nchunks][ns] partial states
  int ns;
  int H, HD, lph_log2;  // heads, H * D, log2(lanes per head) = log2(D / 4)
  float slope;
  // forward: ft (rows gathered), el (gathered), er (per row), out, mz
  // B1:      ft, el gathered; er, mz, dout, out per row; d_er, aux written
  // B2:      dout, aux gathered; el, ft per row; d_ft, d_el written
  const float* ft;
  const float* el;
  const float* er;
  const float* dout;
  float* out;
  float* mz;
  float* aux;
  float* d_ft;
  float* d_el;
  float* d_er;
};

template <typename Idx>
__global__ __launch_bounds__(256) void gat_chunk_rows_kernel(const Idx* __restrict__ indptr, int64_t num_rows,
                                                            int64_t nchunks, int64_t* __restrict__ chunk_row,
                                                            int64_t* __restrict__ prow) {
  const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int64_t p0 = c * kGatChunk;
  int64_t lo = 0, hi = num_rows + 1;  // first k with indptr[k] > p0 (exists: indptr[num_rows] = nnz > p0)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (static_cast<int64_t>(indptr[mid]) > p0)
      hi = mid;
    else
      lo = mid + 1;
  }
  chunk_row[c] = lo - 1;
  prow[2 * c] = prow[2 * c + 1] = -1;
}

// rows without edges: zero their output rows (two outputs of widths wa / wb; either may be null)
template <typename Idx>
__global__ __launch_bounds__(256) void gat_zero_rows_kernel(const Idx* __restrict__ indptr, int64_t num_rows,
                                                           float* __restrict__ a, int wa, float* __restrict__ b, int wb,
                                                           float fill_b0, float fill_b1) {
  const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= num_rows || indptr[r] != indptr[r + 1]) return;
  if (a)
    for (int i = 0; i < wa; ++i) a[r * wa + i] = 0.f;
  if (b)
    for (int i = 0; i < wb; ++i) b[r * wb + i] = (i & 1) ? fill_b1 : fill_b0;
}

template <int LOG2_LPR>
struct Geo {
  static constexpr int LPR = 1 << LOG2_LPR;
  static constexpr int G = 64 / LPR;
  static constexpr int U = LOG2_LPR >= 4 ? 4 : (LOG2_LPR == 3 ? 2 : 1);
};

__device__ __forceinline__ float head_sum(float v, int lph_log2) {
  for (int m = 1; m < (1 << lph_log2); m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float dot4(const F4& a, const F4& b) {
  return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
struct FwdState {
  float m, z;
  F4 acc;
};

__device__ __forceinline__ void fwd_merge(FwdState& s, float m_o, float z_o, const F4& a_o) {
  const float mn = s.m > m_o ? s.m : m_o;
  const float a = s.m == mn ? 1.f : gat_exp(s.m - mn);
  const float b = m_o == mn ? 1.f : gat_exp(m_o - mn);
  s.z = s.z * a + z_o * b;
  s.acc.x = s.acc.x * a + a_o.x * b;
  s.acc.y = s.acc.y * a + a_o.y * b;
  s.acc.z = s.acc.z * a + a_o.z * b;
  s.acc.w = s.acc.w * a + a_o.w * b;
  s.m = mn;
}

template <typename Idx, int LOG2_LPR>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const GatArgs<Idx> p) {
  using GE = Geo<LOG2_LPR>;
  constexpr int LPR = GE::LPR, G = GE::G, U = GE::U;
  const int lane = threadIdx.x & 63;
  const int64_t c = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (c >= p.nchunks) return;
  const int l = lane & (LPR - 1), g = lane >> LOG2_LPR;
  const int col = l * 4;
  const bool active = col < p.HD;
  const int h = active ? (l >> p.lph_log2) : 0;
  const int H = p.H, HD = p.HD;
  const float ninf = -__builtin_huge_valf();
  const int64_t p0 = c * kGatChunk;
  const int64_t p1 = p0 + kGatChunk < p.nnz ? p0 + kGatChunk : p.nnz;
  int64_t row = p.chunk_row[c];
  int64_t rs = static_cast<int64_t>(p.indptr[row]), re = static_cast<int64_t>(p.indptr[row + 1]);
  int64_t pos = p0;
  while (pos < p1) {
    while (re <= pos) {
      ++row;
      rs = re;
      re = static_cast<int64_t>(p.indptr[row + 1]);
    }
    const int64_t b = re < p1 ? re : p1;
    const float er_h = active ? p.er[row * H + h] : 0.f;
    FwdState st{ninf, 0.f, F4{0.f, 0.f, 0.f, 0.f}};
    for (int64_t base = pos; base < b; base += G * U) {
      int64_t src[U];
      bool ok[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t j = base + k * G + g;
        ok[k] = j < b && active;
        src[k] = ok[k] ? static_cast<int64_t>(p.indices[j]) : 0;
      }
      float sv[U];
      F4 f[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        sv[k] = 0.f;
        f[k] = F4{0.f, 0.f, 0.f, 0.f};
        if (ok[k]) {
          sv[k] = p.el[src[k] * H + h];
          f[k] = *reinterpret_cast<const F4*>(p.ft + src[k] * HD + col);
        }
      }
      float bm = ninf;
#pragma unroll
      for (int k = 0; k < U; ++k) {
        float s = sv[k] + er_h;
        s = s > 0.f ? s : s * p.slope;
        sv[k] = ok[k] ? s : ninf;
        bm = bm > sv[k] ? bm : sv[k];
      }
      const float mn = st.m > bm ? st.m : bm;
      const float sc = st.m == mn ? 1.f : gat_exp(st.m - mn);
      st.z *= sc;
      st.acc.x *= sc;
      st.acc.y *= sc;
      st.acc.z *= sc;
      st.acc.w *= sc;
      st.m = mn;
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float pk = ok[k] ? gat_exp(sv[k] - mn) : 0.f;
        st.z += pk;
        st.acc.x = __builtin_fmaf(pk, f[k].x, st.acc.x);
        st.acc.y = __builtin_fmaf(pk, f[k].y, st.acc.y);
        st.acc.z = __builtin_fmaf(pk, f[k].z, st.acc.z);
        st.acc.w = __builtin_fmaf(pk, f[k].w, st.acc.w);
      }
    }
    // merge the G lane groups (every group ends with the merged state)
#pragma unroll
    for (int mk = LPR; mk < 64; mk <<= 1) {
      const float m_o = __shfl_xor(st.m, mk, 64), z_o = __shfl_xor(st.z, mk, 64);
      F4 a_o;
      a_o.x = __shfl_xor(st.acc.x, mk, 64);
      a_o.y = __shfl_xor(st.acc.y, mk, 64);
      a_o.z = __shfl_xor(st.acc.z, mk, 64);
      a_o.w = __shfl_xor(st.acc.w, mk, 64);
      fwd_merge(st, m_o, z_o, a_o);
    }
    const bool head_partial = pos > rs, tail_partial = re > p1;
    if (head_partial || tail_partial) {
      const int64_t slot = 2 * c + (head_partial ? 0 : 1);
      float* pv = p.pval + slot * p.ns;
      if (g == 0 && active) {
        *reinterpret_cast<F4*>(pv + col) = st.acc;
        if ((l & ((1 << p.lph_log2) - 1)) == 0) {
          pv[HD + 2 * h] = st.m;
          pv[HD + 2 * h + 1] = st.z;
        }
      }
      if (lane == 0) p.prow[slot] = row;
    } else if (g == 0 && active) {
      const float rz = 1.f / st.z;
      *reinterpret_cast<F4*>(p.out + row * HD + col) = F4{st.acc.x * rz, st.acc.y * rz, st.acc.z * rz, st.acc.w * rz};
      if ((l & ((1 << p.lph_log2) - 1)) == 0) {
        p.mz[(row * H + h) * 2] = st.m;
        p.mz[(row * H + h) * 2 + 1] = st.z;
      }
    }
    pos = b;
  }
}

// one wavefront per chunk: the row that STARTS in chunk c and runs past its end is finished here from the tail partial
// of c and the head partials of the chunks after it, in chunk order
template <typename Idx>
__global__ __launch_bounds__(64) void gat_fwd_fixup_kernel(const GatArgs<Idx> p) {
  const int64_t c = blockIdx.x;
  const int64_t r = p.prow[2 * c + 1];
  if (r < 0) return;
  const int l = threadIdx.x;
  const int col = l * 4;
  const bool active = col < p.HD;
  const int h = active ? (l >> p.lph_log2) : 0;
  const float* pv = p.pval + (2 * c + 1) * p.ns;
  FwdState st{0.f, 0.f, F4{0.f, 0.f, 0.f, 0.f}};
  if (active) {
    st.acc = *reinterpret_cast<const F4*>(pv + col);
    st.m = pv[p.HD + 2 * h];
    st.z = pv[p.HD + 2 * h + 1];
  }
  for (int64_t cc = c + 1; cc < p.nchunks && p.prow[2 * cc] == r; ++cc) {
    const float* qv = p.pval + (2 * cc) * p.ns;
    if (active) fwd_merge(st, qv[p.HD + 2 * h], qv[p.HD + 2 * h + 1], *reinterpret_cast<const F4*>(qv + col));
  }
  if (active) {
    const float rz = 1.f / st.z;
    *reinterpret_cast<F4*>(p.out + r * p.HD + col) = F4{st.acc.x * rz, st.acc.y * rz, st.acc.z * rz, st.acc.w * rz};
    if ((l & ((1 << p.lph_log2) - 1)) == 0) {
      p.mz[(r * p.H + h) * 2] = st.m;
      p.mz[(r * p.H + h) * 2 + 1] = st.z;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, pass 1 (rows = destination nodes): d_er and the per-node record aux
// ---------------------------------------------------------------------------------------------------------------
template <typename Idx, int LOG2_LPR>
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(const GatArgs<Idx> p) {
  using GE = Geo<LOG2_LPR>;
  constexpr int LPR = GE::LPR, G = GE::G, U = GE::U;
  const int lane = threadIdx.x & 63;
  const int64_t c = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (c >= p.nchunks) return;
  const int l = lane & (LPR - 1), g = lane >> LOG2_LPR;
  const int col = l * 4;
  const bool active = col < p.HD;
  const int h = active ? (l >> p.lph_log2) : 0;
  const bool head_lane = (l & ((1 << p.lph_log2) - 1)) == 0;
  const int H = p.H, HD = p.HD;
  const int64_t p0 = c * kGatChunk;
  const int64_t p1 = p0 + kGatChunk < p.nnz ? p0 + kGatChunk : p.nnz;
  int64_t row = p.chunk_row[c];
  int64_t rs = static_cast<int64_t>(p.indptr[row]), re = static_cast<int64_t>(p.indptr[row + 1]);
  int64_t pos = p0;
  while (pos < p1) {
    while (re <= pos) {
      ++row;
      rs = re;
      re = static_cast<int64_t>(p.indptr[row + 1]);
    }
    const int64_t b = re < p1 ? re : p1;
    float er_h = 0.f, m_h = 0.f, rz = 0.f;
    F4 dO{0.f, 0.f, 0.f, 0.f}, O{0.f, 0.f, 0.f, 0.f};
    if (active) {
      er_h = p.er[row * H + h];
      m_h = p.mz[(row * H + h) * 2];
      rz = 1.f / p.mz[(row * H + h) * 2 + 1];
      dO = *reinterpret_cast<const F4*>(p.dout + row * HD + col);
      O = *reinterpret_cast<const F4*>(p.out + row * HD + col);
    }
    const float t_h = head_sum(dot4(dO, O), p.lph_log2);
    const bool head_partial = pos > rs, tail_partial = re > p1;
    if (!head_partial && g == 0 && active && head_lane)  // the chunk in which the row starts publishes its record
      *reinterpret_cast<F4*>(p.aux + (row * H + h) * 4) = F4{er_h, m_h, rz, t_h};
    float acc = 0.f;
    for (int64_t base = pos; base < b; base += G * U) {
      int64_t src[U];
      bool ok[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t j = base + k * G + g;
        ok[k] = j < b && active;
        src[k] = ok[k] ? static_cast<int64_t>(p.indices[j]) : 0;
      }
      float sv[U];
      F4 f[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        sv[k] = 0.f;
        f[k] = F4{0.f, 0.f, 0.f, 0.f};
        if (ok[k]) {
          sv[k] = p.el[src[k] * H + h];
          f[k] = *reinterpret_cast<const F4*>(p.ft + src[k] * HD + col);
        }
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float dA = head_sum(dot4(dO, f[k]), p.lph_log2);
        const float pre = sv[k] + er_h;
        const float s = pre > 0.f ? pre : pre * p.slope;
        const float a = gat_exp(s - m_h) * rz;
        const float dpre = a * (dA - t_h) * (pre > 0.f ? 1.f : p.slope);
        acc += ok[k] ? dpre : 0.f;
      }
    }
#pragma unroll
    for (int mk = LPR; mk < 64; mk <<= 1) acc += __shfl_xor(acc, mk, 64);
    if (head_partial || tail_partial) {
      const int64_t slot = 2 * c + (head_partial ? 0 : 1);
      if (g == 0 && active && head_lane) p.pval[slot * p.ns + h] = acc;
      if (lane == 0) p.prow[slot] = row;
    } else if (g == 0 && active && head_lane) {
      p.d_er[row * H + h] = acc;
    }
    pos = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, pass 2 (rows = source nodes, out-edge CSR): d_ft and d_el
// ---------------------------------------------------------------------------------------------------------------
template <typename Idx, int LOG2_LPR>
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(const GatArgs<Idx> p) {
  using GE = Geo<LOG2_LPR>;
  constexpr int LPR = GE::LPR, G = GE::G, U = GE::U;
  const int lane = threadIdx.x & 63;
  const int64_t c = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (c >= p.nchunks) return;
  const int l = lane & (LPR - 1), g = lane >> LOG2_LPR;
  const int col = l * 4;
  const bool active = col < p.HD;
  const int h = active ? (l >> p.lph_log2) : 0;
  const bool head_lane = (l & ((1 << p.lph_log2) - 1)) == 0;
  const int H = p.H, HD = p.HD;
  const int64_t p0 = c * kGatChunk;
  const int64_t p1 = p0 + kGatChunk < p.nnz ? p0 + kGatChunk : p.nnz;
  int64_t row = p.chunk_row[c];
  int64_t rs = static_cast<int64_t>(p.indptr[row]), re = static_cast<int64_t>(p.indptr[row + 1]);
  int64_t pos = p0;
  while (pos < p1) {
    while (re <= pos) {
      ++row;
      rs = re;
      re = static_cast<int64_t>(p.indptr[row + 1]);
    }
    const int64_t b = re < p1 ? re : p1;
    float el_h = 0.f;
    F4 f{0.f, 0.f, 0.f, 0.f};
    if (active) {
      el_h = p.el[row * H + h];
      f = *reinterpret_cast<const F4*>(p.ft + row * HD + col);
    }
    float acc_el = 0.f;
    F4 acc{0.f, 0.f, 0.f, 0.f};
    for (int64_t base = pos; base < b; base += G * U) {
      int64_t dst[U];
      bool ok[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t j = base + k * G + g;
        ok[k] = j < b && active;
        dst[k] = ok[k] ? static_cast<int64_t>(p.indices[j]) : 0;
      }
      F4 ax[U], dO[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        ax[k] = F4{0.f, 0.f, 1.f, 0.f};
        dO[k] = F4{0.f, 0.f, 0.f, 0.f};
        if (ok[k]) {
          ax[k] = *reinterpret_cast<const F4*>(p.aux + (dst[k] * H + h) * 4);
          dO[k] = *reinterpret_cast<const F4*>(p.dout + dst[k] * HD + col);
        }
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float dA = head_sum(dot4(dO[k], f), p.lph_log2);
        const float pre = el_h + ax[k].x;
        const float s = pre > 0.f ? pre : pre * p.slope;
        const float a = ok[k] ? gat_exp(s - ax[k].y) * ax[k].z : 0.f;
        acc_el += a * (dA - ax[k].w) * (pre > 0.f ? 1.f : p.slope);
        acc.x = __builtin_fmaf(a, dO[k].x, acc.x);
        acc.y = __builtin_fmaf(a, dO[k].y, acc.y);
        acc.z = __builtin_fmaf(a, dO[k].z, acc.z);
        acc.w = __builtin_fmaf(a, dO[k].w, acc.w);
      }
    }
#pragma unroll
    for (int mk = LPR; mk < 64; mk <<= 1) {
      acc_el += __shfl_xor(acc_el, mk, 64);
      acc.x += __shfl_xor(acc.x, mk, 64);
      acc.y += __shfl_xor(acc.y, mk, 64);
      acc.z += __shfl_xor(acc.z, mk, 64);
      acc.w += __shfl_xor(acc.w, mk, 64);
    }
    const bool head_partial = pos > rs, tail_partial = re > p1;
    if (head_partial || tail_partial) {
      const int64_t slot = 2 * c + (head_partial ? 0 : 1);
      float* pv = p.pval + slot * p.ns;
      if (g == 0 && active) {
        *reinterpret_cast<F4*>(pv + col) = acc;
        if (head_lane) pv[HD + h] = acc_el;
      }
      if (lane == 0) p.prow[slot] = row;
    } else if (g == 0 && active) {
      *reinterpret_cast<F4*>(p.d_ft + row * HD + col) = acc;
      if (head_lane) p.d_el[row * H + h] = acc_el;
    }
    pos = b;
  }
}

// fix-up of the two backward passes: plain sums of the partial states.  Elements [0, wa) of a state go to a[row, :],
// elements [wa, wa + wb) to b[row, :] (wa = 0: only b).
__global__ __launch_bounds__(64) void gat_sum_fixup_kernel(const int64_t* __restrict__ prow, const float* __restrict__ pval,
                                                          int ns, int64_t nchunks, float* __restrict__ a, int wa,
                                                          float* __restrict__ b, int wb) {
  const int64_t c = blockIdx.x;
  const int64_t r = prow[2 * c + 1];
  if (r < 0) return;
  for (int i = threadIdx.x; i < wa + wb; i += 64) {
    float v = pval[(2 * c + 1) * static_cast<int64_t>(ns) + i];
    for (int64_t cc = c + 1; cc < nchunks && prow[2 * cc] == r; ++cc) v += pval[(2 * cc) * static_cast<int64_t>(ns) + i];
    if (i < wa)
      a[r * wa + i] = v;
    else
      b[r * wb + (i - wa)] = v;
  }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

int pad4(int x) { return (x + 3) & ~3; }  // partial states start on 16-byte boundaries

int64_t num_chunks(int64_t nnz) { return (nnz + kGatChunk - 1) / kGatChunk; }

struct Shape {
  int H, D, HD, lph_log2, log2_lpr;
};

// heads H, per-head width D: D a power of two >= 4 and H * D <= 256 (one 16-byte slab per lane, <= 64 lanes per row)
int shape_of(const dgla_tensor* ft, const dgla_tensor* el, const dgla_tensor* er, Shape* s) {
  if (!ft || !el || !er || !ft->data || !el->data || !er->data) return gfail("gat_attention: ft / el / er are required");
  if (ft->ndim != 3 || el->ndim != 3 || er->ndim != 3 || el->shape[2] != 1 || er->shape[2] != 1)
    return gfail("gat_attention: ft must be (N_src, H, D), el (N_src, H, 1), er (N_dst, H, 1)");
  const int64_t H = ft->shape[1], D = ft->shape[2];
  if (el->shape[1] != H || er->shape[1] != H || el->shape[0] != ft->shape[0])
    return gfail("gat_attention: head counts / node counts of ft, el, er differ");
  if (D < 4 || (D & (D - 1)) != 0 || H < 1 || H * D > 256)
    return gfail("gat_attention: needs D a power of two >= 4 and H * D <= 256 (use the composed operators otherwise)");
  s->H = static_cast<int>(H);
  s->D = static_cast<int>(D);
  s->HD = static_cast<int>(H * D);
  s->lph_log2 = 0;
  while ((4 << s->lph_log2) < D) ++s->lph_log2;
  int lanes = s->HD / 4, lg = 2;
  while ((1 << lg) < lanes) ++lg;
  s->log2_lpr = lg;
  return 0;
}

template <typename Idx>
struct Scratch {
  int64_t* chunk_row;
  int64_t* prow;
  float* pval;
  float* aux;
};

size_t scratch_bytes(int64_t nnz, int64_t num_dst, int H, int HD) {
  const int64_t nc = num_chunks(nnz);
  return align256(8 * nc) + align256(16 * nc) + align256(sizeof(float) * 2 * nc * (HD + pad4(2 * H))) +
         align256(sizeof(float) * 4 * num_dst * H);
}

template <typename Idx>
Scratch<Idx> carve(char* ws, int64_t nnz, int H, int HD) {
  const int64_t nc = num_chunks(nnz);
  Scratch<Idx> s;
  s.chunk_row = reinterpret_cast<int64_t*>(ws);
  ws += align256(8 * nc);
  s.prow = reinterpret_cast<int64_t*>(ws);
  ws += align256(16 * nc);
  s.pval = reinterpret_cast<float*>(ws);
  ws += align256(sizeof(float) * 2 * nc * (HD + pad4(2 * H)));
  s.aux = reinterpret_cast<float*>(ws);
  return s;
}

unsigned grid1(int64_t n, int per = 256) { return static_cast<unsigned>((n + per - 1) / per); }

#define DGLA_GAT_DISPATCH(KERNEL, LG, ...)                                                    \
  switch (LG) {                                                                               \
    case 2: hipLaunchKernelGGL((KERNEL<Idx, 2>), __VA_ARGS__); break;                          \
    case 3: hipLaunchKernelGGL((KERNEL<Idx, 3>), __VA_ARGS__); break;                          \
    case 4: hipLaunchKernelGGL((KERNEL<Idx, 4>), __VA_ARGS__); break;                          \
    case 5: hipLaunchKernelGGL((KERNEL<Idx, 5>), __VA_ARGS__); break;                          \
    default: hipLaunchKernelGGL((KERNEL<Idx, 6>), __VA_ARGS__); break;                         \
  }

template <typename Idx>
int forward_typed(const dgla_csr* csc, const Shape& sh, const float* ft, const float* el, const float* er, float slope,
                  float* out, float* mz, char* ws, hipStream_t s) {
  const int64_t nnz = csc->nnz, n = csc->num_rows;
  const Idx* indptr = static_cast<const Idx*>(csc->indptr);
  hipLaunchKernelGGL(gat_zero_rows_kernel<Idx>, dim3(grid1(n)), dim3(256), 0, s, indptr, n, out, sh.HD, mz, 2 * sh.H, 0.f, 1.f);
  if (nnz > 0) {
    const Scratch<Idx> sc = carve<Idx>(ws, nnz, sh.H, sh.HD);
    GatArgs<Idx> a{};
    a.indptr = indptr;
    a.indices = static_cast<const Idx*>(csc->indices);
    a.num_rows = n;
    a.nnz = nnz;
    a.nchunks = num_chunks(nnz);
    a.chunk_row = sc.chunk_row;
    a.prow = sc.prow;
    a.pval = sc.pval;
    a.ns = sh.HD + pad4(2 * sh.H);
    a.H = sh.H;
    a.HD = sh.HD;
    a.lph_log2 = sh.lph_log2;
    a.slope = slope;
    a.ft = ft;
    a.el = el;
    a.er = er;
    a.out = out;
    a.mz = mz;
    hipLaunchKernelGGL(gat_chunk_rows_kernel<Idx>, dim3(grid1(a.nchunks)), dim3(256), 0, s, indptr, n, a.nchunks,
                       sc.chunk_row, sc.prow);
    DGLA_GAT_DISPATCH(gat_fwd_kernel, sh.log2_lpr, dim3(grid1(a.nchunks, 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gat_fwd_fixup_kernel<Idx>, dim3(static_cast<unsigned>(a.nchunks)), dim3(64), 0, s, a);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx>
int backward_typed(const dgla_csr* csc, const dgla_csr* csr, const Shape& sh, const float* ft, const float* el,
                   const float* er, const float* out, const float* mz, const float* dout, float slope, float* d_ft,
                   float* d_el, float* d_er, char* ws, hipStream_t s) {
  const int64_t nnz = csc->nnz, n_dst = csc->num_rows, n_src = csr->num_rows;
  const Idx* ip_in = static_cast<const Idx*>(csc->indptr);
  const Idx* ip_out = static_cast<const Idx*>(csr->indptr);
  hipLaunchKernelGGL(gat_zero_rows_kernel<Idx>, dim3(grid1(n_dst)), dim3(256), 0, s, ip_in, n_dst, d_er, sh.H,
                     static_cast<float*>(nullptr), 0, 0.f, 0.f);
  hipLaunchKernelGGL(gat_zero_rows_kernel<Idx>, dim3(grid1(n_src)), dim3(256), 0, s, ip_out, n_src, d_ft, sh.HD, d_el, sh.H,
                     0.f, 0.f);
  if (nnz > 0) {
    const Scratch<Idx> sc = carve<Idx>(ws, nnz, sh.H, sh.HD);
    GatArgs<Idx> a{};
    a.nnz = nnz;
    a.nchunks = num_chunks(nnz);
    a.chunk_row = sc.chunk_row;
    a.prow = sc.prow;
    a.pval = sc.pval;
    a.H = sh.H;
    a.HD = sh.HD;
    a.lph_log2 = sh.lph_log2;
    a.slope = slope;
    a.ft = ft;
    a.el = el;
    a.er = er;
    a.dout = dout;
    a.out = const_cast<float*>(out);
    a.mz = const_cast<float*>(mz);
    a.aux = sc.aux;
    a.d_ft = d_ft;
    a.d_el = d_el;
    a.d_er = d_er;
    // pass 1: rows = destination nodes
    a.indptr = ip_in;
    a.indices = static_cast<const Idx*>(csc->indices);
    a.num_rows = n_dst;
    a.ns = pad4(sh.H);
    hipLaunchKernelGGL(gat_chunk_rows_kernel<Idx>, dim3(grid1(a.nchunks)), dim3(256), 0, s, ip_in, n_dst, a.nchunks,
                       sc.chunk_row, sc.prow);
    DGLA_GAT_DISPATCH(gat_bwd_dst_kernel, sh.log2_lpr, dim3(grid1(a.nchunks, 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gat_sum_fixup_kernel, dim3(static_cast<unsigned>(a.nchunks)), dim3(64), 0, s, sc.prow, sc.pval, a.ns,
                       a.nchunks, static_cast<float*>(nullptr), 0, d_er, sh.H);
    // pass 2: rows = source nodes
    a.indptr = ip_out;
    a.indices = static_cast<const Idx*>(csr->indices);
    a.num_rows = n_src;
    a.ns = sh.HD + pad4(sh.H);
    hipLaunchKernelGGL(gat_chunk_rows_kernel<Idx>, dim3(grid1(a.nchunks)), dim3(256), 0, s, ip_out, n_src, a.nchunks,
                       sc.chunk_row, sc.prow);
    DGLA_GAT_DISPATCH(gat_bwd_src_kernel, sh.log2_lpr, dim3(grid1(a.nchunks, 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gat_sum_fixup_kernel, dim3(static_cast<unsigned>(a.nchunks)), dim3(64), 0, s, sc.prow, sc.pval, a.ns,
                       a.nchunks, d_ft, sh.HD, d_el, sh.H);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace
}  // namespace dgla

using namespace dgla;

extern "C" {

size_t dgla_gat_attention_workspace_bytes(const dgla_csr* csc, int64_t heads, int64_t dim) {
  if (!csc || csc->nnz <= 0 || heads <= 0 || dim <= 0) return 0;
  return scratch_bytes(csc->nnz, csc->num_rows, static_cast<int>(heads), static_cast<int>(heads * dim));
}

int dgla_gat_attention_forward(const dgla_csr* csc, dgla_dtype dtype, const dgla_tensor* ft, const dgla_tensor* el,
                               const dgla_tensor* er, float negative_slope, const dgla_tensor* out, void* mz,
                               void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!csc || !out || !out->data || !mz) return gfail("gat_attention_forward: csc / out / mz are required");
  if (dtype != DGLA_F32) return gfail("gat_attention: fp32 operands only (use the composed operators otherwise)");
  Shape sh;
  if (shape_of(ft, el, er, &sh)) return -1;
  if (csc->idtype_bits != 32 && csc->idtype_bits != 64) return gfail("idtype must be int32 or int64");
  if (ft->shape[0] != csc->num_cols || er->shape[0] != csc->num_rows || out->ndim != 3 ||
      out->shape[0] != csc->num_rows || out->shape[1] != sh.H || out->shape[2] != sh.D)
    return gfail("gat_attention_forward: tensor shapes do not match the graph");
  if (csc->nnz > 0 && (!workspace || workspace_bytes < dgla_gat_attention_workspace_bytes(csc, sh.H, sh.D)))
    return gfail("gat_attention_forward: workspace too small (dgla_gat_attention_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out->data);
  const float *f = static_cast<const float*>(ft->data), *l = static_cast<const float*>(el->data),
              *r = static_cast<const float*>(er->data);
  return csc->idtype_bits == 32
             ? forward_typed<int32_t>(csc, sh, f, l, r, negative_slope, static_cast<float*>(out->data),
                                      static_cast<float*>(mz), static_cast<char*>(workspace), s)
             : forward_typed<int64_t>(csc, sh, f, l, r, negative_slope, static_cast<float*>(out->data),
                                      static_cast<float*>(mz), static_cast<char*>(workspace), s);
}

int dgla_gat_attention_backward(const dgla_csr* csc, const dgla_csr* csr, dgla_dtype dtype, const dgla_tensor* ft,
                                const dgla_tensor* el, const dgla_tensor* er, const dgla_tensor* out, const void* mz,
                                const dgla_tensor* dout, float negative_slope, const dgla_tensor* d_ft,
                                const dgla_tensor* d_el, const dgla_tensor* d_er, void* workspace, size_t workspace_bytes,
                                void* hip_stream) {
  if (!csc || !csr || !out || !dout || !d_ft || !d_el || !d_er || !mz || !out->data || !dout->data || !d_ft->data ||
      !d_el->data || !d_er->data)
    return gfail("gat_attention_backward: every tensor is required");
  if (dtype != DGLA_F32) return gfail("gat_attention: fp32 operands only (use the composed operators otherwise)");
  Shape sh;
  if (shape_of(ft, el, er, &sh)) return -1;
  if (csc->idtype_bits != csr->idtype_bits || (csc->idtype_bits != 32 && csc->idtype_bits != 64))
    return gfail("gat_attention_backward: the two CSRs must share one id type (int32 or int64)");
  if (csc->nnz != csr->nnz || csc->num_rows != csr->num_cols || csc->num_cols != csr->num_rows)
    return gfail("gat_attention_backward: csr is not the out-edge CSR of csc's graph");
  if (ft->shape[0] != csc->num_cols || er->shape[0] != csc->num_rows)
    return gfail("gat_attention_backward: tensor shapes do not match the graph");
  if (csc->nnz > 0 && (!workspace || workspace_bytes < dgla_gat_attention_workspace_bytes(csc, sh.H, sh.D)))
    return gfail("gat_attention_backward: workspace too small (dgla_gat_attention_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, d_ft->data);
#define DGLA_GAT_BWD(IDX)                                                                                         \
  backward_typed<IDX>(csc, csr, sh, static_cast<const float*>(ft->data), static_cast<const float*>(el->data),     \
                      static_cast<const float*>(er->data), static_cast<const float*>(out->data),                  \
                      static_cast<const float*>(mz), static_cast<const float*>(dout->data), negative_slope,       \
                      static_cast<float*>(d_ft->data), static_cast<float*>(d_el->data),                           \
                      static_cast<float*>(d_er->data), static_cast<char*>(workspace), s)
  return csc->idtype_bits == 32 ? DGLA_GAT_BWD(int32_t) : DGLA_GAT_BWD(int64_t);
#undef DGLA_GAT_BWD
}

}  // extern "C"
