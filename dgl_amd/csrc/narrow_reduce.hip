// g-SpMM with NARROW features (1 ... 8 fp32 output columns) over the in-edge CSR, for gfx950: copy_e, copy_u and the binary
// operators u (+ - * /) e with an edge operand of the same width or one value per `rhs_group` columns.
//
// Same arithmetic as SpMMCsrKernel (src/array/cuda/spmm.cuh:440-520 — one message per edge, reduced into its destination
// row; winners' column / edge ids for max / min).  The merge kernel (spmm_csr.hip.h) gives a
// feature COLUMN to a lane: below 16 columns most of a wave idles and the kernel is bound by issue — 30 G edges/s on the
// ogbn-products-shaped graph whatever the width, 1.84 ms for 250 MB of scalars (profiles/r5/narrow_feature_reductions.jsonl).
// Here a lane owns four consecutive EDGES with all their columns, a wave 256 consecutive CSR positions (a "unit"):
//   1. the row holding the unit's first position by one wave-uniform binary search; the row starts that fall inside the
//      unit are marked in LDS by the lanes reading indptr[r_first + 1 + lane ...] (start bits + the id of the row that
//      starts there; empty rows in between cost nothing: the largest id wins);
//   2. every lane reduces its four edges into runs; a run that begins and ends inside a lane is stored at once (operand
//      rows are loaded first, 16 bytes at a time; edge rows of 4 / 8 columns in position order as whole 1 KB wavefront
//      loads transposed through LDS);
//   3. a segmented scan over the lanes — DPP moves — (the open run at a lane's end is carried to the lane that closes it);
//   4. a row cut by a unit boundary leaves its pieces in the workspace — (head: the piece at the start of a unit, tail:
//      the piece at its end) — and a second, tiny kernel adds the pieces of each such row in position order: a thread per
//      row for up to four pieces, a wavefront per row (64 pieces per step) for hubs.
// Deterministic: the order of the additions depends on the CSR alone.  Rows without an edge keep the reducer's identity
// (0, -inf, +inf) and argument 0, written by a fill kernel in front, as the merge kernel leaves them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdlib>

#include "common.h"

namespace dgla {
namespace {

// Edges per lane (a wave's unit = 64 of them).  Four for every width: 16 (F = 1) and 8 (F <= 4) were built to spread the
// per-unit scans over more edges and measured SLOWER — copy_e sum on C2 0.17 -> 0.21 ms (F = 1), 0.32 -> 0.49 (F = 4): with
// rows of 25 edges nearly every lane then closes a row itself, and 32 KB of row ids per workgroup cost occupancy.
constexpr int nr_lane_edges(int f) { return (void)f, 4; }
constexpr int kNrWaves = 4;                  // waves (units) per workgroup
constexpr uint8_t kNrHead = 1, kNrHeadClosed = 2, kNrTail = 4;
constexpr int kNrInlineSteps = 4;            // pieces of a cut row one thread of the fix-up kernel walks itself

// Partial result of a run of edges: F values (+ the winners' CSR positions for max / min).  P: int inside the unit kernel
// (positions relative to the unit's first one: two DPP moves per column and scan step instead of three, F registers less),
// int64_t in the records and in the fix-up kernel (absolute positions).
template <int RED, int F, typename P = int64_t>
struct Run {
  float v[F];
  P p[RED == kSum ? 1 : F];
};

template <int RED, int F, typename P>
__device__ __forceinline__ void run_reset(Run<RED, F, P>& r) {
#pragma unroll
  for (int c = 0; c < F; ++c) {
    r.v[c] = RED == kSum ? 0.f : (RED == kMax ? -__builtin_huge_valf() : __builtin_huge_valf());
    if constexpr (RED != kSum) r.p[c] = -1;
  }
}

// a = a (+) b with a the EARLIER positions: sums add in that order, max / min keep the earlier winner on a tie
template <int RED, int F, typename P>
__device__ __forceinline__ void run_append(Run<RED, F, P>& a, const Run<RED, F, P>& b) {
#pragma unroll
  for (int c = 0; c < F; ++c) {
    if constexpr (RED == kSum) {
      a.v[c] += b.v[c];
    } else {
      const bool take = RED == kMax ? b.v[c] > a.v[c] : b.v[c] < a.v[c];
      if (take || a.p[c] < 0) {
        if (b.p[c] >= 0) {
          a.v[c] = b.v[c];
          a.p[c] = b.p[c];
        }
      }
    }
  }
}

// Cross-lane moves as DPP modifiers of a VALU move — row_shr:1/2/4/8 inside a row of 16 lanes (0x111 ... 0x118), row_bcast:15
// into rows 1 and 3 (0x142, row mask 0xa), row_bcast:31 into rows 2 and 3 (0x143, 0xc), wave_shr:1 (0x138) — instead of
// ds_bpermute: a scan step is ALU instructions, not an LDS round trip (the scans' chain of round trips was what the waves
// of this kernel waited on).  Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int nr_dpp(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float nr_dpp(float old, float src) {
  return __builtin_bit_cast(float, nr_dpp<CTRL, ROW_MASK>(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int64_t nr_dpp(int64_t old, int64_t src) {
  const uint32_t lo = static_cast<uint32_t>(nr_dpp<CTRL, ROW_MASK>(static_cast<int>(old & 0xffffffffLL), static_cast<int>(src & 0xffffffffLL)));
  const int hi = nr_dpp<CTRL, ROW_MASK>(static_cast<int>(old >> 32), static_cast<int>(src >> 32));
  return (static_cast<int64_t>(hi) << 32) | lo;
}
template <int CTRL, int ROW_MASK, int RED, int F, typename P>
__device__ __forceinline__ Run<RED, F, P> run_dpp(const Run<RED, F, P>& old, const Run<RED, F, P>& r) {
  Run<RED, F, P> o;
#pragma unroll
  for (int c = 0; c < F; ++c) {
    o.v[c] = nr_dpp<CTRL, ROW_MASK>(old.v[c], r.v[c]);
    if constexpr (RED != kSum) o.p[c] = nr_dpp<CTRL, ROW_MASK>(old.p[c], r.p[c]);
  }
  return o;
}
// one step of the segmented inclusive scan: a lane that has a source and no row start yet takes the source's open run in
// front of its own; the "row start seen" flags are OR-ed along
template <int CTRL, int ROW_MASK, int RED, int F, typename P>
__device__ __forceinline__ void scan_step(Run<RED, F, P>& x, int& f, bool has_src) {
  Run<RED, F, P> tx = run_dpp<CTRL, ROW_MASK>(x, x);
  const int tf = nr_dpp<CTRL, ROW_MASK>(f, f);
  if (has_src) {
    if (!f) {
      run_append(tx, x);   // (earlier lanes first)
      x = tx;
    }
    f |= tf;
  }
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void max_step(int64_t& x, bool has_src) {
  const int64_t t = nr_dpp<CTRL, ROW_MASK>(x, x);
  if (has_src && t > x) x = t;
}

struct NrWorkspace {       // per unit: records of the rows it shares with its neighbours
  float* head_v;           // [units][F]
  float* tail_v;
  int64_t* head_p;         // [units][F] (max / min)
  int64_t* tail_p;
  int64_t* head_row;       // [units]
  int64_t* tail_row;
  int64_t* first_row;      // [units] row holding the unit's first position (narrow_plan_kernel)
  uint8_t* flags;          // [units]
  int64_t* long_list;      // [units / kNrInlineSteps + 2] units whose head closes a row cut into many units (narrow_fixup_long_kernel)
  unsigned long long* long_count;
};

// The row that holds a unit's first position — the largest r with indptr[r] <= position: one thread per unit (a search
// per WAVE inside the reduce kernel was 22 dependent loads in front of every unit: 59 rounds of them on this chip).
template <typename Idx>
__global__ __launch_bounds__(256) void narrow_plan_kernel(const Idx* __restrict__ indptr, int64_t num_rows, int64_t units,
                                                         int64_t* __restrict__ first_row, int unit_edges,
                                                         unsigned long long* __restrict__ long_count) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *long_count = 0ull;   // (the fix-up kernel's list of long rows starts empty)
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t u = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; u < units; u += stride) {
    const int64_t base = u * unit_edges;
    int64_t lo = 0, hi = num_rows - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (static_cast<int64_t>(indptr[mid]) <= base)
        lo = mid;
      else
        hi = mid - 1;
    }
    first_row[u] = lo;
  }
}

template <typename Idx>
struct NrOperands {
  const Idx* indptr;
  const Idx* indices;      // column ids (copy_u / binary operators)
  const Idx* eids;         // edge id per position, or null
  const float* ufeat;      // [num_cols][F]
  const float* efeat;      // [nnz][rhs_len]
  float* out;              // [num_rows][F]
  Idx* arg_u;              // winners' column ids (max / min, operators that read u)
  Idx* arg_e;              // winners' edge ids (max / min, operators that read e)
  int op;                  // Op
  int rhs_len;             // columns of an edge row: F, or F / rhs_group
  int rhs_group;           // consecutive output columns sharing one edge value (1: none)
  int mean;                // sum only: store sum / in-degree — the quotient the reference forms after its sum
                           // (python/dgl/ops/spmm.py:109-114), an IEEE division like the merge kernel's DGLA_MEAN
  int stage;               // edge rows of 4 / 8 columns in position order, 16-byte aligned: a unit's rows are fetched as whole
                           // 1 KB wavefront loads and handed to their lanes through LDS (see narrow_reduce_kernel)
  int accumulate;          // sum only: out += result (DGLA_ACCUMULATE: the relation loop of a heterograph, spmm.cuh:528-534);
                           // every row is stored by exactly one thread of one kernel, rows without an edge are not touched
};

template <typename Idx, int RED, int F, int OPK, typename P>
__device__ __forceinline__ void store_row(const NrOperands<Idx>& o, int64_t row, const Run<RED, F, P>& r, int64_t pbase) {
  float den = 1.f;
  if constexpr (RED == kSum) {
    if (o.mean) den = static_cast<float>(static_cast<int64_t>(o.indptr[row + 1]) - static_cast<int64_t>(o.indptr[row]));
  }
#pragma unroll
  for (int c = 0; c < F; ++c) {
    if constexpr (RED == kSum) {
      const float val = o.mean ? r.v[c] / den : r.v[c];
      o.out[row * F + c] = o.accumulate ? o.out[row * F + c] + val : val;
    } else {
      o.out[row * F + c] = r.v[c];
      const int64_t pos = pbase + r.p[c];
      if constexpr (OPK != 0) o.arg_u[row * F + c] = o.indices[pos];   // (OPK 1, 2, 3 read u)
      if constexpr (OPK != 1) o.arg_e[row * F + c] = o.eids ? o.eids[pos] : static_cast<Idx>(pos);
    }
  }
}

// OPK: 0 = copy_e, 1 = copy_u, 2 = u (op) e with an edge row of the output's width, 3 = ... with one edge value per
// `rhs_group` columns.  The operand rows are loaded whole, in front of the arithmetic, with compile-time pitches wherever
// they can be: with a run-time pitch (or the operator's switch between the loads) the compiler gave up the 16-byte loads
// and copy_e at 8 columns went from 0.60 to 1.23 ms.
template <typename Idx, int F, int OPK, bool HAVE_E = false>
__device__ __forceinline__ void message_row(const NrOperands<Idx>& o, int64_t col, int64_t eid, float (&m)[F]) {
  float lv[F], rv[F];   // (HAVE_E: m already holds the edge row)
  if constexpr (OPK != 0) {
#pragma unroll
    for (int c = 0; c < F; ++c) lv[c] = o.ufeat[col * F + c];
  }
  if constexpr (OPK != 1) {
    if constexpr (HAVE_E) {
#pragma unroll
      for (int c = 0; c < F; ++c) rv[c] = m[c];
    } else {
#pragma unroll
      for (int c = 0; c < F; ++c) rv[c] = OPK == 3 ? o.efeat[eid * o.rhs_len + c / o.rhs_group] : o.efeat[eid * F + c];
    }
  }
  if constexpr (OPK == 0) {
#pragma unroll
    for (int c = 0; c < F; ++c) m[c] = rv[c];
  } else if constexpr (OPK == 1) {
#pragma unroll
    for (int c = 0; c < F; ++c) m[c] = lv[c];
  } else {
    const int op = o.op;
#pragma unroll
    for (int c = 0; c < F; ++c)
      m[c] = op == kAdd ? lv[c] + rv[c] : (op == kSub ? lv[c] - rv[c] : (op == kMul ? lv[c] * rv[c] : lv[c] / rv[c]));
  }
}

template <int RED, int F, typename P>
__device__ __forceinline__ void rec_store(float* v, int64_t* p, int64_t unit, const Run<RED, F, P>& r, int64_t pbase) {
#pragma unroll
  for (int c = 0; c < F; ++c) {
    v[unit * F + c] = r.v[c];
    if constexpr (RED != kSum) p[unit * F + c] = r.p[c] < 0 ? static_cast<int64_t>(-1) : pbase + r.p[c];
  }
}

template <int RED, int F>
__device__ __forceinline__ Run<RED, F> rec_load(const float* v, const int64_t* p, int64_t unit) {
  Run<RED, F> r;
#pragma unroll
  for (int c = 0; c < F; ++c) {
    r.v[c] = v[unit * F + c];
    if constexpr (RED != kSum) r.p[c] = p[unit * F + c];
  }
  return r;
}

// Round H of the staged fetch of a unit's edge rows (narrow_reduce_kernel): loads H F / 2 ... H F / 2 + F / 2 - 1 of the
// wavefront = the rows of lanes 32 H ... 32 H + 31 go through the wavefront's half-unit staging area.  Chunk g (16 bytes) sits
// in slot g ^ ((g >> 3) & (F - 1)): the linear writes and the reads at a stride of F chunks are both free of bank conflicts.
typedef float nr_f4 __attribute__((ext_vector_type(4)));   // (a native vector: an array of them stays in registers)
template <int F>
__device__ __forceinline__ int stage_slot(int g) {
  return F == 8 ? ((g & ~7) | ((g ^ (g >> 3)) & 7)) : ((g & ~3) | ((g ^ (g >> 3)) & 3));
}
template <int F, int H>
__device__ __forceinline__ void stage_round(nr_f4* st, const nr_f4 (&t)[F], int lane, float (&msg)[4][F]) {
#pragma unroll
  for (int i = 0; i < F / 2; ++i) st[stage_slot<F>(i * 64 + lane)] = t[H * (F / 2) + i];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool mine = (lane >> 5) == H;
  constexpr int per = F / 4;   // chunks per edge row
#pragma unroll
  for (int c = 0; c < F; ++c) {
    const nr_f4 q = st[stage_slot<F>((lane & 31) * F + c)];   // (every lane reads; the other half's lanes drop the value)
    float* m = &msg[c / per][(c % per) * 4];
    m[0] = (H == 0 || mine) ? q.x : m[0];
    m[1] = (H == 0 || mine) ? q.y : m[1];
    m[2] = (H == 0 || mine) ? q.z : m[2];
    m[3] = (H == 0 || mine) ? q.w : m[3];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <typename Idx, int RED>
__global__ __launch_bounds__(64 * kNrWaves) void narrow_fill_kernel(float* __restrict__ out, Idx* __restrict__ arg_u,
                                                                   Idx* __restrict__ arg_e, int64_t n, Idx arg_empty) {
  const float id = RED == kSum ? 0.f : (RED == kMax ? -__builtin_huge_valf() : __builtin_huge_valf());
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride) {
    out[i] = id;
    if constexpr (RED != kSum) {
      if (arg_u) arg_u[i] = arg_empty;
      if (arg_e) arg_e[i] = arg_empty;
    }
  }
}

template <typename Idx, int RED, int F, int OPK, bool STAGED>
__global__ __launch_bounds__(64 * kNrWaves) void narrow_reduce_kernel(const NrOperands<Idx> o, int64_t num_rows, int64_t nnz,
                                                                     int64_t units, NrWorkspace ws) {
  const Idx* __restrict__ indptr = o.indptr;
  constexpr int kNrLaneEdges = nr_lane_edges(F), kNrUnit = 64 * kNrLaneEdges;
  __shared__ uint32_t s_bits[kNrWaves][kNrUnit / 32];
  __shared__ int64_t s_row[kNrWaves][kNrUnit];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t unit = blockIdx.x * static_cast<int64_t>(kNrWaves) + wave;
  if (unit >= units) return;   // (no workgroup barrier below: the waves of a workgroup are independent)
  const int64_t base = unit * kNrUnit;
  const int64_t uend = base + kNrUnit < nnz ? base + kNrUnit : nnz;

  // the operand rows first: their addresses depend on the positions alone, so they are under way while the rows are found
  const int64_t p0 = base + kNrLaneEdges * lane;
  float msg[kNrLaneEdges][F];
  // Edge rows in position order: a lane's four rows are 64 / 128 consecutive bytes, so a 16-byte load of the wavefront, lane by
  // lane, touches 32 / 64 cache lines and every line is touched by 4 / 8 loads.  Staged instead: the unit's F KB are fetched
  // as F whole 1 KB loads (chunk i * 64 + lane), written to LDS and read back lane-major; the chunk index is XOR-swizzled so
  // that both the linear writes and the reads at a stride of F chunks are free of bank conflicts.
  // The workgroup's staging area holds half a unit per wavefront (two rounds), so that it costs no occupancy.
  static_assert(!STAGED || ((OPK == 0 || OPK == 2) && (F == 4 || F == 8) && kNrLaneEdges == 4), "staged: whole 16-byte chunks per row");
  __shared__ nr_f4 s_stage[STAGED ? kNrWaves * 32 * F : 1];
  if constexpr (STAGED) {
    {
      const nr_f4* __restrict__ src = reinterpret_cast<const nr_f4*>(o.efeat + base * F);
      const int64_t last_chunk = (uend - base) * (F / 4) - 1;
      nr_f4* st = s_stage + wave * 32 * F;
      nr_f4 t[F];
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const int64_t g = i * 64 + lane;   // (the last unit: a chunk past the end re-reads the last one; unused)
        t[i] = src[g < last_chunk ? g : last_chunk];
      }
      int64_t cols[kNrLaneEdges];
#pragma unroll
      for (int j = 0; j < kNrLaneEdges; ++j) cols[j] = OPK != 0 ? static_cast<int64_t>(o.indices[p0 + j < uend ? p0 + j : uend - 1]) : 0;
      stage_round<F, 0>(st, t, lane, msg);
      stage_round<F, 1>(st, t, lane, msg);
#pragma unroll
      for (int j = 0; j < kNrLaneEdges; ++j) {
        message_row<Idx, F, OPK, true>(o, cols[j], p0 + j, msg[j]);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < kNrLaneEdges; ++j) {
      const int64_t pos = p0 + j < uend ? p0 + j : (uend - 1);       // (a position past the end re-reads the last edge; unused)
      const int64_t eid = (OPK != 1 && o.eids) ? static_cast<int64_t>(o.eids[pos]) : pos;
      const int64_t col = OPK != 0 ? static_cast<int64_t>(o.indices[pos]) : 0;
      message_row<Idx, F, OPK>(o, col, eid, msg[j]);
    }
  }
  const int64_t r_first = ws.first_row[unit];   // the row that holds position `base`
  const bool left_open = static_cast<int64_t>(indptr[r_first]) < base;

  // ---- row starts inside the unit -> start bits + row ids in LDS ---------------------------------------------------
  if (lane < kNrUnit / 32) s_bits[wave][lane] = 0u;
  static_assert(32 % kNrLaneEdges == 0, "a lane's start bits lie inside one word");
#pragma unroll
  for (int j = 0; j < kNrLaneEdges; ++j) s_row[wave][kNrLaneEdges * lane + j] = -1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int64_t r = r_first + 1 + lane;; r += 64) {
    bool in = false;
    if (r < num_rows) {
      const int64_t s = static_cast<int64_t>(indptr[r]);
      in = s < uend;
      if (in) {
        const int q = static_cast<int>(s - base);
        atomicOr(&s_bits[wave][q >> 5], 1u << (q & 31));
        atomicMax(reinterpret_cast<long long*>(&s_row[wave][q]), static_cast<long long>(r));
      }
    }
    if (__builtin_amdgcn_ballot_w64(in) != ~0ull) break;   // starts ascend: the first lane past the unit ends the search
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t bits = (s_bits[wave][(lane * kNrLaneEdges) >> 5] >> ((lane * kNrLaneEdges) & 31)) & ((1u << kNrLaneEdges) - 1u);
  int64_t row_at[kNrLaneEdges];
  int64_t lane_last = -1;
#pragma unroll
  for (int j = 0; j < kNrLaneEdges; ++j) {
    row_at[j] = s_row[wave][kNrLaneEdges * lane + j];
    if ((bits >> j) & 1u) lane_last = row_at[j];
  }
  // row in progress at the start of this lane: the last row start in the lanes before it (row ids ascend), or r_first
  const int lr = lane & 15;
  const bool odd_row = ((lane >> 4) & 1) != 0, upper = lane >= 32;
  int64_t incl = lane_last;
  max_step<0x111, 0xf>(incl, lr >= 1);
  max_step<0x112, 0xf>(incl, lr >= 2);
  max_step<0x114, 0xf>(incl, lr >= 4);
  max_step<0x118, 0xf>(incl, lr >= 8);
  max_step<0x142, 0xa>(incl, odd_row);
  max_step<0x143, 0xc>(incl, upper);
  int64_t cur_row;
  {
    const int64_t ex = nr_dpp<0x138, 0xf>(static_cast<int64_t>(-1), incl);   // wave_shr:1; lane 0 keeps -1
    cur_row = ex > r_first ? ex : r_first;
  }

  // ---- the lane's four edges (their operand rows were loaded at the top: the loop below stores the rows it closes, and a
  // load behind a store that may alias it is not moved up — four dependent round trips per lane) ----
  Run<RED, F, int> acc, pre;
  run_reset(acc);
  run_reset(pre);
  int64_t pre_row = -1;
  bool seen = false;
#pragma unroll
  for (int j = 0; j < kNrLaneEdges; ++j) {
    const int64_t pos = p0 + j;
    if ((bits >> j) & 1u) {      // a row starts here: the run in progress ends
      if (!seen) {
        pre = acc;
        pre_row = cur_row;
        seen = true;
      } else {
        store_row<Idx, RED, F, OPK>(o, cur_row, acc, base);
      }
      cur_row = row_at[j];
      run_reset(acc);
    }
    if (pos < uend) {
      Run<RED, F, int> one;
#pragma unroll
      for (int c = 0; c < F; ++c) {
        one.v[c] = msg[j][c];
        if constexpr (RED != kSum) one.p[c] = kNrLaneEdges * lane + j;   // (relative to the unit's first position)
      }
      run_append(acc, one);
    }
  }
  // ---- segmented inclusive scan over the lanes of the runs left open at a lane's end --------------------------------
  Run<RED, F, int> x = acc;
  int f = seen ? 1 : 0;
  scan_step<0x111, 0xf>(x, f, lr >= 1);
  scan_step<0x112, 0xf>(x, f, lr >= 2);
  scan_step<0x114, 0xf>(x, f, lr >= 4);
  scan_step<0x118, 0xf>(x, f, lr >= 8);
  scan_step<0x142, 0xa>(x, f, odd_row);
  scan_step<0x143, 0xc>(x, f, upper);
  Run<RED, F, int> ident;
  run_reset(ident);
  Run<RED, F, int> carry = run_dpp<0x138, 0xf>(ident, x);   // the open run at the end of the lane before (lane 0: none)
  const uint64_t heads = __builtin_amdgcn_ballot_w64(seen);
  const bool first_head_lane = seen && (heads & ((1ull << lane) - 1ull)) == 0ull;
  if (seen) {
    run_append(carry, pre);      // the run this lane's first row start closes: earlier lanes' open run + its own leading edges
    if (first_head_lane && left_open) {
      rec_store<RED, F>(ws.head_v, ws.head_p, unit, carry, base);     // ... began in an earlier unit: a head record
    } else {
      store_row<Idx, RED, F, OPK>(o, pre_row, carry, base);
    }
  }
  // ---- the run open at the unit's end (lane 63 holds it after the scan) -----------------------------------------------
  const int64_t t_row = (static_cast<int64_t>(__builtin_amdgcn_readlane(static_cast<int>(cur_row >> 32), 63)) << 32) |
                        static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cur_row & 0xffffffffLL), 63));
  const bool complete = static_cast<int64_t>(indptr[t_row + 1]) <= uend;
  const bool began_here = heads != 0ull || !left_open;
  if (lane == 63) {
    if (began_here) {
      if (complete)
        store_row<Idx, RED, F, OPK>(o, t_row, x, base);
      else
        rec_store<RED, F>(ws.tail_v, ws.tail_p, unit, x, base);
    } else {
      rec_store<RED, F>(ws.head_v, ws.head_p, unit, x, base);            // the whole unit is a piece of ONE earlier row
    }
  }
  if (lane == 0) {
    uint8_t fl = 0;
    if (left_open) fl |= kNrHead | ((heads != 0ull || complete) ? kNrHeadClosed : 0);
    if (began_here && !complete) fl |= kNrTail;
    ws.flags[unit] = fl;
    ws.head_row[unit] = r_first;
    ws.tail_row[unit] = t_row;
  }
}

// One thread per unit whose head piece CLOSES a row that began earlier: the pieces, walked back to the unit the row began
// in, added in position order.  A row cut into more than kNrInlineSteps units (a hub: 12 000 edges are 47 units, and a
// thread walking them is 47 dependent round trips — longer than the whole unit kernel runs) goes on a list instead.
template <typename Idx, int RED, int F, int OPK>
__global__ __launch_bounds__(256) void narrow_fixup_kernel(const NrOperands<Idx> o, int64_t units, NrWorkspace ws) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t b = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; b < units; b += stride) {
    const uint8_t fl = ws.flags[b];
    if ((fl & (kNrHead | kNrHeadClosed)) != (kNrHead | kNrHeadClosed)) continue;
    int64_t first = -1;   // the unit the row began in, if it is close
    for (int k = 1; k <= kNrInlineSteps && b - k >= 0; ++k) {
      const uint8_t fu = ws.flags[b - k];
      if (!((fu & kNrHead) && !(fu & kNrHeadClosed))) {
        first = b - k;
        break;
      }
    }
    if (first < 0) {
      ws.long_list[atomicAdd(ws.long_count, 1ull)] = b;
      continue;
    }
    Run<RED, F> total = rec_load<RED, F>(ws.head_v, ws.head_p, b);
    for (int64_t u = b - 1; u >= first; --u) {
      Run<RED, F> piece = u > first ? rec_load<RED, F>(ws.head_v, ws.head_p, u) : rec_load<RED, F>(ws.tail_v, ws.tail_p, u);
      run_append(piece, total);   // (unit u > first is a piece of the same row end to end: its head record)
      total = piece;
    }
    store_row<Idx, RED, F, OPK>(o, ws.head_row[b], total, 0);
  }
}

// a (+) b for two partial results of the SAME row whose positions interleave (lanes of the long-row kernel): sums commute;
// max / min keep the better value and, on a tie, the smaller position — what adding them in position order would have kept
template <int RED, int F>
__device__ __forceinline__ void run_merge(Run<RED, F>& a, const Run<RED, F>& b) {
#pragma unroll
  for (int c = 0; c < F; ++c) {
    if constexpr (RED == kSum) {
      a.v[c] += b.v[c];
    } else {
      const bool better = RED == kMax ? b.v[c] > a.v[c] : b.v[c] < a.v[c];
      const bool take = b.p[c] >= 0 && (a.p[c] < 0 || better || (b.v[c] == a.v[c] && b.p[c] < a.p[c]));
      a.v[c] = take ? b.v[c] : a.v[c];
      a.p[c] = take ? b.p[c] : a.p[c];
    }
  }
}

template <int RED, int F>
__device__ __forceinline__ Run<RED, F> run_shfl_xor(const Run<RED, F>& r, int d) {
  Run<RED, F> o;
#pragma unroll
  for (int c = 0; c < F; ++c) {
    o.v[c] = __shfl_xor(r.v[c], d, 64);
    if constexpr (RED != kSum) {
      const uint32_t lo = static_cast<uint32_t>(__shfl_xor(static_cast<int>(r.p[c] & 0xffffffffLL), d, 64));
      const int hi = __shfl_xor(static_cast<int>(r.p[c] >> 32), d, 64);
      o.p[c] = (static_cast<int64_t>(hi) << 32) | lo;
    }
  }
  return o;
}

// One WAVEFRONT per listed unit: 64 pieces of the row per step (lane i: unit b - 1 - i, ...), each lane adds its own pieces
// in position order, the lanes' partial results are merged by a butterfly — a fixed order: the same bits on every run.
template <typename Idx, int RED, int F, int OPK>
__global__ __launch_bounds__(256) void narrow_fixup_long_kernel(const NrOperands<Idx> o, NrWorkspace ws) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
  const int64_t waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  const int64_t n = static_cast<int64_t>(*ws.long_count);
  for (int64_t j = wave; j < n; j += waves) {
    const int64_t b = ws.long_list[j];
    Run<RED, F> part;
    run_reset(part);
    for (int64_t k0 = 1;; k0 += 64) {
      const int64_t u = b - (k0 + lane);
      bool whole = false;
      if (u >= 0) {
        const uint8_t fu = ws.flags[u];
        whole = (fu & kNrHead) && !(fu & kNrHeadClosed);
      }
      const uint64_t stop = __builtin_amdgcn_ballot_w64(!whole);   // (unit 0 is never a piece of an earlier row)
      const int first = stop ? __builtin_ctzll(stop) : 64;
      if (lane <= first && u >= 0) {
        Run<RED, F> piece = lane < first ? rec_load<RED, F>(ws.head_v, ws.head_p, u) : rec_load<RED, F>(ws.tail_v, ws.tail_p, u);
        run_append(piece, part);   // (this step's piece lies in front of the lane's earlier ones)
        part = piece;
      }
      if (stop) break;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const Run<RED, F> other = run_shfl_xor(part, d);
      run_merge(part, other);
    }
    if (lane == 0) {
      const Run<RED, F> head = rec_load<RED, F>(ws.head_v, ws.head_p, b);
      run_merge(part, head);
      store_row<Idx, RED, F, OPK>(o, ws.head_row[b], part, 0);
    }
  }
}

inline size_t nr_align(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

size_t nr_bytes(int64_t nnz, int f, bool cmp) {
  const int64_t unit_edges = 64 * nr_lane_edges(f);
  const size_t units = static_cast<size_t>((nnz + unit_edges - 1) / unit_edges);
  size_t b = 2 * nr_align(units * f * sizeof(float)) + 3 * nr_align(units * sizeof(int64_t)) + nr_align(units) +
             nr_align((units / kNrInlineSteps + 2) * sizeof(int64_t)) + nr_align(sizeof(unsigned long long));
  if (cmp) b += 2 * nr_align(units * f * sizeof(int64_t));
  return b;
}

template <typename Idx, int RED, int F, int OPK>
int nr_launch(const SpmmLaunch& L, char* wsp) {
  const int64_t nnz = L.csr.nnz, n = L.csr.num_rows;
  constexpr int kNrUnit = 64 * nr_lane_edges(F);
  const int64_t units = (nnz + kNrUnit - 1) / kNrUnit;
  NrWorkspace ws;
  char* q = wsp;
  auto take = [&](size_t bytes) {
    char* r = q;
    q += nr_align(bytes);
    return r;
  };
  ws.head_v = reinterpret_cast<float*>(take(units * F * sizeof(float)));
  ws.tail_v = reinterpret_cast<float*>(take(units * F * sizeof(float)));
  ws.head_row = reinterpret_cast<int64_t*>(take(units * sizeof(int64_t)));
  ws.tail_row = reinterpret_cast<int64_t*>(take(units * sizeof(int64_t)));
  ws.first_row = reinterpret_cast<int64_t*>(take(units * sizeof(int64_t)));
  ws.flags = reinterpret_cast<uint8_t*>(take(units));
  ws.long_list = reinterpret_cast<int64_t*>(take((units / kNrInlineSteps + 2) * sizeof(int64_t)));
  ws.long_count = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long)));
  ws.head_p = ws.tail_p = nullptr;
  if (RED != kSum) {
    ws.head_p = reinterpret_cast<int64_t*>(take(units * F * sizeof(int64_t)));
    ws.tail_p = reinterpret_cast<int64_t*>(take(units * F * sizeof(int64_t)));
  }
  NrOperands<Idx> o;
  o.indptr = static_cast<const Idx*>(L.csr.indptr);
  o.indices = static_cast<const Idx*>(L.csr.indices);
  o.eids = static_cast<const Idx*>(L.csr.eids);
  o.ufeat = static_cast<const float*>(L.ufeat);
  o.efeat = static_cast<const float*>(L.efeat);
  o.out = static_cast<float*>(L.out);
  o.arg_u = OPK != 0 ? static_cast<Idx*>(L.arg_u) : nullptr;
  o.arg_e = OPK != 1 ? static_cast<Idx*>(L.arg_e) : nullptr;
  o.op = L.op;
  o.rhs_len = static_cast<int>(L.rhs_len);
  o.rhs_group = L.bcast == kBcRhsGroup ? L.rhs_group : 1;
  o.mean = L.mean ? 1 : 0;
  o.accumulate = L.accumulate ? 1 : 0;
  {
    static const bool stage_on = [] {
      const char* e = std::getenv("DGLA_NARROW_STAGE");
      return !(e && e[0] == '0');
    }();
    o.stage = (stage_on && (OPK == 0 || OPK == 2) && o.eids == nullptr && reinterpret_cast<uintptr_t>(o.efeat) % 16 == 0) ? 1 : 0;
  }
  const int64_t total = n * F;
  const unsigned fill_blocks = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, 8192));
  if (!L.accumulate)
    hipLaunchKernelGGL((narrow_fill_kernel<Idx, RED>), dim3(fill_blocks), dim3(64 * kNrWaves), 0, L.stream, o.out, o.arg_u, o.arg_e,
                       total, static_cast<Idx>(L.arg_empty));
  if (units > 0) {
    hipLaunchKernelGGL((narrow_plan_kernel<Idx>), dim3(static_cast<unsigned>(std::min<int64_t>((units + 255) / 256, 4096))),
                       dim3(256), 0, L.stream, o.indptr, n, units, ws.first_row, kNrUnit, ws.long_count);
    const dim3 grid(static_cast<unsigned>((units + kNrWaves - 1) / kNrWaves));
    constexpr bool kStageable = (OPK == 0 || OPK == 2) && (F == 4 || F == 8) && nr_lane_edges(F) == 4;
    bool staged = false;
    if constexpr (kStageable) {
      if (o.stage) {
        staged = true;
        hipLaunchKernelGGL((narrow_reduce_kernel<Idx, RED, F, OPK, true>), grid, dim3(64 * kNrWaves), 0, L.stream, o, n, nnz, units, ws);
      }
    }
    if (!staged)
      hipLaunchKernelGGL((narrow_reduce_kernel<Idx, RED, F, OPK, false>), grid, dim3(64 * kNrWaves), 0, L.stream, o, n, nnz, units, ws);
    hipLaunchKernelGGL((narrow_fixup_kernel<Idx, RED, F, OPK>),
                       dim3(static_cast<unsigned>(std::min<int64_t>((units + 255) / 256, 4096))), dim3(256), 0, L.stream, o, units,
                       ws);
    if (units > kNrInlineSteps)   // (rows cut into many units, if the graph has any: a list the kernel above filled)
      hipLaunchKernelGGL((narrow_fixup_long_kernel<Idx, RED, F, OPK>),
                         dim3(static_cast<unsigned>(std::min<int64_t>((units / kNrInlineSteps + 3) / 4, 1024))), dim3(256), 0, L.stream, o, ws);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx, int RED, int OPK>
int nr_dispatch_f(const SpmmLaunch& L, char* ws) {
  switch (L.out_len) {
    case 1: return nr_launch<Idx, RED, 1, OPK>(L, ws);
    case 2: return nr_launch<Idx, RED, 2, OPK>(L, ws);
    case 3: return nr_launch<Idx, RED, 3, OPK>(L, ws);
    case 4: return nr_launch<Idx, RED, 4, OPK>(L, ws);
    case 5: return nr_launch<Idx, RED, 5, OPK>(L, ws);
    case 6: return nr_launch<Idx, RED, 6, OPK>(L, ws);
    case 7: return nr_launch<Idx, RED, 7, OPK>(L, ws);
    default: return nr_launch<Idx, RED, 8, OPK>(L, ws);
  }
}

template <typename Idx, int OPK>
int nr_dispatch_r(const SpmmLaunch& L, char* ws) {
  switch (L.red) {
    case kSum: return nr_dispatch_f<Idx, kSum, OPK>(L, ws);
    case kMax: return nr_dispatch_f<Idx, kMax, OPK>(L, ws);
    default: return nr_dispatch_f<Idx, kMin, OPK>(L, ws);
  }
}

template <typename Idx>
int nr_dispatch(const SpmmLaunch& L, char* ws) {
  if (L.op == kCopyRhs) return nr_dispatch_r<Idx, 0>(L, ws);
  if (L.op == kCopyLhs) return nr_dispatch_r<Idx, 1>(L, ws);
  if (L.bcast == kBcRhsGroup && L.rhs_group > 1) return nr_dispatch_r<Idx, 3>(L, ws);
  return nr_dispatch_r<Idx, 2>(L, ws);
}

}  // namespace

// fp32, 1 ... 8 output columns, one relation, plain store, operands of the output's width (an edge operand may be one value
// per `rhs_group` columns): the shapes this file takes (DGLA_NARROW_REDUCE=0: none)
bool narrow_reduce_eligible(const SpmmLaunch& L) {
  static const bool on = [] {
    const char* e = std::getenv("DGLA_NARROW_REDUCE");
    return !(e && e[0] == '0');
  }();
  if (!on || L.dtype != 0 /* DGLA_F32 */ || L.out_len < 1 || L.out_len > 8 || (L.accumulate && (L.red != kSum || L.mean)) ||
      (L.mean && L.red != kSum) ||
      L.prepare_only || L.rel != nullptr || L.rhs_mask || L.csr.nnz <= 0 || L.csr.num_rows <= 0)
    return false;
  if (L.op == kDot) return false;
  const bool use_l = op_uses_lhs(L.op), use_r = op_uses_rhs(L.op);
  if (use_l && (L.lhs_len != L.out_len || L.csr.indices == nullptr)) return false;
  if (use_r) {
    if (L.bcast == kBcNone) {
      if (L.rhs_len != L.out_len) return false;
    } else if (L.bcast == kBcRhsGroup) {
      if (L.rhs_group < 1 || L.rhs_len * L.rhs_group != L.out_len) return false;
    } else {
      return false;
    }
  } else if (L.bcast != kBcNone) {
    return false;
  }
  return true;
}

size_t narrow_reduce_workspace_bytes(const SpmmLaunch& L) { return nr_bytes(L.csr.nnz, static_cast<int>(L.out_len), L.red != kSum); }

static std::atomic<int64_t> g_narrow_calls{0};
int64_t narrow_reduce_calls() { return g_narrow_calls.load(); }

// `ws`: narrow_reduce_workspace_bytes(L) bytes, outside the merge plan's region (the plan of the graph stays valid)
int launch_narrow_reduce(const SpmmLaunch& L, void* ws) {
  g_narrow_calls.fetch_add(1);
  if (L.red != kSum && ((op_uses_rhs(L.op) && !L.arg_e) || (op_uses_lhs(L.op) && !L.arg_u))) {
    last_error() = "arg_u / arg_e are required for max/min";
    return -1;
  }
  return L.csr.idbits == 32 ? nr_dispatch<int32_t>(L, static_cast<char*>(ws)) : nr_dispatch<int64_t>(L, static_cast<char*>(ws));
}

}  // namespace dgla
