// g-SpMM on CSR for gfx950 (MI355X): merge-path row-blocked kernel.
//
// Replaces, from scratch, the reference's SpMMCsrKernel (src/array/cuda/spmm.cuh:496-543,
// host :700-732) and its cuSPARSE route (spmm.cuh:196-284, chosen at spmm.cu:37-58).
//
// Decomposition (why it looks nothing like the reference's thread-per-(row, column) loop):
//  * The CSR is cut by MERGE PATH into units of kWaveItems = 512 "items" (an item is one
//    edge or one row end).  Every wavefront gets exactly one unit, so a 17k-degree row and
//    a run of isolated nodes cost the same per wave: no degree imbalance, no tail.
//    The unit boundaries ("plan", one int64 per wave) are found by a binary search over
//    indptr[r] + r in a tiny pre-kernel and can be cached by the caller per graph.
//  * A wave stages its unit's column indices, row ends (and edge ids) in LDS with
//    coalesced loads, then splits the 64 lanes into G = 64 / LPE lane groups, LPE being the
//    number of lanes needed to cover one feature row with 16-byte loads (F = 100 fp32 ->
//    25 of 32 lanes live, G = 2).  Each group walks its own contiguous sub-range of the
//    unit (found by a second merge search in LDS): for every edge all lanes of the group
//    read the same column id from LDS (broadcast) and gather one 16-byte piece of the
//    neighbour's feature row, U edges in flight per lane plus the next batch prefetched.
//  * Partial sums live in registers; a row end flushes them with one coalesced row store.
//    Rows that straddle group boundaries hand a "carry" (head part) and a "tail" to a
//    workspace; a small fix-up kernel combines them in position order.  No atomics:
//    results are run-to-run deterministic and arg-max / arg-min ties resolve to the lowest
//    CSR position exactly like the reference's sequential loop (functor.cuh:246-254).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace dgla {

template <typename Idx>
struct SpmmParams {
  const Idx* indptr;
  const Idx* indices;
  const Idx* eids;
  int64_t num_rows, nnz, num_waves;
  int wave_items;       // merge items per wavefront: kWaveItems, or a smaller power of two on small graphs
  const int64_t* plan;  // [num_waves + 1] row coordinate of each unit boundary
  const void* ufeat;
  const void* efeat;
  void* out;
  Idx* arg_u;
  Idx* arg_e;
  int out_len, lhs_len, rhs_len;
  int log2_lpe;   // lanes per feature row = 1 << log2_lpe
  int rhs_group;  // kBcRhsGroup
  BcastDims bd;   // kBcGeneral
  int accumulate;
  int rhs_neg, rhs_div;  // sub / div on the add / mul instantiations
  int rhs_mask;          // mul on kBcRhsGroup: rhs words are BIT masks, bit (k mod bits-per-word) gates output column k
  int stage_w;           // scalar edge weights (rhs_len == 1, 4-byte type, sum): staged in LDS with the unit's column ids
  int red_min;           // min on the max instantiation
  int arg_empty;  // arg_u / arg_e of an output element no edge won: 0 (g-SpMM) or -1 (segment reduce)
  int mean;       // reduce == sum only: divide every row by max(its edge count, 1) before storing
  uint32_t tune;  // kTune* bits (common.h)
  // split layouts of ufeat (kTuneSplit): side copies of the parts of every row that make a gather
  // touch one 128-byte line too many; `ufeat` stays the caller's tensor.  split_main == 0: plain layout.
  // `split_meta` (device, may be NULL = always): the locality probe's counters {local, sampled}
  // — the side copy is made and used only when the graph is not in a locality-preserving order
  // (split_wanted(): fewer than 15/16 of the sampled edges local).
  const void* umain;
  const void* utail;
  const unsigned* split_meta;
  int split_main, split_tail;
  // edge layout (split_edge_lines > 0; see spmm_split_edges_kernel): a row of 128 k + t bytes keeps
  // its k - 1 line-aligned interior lines in `ufeat`; `umain` holds ONE 128-byte line per row (the
  // ragged head + the part of the ragged tail that completes the line), `utail` the last t bytes.
  int split_edge_lines;  // k - 1 (0: classic layout / none)
  int split_t16;         // t / 16
  int split_base16;      // (address of ufeat mod 128) / 16: where row 0 starts inside its line
  // straddle layout (split_straddle_slack >= 0; see spmm_straddle_rows_kernel): rows of RB bytes at
  // 8-byte alignment; a row that starts more than `slack` bytes into its line touches one line more
  // than ceil(RB / 128) and is gathered from its line-aligned copy in `umain` (pitch split_main)
  int split_straddle_slack;  // -1: not in use
  int split_row_bytes, split_base_bytes;
  // stacked multi-relation form (MULTI kernels only)
  const uint8_t* rel;
  const void* const* xtab;
  const void* const* wtab;
  int num_rel;
  // fix-up workspace, one slot per lane group: slot = wave * G + group — or, with `wave_slots`
  // (sum reducer, two lane groups per wave), ONE slot per wave: the two groups settle the row that
  // crosses between them in registers (see the end of spmm_csr_merge_kernel)
  int wave_slots;
  // in-kernel fix-up (round 4): arrival counters, one per slot, all zero between launches; NULL = the
  // separate fix-up kernel runs (feature chunks > 1, or more lane groups per wave than the counters cover)
  unsigned* fix_count;
  int64_t* carry_row;  // row id of the slot's carry-out, or -1
  void* carry_val;     // [slots, out_len] accumulator type: head part of a straddling row
  void* tail_val;      // [slots, out_len] accumulator type: tail part of a straddling row
  Idx* carry_argu;     // max / min only
  Idx* carry_arge;
  Idx* tail_argu;
  Idx* tail_arge;
};

// ---------------------------------------------------------------------------------------
// Plan: plan[w] = number of rows that end strictly before merge diagonal w * kWaveItems.
// Row r's end item sits at merge position indptr[r + 1] + r, so the answer is the largest
// i in [0, N] with indptr[i] + i <= d.
// ---------------------------------------------------------------------------------------
template <typename Idx>
__global__ void spmm_merge_plan_kernel(const Idx* __restrict__ indptr, int64_t num_rows,
                                       int64_t nnz, int64_t num_waves,
                                       int64_t* __restrict__ plan, int wave_items) {
  const int64_t w = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (w > num_waves) return;
  int64_t d = w * wave_items;
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

// Non-temporal store of one lane access (16 / 8 / 4 / 2 bytes).
template <typename DT, int VEC>
__device__ __forceinline__ void store_nt(DT* dst, const VecT<DT, VEC>& v) {
  constexpr int B = sizeof(DT) * VEC;
  if constexpr (B == 16) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(*reinterpret_cast<const u4*>(&v), reinterpret_cast<u4*>(dst));
  } else if constexpr (B == 8) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const u2*>(&v), reinterpret_cast<u2*>(dst));
  } else if constexpr (B == 4) {
    __builtin_nontemporal_store(*reinterpret_cast<const uint32_t*>(&v), reinterpret_cast<uint32_t*>(dst));
  } else {
    *reinterpret_cast<VecT<DT, VEC>*>(dst) = v;
  }
}

// Non-temporal load of one lane access: for operands read exactly once in position order (the
// edge operand without an edge-id map, the rows of a segment reduce).  Measured on the load shape
// of this kernel (benchmarks/micro/seq_rows.hip): 6.3 TB/s with default loads, 6.9-7.1 with these.
template <typename DT, int VEC>
__device__ __forceinline__ VecT<DT, VEC> load_nt(const DT* src) {
  constexpr int B = sizeof(DT) * VEC;
  VecT<DT, VEC> v;
  if constexpr (B == 16) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<u4*>(&v) = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src));
  } else if constexpr (B == 8) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u2*>(&v) = __builtin_nontemporal_load(reinterpret_cast<const u2*>(src));
  } else if constexpr (B == 4) {
    *reinterpret_cast<uint32_t*>(&v) = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src));
  } else {
    v = *reinterpret_cast<const VecT<DT, VEC>*>(src);
  }
  return v;
}

// Load of one lane access through a pointer the compiler cannot see the address space of (the operand
// tables of the stacked kernels live in LDS): an explicit GLOBAL load.  As a generic pointer it became
// flat_load, whose completion the compiler cannot count — every reduction of the stacked kernels then
// waited for vmcnt(0), i.e. drained the prefetched batch (seen in the disassembly, round 3).
template <typename DT, int VEC>
__device__ __forceinline__ VecT<DT, VEC> load_global(const DT* src) {
  constexpr int B = sizeof(DT) * VEC;
  VecT<DT, VEC> v;
  if constexpr (B % 16 == 0) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u4* gptr;
#pragma unroll
    for (int i = 0; i < B / 16; ++i) reinterpret_cast<u4*>(&v)[i] = ((gptr)src)[i];
  } else if constexpr (B == 8) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u2*>(&v) = *(const __attribute__((address_space(1))) u2*)src;
  } else if constexpr (B == 4) {
    *reinterpret_cast<uint32_t*>(&v) = *(const __attribute__((address_space(1))) uint32_t*)src;
  } else {
    static_assert(B == 2, "lane access of 2, 4, 8 or a multiple of 16 bytes");
    *reinterpret_cast<uint16_t*>(&v) = *(const __attribute__((address_space(1))) uint16_t*)src;
  }
  return v;
}

// Split layouts (kTuneSplit).  A feature row of RB bytes that is not a multiple of the 128-byte L2
// line straddles one line more than ceil(RB / 128) when gathered (F = 100 fp32: 400 B -> always 4
// lines = 512 B of fabric traffic per edge).  Two side copies remove the extra line: the EDGE layout
// (16-byte lanes, rows of two or more whole lines) and the STRADDLE layout (8-byte lanes).  Round 2's
// whole-row copy ("classic" layout) and round 3's column-sliced tail pass were measured and removed in
// round 4 (docs/DESIGN_detail_r1_r5.md §3.1 keeps the numbers: classic 0.33 ms of copy against the edge layout's 0.10;
// tail pass -5 % .. +3.5 %).
typedef uint32_t piece16_t __attribute__((ext_vector_type(4)));

// Should this launch use the side copy?  (device side; uniform)  The copies move only the rows' ragged
// ends (edge layout: 0.10 ms on C2) or only the straddling rows — they pay for themselves even on a
// graph in community order (variant L, 81 % of the sampled edges local: 3.81 ms with the copy against
// 4.03 ms without), so the copy is declined only when practically every gather stays inside the
// window (>= 15/16 local).
__device__ __forceinline__ bool split_wanted(const unsigned* __restrict__ meta) {
  if (meta == nullptr) return true;
  const unsigned local = meta[0], sampled = meta[1];
  return 16u * local < 15u * sampled;
}

// Edge layout (round 3; default for rows of 128 k + t bytes with k >= 2).  Row r of X starts at byte
// RB r, i.e. o = (t r) mod 128 bytes into a 128-byte line: [RB r, RB r + 128 - o) is a ragged HEAD,
// then k - 1 WHOLE aligned lines, then a ragged tail of o + t bytes.  Only the ragged ends are
// copied: S1[r] (one aligned 128-byte line per row) = head ++ first o bytes of the tail, S2[r] = the
// last t bytes (dense; small enough for the Infinity Cache).  A gather then touches k - 1 lines of X
// in place + 1 line of S1 + a cached piece, like the classic layout, but the copy moves
// (128 + t) / RB of X instead of all of it (F = 100 fp32: 0.67 GB instead of 1.96 GB per call,
// 0.33 -> 0.12 ms): the boundary line between two rows is read once and feeds both rows' S1 lines.
// (o also counts the offset of X itself inside its line, `base16`, so that the in-place pieces are
// whole lines for any 16-byte-aligned X.)  Piece p of the (8 + t16) side pieces of row r:  p < 8: S1 piece p  <-  row piece (p < 8 - o16 ? p :
// p + 8 (k - 1));  p >= 8: S2 piece p - 8  <-  row piece RB/16 - t16 + (p - 8).
template <int K>
__global__ __launch_bounds__(256) void spmm_split_edges_kernel(
    const piece16_t* __restrict__ x, piece16_t* __restrict__ s1, piece16_t* __restrict__ s2,
    int64_t num_rows, int row_pieces, int t16, int interior_pieces, unsigned magic,
    const unsigned* __restrict__ meta, unsigned base16) {
  if (!split_wanted(meta)) return;
  const int side = 8 + t16;
  const int64_t total = num_rows * side;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * (256 * K); base < total;
       base += static_cast<int64_t>(gridDim.x) * (256 * K)) {
    const int64_t r0 = base / side;
    const unsigned p0 = static_cast<unsigned>(base - r0 * side);
    piece16_t v[K];
    int64_t row[K];
    unsigned pp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int64_t i = base + k * 256 + threadIdx.x;
      if (i >= total) i = total - 1;
      const unsigned loc = p0 + static_cast<unsigned>(i - base);
      const unsigned dr = __umulhi(loc, magic);
      const unsigned p = loc - dr * static_cast<unsigned>(side);
      const int64_t r = r0 + dr;
      const unsigned o16 = (static_cast<unsigned>(t16) * static_cast<unsigned>(r & 7) + base16) & 7u;
      const unsigned j = p < 8u ? (p < 8u - o16 ? p : p + static_cast<unsigned>(interior_pieces))
                                : static_cast<unsigned>(row_pieces - t16) + (p - 8u);
      row[k] = r;
      pp[k] = p;
      v[k] = __builtin_nontemporal_load(x + r * row_pieces + j);  // read once
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (base + k * 256 + threadIdx.x >= total) continue;
      if (pp[k] < 8u)
        s1[row[k] * 8 + pp[k]] = v[k];
      else
        s2[row[k] * t16 + (pp[k] - 8u)] = v[k];
    }
  }
}

// Straddle layout (round 3; rows of RB bytes, RB a multiple of 8 but not of 16 — bf16 / fp16 F = 100:
// 200 bytes — gathered with 8-byte lane accesses).  Such a row needs ceil(RB / 128) lines but touches
// one more whenever it starts more than slack = ceil(RB / 128) * 128 - RB bytes into its line (200
// bytes: slack 56, every second row, 2.5 requests per edge instead of 2 — and the kernel runs at
// the request rate).  Only THOSE rows are copied, each to the line-aligned slot of its row index in
// a side array of pitch ceil(RB / 128) * 128; the others are gathered in place.  One thread per
// 8-byte piece of a copied row; rows that stay in place cost one index computation.
typedef uint32_t piece8_t __attribute__((ext_vector_type(2)));
template <int UNUSED>  // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(256) void spmm_straddle_rows_kernel(
    const piece8_t* __restrict__ x, piece8_t* __restrict__ side, int64_t num_rows, int row_pieces,
    int pitch_pieces, int slack, unsigned base_bytes, const unsigned* __restrict__ meta) {
  if (!split_wanted(meta)) return;
  const int64_t total = num_rows * row_pieces;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / row_pieces;
    const int j = static_cast<int>(i - r * row_pieces);
    const unsigned o = static_cast<unsigned>((static_cast<uint64_t>(r) * (row_pieces * 8) + base_bytes) & 127u);
    if (o > static_cast<unsigned>(slack)) side[r * pitch_pieces + j] = __builtin_nontemporal_load(x + i);
  }
}

// Locality probe (once per graph, next to the merge plan): of the edges of every `stride`-th
// unit, how many have their column within `window` rows of the unit's own position (its
// middle row, rescaled to column ids when the matrix is not square)?  Graphs in a
// locality-preserving order (METIS / community order) re-use gathered rows in L2 / Infinity
// Cache and lose more to the re-layout copy than the split gather saves; uniformly random
// neighbour sets gain.  meta[0] += local edges, meta[1] += sampled edges.
template <typename Idx>
__global__ __launch_bounds__(64) void spmm_locality_probe_kernel(
    const Idx* __restrict__ indices, const int64_t* __restrict__ plan, int64_t num_rows,
    int64_t num_cols, int64_t nnz, int64_t num_waves, int64_t stride, int64_t window,
    unsigned* __restrict__ meta, int wave_items) {
  const int64_t w = static_cast<int64_t>(blockIdx.x) * stride;
  if (w >= num_waves) return;
  const int64_t total = num_rows + nnz;
  const int64_t d0 = w * wave_items;
  int64_t d1 = d0 + wave_items;
  if (d1 > total) d1 = total;
  const int64_t i0 = plan[w], i1 = plan[w + 1];
  const int64_t j0 = d0 - i0;
  const int nE = static_cast<int>((d1 - i1) - j0);
  const int64_t mid_row = i0 + (i1 - i0) / 2;
  const int64_t pivot = num_rows == num_cols
                            ? mid_row
                            : static_cast<int64_t>(static_cast<double>(mid_row) * num_cols / (num_rows > 0 ? num_rows : 1));
  unsigned local = 0;
  for (int e = threadIdx.x; e < nE; e += 64) {
    const int64_t c = static_cast<int64_t>(indices[j0 + e]);
    const int64_t d = c > pivot ? c - pivot : pivot - c;
    local += d <= window ? 1u : 0u;
  }
  for (int m = 32; m >= 1; m >>= 1) local += __shfl_xor(local, m, 64);
  if (threadIdx.x == 0 && nE > 0) {
    atomicAdd(meta, local);
    atomicAdd(meta + 1, static_cast<unsigned>(nE));
  }
}

// ---------------------------------------------------------------------------------------
template <int OP, typename A>
__device__ __forceinline__ A apply_op(A l, A r) {
  if constexpr (OP == kAdd) return l + r;
  if constexpr (OP == kSub) return l - r;
  if constexpr (OP == kMul) return l * r;
  if constexpr (OP == kDiv) return l / r;
  if constexpr (OP == kCopyLhs) return l;
  return r;  // kCopyRhs
}

// The binary functor's result is rounded to the storage type before it is reduced, as in
// the reference (`DType out = BinaryOp::Call(...)`, spmm.cuh:524); a no-op for fp32/fp64.
template <typename DT>
__device__ __forceinline__ typename Acc<DT>::type round_to_storage(typename Acc<DT>::type v) {
  if constexpr (sizeof(DT) == 2)
    return to_acc<DT>(from_acc<DT>(v));
  else
    return v;
}

// Written-through ("sc1") accesses of the in-kernel fix-up, on EXPLICIT global pointers: as generic pointers
// (values that travelled through lambda parameters) some instantiations turned them into flat_ instructions
// (tests/test_isa_audit.py).
template <typename T>
__device__ __forceinline__ T ld_sc1(const T* p) {
  return __hip_atomic_load((const __attribute__((address_space(1))) T*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_sc1(T* p, T v) {
  __hip_atomic_store((__attribute__((address_space(1))) T*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MULTI: the CSR is a row-wise concatenation of several relations (SpMMCsrHetero's sum over
// relations sharing a destination type, src/array/cuda/spmm_hetero.cu:150-158, fused into
// one launch): every edge carries its relation id, and the relation's operand base pointers
// are staged in LDS next to the column ids, so the gather loop reads (column, base) pairs.
// Bit k (mod the word's width) of a mask word stored in a feature-typed slot (dgla_spmm_cmp_mask writes them).
template <typename DT>
__device__ __forceinline__ bool mask_bit(DT w, int k) {
  if constexpr (sizeof(DT) == 2) {
    uint16_t b;
    __builtin_memcpy(&b, &w, 2);
    return (b >> (k & 15)) & 1;
  } else if constexpr (sizeof(DT) == 4) {
    uint32_t b;
    __builtin_memcpy(&b, &w, 4);
    return (b >> (k & 31)) & 1;
  } else {
    uint64_t b;
    __builtin_memcpy(&b, &w, 8);
    return (b >> (k & 63)) & 1;
  }
}

// `sub` and `div` are run-time variants of the add / mul instantiations (p.rhs_neg: l + (-r), exact;
// p.rhs_div: l / r instead of l * r) and `accumulate` is a run-time switch of the row store — a third
// of the code objects for wave-uniform scalar branches (round 4).
template <typename Idx, typename DT, int VEC, int OP, int RED, int BC, int U, bool MULTI = false,
          bool NTR = false>
__global__ __launch_bounds__(64 * kWavesPerBlock) void spmm_csr_merge_kernel(
    const SpmmParams<Idx> p) {
  using A = typename Acc<DT>::type;
  constexpr bool UL = op_uses_lhs(OP), UR = op_uses_rhs(OP);
  constexpr bool ARG = RED != kSum;
  // what max / min record about the winning edge: its column (arg_u) and its edge id (arg_e) —
  // in the stacked form its POSITION in the stacked CSR (in the arg_e channel), from which the
  // caller derives column, relation-local edge id and the relation (= node / edge type trackers)
  constexpr bool TU = UL && !MULTI, TE = UR || MULTI;
  // rhs element(s) per lane and edge: a full vector only when rhs is laid out like out
  constexpr int RV = (BC == kBcNone) ? VEC : 1;

  // column ids only where the operator gathers lhs rows; edge ids only when the CSR carries an
  // edge-id map (DYNAMIC LDS, sized by the launch: a map-free copy_rhs — segment reduce, copy_e —
  // staged 16 KB of positions it can compute, and with int64 ids that alone held it to 16 waves per CU)
  __shared__ int s_cols[UL ? kWavesPerBlock : 1][UL ? kWaveItems : 1];
  __shared__ int s_rend[kWavesPerBlock][kWaveItems + 2];
  extern __shared__ __align__(8) unsigned char s_dyn[];
  // stacked form: one relation BYTE per staged edge + the relations' operand base pointers once per
  // workgroup (an 8-byte pointer per edge had doubled the LDS footprint: 20 instead of 28 waves per CU)
  __shared__ uint8_t s_rel[MULTI ? kWavesPerBlock : 1][MULTI ? kWaveItems : 1];
  __shared__ const DT* s_tx[(MULTI && UL) ? 256 : 1];
  __shared__ const DT* s_tw[(MULTI && UR) ? 256 : 1];
  if constexpr (MULTI) {
    for (int i = threadIdx.x; i < p.num_rel; i += 64 * kWavesPerBlock) {
      if constexpr (UL) s_tx[i] = static_cast<const DT*>(p.xtab[i]);
      if constexpr (UR) s_tw[i] = static_cast<const DT*>(p.wtab[i]);
    }
  }

  const int wib = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  // XCD-contiguous unit order: the dispatcher deals workgroups round-robin over the 8 XCDs
  // (block b -> XCD b % 8), so consecutive units -- neighbouring destination rows, whose
  // neighbour sets overlap on any graph with locality -- would land in 8 different L2s.
  // Remapped, XCD x walks one contiguous eighth of the merge path and its 4 MiB L2 holds one
  // window of X instead of a copy of everybody's.  Pure index math: correct under any placement.
  unsigned blk = blockIdx.x;
  if ((p.tune & kTuneXcd) && gridDim.y == 1) {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u;
    const unsigned x = blk & 7u, i = blk >> 3;
    blk = x * q + (x < r ? x : r) + i;
  }
  const int64_t w = static_cast<int64_t>(blk) * kWavesPerBlock + wib;
  const bool has_eid = p.eids != nullptr;
  // scalar edge weights staged in LDS with the unit's column ids (u_mul_e / u_add_e + sum, 4-byte types): one load per EDGE in
  // the staging phase instead of one global instruction per gather batch in the main loop
  constexpr bool kStageW = UL && UR && !MULTI && !NTR && BC == kBcRhsGroup && RED == kSum && sizeof(DT) == 4;
  [[maybe_unused]] const bool stage_w = kStageW && p.stage_w != 0;
  // NTR (spmm_nt_stream(): copy_rhs over long rows without an edge-id map, i.e. a readout-like segment reduce):
  // the edge operand is loaded non-temporally.  A compile-time switch: selecting the load flavour per load at run
  // time put a branch between the prefetch loads and made the compiler drain them (vmcnt(0)) before
  // every reduction instead of waiting with a count.

  int64_t i0 = 0, j0 = 0;
  int R = 0, nE = 0;
  if (w < p.num_waves) {
    const int64_t total = p.num_rows + p.nnz;
    const int64_t d0 = w * p.wave_items;
    int64_t d1 = d0 + p.wave_items;
    if (d1 > total) d1 = total;
    i0 = p.plan[w];
    const int64_t i1 = p.plan[w + 1];
    j0 = d0 - i0;
    R = static_cast<int>(i1 - i0);
    nE = static_cast<int>((d1 - i1) - j0);
    const int items = nE + R;  // <= kWaveItems
    // Stage the unit in LDS.  All loads are issued back to back (addresses clamped instead
    // of predicated) so one HBM round trip covers the unit's nE column ids, its R row ends
    // (indptr[i0 + 1 .. i1]) and, when the operator reads edge features, its edge ids.
    //   s_rend[t]     = local edge offset where local row t starts
    //   s_rend[t + 1] = where it ends;  s_rend[0] < 0 means row i0 began in an earlier unit.
    if (items > 0) {
      Idx itemv[kWaveItems / 64];
      Idx eidv[kWaveItems / 64];
      [[maybe_unused]] uint32_t wv[kStageW ? kWaveItems / 64 : 1];   // scalar edge weights of the unit (stage_w)
      uint8_t relv[MULTI ? kWaveItems / 64 : 1];
#pragma unroll
      for (int k = 0; k < kWaveItems / 64; ++k) {
        if (64 * k >= p.wave_items) break;  // (uniform) small graphs run shorter units
        int it = lane + 64 * k;
        if (it >= items) it = items - 1;
        // column ids are only needed to gather ufeat rows (and to name arg_u): copy_rhs /
        // segment reduce never reads them (p.indices may be NULL there)
        const Idx* src = p.indptr + (i0 + 1 + (it < nE ? 0 : it - nE));
        if constexpr (UL) src = it < nE ? p.indices + (j0 + it) : src;
        itemv[k] = *src;  // (non-temporal loads of the index streams measured neutral: the bit was removed in round 4)
        if constexpr (MULTI) relv[k] = it < nE ? p.rel[j0 + it] : uint8_t(0);
        if constexpr (UR) {
          if (has_eid) {
            const int ie = it < nE ? it : (nE > 0 ? nE - 1 : 0);
            // (explicit global load: as `cond ? p.eids[..] : p.arg_empty` the compiler selected between a global and a
            // kernel-argument ADDRESS and loaded through a flat pointer — tests/test_isa_audit.py)
            eidv[k] = nE > 0 ? load_global<Idx, 1>(p.eids + (j0 + ie)).v[0] : static_cast<Idx>(p.arg_empty);
          }
          if constexpr (kStageW) {
            if (stage_w && !has_eid)   // position order: one coalesced load with the column ids
              wv[k] = nE > 0 ? load_global<uint32_t, 1>(static_cast<const uint32_t*>(p.efeat) + (j0 + (it < nE ? it : nE - 1))).v[0] : 0u;
          }
        }
      }
      if constexpr (kStageW) {
        if (stage_w && has_eid) {   // behind the edge ids: the unit's weights gathered once per edge (not once per lane and edge)
#pragma unroll
          for (int k = 0; k < kWaveItems / 64; ++k) {
            if (64 * k >= p.wave_items) break;
            wv[k] = nE > 0 ? load_global<uint32_t, 1>(static_cast<const uint32_t*>(p.efeat) + static_cast<int64_t>(eidv[k])).v[0] : 0u;
          }
        }
      }
      const int64_t first = static_cast<int64_t>(p.indptr[i0]) - j0;
#pragma unroll
      for (int k = 0; k < kWaveItems / 64; ++k) {
        if (64 * k >= p.wave_items) break;
        const int it = lane + 64 * k;
        if (it < nE) {
          if constexpr (UL) s_cols[wib][it] = static_cast<int>(itemv[k]);
          if constexpr (UR) {
            bool staged = false;
            if constexpr (kStageW) {
              if (stage_w) {   // (the weights take the edge ids' place: a sum names no edge)
                reinterpret_cast<uint32_t*>(s_dyn)[wib * kWaveItems + it] = wv[k];
                staged = true;
              }
            }
            if (!staged && has_eid) reinterpret_cast<Idx*>(s_dyn)[wib * kWaveItems + it] = eidv[k];
          }
          if constexpr (MULTI) s_rel[wib][it] = relv[k];
        } else if (it < items) {
          s_rend[wib][it - nE + 1] = static_cast<int>(static_cast<int64_t>(itemv[k]) - j0);
        }
      }
      if (lane == 0) s_rend[wib][0] = first < 0 ? -1 : static_cast<int>(first);
    } else if (lane == 0) {
      s_rend[wib][0] = 0;
    }
  }
  __syncthreads();
  if (w >= p.num_waves) return;

  const int lpe = 1 << p.log2_lpe;
  const int g = lane >> p.log2_lpe;
  const int lg = lane & (lpe - 1);
  const int G = 64 >> p.log2_lpe;
  const int Tg = p.wave_items >> (6 - p.log2_lpe);
  const int F = p.out_len;
  const int k0 = (static_cast<int>(blockIdx.y) * 64 + lg) * VEC;  // first feature of this lane
  if (k0 >= F) return;

  // explicit LDS pointers: as plain pointers some instantiations (general broadcast + arg outputs) lost the
  // address space and read the winners' column ids with flat_load, which the compiler cannot count
  typedef const __attribute__((address_space(3))) int* lds_int_ptr;
  typedef const __attribute__((address_space(3))) Idx* lds_idx_ptr;
  const lds_int_ptr cols = (lds_int_ptr)s_cols[UL ? wib : 0];
  const lds_int_ptr rend = (lds_int_ptr)s_rend[wib];
  const lds_idx_ptr eidl = (lds_idx_ptr)s_dyn + wib * kWaveItems;  // (valid only with has_eid)

  // ---- split the unit between the lane groups (merge search in LDS) -------------------
  const int items = R + nE;
  int dlo = g * Tg, dhi = dlo + Tg;
  if (dlo > items) dlo = items;
  if (dhi > items) dhi = items;
  auto rows_before = [&](int d) {
    // number of row ends at merge position < d; row end t sits at rend[t + 1] + t
    int lo = 0, hi = R;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rend[mid + 1] + mid < d)
        lo = mid + 1;
      else
        hi = mid;
    }
    return lo;
  };
  const int t_s = rows_before(dlo), t_e = rows_before(dhi);
  const int e_s = dlo - t_s, e_e = dhi - t_e;
  // wave-level slots (sum, G == 2): group 1 keeps the tail of the row that crosses in from group 0
  // in registers; the two parts meet at the end of the kernel
  const bool wslots = (RED == kSum) && p.wave_slots != 0;
  const int64_t slot = wslots ? w : w * G + g;

  // ---- per-lane operand offsets -------------------------------------------------------
  int lo_off = k0, ro_off = k0;
  if constexpr (BC == kBcRhsGroup) ro_off = k0 / p.rhs_group;
  if constexpr (BC == kBcGeneral) bcast_offsets(p.bd, k0, &lo_off, &ro_off);
  const DT* __restrict__ X = static_cast<const DT*>(p.ufeat) + lo_off;
  const DT* __restrict__ Wt = static_cast<const DT*>(p.efeat) + ro_off;
  const int64_t lhs_len = p.lhs_len;
  const int64_t rhs_len = p.rhs_len;
  // edge layout (wave-uniform switch): per-lane constants of the address select in load_batch
  constexpr int E16 = 16 / static_cast<int>(sizeof(DT));  // elements per 16-byte piece
  bool edge_layout = false;
  [[maybe_unused]] unsigned ej = 0u;       // this lane's piece index in the row (huge: never in place)
  [[maybe_unused]] int e_piece_lo = 0, e_piece_hi = 0;  // side-array element offset when the piece is in the head / tail
  [[maybe_unused]] int e_pitch = 0;        // side-array pitch of this lane in elements
  [[maybe_unused]] const DT* e_side = nullptr;
  bool straddle = false;  // (wave-uniform)
  if (p.split_main > 0 && p.split_straddle_slack >= 0) {
    if constexpr (UL && !MULTI && VEC * sizeof(DT) == 8) straddle = split_wanted(p.split_meta);
  } else if (p.split_main > 0 && p.split_edge_lines > 0 && split_wanted(p.split_meta)) {
    {
      if constexpr (UL && !MULTI && VEC * sizeof(DT) == 16) {
        edge_layout = true;
        const int j = lo_off / E16, rp = p.lhs_len / E16;
        if (j >= rp - p.split_t16) {  // the dense tail array: a fixed place for this lane
          ej = 0x40000000u;
          e_side = static_cast<const DT*>(p.utail);
          e_pitch = p.split_t16 * E16;
          e_piece_lo = e_piece_hi = (j - (rp - p.split_t16)) * E16;
        } else {
          ej = static_cast<unsigned>(j);
          e_side = static_cast<const DT*>(p.umain);
          e_pitch = 8 * E16;
          e_piece_lo = j * E16;
          e_piece_hi = (j - 8 * p.split_edge_lines) * E16;
        }
      }
    }
  }
  [[maybe_unused]] const unsigned e_in_place = 8u * static_cast<unsigned>(p.split_edge_lines);
  [[maybe_unused]] const unsigned e_t16 = static_cast<unsigned>(p.split_t16);
  [[maybe_unused]] const unsigned e_base16 = static_cast<unsigned>(p.split_base16);

  using XV = VecT<DT, VEC>;
  using WV = VecT<DT, RV>;
  struct Batch {
    XV x[UL ? U : 1];
    WV w[UR ? U : 1];
  };
  auto load_batch = [&](int e, Batch& b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int ee = e + u;
      if (ee >= e_e) ee = e_e - 1;  // clamp: re-load a valid edge, ignored when reducing
      if constexpr (UL) {
        const int64_t c = cols[ee];
        const DT* xb = X;
        if constexpr (MULTI) xb = s_tx[s_rel[wib][ee]] + lo_off;
        bool done = false;
        if constexpr (!MULTI && VEC * sizeof(DT) == 16) {
          if (edge_layout) {  // (wave-uniform)
            // row c starts o16 pieces into a line: pieces [8 - o16, 8 - o16 + 8 (k - 1)) are whole
            // aligned lines of X itself, the ragged ends live in the side line of the row
            const unsigned o16 = (e_t16 * static_cast<unsigned>(c) + e_base16) & 7u;
            const unsigned head = 8u - o16;
            const bool in_place = (ej - head) < e_in_place;
            const int side_off = ej < head ? e_piece_lo : e_piece_hi;
            const DT* base = in_place ? X : e_side;
            const int64_t pitch = in_place ? lhs_len : static_cast<int64_t>(e_pitch);
            const int add = in_place ? 0 : side_off;
            b.x[u] = *reinterpret_cast<const XV*>(base + (c * pitch + add));
            done = true;
          }
        }
        if constexpr (!MULTI && VEC * sizeof(DT) == 8) {
          if (straddle) {
            // row c starts o bytes into its line; past the slack it would touch one line too many:
            // take its line-aligned copy
            const unsigned o = static_cast<unsigned>((static_cast<uint64_t>(c) * static_cast<unsigned>(p.split_row_bytes) +
                                                      static_cast<unsigned>(p.split_base_bytes)) & 127u);
            const bool from_side = o > static_cast<unsigned>(p.split_straddle_slack);
            const DT* base = from_side ? static_cast<const DT*>(p.umain) + lo_off : xb;
            const int64_t pitch = from_side ? static_cast<int64_t>(p.split_main) : lhs_len;
            b.x[u] = *reinterpret_cast<const XV*>(base + c * pitch);
            done = true;
          }
        }
        if constexpr (MULTI)
          b.x[u] = load_global<DT, VEC>(xb + c * lhs_len);
        else if (!done)
          b.x[u] = *reinterpret_cast<const XV*>(xb + c * lhs_len);
      }
      if constexpr (UR) {
        const int64_t eid = has_eid ? static_cast<int64_t>(eidl[ee]) : j0 + ee;  // no map: edge id == position
        const DT* wb = Wt;
        if constexpr (MULTI) wb = s_tw[s_rel[wib][ee]] + ro_off;
        if constexpr (NTR)
          b.w[u] = load_nt<DT, RV>(wb + eid * rhs_len);  // position order: every piece read once
        else if constexpr (MULTI)
          b.w[u] = load_global<DT, RV>(wb + eid * rhs_len);
        else
          b.w[u] = *reinterpret_cast<const WV*>(wb + eid * rhs_len);
      }
    }
  };

  // ---- masked launch (dgla_spmm_csr_masked: rhs words are bit masks): the mask word is fetched ONE BATCH AHEAD of the row
  // gather, and a lane whose VEC columns are all switched off for an edge reads a fixed, cache-resident address instead of
  // its piece of the row: a 128-byte line of the gathered operand is requested only when one of its columns is wanted
  // (max / min backward: an edge wins ~F / degree columns, so 2.3 instead of 4 lines per edge at C2).
  constexpr bool kGated = UL && UR && !MULTI && OP == kMul && BC == kBcRhsGroup && RED == kSum && RV == 1;
  [[maybe_unused]] auto load_w_only = [&](int e, Batch& b) {
    if constexpr (kGated) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int ee = e + u;
        if (ee >= e_e) ee = e_e - 1;
        const int64_t eid = has_eid ? static_cast<int64_t>(eidl[ee]) : j0 + ee;
        b.w[u] = *reinterpret_cast<const WV*>(Wt + eid * rhs_len);
      }
    }
  };
  [[maybe_unused]] const XV* const gate_dummy =
      reinterpret_cast<const XV*>(edge_layout ? static_cast<const DT*>(p.umain) : static_cast<const DT*>(p.ufeat));   // (the first piece of the operand: always inside it)
  [[maybe_unused]] auto load_x_gated = [&](int e, Batch& b) {
    if constexpr (kGated) {
      constexpr int BITS = 8 * static_cast<int>(sizeof(DT));
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int ee = e + u;
        if (ee >= e_e) ee = e_e - 1;
        const int64_t c = cols[ee];
        const XV* ptr;
        bool done = false;
        if constexpr (VEC * sizeof(DT) == 16) {
          if (edge_layout) {
            const unsigned o16 = (e_t16 * static_cast<unsigned>(c) + e_base16) & 7u;
            const unsigned head = 8u - o16;
            const bool in_place = (ej - head) < e_in_place;
            const int side_off = ej < head ? e_piece_lo : e_piece_hi;
            const DT* base = in_place ? X : e_side;
            const int64_t pitch = in_place ? lhs_len : static_cast<int64_t>(e_pitch);
            const int add = in_place ? 0 : side_off;
            ptr = reinterpret_cast<const XV*>(base + (c * pitch + add));
            done = true;
          }
        }
        if constexpr (VEC * sizeof(DT) == 8) {
          if (straddle) {
            const unsigned o = static_cast<unsigned>((static_cast<uint64_t>(c) * static_cast<unsigned>(p.split_row_bytes) +
                                                      static_cast<unsigned>(p.split_base_bytes)) & 127u);
            const bool from_side = o > static_cast<unsigned>(p.split_straddle_slack);
            const DT* base = from_side ? static_cast<const DT*>(p.umain) + lo_off : X;
            const int64_t pitch = from_side ? static_cast<int64_t>(p.split_main) : lhs_len;
            ptr = reinterpret_cast<const XV*>(base + c * pitch);
            done = true;
          }
        }
        if (!done) ptr = reinterpret_cast<const XV*>(X + c * lhs_len);
        uint64_t wb = 0;
        __builtin_memcpy(&wb, &b.w[u].v[0], sizeof(DT));
        const bool wanted = ((wb >> (k0 & (BITS - 1))) & ((1u << VEC) - 1u)) != 0u;
        b.x[u] = *(wanted ? ptr : gate_dummy);
      }
    }
  };

  // load_batch with the edge operand read from the staged weights (stage_w): no global load for it
  [[maybe_unused]] auto load_batch_sw = [&](int e, Batch& b) {
    if constexpr (kStageW) {
      typedef const __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
      const lds_u32_ptr sw = (lds_u32_ptr)s_dyn + wib * kWaveItems;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int ee = e + u;
        if (ee >= e_e) ee = e_e - 1;
        const int64_t c = cols[ee];
        bool done = false;
        if constexpr (VEC * sizeof(DT) == 16) {
          if (edge_layout) {
            const unsigned o16 = (e_t16 * static_cast<unsigned>(c) + e_base16) & 7u;
            const unsigned head = 8u - o16;
            const bool in_place = (ej - head) < e_in_place;
            const int side_off = ej < head ? e_piece_lo : e_piece_hi;
            const DT* base = in_place ? X : e_side;
            const int64_t pitch = in_place ? lhs_len : static_cast<int64_t>(e_pitch);
            const int add = in_place ? 0 : side_off;
            b.x[u] = *reinterpret_cast<const XV*>(base + (c * pitch + add));
            done = true;
          }
        }
        if constexpr (VEC * sizeof(DT) == 8) {
          if (straddle) {
            const unsigned o = static_cast<unsigned>((static_cast<uint64_t>(c) * static_cast<unsigned>(p.split_row_bytes) +
                                                      static_cast<unsigned>(p.split_base_bytes)) & 127u);
            const bool from_side = o > static_cast<unsigned>(p.split_straddle_slack);
            const DT* base = from_side ? static_cast<const DT*>(p.umain) + lo_off : X;
            const int64_t pitch = from_side ? static_cast<int64_t>(p.split_main) : lhs_len;
            b.x[u] = *reinterpret_cast<const XV*>(base + c * pitch);
            done = true;
          }
        }
        if (!done) b.x[u] = *reinterpret_cast<const XV*>(X + c * lhs_len);
        const uint32_t wbits = sw[ee];
        __builtin_memcpy(&b.w[u].v[0], &wbits, 4);
      }
    }
  };

  A acc[VEC];
  int best[ARG ? VEC : 1];
  // RED is kSum or kMax; the kMax instantiation also runs min (p.red_min: wave-uniform compare direction)
  const A ident = red_identity<DT>(RED == kSum ? kSum : (p.red_min ? kMin : kMax));
  [[maybe_unused]] const bool red_min = p.red_min != 0;
  int cnt = 0;
  auto reset = [&]() {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[v] = ident;
      if constexpr (ARG) best[v] = -1;
    }
    cnt = 0;
  };
  reset();

  // the arg_e value of the unit's edge bi (bi < 0: no edge won)
  auto edge_name = [&](int bi) -> Idx {
    if (bi < 0) return static_cast<Idx>(p.arg_empty);
    if constexpr (MULTI)
      return static_cast<Idx>(j0 + bi);
    else if constexpr (UR)
      return has_eid ? eidl[bi] : static_cast<Idx>(j0 + bi);
    else
      return Idx(0);
  };
  DT* __restrict__ out = static_cast<DT*>(p.out);
  // Does local row t_s have edges before this group's range?  Then earlier groups hold
  // carries for it and its total is assembled by the fix-up kernel from `tail_val`.
  bool first_is_tail = rend[t_s] < e_s;
  [[maybe_unused]] A treg[VEC];       // wslots, group 1: the crossing row's part inside this group
  [[maybe_unused]] bool have_tail = false;
  [[maybe_unused]] int tail_t = 0;    // its local row

  // ---- partial results of rows that straddle slots -------------------------------------------------------
  // Separate fix-up kernel (fix_count == NULL): plain stores, combined after the launch.  In-kernel fix-up:
  // the partials are written THROUGH (relaxed agent-scope stores = `sc1`), every slot that holds a part of
  // row R takes a ticket on the counter of the slot R ends in (after `s_waitcnt vmcnt(0)`: its part is out),
  // and the slot that draws the last ticket reads the parts back (`sc1` loads) IN SLOT ORDER — the same
  // order, hence the same bits, as the fix-up kernel — and writes the row.  (MI355X_MICROARCH.md, "valid
  // forms": sc1 payload -> vmcnt(0) -> agent atomic; consumer sc1 loads.)  No waiting anywhere.
  const bool fuse = p.fix_count != nullptr;
  [[maybe_unused]] int64_t tail_pending = -1;  // row whose tail part this slot has stored
  auto put_part = [&](A* dst, const A (&vals)[VEC]) {
    if (fuse) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) st_sc1(dst + v, vals[v]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) dst[v] = vals[v];
    }
  };
  auto put_arg = [&](Idx* dst, Idx val) {
    if (fuse)
      st_sc1(dst, val);
    else
      *dst = val;
  };

  auto flush = [&](int t) {
    const int64_t row = i0 + t;
    if (first_is_tail && RED == kSum && wslots && g == 1) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) treg[v] = acc[v];
      have_tail = true;
      tail_t = t;
      first_is_tail = false;
    } else if (first_is_tail) {
      put_part(static_cast<A*>(p.tail_val) + slot * F + k0, acc);
      if constexpr (ARG) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int bi = best[v];
          if constexpr (TU) put_arg(p.tail_argu + slot * F + k0 + v, bi >= 0 ? static_cast<Idx>(cols[bi]) : static_cast<Idx>(p.arg_empty));
          if constexpr (TE) put_arg(p.tail_arge + slot * F + k0 + v, edge_name(bi));
        }
      }
      tail_pending = row;
      first_is_tail = false;
    } else {
      DT* o = out + row * F + k0;
      VecT<DT, VEC> ov;
      if (p.accumulate) {
        ov = *reinterpret_cast<VecT<DT, VEC>*>(o);
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(to_acc<DT>(ov.v[v]) + acc[v]);
      } else if (RED == kSum && p.mean) {
        // fused `mean` (python/dgl/ops/spmm.py:109-114: sum, then / clamp(in_degree, 1)): a row
        // written here lies wholly inside this group, so `cnt` is its in-degree.  Same two
        // roundings as the reference: the sum to the storage type, then the quotient.
        // the reference casts the degree to the feature dtype before dividing (`F.astype(deg,
        // F.dtype(ret))`): in 16-bit storage a degree above 256 (bf16) / 2048 (fp16) is rounded
        const A den = round_to_storage<DT>(static_cast<A>(cnt > 1 ? cnt : 1));
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(round_to_storage<DT>(acc[v]) / den);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(acc[v]);
      }
      *reinterpret_cast<VecT<DT, VEC>*>(o) = ov;  // (non-temporal row stores measured neutral: bit removed in round 4)
      if constexpr (ARG) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const int bi = best[v];
          if constexpr (TU) p.arg_u[row * F + k0 + v] = bi >= 0 ? static_cast<Idx>(cols[bi]) : static_cast<Idx>(p.arg_empty);
          if constexpr (TE) p.arg_e[row * F + k0 + v] = edge_name(bi);
        }
      }
    }
    reset();
  };

  // Take this slot's ticket for row R (its part is stored); the last ticket of the run combines and writes.
  auto finish_run = [&](int64_t R) {
    const int64_t ra = static_cast<int64_t>(p.indptr[R]), rb = static_cast<int64_t>(p.indptr[R + 1]);
    const int sh_w = 31 - __builtin_clz(static_cast<unsigned>(p.wave_items));
    auto slot_of = [&](int64_t pos) -> int64_t {  // slot that holds merge position `pos`
      const int64_t u = pos >> sh_w;
      if (wslots) return u;
      const int sh_g = 31 - __builtin_clz(static_cast<unsigned>(Tg));
      return u * G + ((pos - (u << sh_w)) >> sh_g);
    };
    const int64_t s_first = slot_of(ra + R), s_last = slot_of(rb + R);  // first edge / row-end item of R
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's written-through parts have left
    unsigned ticket = 0;
    if (lg == 0)
      ticket = __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned*)(p.fix_count + s_last), 1u,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, lane - lg, 64);
    if (ticket != static_cast<unsigned>(s_last - s_first)) return;  // somebody else draws the last ticket
    if (lg == 0) st_sc1(p.fix_count + s_last, 0u);
    const A* cv = static_cast<const A*>(p.carry_val);
    const A* tv = static_cast<const A*>(p.tail_val);
    auto get = [&](const A* src) { return ld_sc1(src); };
    auto geti = [&](const Idx* src) { return ld_sc1(src); };
    A tot[VEC];
    [[maybe_unused]] Idx au[VEC], ae[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      tot[v] = get(cv + s_first * F + k0 + v);
      if constexpr (ARG) {
        if constexpr (TU) au[v] = geti(p.carry_argu + s_first * F + k0 + v);
        if constexpr (TE) ae[v] = geti(p.carry_arge + s_first * F + k0 + v);
      }
    }
    auto combine = [&](const A* vals, const Idx* pu, const Idx* pe, int64_t base) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const A val = get(vals + base + v);
        if constexpr (RED == kSum) {
          tot[v] += val;
        } else {
          if (red_min ? tot[v] > val : tot[v] < val) {
            tot[v] = val;
            if constexpr (TU) au[v] = geti(pu + base + v);
            if constexpr (TE) ae[v] = geti(pe + base + v);
          }
        }
      }
    };
    for (int64_t q = s_first + 1; q < s_last; ++q) combine(cv, p.carry_argu, p.carry_arge, q * F + k0);
    combine(tv, p.tail_argu, p.tail_arge, s_last * F + k0);
    DT* o = out + R * F + k0;
    VecT<DT, VEC> ov;
    if (RED == kSum && p.mean) {
      const int64_t deg = rb - ra;
      const A den = round_to_storage<DT>(static_cast<A>(deg > 1 ? deg : 1));
#pragma unroll
      for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(round_to_storage<DT>(tot[v]) / den);
    } else if (p.accumulate) {
      ov = *reinterpret_cast<VecT<DT, VEC>*>(o);
#pragma unroll
      for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(to_acc<DT>(ov.v[v]) + tot[v]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(tot[v]);
    }
    *reinterpret_cast<VecT<DT, VEC>*>(o) = ov;
    if constexpr (ARG) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if constexpr (TU) p.arg_u[R * F + k0 + v] = au[v];
        if constexpr (TE) p.arg_e[R * F + k0 + v] = ae[v];
      }
    }
  };

  int t = t_s;
  int next_end = (t < R) ? rend[t + 1] : 0x7fffffff;
  [[maybe_unused]] const bool rhs_neg = p.rhs_neg != 0, rhs_div = p.rhs_div != 0, rhs_mask = p.rhs_mask != 0;

  auto reduce_batch = [&](int e, const Batch& b) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ee = e + u;
      if (ee < e_e) {
        while (ee >= next_end) {
          flush(t);
          ++t;
          next_end = (t < R) ? rend[t + 1] : 0x7fffffff;
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          A l = A(0), r = A(0);
          if constexpr (UL) l = to_acc<DT>(b.x[u].v[v]);
          if constexpr (UR) r = to_acc<DT>(b.w[u].v[RV == 1 ? 0 : v]);
          A raw;
          if constexpr (OP == kAdd)
            raw = l + (rhs_neg ? -r : r);     // sub: l + (-r) == l - r bit for bit
          else if constexpr (OP == kMul && BC == kBcRhsGroup)
            // rhs_mask (dgla_spmm_csr_masked, wave-uniform): the lane's rhs word holds one bit per output column
            // of its group; a cleared bit drops the edge for that column (a select, not a product: inf and NaN
            // of a dropped edge must not reach the sum)
            raw = rhs_mask ? (mask_bit<DT>(b.w[u].v[0], k0 + v) ? l : A(0)) : (rhs_div ? l / r : l * r);
          else if constexpr (OP == kMul)
            raw = rhs_div ? l / r : l * r;    // (wave-uniform)
          else
            raw = apply_op<OP, A>(l, r);
          const A val = round_to_storage<DT>(raw);
          if constexpr (RED == kSum) {
            acc[v] += val;
          } else {
            if (red_min ? acc[v] > val : acc[v] < val) {  // strict, as the reference (functor.cuh:246-254)
              acc[v] = val;
              best[v] = ee;
            }
          }
        }
        ++cnt;
      }
    }
  };

  // ---- main loop: U gathers in flight per lane, next batch issued before reducing -----
  // The prefetch is unconditional (load_batch clamps to the group's last edge) so the
  // loop body is branch-free up to the reduction and the compiler can wait with a counted
  // vmcnt(U) instead of draining the prefetched batch.
  bool gated_done = false;
  if constexpr (kGated) {
    if (rhs_mask && VEC <= static_cast<int>(8 * sizeof(DT)) && (p.tune & kTuneNoGate) == 0) {  // (wave-uniform)
      gated_done = true;
      if (e_s < e_e) {
        // three batches in rotation: mask words two batches ahead of the reduction, rows one batch ahead
        int e = e_s;
        Batch b0, b1, b2;
        load_w_only(e, b0);
        load_w_only(e + U, b1);
        load_x_gated(e, b0);
        for (; e < e_e; e += 3 * U) {
          load_w_only(e + 2 * U, b2);
          load_x_gated(e + U, b1);
          reduce_batch(e, b0);
          load_w_only(e + 3 * U, b0);
          load_x_gated(e + 2 * U, b2);
          reduce_batch(e + U, b1);
          load_w_only(e + 4 * U, b1);
          load_x_gated(e + 3 * U, b0);
          reduce_batch(e + 2 * U, b2);
        }
      }
    }
  }
  if constexpr (kStageW) {
    if (stage_w && !gated_done) {   // (wave-uniform)
      gated_done = true;
      if (e_s < e_e) {
        int e = e_s;
        Batch ba, bb;
        load_batch_sw(e, ba);
        for (; e < e_e; e += 2 * U) {
          load_batch_sw(e + U, bb);
          reduce_batch(e, ba);
          load_batch_sw(e + 2 * U, ba);
          reduce_batch(e + U, bb);
        }
      }
    }
  }
  if (!gated_done && e_s < e_e) {
    int e = e_s;
    Batch ba, bb;  // ping-pong: copying a batch would wait for its loads
    load_batch(e, ba);
    for (; e < e_e; e += 2 * U) {
      load_batch(e + U, bb);
      reduce_batch(e, ba);
      load_batch(e + 2 * U, ba);
      reduce_batch(e + U, bb);
    }
  }
  while (t < t_e) {  // row ends after the last edge (includes zero-degree rows)
    flush(t);
    ++t;
  }

  if constexpr (RED == kSum) {
    if (wslots) {
      // ---- the two groups of the wave meet (lane lg of group 1 <-> lane lg of group 0) -----------
      // c0 = group 0's carry-out.  Group 1 either closed that row (have_tail: total = c0 + treg, in
      // position order) or lies wholly inside it (no row end: the wave's carry-out is c0 + its own
      // part).  A total whose row began in THIS unit is final and is written here; one that began
      // in an earlier unit is the wave's tail for the fix-up kernel.
      A c0[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) c0[v] = __shfl_xor(acc[v], 32, 64);
      const int cnt0 = __shfl_xor(cnt, 32, 64);
      if (g == 1) {
        const bool g0_carry = cnt0 > 0;
        int64_t crow = -1;
        if (have_tail) {  // (implies g0_carry)
          A tot[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) tot[v] = c0[v] + treg[v];
          const bool began_here = tail_t > 0 || rend[0] >= 0;
          if (began_here) {
            DT* o = out + (i0 + tail_t) * F + k0;
            VecT<DT, VEC> ov;
            if (p.accumulate) {
              ov = *reinterpret_cast<VecT<DT, VEC>*>(o);
#pragma unroll
              for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(to_acc<DT>(ov.v[v]) + tot[v]);
            } else if (p.mean) {
              const int deg = rend[tail_t + 1] - rend[tail_t];
              const A den = round_to_storage<DT>(static_cast<A>(deg > 1 ? deg : 1));
#pragma unroll
              for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(round_to_storage<DT>(tot[v]) / den);
            } else {
#pragma unroll
              for (int v = 0; v < VEC; ++v) ov.v[v] = from_acc<DT>(tot[v]);
            }
            *reinterpret_cast<VecT<DT, VEC>*>(o) = ov;
          } else {
            put_part(static_cast<A*>(p.tail_val) + slot * F + k0, tot);
            tail_pending = i0 + tail_t;
          }
          if (cnt > 0) {  // group 1's own unfinished last row
            crow = i0 + t;
            put_part(static_cast<A*>(p.carry_val) + slot * F + k0, acc);
          }
        } else if (g0_carry || cnt > 0) {
          // no crossing row closed in group 1: either group 0 ended on a row end (then this is group
          // 1's own carry) or the row runs through all of group 1 (then its parts add up, in order)
          crow = i0 + t;
          A cvv[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) cvv[v] = g0_carry ? (cnt > 0 ? c0[v] + acc[v] : c0[v]) : acc[v];
          put_part(static_cast<A*>(p.carry_val) + slot * F + k0, cvv);
        }
        if (!fuse && lg == 0) p.carry_row[slot] = crow;
        if (fuse && crow >= 0) finish_run(crow);
      }
      if (fuse && tail_pending >= 0) finish_run(tail_pending);
      return;
    }
  }
  // ---- carry-out: head part of the row that continues in the next group ---------------
  const bool has_carry = cnt > 0;
  if (!fuse && lg == 0 && blockIdx.y == 0) p.carry_row[slot] = has_carry ? i0 + t : int64_t(-1);
  if (has_carry) {
    put_part(static_cast<A*>(p.carry_val) + slot * F + k0, acc);
    if constexpr (ARG) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int bi = best[v];
        if constexpr (TU) put_arg(p.carry_argu + slot * F + k0 + v, bi >= 0 ? static_cast<Idx>(cols[bi]) : static_cast<Idx>(p.arg_empty));
        if constexpr (TE) put_arg(p.carry_arge + slot * F + k0 + v, edge_name(bi));
      }
    }
  }
  if (fuse) {
    if (tail_pending >= 0) finish_run(tail_pending);
    if (has_carry) finish_run(i0 + t);
  }
}

// ---------------------------------------------------------------------------------------
// Fix-up: for every maximal run of slots s .. s2-1 carrying the same row, combine the
// carries in slot (= CSR position) order, then the tail held by slot s2, and write the row.
// One 64-lane block per slot; non-leaders exit at once.
// ---------------------------------------------------------------------------------------
template <typename Idx, typename DT, int OP, int RED, bool MULTI = false>
__global__ __launch_bounds__(64) void spmm_csr_fixup_kernel(const SpmmParams<Idx> p,
                                                            int64_t num_slots) {
  using A = typename Acc<DT>::type;
  constexpr bool UL = op_uses_lhs(OP) && !MULTI, UR = op_uses_rhs(OP) || MULTI;  // tracked channels
  constexpr bool ARG = RED != kSum;
  // grid-stride over the slots: one block per slot would need num_slots * 64 threads, which
  // passes HIP's 2^32 threads-per-launch limit for narrow features on billion-edge graphs
  // (tests/test_gpu_large.py)
  for (int64_t s = blockIdx.x; s < num_slots; s += gridDim.x) {
    const int64_t row = p.carry_row[s];
    if (row < 0) continue;
    if (s > 0 && p.carry_row[s - 1] == row) continue;
    // end of the run: 64 slots per step (a row of a few-segments reduction, or a hub row, spans
    // hundreds of slots: walking them one dependent load at a time was most of this kernel's time)
    int64_t s2 = s + 1;
    for (;;) {
      const int64_t q = s2 + threadIdx.x;
      const bool same = q < num_slots && p.carry_row[q] == row;
      const uint64_t m = __ballot(same);
      if (m != ~uint64_t(0)) {
        s2 += __builtin_ctzll(~m);
        break;
      }
      s2 += 64;
    }
    // slot s2 holds the tail (the group in which the row ends); it always exists because a
    // row with a carry has its row-end item in a later slot.
    const int F = p.out_len;
    const A* cv = static_cast<const A*>(p.carry_val);
    const A* tv = static_cast<const A*>(p.tail_val);
    DT* out = static_cast<DT*>(p.out);
    for (int k = threadIdx.x; k < F; k += 64) {
      A acc = cv[s * F + k];
      Idx au = 0, ae = 0;
      if constexpr (ARG) {
        if constexpr (UL) au = p.carry_argu[s * F + k];
        if constexpr (UR) ae = p.carry_arge[s * F + k];
      }
      // (max / min: the winners' ids are loaded WITH the values, not after the compare — a dependent load per taken
      // carry put two to three memory latencies in series per row: this kernel took 0.163 ms against 0.054 ms for sum)
      auto combine = [&](A val, Idx cu, Idx ce) {
        if constexpr (RED == kSum) {
          acc += val;
        } else {
          const bool take = p.red_min ? (acc > val) : (acc < val);
          if (take) {
            acc = val;
            if constexpr (UL) au = cu;
            if constexpr (UR) ae = ce;
          }
        }
      };
      // the tail's operands: asked for now, used last
      const A tval = tv[s2 * F + k];
      Idx tu = 0, te = 0;
      if constexpr (ARG) {
        if constexpr (UL) tu = p.tail_argu[s2 * F + k];
        if constexpr (UR) te = p.tail_arge[s2 * F + k];
      }
      // same order as ever, but eight carries are in flight at a time
      int64_t q = s + 1;
      for (; q + 8 <= s2; q += 8) {
        A v[8];
        Idx vu[8], ve[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = cv[(q + u) * F + k];
          vu[u] = ve[u] = 0;
          if constexpr (ARG) {
            if constexpr (UL) vu[u] = p.carry_argu[(q + u) * F + k];
            if constexpr (UR) ve[u] = p.carry_arge[(q + u) * F + k];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) combine(v[u], vu[u], ve[u]);
      }
      for (; q < s2; ++q) {
        Idx cu = 0, ce = 0;
        if constexpr (ARG) {
          if constexpr (UL) cu = p.carry_argu[q * F + k];
          if constexpr (UR) ce = p.carry_arge[q * F + k];
        }
        combine(cv[q * F + k], cu, ce);
      }
      combine(tval, tu, te);
      const int64_t o = row * F + k;
      if (RED == kSum && p.mean) {
        const int64_t deg = static_cast<int64_t>(p.indptr[row + 1]) - static_cast<int64_t>(p.indptr[row]);
        acc = round_to_storage<DT>(acc) / round_to_storage<DT>(static_cast<A>(deg > 1 ? deg : 1));
      }
      if (p.accumulate)
        out[o] = from_acc<DT>(to_acc<DT>(out[o]) + acc);
      else
        out[o] = from_acc<DT>(acc);
      if constexpr (ARG) {
        if constexpr (UL) p.arg_u[o] = au;
        if constexpr (UR) p.arg_e[o] = ae;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Host side: workspace carving and launch.
// ---------------------------------------------------------------------------------------
// Lane groups per wave the in-kernel fix-up's counters cover (F >= 29 fp32 / 57 16-bit columns: 8+ lanes per row)
constexpr int kFuseMaxGroups = 8;

// In-kernel fix-up (round 4): for graphs of fewer than kFuseMaxWaves units, where a g-SpMM call is launch-bound
// and one launch fewer is worth more than what the tickets cost (mini-batch step, eager: 1.66 -> 1.54 ms).
// On the headline graph (121 k units) every wave pays two returning device-scope atomics and a drained store
// queue at its end: merge kernel 4.37 -> 4.51 ms against 0.053 ms for the separate kernel — measured, so large
// graphs keep the second launch.  DGLA_SPMM_FUSE_FIXUP=0 / 1 forces either (A/B and tests; read per call).
// LONG rows (>= 64 edges on average: a readout-like segment reduce, 15 M rows into 64 segments) keep the second
// launch too: a row that spans hundreds of units is combined by ONE wave walking all of its slots (segment sum
// 1.04 -> 1.23 ms, max 1.29 -> 2.24 ms with the in-kernel form; the fix-up kernel spreads that over a workgroup).
constexpr int64_t kFuseMaxWaves = 32768;
inline bool spmm_fuse_fixup_enabled(int64_t num_waves, int64_t rows, int64_t nnz) {
  const char* e = getenv("DGLA_SPMM_FUSE_FIXUP");
  if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  return num_waves < kFuseMaxWaves && nnz < 64 * rows;
}

struct SpmmGeometry {
  int vec;       // elements per lane access
  int log2_lpe;  // lanes per feature row
  int groups;    // lane groups per wave
  int chunks;    // grid.y: feature chunks of 64 * vec
  int wave_slots;  // one fix-up slot per wave (see SpmmParams::wave_slots)
  int wave_items;  // merge items per wavefront (see below)
  int64_t num_waves, num_slots;
  size_t off_plan, off_meta, off_fixcnt, off_carry_row, off_carry_val, off_tail_val, off_carry_argu,
      off_carry_arge, off_tail_argu, off_tail_arge, total;
  // split-row layout (0 = not used): bytes of a row kept in the main / tail array
  int split_main_bytes, split_tail_bytes;
  size_t off_split_main, off_split_tail;
  int split_edge_lines;  // > 0: edge layout (main = one 128-byte side line per row, k - 1 lines stay in place)
  bool split_straddle;   // straddle layout (8-byte lanes): main = line-aligned row slots of pitch split_main_bytes
};

// Half-width, in rows, of the window the locality probe counts as "local": 2 x 64 Ki rows of
// 400 bytes = 52 MB, a fifth of the 256 MiB Infinity Cache.
constexpr int64_t kSplitProbeWindowRows = 65536;

// Below this many 512-item units a graph is "small": its units are shortened (spmm_geometry).
constexpr int64_t kSmallGraphUnits = 4096;  // 16 per CU.  (DGLA_SMALL_GRAPH_UNITS overrides; measured at C3 size,
                                            // 5.2 k units: 16384 / 65536 make u_mul_e_sum 0.159 -> 0.168 / 0.184 ms)

// Smallest gathered operand the split layouts are used for (bytes; DGLA_SPLIT_MIN_MB overrides, for
// experiments and tests).  Measured on C2's rows with fewer distinct columns
// (DGLA_SPLIT_MIN_MB = 1 against 64): X = 61 MB 4.23 -> 3.92 ms (its 2.4 MB of tails then live in
// L2), X = 15 MB and 4 MB unchanged (everything is cache-resident either way) -> 16 MB.
inline int64_t spmm_split_min_bytes() {
  const char* e = getenv("DGLA_SPLIT_MIN_MB");
  const int64_t mb = e && atoll(e) > 0 ? atoll(e) : 16;
  return mb << 20;
}

// Shape-only eligibility of the split-row layout: 16-byte lane accesses cover the row in one
// chunk, the row is longer than one 128-byte line and not a whole number of lines.
inline bool spmm_split_shape_ok(const SpmmLaunch& L, size_t elem_bytes) {
  if (!(L.tune & kTuneSplit) || L.rel != nullptr || !op_uses_lhs(L.op)) return false;
  if (L.bcast == kBcGeneral || L.lhs_len != L.out_len) return false;
  const int64_t rb = L.lhs_len * static_cast<int64_t>(elem_bytes);
  if (rb % 8 || rb > 1024 || rb < 128 || rb % 128 == 0) return false;
  if (rb % 16) {
    // 8-byte-aligned rows (straddle layout): only worth it when rows can straddle an extra line at
    // all, i.e. the slack is smaller than the largest start offset (120)
    const int64_t slack = (rb + 127) / 128 * 128 - rb;
    if (slack >= 120) return false;
  } else if (rb < 256) {
    return false;  // edge layout: rows of two or more whole lines (a 144 .. 240-byte row touches 2 lines either way)
  }
  // pays off only when rows are re-read (average in-degree) and X does not fit the caches
  if (L.csr.nnz < 4 * L.csr.num_cols) return false;
  if (L.csr.num_cols * rb < spmm_split_min_bytes()) return false;
  return true;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline SpmmGeometry spmm_geometry(int64_t num_rows, int64_t nnz, int64_t out_len, int vec,
                                  size_t acc_bytes, int idbytes, bool with_arg,
                                  int64_t split_rows = 0, int64_t split_row_bytes = 0) {
  SpmmGeometry g;
  g.vec = vec;
  int64_t lanes = (out_len + vec - 1) / vec;
  if (lanes > 64) lanes = 64;
  int l2 = 0;
  while ((1 << l2) < lanes) ++l2;
  g.log2_lpe = l2;
  g.groups = 64 >> l2;
  g.chunks = static_cast<int>((out_len + 64 * vec - 1) / (64 * vec));
  // Unit size.  512 items keep 8 + 8 gathers in flight per lane over 32 batches — right when every
  // CU holds many waves.  A graph with fewer than ~16 units per CU (a sampled mini-batch block: 180 k
  // items = 350 units) leaves most CUs with ONE wave whose 32 dependent batches each wait a full
  // memory round trip (66 us for a 20 us job): such graphs get shorter units, down to 64 items.
  g.wave_items = kWaveItems;
  static const int64_t small_units = [] {
    const char* e = getenv("DGLA_SMALL_GRAPH_UNITS");
    return e && atoll(e) > 0 ? atoll(e) : kSmallGraphUnits;
  }();
  while (g.wave_items > 64 && (num_rows + nnz + g.wave_items - 1) / g.wave_items < small_units)
    g.wave_items >>= 1;
  g.num_waves = (num_rows + nnz + g.wave_items - 1) / g.wave_items;
  // sum reducer with two lane groups per wave (17 .. 32 lanes per feature row, e.g. F = 100 fp32):
  // one fix-up slot per wave instead of one per group
  g.wave_slots = (!with_arg && g.groups == 2 && g.chunks == 1) ? 1 : 0;
  g.num_slots = g.num_waves * (g.wave_slots ? 1 : g.groups);
  size_t off = 0;
  g.off_plan = off;
  off = align_up(off + sizeof(int64_t) * (g.num_waves + 1), 256);
  g.off_meta = off;  // locality probe counters {local, sampled}; lives and dies with the plan
  off += 256;
  // arrival counters of the in-kernel fix-up: one per slot for up to kFuseMaxGroups lane groups per wave; a
  // function of (rows, nnz) alone like the plan, zeroed with it, and left zero by every launch
  g.off_fixcnt = off;
  off = align_up(off + sizeof(unsigned) * g.num_waves * kFuseMaxGroups, 256);
  g.off_carry_row = off;
  off = align_up(off + sizeof(int64_t) * g.num_slots, 256);
  g.off_carry_val = off;
  off = align_up(off + acc_bytes * g.num_slots * out_len, 256);
  g.off_tail_val = off;
  off = align_up(off + acc_bytes * g.num_slots * out_len, 256);
  g.off_carry_argu = g.off_carry_arge = g.off_tail_argu = g.off_tail_arge = off;
  if (with_arg) {
    const size_t a = align_up(static_cast<size_t>(idbytes) * g.num_slots * out_len, 256);
    g.off_carry_argu = off;
    g.off_carry_arge = off + a;
    g.off_tail_argu = off + 2 * a;
    g.off_tail_arge = off + 3 * a;
    off += 4 * a;
  }
  g.split_main_bytes = g.split_tail_bytes = 0;
  g.off_split_main = g.off_split_tail = off;
  g.split_edge_lines = 0;
  g.split_straddle = false;
  if (split_rows > 0 && split_row_bytes % 16 != 0) {  // 8-byte-aligned rows: the straddle layout
    g.split_straddle = true;
    g.split_main_bytes = static_cast<int>((split_row_bytes + 127) / 128 * 128);
    g.off_split_main = g.off_split_tail = off;
    off = align_up(off + static_cast<size_t>(split_rows) * g.split_main_bytes, 256);
    g.off_split_tail = off;
  } else if (split_rows > 0) {  // edge layout (rows of two or more whole lines): copy the ragged ends only
    g.split_tail_bytes = static_cast<int>(split_row_bytes % 128);
    g.split_edge_lines = static_cast<int>(split_row_bytes / 128) - 1;
    g.split_main_bytes = 128;  // the side line of a row
    g.off_split_main = off;
    off = align_up(off + static_cast<size_t>(split_rows) * g.split_main_bytes, 256);
    g.off_split_tail = off;
    off = align_up(off + static_cast<size_t>(split_rows) * g.split_tail_bytes, 256);
  }
  g.total = off;
  return g;
}

template <typename Idx>
inline int launch_plan(const SpmmLaunch& L, const SpmmGeometry& g) {
  if (L.plan_valid) return 0;
  char* ws = static_cast<char*>(L.workspace);
  const int64_t n = g.num_waves + 1;
  const int threads = 256;
  const unsigned blocks = static_cast<unsigned>((n + threads - 1) / threads);
  hipLaunchKernelGGL(spmm_merge_plan_kernel<Idx>, dim3(blocks), dim3(threads), 0, L.stream,
                     static_cast<const Idx*>(L.csr.indptr), L.csr.num_rows, L.csr.nnz,
                     g.num_waves, reinterpret_cast<int64_t*>(ws + g.off_plan), g.wave_items);
  DGLA_CHECK_HIP(hipGetLastError());
  DGLA_CHECK_HIP(hipMemsetAsync(ws + g.off_fixcnt, 0, sizeof(unsigned) * g.num_waves * kFuseMaxGroups, L.stream));
  {
    // locality probe over ~4096 evenly spaced units, made with every plan (a later call on the
    // same workspace may be the first one whose shape is eligible for the split-row layout)
    // ... unless NO shape could ever be eligible on this graph (spmm_split_shape_ok's graph-side
    // conditions with the widest row, 1024 bytes): a sampled mini-batch block pays 35 us for the
    // probe's few thousand same-address atomics and can never use their answer
    const bool split_possible = L.csr.nnz >= 4 * L.csr.num_cols && L.csr.num_cols * 1024 >= spmm_split_min_bytes();
    unsigned* meta = reinterpret_cast<unsigned*>(ws + g.off_meta);
    if (split_possible) DGLA_CHECK_HIP(hipMemsetAsync(meta, 0, 8, L.stream));
    const int64_t stride = std::max<int64_t>(1, g.num_waves / 4096);
    const int64_t probes = (g.num_waves + stride - 1) / stride;
    const int64_t window = kSplitProbeWindowRows;
    if (split_possible && L.csr.indices != nullptr && L.csr.nnz > 0)
      hipLaunchKernelGGL(spmm_locality_probe_kernel<Idx>, dim3(static_cast<unsigned>(probes)),
                       dim3(64), 0, L.stream, static_cast<const Idx*>(L.csr.indices),
                       reinterpret_cast<const int64_t*>(ws + g.off_plan), L.csr.num_rows,
                       L.csr.num_cols, L.csr.nnz, g.num_waves, stride, window, meta, g.wave_items);
    DGLA_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

template <typename Idx, typename DT>
inline SpmmParams<Idx> make_params(const SpmmLaunch& L, const SpmmGeometry& g) {
  char* ws = static_cast<char*>(L.workspace);
  SpmmParams<Idx> p;
  p.indptr = static_cast<const Idx*>(L.csr.indptr);
  p.indices = static_cast<const Idx*>(L.csr.indices);
  p.eids = static_cast<const Idx*>(L.csr.eids);
  p.num_rows = L.csr.num_rows;
  p.nnz = L.csr.nnz;
  p.num_waves = g.num_waves;
  p.wave_items = g.wave_items;
  p.plan = reinterpret_cast<const int64_t*>(ws + g.off_plan);
  p.ufeat = L.ufeat;
  p.efeat = L.efeat;
  p.out = L.out;
  p.arg_u = static_cast<Idx*>(L.arg_u);
  p.arg_e = static_cast<Idx*>(L.arg_e);
  p.out_len = static_cast<int>(L.out_len);
  p.lhs_len = static_cast<int>(L.lhs_len);
  p.rhs_len = static_cast<int>(L.rhs_len);
  p.log2_lpe = g.log2_lpe;
  p.rhs_group = L.rhs_group > 0 ? L.rhs_group : 1;
  p.bd = L.bdims;
  p.accumulate = L.accumulate ? 1 : 0;
  p.rhs_neg = L.op == kSub ? 1 : 0;
  p.rhs_div = L.op == kDiv ? 1 : 0;
  p.rhs_mask = L.rhs_mask ? 1 : 0;
  p.stage_w = 0;   // (launch_spmm_one decides)
  p.red_min = L.red == kMin ? 1 : 0;
  p.arg_empty = L.arg_empty;
  p.mean = L.mean ? 1 : 0;
  p.tune = L.tune;
  p.umain = p.utail = nullptr;
  p.split_meta = nullptr;
  p.split_main = p.split_tail = 0;
  p.split_edge_lines = p.split_t16 = p.split_base16 = 0;
  p.split_straddle_slack = -1;
  p.split_row_bytes = p.split_base_bytes = 0;
  if (g.split_straddle) {
    p.umain = ws + g.off_split_main;
    if (!(L.tune & kTuneSplitForce) && !L.split_keep)
      p.split_meta = reinterpret_cast<const unsigned*>(ws + g.off_meta);
    p.split_main = g.split_main_bytes / static_cast<int>(sizeof(DT));
    p.split_row_bytes = static_cast<int>(L.lhs_len * static_cast<int64_t>(sizeof(DT)));
    p.split_straddle_slack = g.split_main_bytes - p.split_row_bytes;
    p.split_base_bytes = static_cast<int>(reinterpret_cast<uintptr_t>(L.ufeat) & 127u);
  } else if (g.split_main_bytes > 0) {
    p.umain = ws + g.off_split_main;
    p.utail = ws + g.off_split_tail;
    if (!(L.tune & kTuneSplitForce) && !L.split_keep)
      p.split_meta = reinterpret_cast<const unsigned*>(ws + g.off_meta);
    p.split_main = g.split_main_bytes / static_cast<int>(sizeof(DT));
    p.split_tail = g.split_tail_bytes / static_cast<int>(sizeof(DT));
    p.split_edge_lines = g.split_edge_lines;
    p.split_t16 = g.split_tail_bytes / 16;
    p.split_base16 = static_cast<int>((reinterpret_cast<uintptr_t>(L.ufeat) & 127u) >> 4);
  }
  p.rel = static_cast<const uint8_t*>(L.rel);
  p.xtab = L.ufeat_tab;
  p.wtab = L.efeat_tab;
  p.num_rel = L.num_rel;
  p.wave_slots = g.wave_slots;
  p.fix_count = (g.chunks == 1 && (g.wave_slots || g.groups <= kFuseMaxGroups) && spmm_fuse_fixup_enabled(g.num_waves, L.csr.num_rows, L.csr.nnz))
                    ? reinterpret_cast<unsigned*>(ws + g.off_fixcnt)
                    : nullptr;
  p.carry_row = reinterpret_cast<int64_t*>(ws + g.off_carry_row);
  p.carry_val = ws + g.off_carry_val;
  p.tail_val = ws + g.off_tail_val;
  p.carry_argu = reinterpret_cast<Idx*>(ws + g.off_carry_argu);
  p.carry_arge = reinterpret_cast<Idx*>(ws + g.off_carry_arge);
  p.tail_argu = reinterpret_cast<Idx*>(ws + g.off_tail_argu);
  p.tail_arge = reinterpret_cast<Idx*>(ws + g.off_tail_arge);
  return p;
}

// Gathers in flight per lane (x2 with the prefetched batch).
constexpr int kSpmmUnroll = 4;

// Non-temporal edge-operand stream: copy_rhs without an edge-id map and with LONG rows (>= 64 edges on average, e.g. a graph
// readout).  Interleaved A/B (profiles/r3/nt_stream_ab.jsonl, 15.5 M rows x 400 B): 64 segments sum -8 %,
// max -5 %; 612 k segments (25 rows each) sum -2 %, max +9 %; 2.4 M segments (6 rows each) +7 % / +10 %:
// next to many output rows the non-temporal stream loses.
inline bool spmm_nt_stream(const SpmmLaunch& L) {
  return L.csr.eids == nullptr && L.csr.nnz >= 64 * L.csr.num_rows;
}

template <typename Idx, typename DT, int VEC, int OP, int RED, int BC>
inline int launch_spmm_one(const SpmmLaunch& L, const SpmmGeometry& g) {
  SpmmParams<Idx> p = make_params<Idx, DT>(L, g);
  const unsigned blocks =
      static_cast<unsigned>((g.num_waves + kWavesPerBlock - 1) / kWavesPerBlock);
  if (g.split_straddle && !L.split_valid) {
    char* ws = static_cast<char*>(L.workspace);
    const int row_pieces = p.split_row_bytes / 8;
    const int64_t total = L.csr.num_cols * row_pieces;
    const unsigned sblocks = static_cast<unsigned>(std::min<int64_t>((total + 255) / 256, int64_t(1) << 20));
    hipLaunchKernelGGL(spmm_straddle_rows_kernel<0>, dim3(sblocks), dim3(256), 0, L.stream,
                       static_cast<const piece8_t*>(L.ufeat), reinterpret_cast<piece8_t*>(ws + g.off_split_main),
                       L.csr.num_cols, row_pieces, g.split_main_bytes / 8, p.split_straddle_slack,
                       static_cast<unsigned>(p.split_base_bytes), p.split_meta);
    DGLA_CHECK_HIP(hipGetLastError());
  } else if (g.split_edge_lines > 0 && !L.split_valid) {
    char* ws = static_cast<char*>(L.workspace);
    const int t16 = g.split_tail_bytes / 16, side = 8 + t16;
    const int row_pieces = static_cast<int>(L.lhs_len * static_cast<int64_t>(sizeof(DT)) / 16);
    const int64_t total = L.csr.num_cols * side;
    constexpr int K = 4;
    const unsigned sblocks =
        static_cast<unsigned>(std::min<int64_t>((total + 256 * K - 1) / (256 * K), int64_t(1) << 20));
    const unsigned magic = 0xFFFFFFFFu / static_cast<unsigned>(side) + 1u;
    hipLaunchKernelGGL(spmm_split_edges_kernel<K>, dim3(sblocks), dim3(256), 0, L.stream,
                       static_cast<const piece16_t*>(L.ufeat),
                       reinterpret_cast<piece16_t*>(ws + g.off_split_main),
                       reinterpret_cast<piece16_t*>(ws + g.off_split_tail), L.csr.num_cols, row_pieces,
                       t16, 8 * g.split_edge_lines, magic, p.split_meta,
                       static_cast<unsigned>(p.split_base16));
    DGLA_CHECK_HIP(hipGetLastError());
  }
  if (L.prepare_only) return 0;  // the producer's half of the call: plan + side copy are in the workspace
  // dynamic LDS: the unit's edge ids, only when the operator reads edge features through a map
  unsigned dyn_lds = (op_uses_rhs(OP) && L.csr.eids != nullptr)
                         ? static_cast<unsigned>(kWavesPerBlock * kWaveItems * sizeof(Idx))
                         : 0u;
  if constexpr (op_uses_lhs(OP) && op_uses_rhs(OP) && BC == kBcRhsGroup && RED == kSum && sizeof(DT) == 4) {
    // scalar edge weights (one 4-byte value per edge), plain sum: staged in LDS with the unit's column ids
    if (L.rel == nullptr && L.rhs_len == 1 && !L.rhs_mask && L.efeat != nullptr && (L.tune & kTuneNoStageW) == 0) {
      p.stage_w = 1;
      dyn_lds = std::max<unsigned>(dyn_lds, static_cast<unsigned>(kWavesPerBlock * kWaveItems * sizeof(uint32_t)));
    }
  }
  const ProfileEvents pe = profile_events();
  if (pe.before) DGLA_CHECK_HIP(hipEventRecord(pe.before, L.stream));
  if (L.rel != nullptr) {
    // stacked multi-relation launch: the operator subset the fused hetero path uses
    if constexpr ((OP == kCopyLhs || OP == kCopyRhs || OP == kMul) && BC != kBcGeneral) {
      if (L.op == kDiv) {
        last_error() = "stacked SpMM supports copy_lhs / copy_rhs / mul only";
        return -1;
      }
      // (max / min over the stacked edges: earlier relations win ties, like the reference's running
      // compare relation by relation, spmm.cuh:552-606; arg_e receives stacked positions)
      hipLaunchKernelGGL((spmm_csr_merge_kernel<Idx, DT, VEC, OP, RED, BC, kSpmmUnroll, true>),
                         dim3(blocks, g.chunks), dim3(64 * kWavesPerBlock), dyn_lds, L.stream, p);
      if constexpr (RED != kSum) {
        DGLA_CHECK_HIP(hipGetLastError());
        if (pe.after) DGLA_CHECK_HIP(hipEventRecord(pe.after, L.stream));
        if (p.fix_count) return 0;  // straddling rows were finished inside the launch
        hipLaunchKernelGGL((spmm_csr_fixup_kernel<Idx, DT, OP, RED, true>),
                           dim3(static_cast<unsigned>(std::min<int64_t>(g.num_slots, int64_t(1) << 24))),
                           dim3(64), 0, L.stream, p, g.num_slots);
        DGLA_CHECK_HIP(hipGetLastError());
        return 0;
      }
    } else {
      last_error() = "stacked SpMM supports copy_lhs / copy_rhs / mul only";
      return -1;
    }
  } else if (OP == kCopyRhs && BC == kBcNone && !L.accumulate && spmm_nt_stream(L)) {
    if constexpr (OP == kCopyRhs && BC == kBcNone)
      hipLaunchKernelGGL((spmm_csr_merge_kernel<Idx, DT, VEC, OP, RED, BC, kSpmmUnroll, false, true>),
                         dim3(blocks, g.chunks), dim3(64 * kWavesPerBlock), dyn_lds, L.stream, p);
  } else {
    hipLaunchKernelGGL((spmm_csr_merge_kernel<Idx, DT, VEC, OP, RED, BC, kSpmmUnroll, false>),
                       dim3(blocks, g.chunks), dim3(64 * kWavesPerBlock), dyn_lds, L.stream, p);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  if (pe.after) DGLA_CHECK_HIP(hipEventRecord(pe.after, L.stream));
  if (p.fix_count) return 0;  // straddling rows were finished inside the launch
  hipLaunchKernelGGL((spmm_csr_fixup_kernel<Idx, DT, OP, RED>),
                     dim3(static_cast<unsigned>(std::min<int64_t>(g.num_slots, int64_t(1) << 24))),
                     dim3(64), 0, L.stream, p, g.num_slots);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx, typename DT, int VEC, int OP, int RED>
inline int launch_spmm_bc(const SpmmLaunch& L, const SpmmGeometry& g) {
  if constexpr (!op_uses_rhs(OP) || OP == kCopyRhs) {
    return launch_spmm_one<Idx, DT, VEC, OP, RED, kBcNone>(L, g);
  } else {
    switch (L.bcast) {
      case kBcNone: return launch_spmm_one<Idx, DT, VEC, OP, RED, kBcNone>(L, g);
      case kBcRhsGroup: return launch_spmm_one<Idx, DT, VEC, OP, RED, kBcRhsGroup>(L, g);
      default:
        if constexpr (VEC == 1) return launch_spmm_one<Idx, DT, 1, OP, RED, kBcGeneral>(L, g);
    }
    last_error() = "internal: general broadcast needs the scalar path";
    return -1;
  }
}

template <typename Idx, typename DT, int VEC, int OP>
inline int launch_spmm_red(const SpmmLaunch& L, const SpmmGeometry& g) {
  switch (L.red) {
    case kSum: return launch_spmm_bc<Idx, DT, VEC, OP, kSum>(L, g);
    case kMax:
    case kMin: return launch_spmm_bc<Idx, DT, VEC, OP, kMax>(L, g);  // min: p.red_min
  }
  last_error() = "unsupported SpMM reducer";
  return -1;
}

template <typename Idx, typename DT, int VEC>
inline int launch_spmm_op(const SpmmLaunch& L, const SpmmGeometry& g) {
  switch (L.op) {
    case kAdd:
    case kSub: return launch_spmm_red<Idx, DT, VEC, kAdd>(L, g);  // sub: p.rhs_neg
    case kMul:
    case kDiv: return launch_spmm_red<Idx, DT, VEC, kMul>(L, g);  // div: p.rhs_div
    case kCopyLhs: return launch_spmm_red<Idx, DT, VEC, kCopyLhs>(L, g);
    case kCopyRhs: return launch_spmm_red<Idx, DT, VEC, kCopyRhs>(L, g);
  }
  last_error() = "unsupported SpMM binary operator";
  return -1;
}

// Chooses the access width: 16-byte lane accesses when every row start is 16-byte aligned
// and the operand layout allows it, else 8-byte accesses under the same test (bf16 / fp16
// rows of F % 4 == 0 elements such as F = 100, fp32 rows of even length), else element-wise.
template <typename DT>
inline bool spmm_vec_fits(const SpmmLaunch& L, int vec) {
  const uintptr_t mask = sizeof(DT) * vec - 1;
  auto aligned = [mask](const void* p) { return (reinterpret_cast<uintptr_t>(p) & mask) == 0; };
  if (L.out_len % vec) return false;
  if (!aligned(L.out)) return false;
  if (op_uses_lhs(L.op) && (L.lhs_len % vec || !aligned(L.ufeat))) return false;
  if (op_uses_rhs(L.op) && L.bcast == kBcNone && (L.rhs_len % vec || !aligned(L.efeat))) return false;
  if (L.bcast == kBcRhsGroup && L.rhs_group % vec) return false;
  return true;
}

template <typename DT>
inline int spmm_pick_vec(const SpmmLaunch& L) {
  constexpr int full = 16 / sizeof(DT);
  constexpr int half = full / 2;
  if (L.bcast == kBcGeneral) return 1;
  if (spmm_vec_fits<DT>(L, full)) return full;
  if (half > 1 && spmm_vec_fits<DT>(L, half)) return half;
  return 1;
}

template <typename Idx, typename DT>
inline int launch_spmm_vec(const SpmmLaunch& L, const SpmmGeometry& g, int vec) {
  constexpr int full = 16 / sizeof(DT);
  constexpr int half = full / 2;
  if (vec == full) return launch_spmm_op<Idx, DT, full>(L, g);
  if constexpr (half > 1) {
    if (vec == half) return launch_spmm_op<Idx, DT, half>(L, g);
  }
  return launch_spmm_op<Idx, DT, 1>(L, g);
}

template <typename DT>
inline int launch_spmm_csr_typed(const SpmmLaunch& L) {
  using A = typename Acc<DT>::type;
  const int vec = spmm_pick_vec<DT>(L);
  SpmmGeometry g = spmm_geometry(L.csr.num_rows, L.csr.nnz, L.out_len, vec, sizeof(A),
                                 L.csr.idbits / 8, L.red != kSum);
  const int64_t rb_l = L.lhs_len * static_cast<int64_t>(sizeof(DT));
  const bool vec_ok = rb_l % 16 == 0 ? vec * sizeof(DT) == 16 : vec * sizeof(DT) == 8;
  if (vec_ok && g.chunks == 1 && spmm_split_shape_ok(L, sizeof(DT))) {
    // same carve-up plus the two re-laid-out copies of X; used only if the caller's workspace
    // has room (dgla_spmm_csr_workspace_bytes accounts for it), else the plain layout runs
    const SpmmGeometry gs = spmm_geometry(L.csr.num_rows, L.csr.nnz, L.out_len, vec, sizeof(A),
                                          L.csr.idbits / 8, L.red != kSum, L.csr.num_cols,
                                          L.lhs_len * static_cast<int64_t>(sizeof(DT)));
    if (L.workspace && L.workspace_bytes >= gs.total) g = gs;
  }
  if (L.workspace_bytes < g.total || (g.total && !L.workspace)) {
    last_error() = "SpMM workspace too small: need " + std::to_string(g.total) + " bytes";
    return -1;
  }
  if (L.csr.idbits == 32) {
    if (launch_plan<int32_t>(L, g)) return -1;
    return launch_spmm_vec<int32_t, DT>(L, g, vec);
  } else {
    if (launch_plan<int64_t>(L, g)) return -1;
    return launch_spmm_vec<int64_t, DT>(L, g, vec);
  }
}

// Upper bound over all access widths, so the answer does not depend on pointer alignment
// and one allocation serves every later call on this (graph, feature width).
template <typename DT>
inline size_t spmm_csr_workspace_typed(const SpmmLaunch& L) {
  using A = typename Acc<DT>::type;
  constexpr int full = 16 / sizeof(DT);
  size_t best = 0;
  const bool split = spmm_split_shape_ok(L, sizeof(DT));
  for (int vec : {1, full / 2 > 1 ? full / 2 : 1, full}) {
    const size_t t = spmm_geometry(L.csr.num_rows, L.csr.nnz, L.out_len, vec, sizeof(A),
                                   L.csr.idbits / 8, L.red != kSum, split ? L.csr.num_cols : 0,
                                   L.lhs_len * static_cast<int64_t>(sizeof(DT)))
                         .total;
    if (t > best) best = t;
  }
  return best;
}

}  // namespace dgla
