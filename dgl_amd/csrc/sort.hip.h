// Bounded-key stable sort and device-wide exclusive scan for gfx950 — the library's own, used by the COO -> CSR
// conversion (coo2csr.hip) and the sampling pipeline (sampling.hip) in place of a vendor radix sort.
//
// Reference: COOSort (src/array/cuda/coo_sort.cu: a radix sort of the row keys carrying the edge permutation) +
// cusparseXcoo2csr (src/array/cuda/coo2csr.cu:28-110).  Keys here are node ids bounded by the node count (22 bits at
// ogbn-products size, 27 at papers100M), which a general 32/64-bit radix sort does not exploit.
//
// Algorithm: most-significant-digit bucket sort, nl = ceil(key_bits / 9) levels of <= 9-bit digits, stable.
//   * an element travels as ONE 64-bit word  [remaining key bits | column id | position]  — each level SHEDS the digit it
//     has sorted by (the bucket implies it), so 22 + 22 + 26 bits never need more than 64 (128-bit elements when they do);
//   * a level is three launches over ITEMS (pieces of <= kItemLen elements of one bucket, one wavefront each, so a hub
//     bucket is spread over many waves):  histogram of the level's digit per item -> exclusive scan (bucket-major,
//     item-minor: stable) -> scatter;
//   * the scatter ranks a tile of 1 024 elements exactly and stably with LDS match masks: per 64-element row every lane
//     ORs its lane bit into mask[digit] (one ds_or_b64), reads the word back, and rank = popcount(mask & lanes below) —
//     one LDS round trip per row, no per-bit ballots, no data-dependent loop; the tile is then staged in LDS in bucket
//     order and written out as runs (consecutive lanes -> consecutive addresses of one bucket);
//   * the LAST level's bins are the rows themselves: its scan writes indptr, its scatter unpacks (column, position)
//     into indices / edge ids — no separate compress pass, no sorted-key array is ever written.
// Traffic at C2 size (3 levels): 4 + 16 (level 1) + 24 + 24 bytes per edge; measured 1.64 ms at 61.9 M edges (int32), see
// profiles/r6/coo2csr_own_sort.jsonl for the kernel table and the variants that were measured and dropped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace dgla {
namespace msd {

constexpr int kTileRows = 16;               // 64-element rows per tile
constexpr int kTile = 64 * kTileRows;       // 1 024 elements staged in LDS at a time
constexpr int kItemLen = 8192;              // elements per item (one wavefront) of a large input: a multiple of kTile
constexpr int64_t kSmallInput = 1 << 19;    // inputs up to this size use one-tile items (more wavefronts, fewer tiles each)
inline int item_len_for(int64_t n) { return n <= kSmallInput ? kTile : kItemLen; }
#ifndef DGLA_SORT_MAX_DIGIT
#define DGLA_SORT_MAX_DIGIT 9
#endif
constexpr int kMaxDigit = DGLA_SORT_MAX_DIGIT;   // bits per level of the COO -> CSR sort
constexpr int kMaxNB = 1 << kMaxDigit;           // bins per level
constexpr int kKeysDigit = 9;                    // bits per pass of the keys-only sort (small inputs: fused one-block scan)
constexpr int kMaxLevels = 8;

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

inline int bits_for(int64_t n) {  // bits needed for values in [0, n)
  int b = 1;
  while ((int64_t(1) << b) < n) ++b;
  return b;
}

// ------------------------------------------------------------------------------------------------------------------
// device-wide exclusive scan (int32 / int64), three launches: per-block sums, scan of the sums, per-block scan + offset
// ------------------------------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256, kScanPer = 16, kScanChunk = kScanBlock * kScanPer;

// Inclusive sum over the 64 lanes.  32-bit values: six DPP-modified adds — row_shr:1/2/4/8 inside a row of 16 lanes
// (0x111 .. 0x118; lanes without a source take the 0 passed as `old`), row_bcast:15 into rows 1 and 3 (0x142, row mask
// 0xa), row_bcast:31 into rows 2 and 3 (0x143, 0xc) — instead of six ds_bpermute round trips (csrc/narrow_reduce.hip uses
// the same moves); wider values go through shuffles.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_from_zero(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_scan32(int v) {
  v += dpp_from_zero<0x111, 0xf>(v);
  v += dpp_from_zero<0x112, 0xf>(v);
  v += dpp_from_zero<0x114, 0xf>(v);
  v += dpp_from_zero<0x118, 0xf>(v);
  v += dpp_from_zero<0x142, 0xa>(v);
  v += dpp_from_zero<0x143, 0xc>(v);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
  if constexpr (sizeof(T) == 4) {
    return static_cast<T>(wave_inclusive_scan32(static_cast<int>(v)));
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const T o = __shfl_up(v, d, 64);
      if ((threadIdx.x & 63) >= d) v += o;
    }
    return v;
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(const TI* __restrict__ in, int64_t n, TO* __restrict__ sums) {
  __shared__ TO part[kScanBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk;
  TO s = 0;
  for (int k = 0; k < kScanPer; ++k) {
    const int64_t i = base + k * kScanBlock + threadIdx.x;
    if (i < n) s += static_cast<TO>(in[i]);
  }
  s = wave_inclusive_scan(s);
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    TO t = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) t += part[w];
    sums[blockIdx.x] = t;
  }
}

template <typename TO>
__global__ __launch_bounds__(1024) void scan_sums_kernel(TO* __restrict__ sums, int64_t nb) {
  // one block: exclusive scan of nb block sums in place (nb is small: n / 4096)
  __shared__ TO wsum[16];
  __shared__ TO carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const TO v = i < nb ? sums[i] : TO(0);
    const TO incl = wave_inclusive_scan(v);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    TO off = carry_s;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < nb) sums[i] = off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + incl;
    __syncthreads();
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const TI* __restrict__ in, int64_t n, const TO* __restrict__ sums,
                                                               TO* __restrict__ out) {
  // thread t owns kScanPer CONSECUTIVE elements (in and out may alias: every element is read before it is written)
  __shared__ TO part[kScanBlock / 64];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * kScanPer;
  TO v[kScanPer];
  TO s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    v[k] = base + k < n ? static_cast<TO>(in[base + k]) : TO(0);
    s += v[k];
  }
  const TO incl = wave_inclusive_scan(s);
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
  __syncthreads();
  TO off = sums[blockIdx.x] + incl - s;
  for (int w = 0; w < (threadIdx.x >> 6); ++w) off += part[w];
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) {
    if (base + k < n) out[base + k] = off;
    off += v[k];
  }
}

// n <= kScanSmall: one workgroup, one launch (thread t owns a contiguous piece)
constexpr int kScanSmall = 32768;
template <typename TI, typename TO>
__global__ __launch_bounds__(1024) void scan_small_kernel(const TI* __restrict__ in, int n, TO* __restrict__ out) {
  // round k: thread t owns element k * 1024 + t (coalesced); a block scan per round, the running total carried in LDS
  __shared__ TO wsum[16];
  __shared__ TO carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const TO v = i < n ? static_cast<TO>(in[i]) : TO(0);
    const TO incl = wave_inclusive_scan(v);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    TO off = carry_s;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < n) out[i] = off + incl - v;   // (in == out allowed: element i is read before it is written, by the same thread)
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + incl;
    __syncthreads();
  }
}

inline size_t scan_temp_bytes(int64_t n, size_t elem) { return align256(((n + kScanChunk - 1) / kScanChunk + 1) * elem); }

// out[i] = sum of in[0 .. i) for i in [0, n); in == out allowed.  temp: scan_temp_bytes(n, sizeof(TO)).
template <typename TI, typename TO>
int exclusive_scan(const TI* in, TO* out, int64_t n, void* temp, hipStream_t s) {
  if (n <= 0) return 0;
  if (n <= kScanSmall) {
    hipLaunchKernelGGL((scan_small_kernel<TI, TO>), dim3(1), dim3(1024), 0, s, in, static_cast<int>(n), out);
    DGLA_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const int64_t nb = (n + kScanChunk - 1) / kScanChunk;
  TO* sums = static_cast<TO*>(temp);
  hipLaunchKernelGGL((scan_block_sums_kernel<TI, TO>), dim3(static_cast<unsigned>(nb)), dim3(kScanBlock), 0, s, in, n, sums);
  hipLaunchKernelGGL((scan_sums_kernel<TO>), dim3(1), dim3(1024), 0, s, sums, nb);
  hipLaunchKernelGGL((scan_apply_kernel<TI, TO>), dim3(static_cast<unsigned>(nb)), dim3(kScanBlock), 0, s, in, n, sums, out);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// elements
// ------------------------------------------------------------------------------------------------------------------
struct Wide {  // 128-bit element for graphs whose key + column + position bits exceed 64
  uint64_t k;  // remaining key bits
  uint64_t v;  // column << pb | position
};

struct Fmt {
  int cb, pb;  // column bits, position bits (0 / 0 for a keys-only sort)
};

template <bool WIDE>
struct El;
template <>
struct El<false> {
  using T = uint64_t;
  static __device__ __forceinline__ T make(uint64_t key, uint64_t col, uint64_t pos, const Fmt& f) {
    return (f.cb + f.pb < 64 ? key << (f.cb + f.pb) : 0) | (col << f.pb) | pos;
  }
  static __device__ __forceinline__ uint32_t digit(T x, int kb, int b, const Fmt& f) {
    return static_cast<uint32_t>(x >> (kb - b + f.cb + f.pb)) & ((1u << b) - 1u);
  }
  static __device__ __forceinline__ T shed(T x, int kb, int b, const Fmt& f) {
    const int keep = kb - b + f.cb + f.pb;
    return keep >= 64 ? x : x & ((uint64_t(1) << keep) - 1);
  }
  static __device__ __forceinline__ uint64_t col(T x, const Fmt& f) { return (x >> f.pb) & ((uint64_t(1) << f.cb) - 1); }
  static __device__ __forceinline__ uint64_t pos(T x, const Fmt& f) { return x & ((uint64_t(1) << f.pb) - 1); }
};
template <>
struct El<true> {
  using T = Wide;
  static __device__ __forceinline__ T make(uint64_t key, uint64_t col, uint64_t pos, const Fmt& f) {
    return Wide{key, (col << f.pb) | pos};
  }
  static __device__ __forceinline__ uint32_t digit(const T& x, int kb, int b, const Fmt&) {
    return static_cast<uint32_t>(x.k >> (kb - b)) & ((1u << b) - 1u);
  }
  static __device__ __forceinline__ T shed(const T& x, int kb, int b, const Fmt&) {
    return Wide{x.k & ((uint64_t(1) << (kb - b)) - 1), x.v};
  }
  static __device__ __forceinline__ uint64_t col(const T& x, const Fmt& f) { return f.pb >= 64 ? 0 : (x.v >> f.pb); }
  static __device__ __forceinline__ uint64_t pos(const T& x, const Fmt& f) {
    return f.pb >= 64 ? x.v : (x.v & ((uint64_t(1) << f.pb) - 1));
  }
};

struct Item {
  int64_t start;  // first element (position in the level's input array)
  int32_t len;
  int32_t seg;    // bucket of the previous levels this piece belongs to
};

enum InMode : int { kInCoo = 0, kInKeys = 1, kInElem = 2 };   // kInKeys: bare keys, travelling whole (the LSD keys-only sort)
enum OutMode : int { kOutElem = 0, kOutCsr = 1, kOutKeys = 2 };

template <typename Idx>
struct LevelArgs {
  // input of the level
  const Idx* row;      // kInCoo / kInKeys: the keys
  const Idx* col;      // kInCoo
  const void* src;     // kInElem: elements
  // output
  void* dst;           // kOutElem
  Idx* indices;        // kOutCsr
  Idx* eids_out;       // kOutCsr
  const Idx* eids_in;  // kOutCsr: optional edge ids of the COO (gathered through the position)
  Idx* keys_out;       // kOutKeys
  Idx* indptr;         // last level (kOutCsr): written by the scan
  int64_t num_rows;
  // geometry
  int64_t nnz;
  int kb;              // key bits still in the input elements (this level's digit included)
  int b;               // digit bits of this level
  int level;           // 0-based
  int shift;           // kInCoo / kInKeys: the digit is (key >> shift) & (2^b - 1)
  int item_len;        // elements per item (kTile or kItemLen)
  Fmt fmt;
  // items
  const Item* items;   // level > 0
  const int32_t* n_items_dev;  // level > 0: number of items (device)
  int64_t n_items_host;        // level 0: number of items
  int64_t num_segs;            // buckets entering this level
  const int64_t* seg;          // [num_segs + 1] bucket starts entering this level (level 0: null)
  int64_t* seg_next;           // [num_segs * NB + 1] bucket starts leaving this level (not for the last level)
  const int32_t* item_base;    // level > 0: [num_segs + 1] first item of every bucket
  uint64_t* counts;            // level 0: [NB][n_items] (bin-major); level > 0: [n_items][NB]; counts, then prefixes
  int64_t* bin_total;          // level 0: [NB]
};

// What one element of the level's input is in memory: the key (+ column) of the COO, or a packed word.  Loading
// (unconditional, so that a tile's loads are all in flight together) is kept apart from splitting.
template <bool WIDE, int IN, typename Idx>
struct Raw {
  typename El<WIDE>::T x;
};
template <bool WIDE, typename Idx>
struct Raw<WIDE, kInCoo, Idx> {
  Idx row, col;
};
template <bool WIDE, typename Idx>
struct Raw<WIDE, kInKeys, Idx> {
  Idx row;
};

template <bool WIDE, int IN, typename Idx>
__device__ __forceinline__ Raw<WIDE, IN, Idx> load_raw(const LevelArgs<Idx>& a, int64_t i) {
  Raw<WIDE, IN, Idx> r;
  if constexpr (IN == kInElem) {
    r.x = static_cast<const typename El<WIDE>::T*>(a.src)[i];
  } else {
    r.row = a.row[i];
    if constexpr (IN == kInCoo) r.col = a.col[i];
  }
  return r;
}

// this level's digit split off: *d = the digit, return = the element WITHOUT it (a word never holds more than
// key_bits - b[0] key bits: level 0 takes the digit straight from the key)
template <bool WIDE, int IN, typename Idx>
__device__ __forceinline__ typename El<WIDE>::T split(const LevelArgs<Idx>& a, const Raw<WIDE, IN, Idx>& r, int64_t i, uint32_t* d) {
  if constexpr (IN == kInElem) {
    *d = El<WIDE>::digit(r.x, a.kb, a.b, a.fmt);
    return El<WIDE>::shed(r.x, a.kb, a.b, a.fmt);
  } else {
    const uint64_t key = static_cast<uint64_t>(r.row);
    *d = static_cast<uint32_t>(key >> a.shift) & ((1u << a.b) - 1u);
    if constexpr (IN == kInCoo)   // (most significant digit first: shift = the bits that stay)
      return El<WIDE>::make(key & ((uint64_t(1) << a.shift) - 1), static_cast<uint64_t>(r.col), static_cast<uint64_t>(i), a.fmt);
    else
      return El<WIDE>::make(key, 0, 0, Fmt{0, 0});
  }
}

template <typename Idx>
__device__ __forceinline__ bool get_item(const LevelArgs<Idx>& a, int64_t it, Item* out) {
  if (a.level == 0) {
    if (it >= a.n_items_host) return false;
    out->start = it * a.item_len;
    const int64_t rest = a.nnz - out->start;
    out->len = static_cast<int32_t>(rest < a.item_len ? rest : a.item_len);
    out->seg = 0;
    return true;
  }
  if (it >= *a.n_items_dev) return false;
  *out = a.items[it];
  return true;
}

// ---- histogram of the level's digit, one wavefront per item ------------------------------------------------------------
template <bool WIDE, int IN, typename Idx>
__global__ __launch_bounds__(64) void msd_hist_kernel(const LevelArgs<Idx> a) {
  __shared__ uint32_t hist[kMaxNB];
  Item it;
  if (!get_item(a, blockIdx.x, &it)) return;
  const int lane = threadIdx.x;
  const int nb = 1 << a.b;
  for (int d = lane; d < nb; d += 64) hist[d] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < it.len; i0 += 512) {   // 8 rows of loads in flight, then the LDS adds
    uint32_t dv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 64 + lane;
      const int64_t j = it.start + (i < it.len ? i : it.len - 1);
      if constexpr (IN == kInElem)
        dv[k] = El<WIDE>::digit(static_cast<const typename El<WIDE>::T*>(a.src)[j], a.kb, a.b, a.fmt);
      else
        dv[k] = static_cast<uint32_t>(static_cast<uint64_t>(a.row[j]) >> a.shift) & (nb - 1);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i0 + k * 64 + lane < it.len) atomicAdd(&hist[dv[k]], 1u);
  }
  __syncthreads();
  if (a.level == 0) {
    for (int d = lane; d < nb; d += 64) a.counts[static_cast<int64_t>(d) * a.n_items_host + blockIdx.x] = hist[d];
  } else {
    for (int d = lane; d < nb; d += 64) a.counts[static_cast<int64_t>(blockIdx.x) * nb + d] = hist[d];
  }
}

// ---- level 0 scan: one wavefront per bin over the items, then one wavefront over the bins ------------------------------
template <typename Idx>
__global__ __launch_bounds__(64) void msd_scan0_bins_kernel(const LevelArgs<Idx> a) {
  const int d = blockIdx.x, lane = threadIdx.x;
  uint64_t* c = a.counts + static_cast<int64_t>(d) * a.n_items_host;
  int64_t run = 0;
  for (int64_t base = 0; base < a.n_items_host; base += 64) {
    const int64_t i = base + lane;
    const uint32_t v = i < a.n_items_host ? static_cast<uint32_t>(c[i]) : 0u;   // (an item has <= kItemLen elements)
    const uint32_t incl = wave_inclusive_scan(v);
    if (i < a.n_items_host) c[i] = static_cast<uint64_t>(run) + incl - v;
    run += __shfl(incl, 63, 64);
  }
  if (lane == 0) a.bin_total[d] = run;
}

// bucket starts leaving the level (or indptr on the last level) from per-bin totals; one wavefront.
// `base`: start of the bucket being split; bins of the LAST level are rows (row = (seg << b) | bin).
template <typename Idx>
__device__ __forceinline__ void write_bin_starts(const LevelArgs<Idx>& a, int64_t seg, int64_t base, const int64_t* tot /*[rounds]*/,
                                                 int lane, bool last_level) {
  const int nb = 1 << a.b;
  int64_t carry = base;
  for (int q = 0; q * 64 < nb; ++q) {
    const int d = q * 64 + lane;
    const int64_t v = d < nb ? tot[q] : 0;
    const int64_t incl = wave_inclusive_scan(v);
    const int64_t excl = carry + incl - v;
    if (d < nb) {
      if (last_level) {
        const int64_t r = (seg << a.b) | d;
        if (a.indptr && r <= a.num_rows) a.indptr[r] = static_cast<Idx>(excl);
      }
      if (a.seg_next) a.seg_next[seg * nb + d] = excl;
    }
    carry += __shfl(incl, 63, 64);
  }
}

template <typename Idx>
__global__ __launch_bounds__(64) void msd_scan0_top_kernel(const LevelArgs<Idx> a, int last_level) {
  const int lane = threadIdx.x, nb = 1 << a.b;
  int64_t tot[kMaxNB / 64];
  for (int q = 0; q * 64 < nb; ++q) tot[q] = q * 64 + lane < nb ? a.bin_total[q * 64 + lane] : 0;
  write_bin_starts(a, 0, 0, tot, lane, last_level != 0);
  if (lane == 0) {
    if (a.seg_next) a.seg_next[nb] = a.nnz;
    if (last_level && a.indptr) a.indptr[a.num_rows] = static_cast<Idx>(a.nnz);
  }
}

// level 0 with few items (small inputs: the block builder's id sorts): both steps in ONE launch, one thread per bin
constexpr int kScan0FusedItems = 16;   // (more items: one wave per bin in parallel beats one workgroup, even counting the second launch)
template <typename Idx>
__global__ __launch_bounds__(512) void msd_scan0_fused_kernel(const LevelArgs<Idx> a, int last_level) {   // nb <= 512
  // step 1: wave w takes bins w, w + 8, ...: exclusive prefix of the bin's per-item counts in 64-item chunks (coalesced
  // loads, one wave scan per chunk) -> the counts become prefixes in place, the bin totals go to LDS
  __shared__ int64_t total[512];
  __shared__ int64_t wsum[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nb = 1 << a.b;
  // (the (bin, chunk) steps of a wave form one sequence; the loads run kAhead steps ahead of the scans: 32 bins of
  // dependent global round trips were 20 us of a 90 k-key sort)
  {
    constexpr int kAhead = 4;
    const int nch = static_cast<int>((a.n_items_host + 63) / 64);
    const int nbw = (nb - wave + 7) / 8;                 // bins of this wave
    const int steps = nbw * nch;
    auto load = [&](int st) -> uint32_t {
      if (st >= steps) return 0u;
      const int d = wave + 8 * (st / nch);
      const int64_t i = static_cast<int64_t>(st % nch) * 64 + lane;
      return i < a.n_items_host ? static_cast<uint32_t>(a.counts[static_cast<int64_t>(d) * a.n_items_host + i]) : 0u;
    };
    uint32_t pre[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) pre[k] = load(k);
    int64_t run = 0;
    for (int st = 0; st < steps; st += kAhead) {
      uint32_t cur[kAhead];
#pragma unroll
      for (int k = 0; k < kAhead; ++k) cur[k] = pre[k];
#pragma unroll
      for (int k = 0; k < kAhead; ++k) pre[k] = load(st + kAhead + k);
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        const int s1 = st + k;
        if (s1 >= steps) break;
        const int d = wave + 8 * (s1 / nch), ch = s1 % nch;
        if (ch == 0) run = 0;
        const int64_t i = static_cast<int64_t>(ch) * 64 + lane;
        const uint32_t v = cur[k];
        const uint32_t incl = wave_inclusive_scan(v);
        if (i < a.n_items_host) a.counts[static_cast<int64_t>(d) * a.n_items_host + i] = static_cast<uint64_t>(run) + incl - v;
        run += __shfl(incl, 63, 64);
        if (ch == nch - 1 && lane == 0) total[d] = run;
      }
    }
  }
  __syncthreads();
  // step 2: exclusive scan over the bins
  const int d = threadIdx.x;
  const int64_t run = d < nb ? total[d] : 0;
  const int64_t incl = wave_inclusive_scan(run);
  if ((d & 63) == 63) wsum[d >> 6] = incl;
  __syncthreads();
  int64_t excl = incl - run;
  for (int w = 0; w < (d >> 6); ++w) excl += wsum[w];
  if (d < nb) {
    if (last_level && a.indptr && d <= a.num_rows) a.indptr[d] = static_cast<Idx>(excl);
    if (a.seg_next) a.seg_next[d] = excl;
  }
  if (d == 0) {
    if (a.seg_next) a.seg_next[nb] = a.nnz;
    if (last_level && a.indptr) a.indptr[a.num_rows] = static_cast<Idx>(a.nnz);
  }
}

// ---- level > 0 scan: one wavefront per bucket (its items are consecutive) ---------------------------------------------
template <typename Idx>
__global__ __launch_bounds__(256) void msd_scan_seg_kernel(const LevelArgs<Idx> a, int last_level) {
  const int64_t s = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (s >= a.num_segs) return;
  const int lane = threadIdx.x & 63, nb = 1 << a.b;
  int64_t run[kMaxNB / 64];
  for (int q = 0; q < kMaxNB / 64; ++q) run[q] = 0;
  const int32_t j0 = a.item_base[s], j1 = a.item_base[s + 1];
  for (int32_t j = j0; j < j1; ++j) {
    uint64_t* c = a.counts + static_cast<int64_t>(j) * nb;
    for (int q = 0; q * 64 < nb; ++q) {
      const int d = q * 64 + lane;
      if (d < nb) {
        const uint64_t v = c[d];
        c[d] = static_cast<uint64_t>(run[q]);
        run[q] += static_cast<int64_t>(v);
      }
    }
  }
  write_bin_starts(a, s, a.seg[s], run, lane, last_level != 0);
  if (s == a.num_segs - 1 && lane == 0) {
    if (a.seg_next) a.seg_next[a.num_segs * nb] = a.nnz;
    if (last_level && a.indptr) a.indptr[a.num_rows] = static_cast<Idx>(a.nnz);
  }
}

// ---- items of the next level: pieces of <= kItemLen elements of every bucket ------------------------------------------
static __global__ __launch_bounds__(256) void msd_item_count_kernel(const int64_t* __restrict__ seg, int64_t num_segs, int item_len,
                                                                   int32_t* __restrict__ cnt) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s > num_segs) return;
  cnt[s] = s < num_segs ? static_cast<int32_t>((seg[s + 1] - seg[s] + item_len - 1) / item_len) : 0;
}

static __global__ __launch_bounds__(256) void msd_item_fill_kernel(const int64_t* __restrict__ seg, int64_t num_segs, int item_len,
                                                                  const int32_t* __restrict__ item_base, Item* __restrict__ items) {
  const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= num_segs) return;
  const int64_t b0 = seg[s], len = seg[s + 1] - b0;
  int32_t j = item_base[s];
  for (int64_t o = 0; o < len; o += item_len, ++j) {
    Item it;
    it.start = b0 + o;
    it.len = static_cast<int32_t>(len - o < item_len ? len - o : item_len);
    it.seg = static_cast<int32_t>(s);
    items[j] = it;
  }
}

// ---- scatter: one wavefront per item, tiles of kTile elements -------------------------------------------------------------
template <bool WIDE>
inline size_t scatter_lds_bytes(int b) {
  return sizeof(typename El<WIDE>::T) * kTile + (8 + 8 + 4) * (size_t(1) << b) + 2 * kTile;
}

template <bool WIDE, int IN, int OUT, typename Idx>
__global__ __launch_bounds__(64) void msd_scatter_kernel(const LevelArgs<Idx> a) {
  using E = El<WIDE>;
  using T = typename E::T;
  // LDS sized by the level's bin count (scatter_lds_bytes): the tables are what limits the waves per CU
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int nbins = 1 << a.b;
  T* tile = reinterpret_cast<T*>(lds_raw);                                  // the tile in bin order (without this level's digit)
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(tile + kTile);
  int64_t* gcur = reinterpret_cast<int64_t*>(mask + nbins);                 // where the bin's next element goes in the output
  uint32_t* start = reinterpret_cast<uint32_t*>(gcur + nbins);              // next free slot of the bin inside the tile
  uint16_t* tile_d = reinterpret_cast<uint16_t*>(start + nbins);            // the bin of every slot
  Item it;
  if (!get_item(a, blockIdx.x, &it)) return;
  const int lane = threadIdx.x;
  const int nb = 1 << a.b;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int d = lane; d < nb; d += 64) {
    mask[d] = 0ull;
    // first output position of (this item, bin d): the bin's start + what earlier items of the bucket put there
    const int64_t r = (static_cast<int64_t>(it.seg) << a.b) | d;       // (last level: the row this bin is)
    const int64_t bin0 = a.seg_next ? a.seg_next[static_cast<int64_t>(it.seg) * nb + d]
                                    : (r <= a.num_rows ? static_cast<int64_t>(a.indptr[r]) : a.nnz);
    gcur[d] = bin0 + static_cast<int64_t>(a.level == 0 ? a.counts[static_cast<int64_t>(d) * a.n_items_host + blockIdx.x]
                                                       : a.counts[static_cast<int64_t>(blockIdx.x) * nb + d]);
  }
  __syncthreads();
  for (int t0 = 0; t0 < it.len; t0 += kTile) {
    const int nt = it.len - t0 < kTile ? it.len - t0 : kTile;
    T x[kTileRows];
    uint32_t dg[kTileRows];
    {
      Raw<WIDE, IN, Idx> raw[kTileRows];
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) {   // every load of the tile in flight (indices past the end re-read the last element)
        const int i = r * 64 + lane;
        raw[r] = load_raw<WIDE, IN>(a, it.start + t0 + (i < nt ? i : nt - 1));
      }
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) {
        const int i = r * 64 + lane;
        x[r] = split<WIDE, IN>(a, raw[r], it.start + t0 + i, &dg[r]);
        if (i >= nt) dg[r] = 0xffffffffu;
      }
    }
    for (int d = lane; d < nb; d += 64) start[d] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTileRows; ++r)
      if (dg[r] != 0xffffffffu) atomicAdd(&start[dg[r]], 1u);
    __syncthreads();
    {  // exclusive scan of the tile histogram -> first slot of every bin
      uint32_t carry = 0;
      for (int q = 0; q * 64 < nb; ++q) {
        const int d = q * 64 + lane;
        const uint32_t v = d < nb ? start[d] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (d < nb) {
          start[d] = carry + incl - v;
          gcur[d] -= carry + incl - v;   // from here to the end of the tile: output position of slot j of bin d = gcur[d] + j
        }
        carry += __shfl(incl, 63, 64);
      }
    }
    __syncthreads();
    // stable rank, row by row: lanes of one bin find each other through the bin's 64-bit mask word.  One wavefront per
    // workgroup and the LDS executes a wave's instructions in order, so a row needs ONE wait (for the two reads): the
    // leader's stores of row r, the ORs of row r + 1 and its reads are issued back to back; the compiler keeps their
    // program order (same addresses), the signal fences only stop it from moving them across rows.
    uint32_t slot[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
      const bool ok = dg[r] != 0xffffffffu;
      unsigned long long m = 0ull;
      uint32_t base = 0;
      if (ok) {
        __hip_atomic_fetch_or(&mask[dg[r]], 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        m = __hip_atomic_load(&mask[dg[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        base = __hip_atomic_load(&start[dg[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      slot[r] = base + __popcll(m & lt);
      if (ok && (m & lt) == 0ull) {  // lowest lane of the bin in this row
        __hip_atomic_store(&start[dg[r]], base + static_cast<uint32_t>(__popcll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&mask[dg[r]], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kTileRows; ++r)
      if (dg[r] != 0xffffffffu) {
        tile[slot[r]] = x[r];
        tile_d[slot[r]] = static_cast<uint16_t>(dg[r]);
      }
    __syncthreads();
    // write-out in slot order: consecutive lanes -> consecutive addresses of one bin
#pragma unroll 4
    for (int r = 0; r < kTileRows; ++r) {
      const int j = r * 64 + lane;
      if (j < nt) {
        const T y = tile[j];
        const uint32_t d = tile_d[j];
        const int64_t p = gcur[d] + j;
        if constexpr (OUT == kOutElem) {
          static_cast<T*>(a.dst)[p] = y;
        } else if constexpr (OUT == kOutCsr) {
          const uint64_t ps = E::pos(y, a.fmt);
          a.indices[p] = static_cast<Idx>(E::col(y, a.fmt));
          a.eids_out[p] = a.eids_in ? a.eids_in[ps] : static_cast<Idx>(ps);
        } else {
          if constexpr (WIDE)   // keys-only: the word IS the key
            a.keys_out[p] = static_cast<Idx>(y.k);
          else
            a.keys_out[p] = static_cast<Idx>(y);
        }
      }
    }
    __syncthreads();
    for (int d = lane; d < nb; d += 64) gcur[d] += start[d];   // (start[d] = one past the bin's last slot of this tile)
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
struct Plan {
  int key_bits = 0, nl = 0, b[kMaxLevels] = {0};
  Fmt fmt{0, 0};
  bool wide = false;
  int64_t nnz = 0;
  int64_t segs[kMaxLevels + 1] = {0};    // buckets entering level l
  int64_t max_items[kMaxLevels] = {0};   // launch bound of level l
  size_t bytes = 0;
  // offsets into the workspace
  size_t off_buf[2] = {0, 0}, off_counts = 0, off_bin_total = 0, off_seg[kMaxLevels + 1] = {0}, off_items = 0,
         off_item_base = 0, off_item_cnt = 0, off_scan_tmp = 0;
};

// key_bits: keys lie in [0, 2^key_bits); cb / pb: bits of the column ids and of the positions (0 / 0: keys only)
inline Plan make_plan(int64_t nnz, int key_bits, int cb, int pb) {
  Plan p;
  p.nnz = nnz;
  p.key_bits = key_bits < 1 ? 1 : key_bits;
  p.nl = (p.key_bits + kMaxDigit - 1) / kMaxDigit;
  const int base = p.key_bits / p.nl, extra = p.key_bits % p.nl;
  for (int l = 0; l < p.nl; ++l) p.b[l] = base + (l < extra ? 1 : 0);
  p.fmt = Fmt{cb, pb};
  p.wide = p.key_bits - p.b[0] + cb + pb > 64;   // (b[0] is set above: a word holds the key WITHOUT the first digit)
  p.segs[0] = 1;
  for (int l = 0; l < p.nl; ++l) p.segs[l + 1] = p.segs[l] << p.b[l];
  int64_t worst_items = 0, worst_segs = 0, worst_counts = 0;
  for (int l = 0; l < p.nl; ++l) {
    p.max_items[l] = (nnz + item_len_for(nnz) - 1) / item_len_for(nnz) + (l == 0 ? 0 : p.segs[l]);
    if (p.max_items[l] > worst_items) worst_items = p.max_items[l];
    if (l > 0 && p.segs[l] > worst_segs) worst_segs = p.segs[l];
    if ((p.max_items[l] << p.b[l]) > worst_counts) worst_counts = p.max_items[l] << p.b[l];
  }
  const size_t esz = p.wide ? sizeof(Wide) : sizeof(uint64_t);
  size_t o = 0;
  const int nbuf = p.nl >= 3 ? 2 : (p.nl == 2 ? 1 : 0);
  for (int k = 0; k < 2; ++k) {
    p.off_buf[k] = o;
    if (k < nbuf) o += align256(esz * nnz);
  }
  p.off_counts = o;
  o += align256(sizeof(uint64_t) * static_cast<size_t>(worst_counts));
  p.off_bin_total = o;
  o += align256(sizeof(int64_t) * kMaxNB);
  for (int l = 1; l < p.nl; ++l) {
    p.off_seg[l] = o;
    o += align256(sizeof(int64_t) * (p.segs[l] + 1));
  }
  p.off_items = o;
  o += align256(sizeof(Item) * static_cast<size_t>(worst_items + 1));
  p.off_item_base = o;
  o += align256(sizeof(int32_t) * (worst_segs + 2));
  p.off_item_cnt = o;
  o += align256(sizeof(int32_t) * (worst_segs + 2));
  p.off_scan_tmp = o;
  o += scan_temp_bytes(worst_segs + 2, sizeof(int32_t));
  p.bytes = o;
  return p;
}

// COO -> CSR: every level of the plan; a0 carries the caller's arrays (row / col / eids / outputs / num_rows)
template <bool WIDE, typename Idx>
int run_levels(const Plan& p, LevelArgs<Idx> a0, char* ws, hipStream_t s) {
  int kb = p.key_bits;
  for (int l = 0; l < p.nl; ++l) {
    LevelArgs<Idx> a = a0;
    const bool last = l == p.nl - 1;
    a.nnz = p.nnz;
    a.kb = kb;
    a.b = p.b[l];
    a.level = l;
    a.item_len = item_len_for(p.nnz);
    a.shift = kb - p.b[l];
    a.fmt = p.fmt;
    a.num_segs = p.segs[l];
    a.counts = reinterpret_cast<uint64_t*>(ws + p.off_counts);
    a.bin_total = reinterpret_cast<int64_t*>(ws + p.off_bin_total);
    a.seg = l > 0 ? reinterpret_cast<const int64_t*>(ws + p.off_seg[l]) : nullptr;
    a.seg_next = last ? nullptr : reinterpret_cast<int64_t*>(ws + p.off_seg[l + 1]);
    a.src = l > 0 ? ws + p.off_buf[(l - 1) & 1] : nullptr;
    a.dst = last ? nullptr : ws + p.off_buf[l & 1];
    if (!last) a.indptr = nullptr;  // only the last level's bins are rows
    const unsigned grid = static_cast<unsigned>(p.max_items[l] > 0 ? p.max_items[l] : 1);
    if (l == 0) {
      a.n_items_host = (p.nnz + a.item_len - 1) / a.item_len;
      hipLaunchKernelGGL((msd_hist_kernel<WIDE, kInCoo, Idx>), dim3(grid), dim3(64), 0, s, a);
      if (a.n_items_host <= kScan0FusedItems && a.b <= 9) {
        hipLaunchKernelGGL(msd_scan0_fused_kernel<Idx>, dim3(1), dim3(512), 0, s, a, last ? 1 : 0);
      } else {
        hipLaunchKernelGGL(msd_scan0_bins_kernel<Idx>, dim3(1u << a.b), dim3(64), 0, s, a);
        hipLaunchKernelGGL(msd_scan0_top_kernel<Idx>, dim3(1), dim3(64), 0, s, a, last ? 1 : 0);
      }
    } else {
      int32_t* cnt = reinterpret_cast<int32_t*>(ws + p.off_item_cnt);
      int32_t* ibase = reinterpret_cast<int32_t*>(ws + p.off_item_base);
      Item* items = reinterpret_cast<Item*>(ws + p.off_items);
      const unsigned gs = static_cast<unsigned>((a.num_segs + 1 + 255) / 256);
      hipLaunchKernelGGL(msd_item_count_kernel, dim3(gs), dim3(256), 0, s, a.seg, a.num_segs, a.item_len, cnt);
      if (exclusive_scan<int32_t, int32_t>(cnt, ibase, a.num_segs + 1, ws + p.off_scan_tmp, s)) return -1;
      hipLaunchKernelGGL(msd_item_fill_kernel, dim3(gs), dim3(256), 0, s, a.seg, a.num_segs, a.item_len, ibase, items);
      a.items = items;
      a.item_base = ibase;
      a.n_items_dev = ibase + a.num_segs;
      a.n_items_host = 0;
      hipLaunchKernelGGL((msd_hist_kernel<WIDE, kInElem, Idx>), dim3(grid), dim3(64), 0, s, a);
      hipLaunchKernelGGL(msd_scan_seg_kernel<Idx>, dim3(static_cast<unsigned>((a.num_segs + 3) / 4)), dim3(256), 0, s, a,
                         last ? 1 : 0);
    }
#define DGLA_MSD_SCATTER(IN, OUT) \
  hipLaunchKernelGGL((msd_scatter_kernel<WIDE, IN, OUT, Idx>), dim3(grid), dim3(64), scatter_lds_bytes<WIDE>(a.b), s, a)
    if (l == 0 && last) {
      DGLA_MSD_SCATTER(kInCoo, kOutCsr);
    } else if (l == 0) {
      DGLA_MSD_SCATTER(kInCoo, kOutElem);
    } else if (last) {
      DGLA_MSD_SCATTER(kInElem, kOutCsr);
    } else {
      DGLA_MSD_SCATTER(kInElem, kOutElem);
    }
#undef DGLA_MSD_SCATTER
    kb -= p.b[l];
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// keys-only sort (the block builder's source-node ids, sampling.hip): least-significant digit first with the same
// histogram / scan / scatter launches — every pass is ONE bucket (no bucket tables), keys travel whole, so wide and
// sparse key ranges cost passes, not memory.  key_bits <= 0: all 31 / 63 value bits of the id type.
// ------------------------------------------------------------------------------------------------------------------
struct KeysPlan {
  int passes = 0, b[kMaxLevels] = {0};
  int64_t n = 0, n_items = 0;
  size_t off_tmp = 0, off_counts = 0, off_bin_total = 0, off_starts = 0, bytes = 0;
};

inline KeysPlan make_keys_plan(int64_t n, int key_bits, size_t key_size) {
  KeysPlan p;
  p.n = n;
  if (key_bits <= 0 || key_bits > static_cast<int>(key_size * 8 - 1)) key_bits = static_cast<int>(key_size * 8 - 1);
  p.passes = (key_bits + kKeysDigit - 1) / kKeysDigit;
  const int base = key_bits / p.passes, extra = key_bits % p.passes;
  for (int l = 0; l < p.passes; ++l) p.b[l] = base + (l < extra ? 1 : 0);
  p.n_items = (n + item_len_for(n) - 1) / item_len_for(n);
  size_t o = 0;
  p.off_tmp = o;
  o += align256(key_size * static_cast<size_t>(n));
  p.off_counts = o;
  o += align256(sizeof(uint64_t) * static_cast<size_t>(p.n_items) * kMaxNB);
  p.off_bin_total = o;
  o += align256(sizeof(int64_t) * kMaxNB);
  p.off_starts = o;
  o += align256(sizeof(int64_t) * (kMaxNB + 1));
  p.bytes = o;
  return p;
}

// Up to 16 384 keys of at most 32 bits: ONE workgroup, the whole LSD radix sort in LDS (the mini-batch block builder
// sorts a few thousand ids per layer, where launch count is the whole cost).  16 wavefronts; wave w owns the 1 024 keys
// [1024 w, 1024 w + 1024) as 16 rows of 64, in order (stable).  A pass of 8 bits: per-wave histogram -> exclusive scan in
// (bin, wave) order -> every wave ranks its rows one by one (lanes with my digit from 8 ballots, the wave's cursor of the
// bin from LDS, the lowest lane of the bin advances it) and stores the key into the other LDS buffer.
constexpr int kSmallKeys = 16384;
constexpr int kSmallRows = kSmallKeys / 1024;   // 64-key rows per wavefront
constexpr size_t kSmallRadixLds = 2 * kSmallKeys * sizeof(uint32_t) + 16 * 256 * sizeof(uint32_t);
template <typename Idx>
__global__ __launch_bounds__(1024) void small_radix_kernel(const Idx* __restrict__ in, Idx* __restrict__ out, int n, int key_bits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  uint32_t* buf0 = reinterpret_cast<uint32_t*>(lds_raw);
  uint32_t* buf1 = buf0 + kSmallKeys;
  uint32_t* cur = buf1 + kSmallKeys;            // [16][256] per-wave counts, then cursors
  __shared__ uint32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int i = tid; i < kSmallKeys; i += 1024) buf0[i] = i < n ? static_cast<uint32_t>(in[i]) : 0xffffffffu;
  __syncthreads();
  uint32_t* src = buf0;
  uint32_t* dst = buf1;
  for (int shift = 0; shift < key_bits; shift += 8) {
    for (int i = tid; i < 16 * 256; i += 1024) cur[i] = 0;
    __syncthreads();
    uint32_t key[kSmallRows], dg[kSmallRows];
    unsigned long long same[kSmallRows];
#pragma unroll
    for (int r = 0; r < kSmallRows; ++r) {
      key[r] = src[wave * (64 * kSmallRows) + r * 64 + lane];
      dg[r] = (key[r] >> shift) & 255u;
      unsigned long long m = ~0ull;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool bit = (dg[r] >> k) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
      }
      same[r] = m;
      if ((m & lt) == 0ull) atomicAdd(&cur[wave * 256 + dg[r]], static_cast<uint32_t>(__popcll(m)));   // (wave-private row)
    }
    __syncthreads();
    {  // exclusive scan of the 4 096 counts in (bin, wave) order: thread t owns bin t / 4, waves 4 (t % 4) .. + 3
      const int d = tid >> 2, w0 = (tid & 3) * 4;
      uint32_t v[4], s4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = cur[(w0 + k) * 256 + d];
        s4 += v[k];
      }
      const uint32_t incl = wave_inclusive_scan(s4);
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      uint32_t off = incl - s4;
      for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        cur[(w0 + k) * 256 + d] = off;
        off += v[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSmallRows; ++r) {
      uint32_t* c = &cur[wave * 256 + dg[r]];
      const uint32_t base = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      dst[base + __popcll(same[r] & lt)] = key[r];
      if ((same[r] & lt) == 0ull)
        __hip_atomic_store(c, base + static_cast<uint32_t>(__popcll(same[r])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    }
    __syncthreads();
    uint32_t* t = src;
    src = dst;
    dst = t;
  }
  for (int i = tid; i < n; i += 1024) out[i] = static_cast<Idx>(src[i]);
}

// sorted copy of `keys` into `out` (out != keys); ws: make_keys_plan(n, key_bits, sizeof(Idx)).bytes
template <typename Idx>
int sort_keys(const Idx* keys, Idx* out, int64_t n, int key_bits, char* ws, hipStream_t s) {
  if (n <= 0) return 0;
  if (key_bits <= 0 || key_bits > static_cast<int>(sizeof(Idx) * 8 - 1)) key_bits = static_cast<int>(sizeof(Idx) * 8 - 1);
  if (n <= kSmallKeys && key_bits <= 32) {
    static bool lds_ok = [] {   // more than the default 64 KiB of dynamic LDS: asked for once per process
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&small_radix_kernel<Idx>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSmallRadixLds)) == hipSuccess;
    }();
    if (lds_ok) {
      hipLaunchKernelGGL(small_radix_kernel<Idx>, dim3(1), dim3(1024), kSmallRadixLds, s, keys, out, static_cast<int>(n), key_bits);
      DGLA_CHECK_HIP(hipGetLastError());
      return 0;
    }
    (void)hipGetLastError();
  }
  const KeysPlan p = make_keys_plan(n, key_bits, sizeof(Idx));
  Idx* tmp = reinterpret_cast<Idx*>(ws + p.off_tmp);
  const Idx* in = keys;
  int shift = 0;
  for (int l = 0; l < p.passes; ++l) {
    Idx* dst = ((p.passes - 1 - l) & 1) ? tmp : out;   // the last pass lands in `out`
    LevelArgs<Idx> a{};
    a.row = in;
    a.keys_out = dst;
    a.nnz = n;
    a.kb = 0;
    a.b = p.b[l];
    a.level = 0;
    a.item_len = item_len_for(n);
    a.shift = shift;
    a.fmt = Fmt{0, 0};
    a.num_segs = 1;
    a.n_items_host = p.n_items;
    a.counts = reinterpret_cast<uint64_t*>(ws + p.off_counts);
    a.bin_total = reinterpret_cast<int64_t*>(ws + p.off_bin_total);
    a.seg_next = reinterpret_cast<int64_t*>(ws + p.off_starts);
    const unsigned grid = static_cast<unsigned>(p.n_items);
    hipLaunchKernelGGL((msd_hist_kernel<false, kInKeys, Idx>), dim3(grid), dim3(64), 0, s, a);
    if (a.n_items_host <= kScan0FusedItems) {
      hipLaunchKernelGGL(msd_scan0_fused_kernel<Idx>, dim3(1), dim3(512), 0, s, a, 0);
    } else {
      hipLaunchKernelGGL(msd_scan0_bins_kernel<Idx>, dim3(1u << a.b), dim3(64), 0, s, a);
      hipLaunchKernelGGL(msd_scan0_top_kernel<Idx>, dim3(1), dim3(64), 0, s, a, 0);
    }
    hipLaunchKernelGGL((msd_scatter_kernel<false, kInKeys, kOutKeys, Idx>), dim3(grid), dim3(64), scatter_lds_bytes<false>(a.b), s, a);
    in = dst;
    shift += p.b[l];
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace msd
}  // namespace dgla
