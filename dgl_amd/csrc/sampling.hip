// Uniform neighbour sampling and block construction on the device: the two
// steps in front of the g-SpMM in mini-batch GraphSAGE (BASELINE config 4).
//
// Reference: CSRRowWiseSamplingUniform<kDGLCUDA> (src/array/cuda/rowwise_sampling.cu:43-330:
// degree kernel + prefix sum + one warp per row, reservoir "algorithm R" with atomics in the
// without-replacement case) behind dgl.sampling.sample_neighbors
// (python/dgl/sampling/neighbor.py:222-395), and ToBlock<kDGLCUDA>
// (src/graph/transform/cuda/cuda_to_block.cu + cuda_map_edges.cuh: an ordered device hash table
// that renumbers the source nodes) behind dgl.to_block.
//
// MI355X-first choices:
//  * Sampling without replacement is Floyd's subset algorithm run by ONE LANE per seed row:
//    exactly `fanout` distinct positions in O(fanout) draws, no atomics, no retries, and the
//    same picks whatever the launch geometry (counter-based generator keyed by (seed, row,
//    draw)) — a run is reproducible from its seed, which the reference's per-block Philox
//    streams are not across geometries.  64 independent rows per wavefront keep the lanes busy
//    for fanouts of 5-25; the gathers of the picked neighbours are the only memory traffic.
//  * Renumbering uses a dense node -> local id map instead of a hash table: with 288 GB of HBM
//    a 4-byte entry per node is affordable even for papers100M (444 MB) and turns every
//    lookup into one load.  The map lives in caller-owned scratch, is all -1 between calls and
//    is cleaned by touching only the entries the call used.  Local ids are deterministic:
//    seeds first in the given order (the block's destination nodes are its first source
//    nodes, dgl.to_block's include_dst_in_src), then the other sampled nodes by ascending id.
#include "../../include/dgl_amd.h"

#include <cstring>

#include "common.h"
#include "sort.hip.h"

namespace dgla {
namespace {

int sfail(const std::string& m) {
  last_error() = m;
  return -1;
}

constexpr int kMaxFanout = 128;

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// uniform integer in [0, n) from the (seed, row, draw) counter: multiply-high reduction
__device__ __forceinline__ uint64_t draw(uint64_t seed, uint64_t row, uint32_t k, uint64_t n) {
  const uint64_t r = mix64(mix64(seed ^ (row * 0xD1B54A32D192ED03ull)) + k);
  return __umul64hi(r, n);
}

template <typename Idx>
__global__ __launch_bounds__(256) void sample_count_kernel(const Idx* __restrict__ indptr,
                                                           const Idx* __restrict__ seeds,
                                                           int64_t num_seeds, int fanout, int replace,
                                                           Idx* __restrict__ counts,
                                                           const int64_t* __restrict__ num_valid = nullptr) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i > num_seeds) return;
  if (i == num_seeds || (num_valid != nullptr && i >= *num_valid)) {
    counts[i] = 0;  // so that an exclusive scan over num_seeds + 1 entries ends with the total;
    return;         // padded form: seeds past the device-side count are padding and pick nothing
  }
  const int64_t r = static_cast<int64_t>(seeds[i]);
  const int64_t deg = static_cast<int64_t>(indptr[r + 1]) - static_cast<int64_t>(indptr[r]);
  int64_t c = deg < fanout ? deg : fanout;
  if (replace) c = deg == 0 ? 0 : fanout;
  if (fanout < 0) c = deg;  // fanout = -1: all neighbours (neighbor.py:259-261)
  counts[i] = static_cast<Idx>(c);
}

// FM: capacity of the position array.  FM <= 32: the array lives in REGISTERS (every loop over it is
// unrolled to its static bound and predicated) — with the run-time bound of kMaxFanout it sat in
// scratch memory, and Floyd's duplicate check (m reads per draw) made the kernel 30-50 us for a
// few thousand seeds; larger fanouts keep the general form.
template <typename Idx, int FM>
__global__ __launch_bounds__(256) void sample_pick_kernel(const Idx* __restrict__ indptr,
                                                          const Idx* __restrict__ indices,
                                                          const Idx* __restrict__ eids,
                                                          const Idx* __restrict__ seeds,
                                                          int64_t num_seeds, int fanout, int replace,
                                                          uint64_t rng_seed,
                                                          const Idx* __restrict__ out_indptr,
                                                          Idx* __restrict__ out_src,
                                                          Idx* __restrict__ out_eids,
                                                          const int64_t* __restrict__ num_valid = nullptr,
                                                          const int64_t* __restrict__ rng_counter = nullptr) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= num_seeds) return;
  if (num_valid != nullptr && i >= *num_valid) return;
  // padded form: the call's stream position comes from device memory, so that a captured launch
  // draws fresh neighbours on every replay
  if (rng_counter != nullptr) rng_seed += static_cast<uint64_t>(*rng_counter) * 0x9E3779B97F4A7C15ull;
  const int64_t r = static_cast<int64_t>(seeds[i]);
  const int64_t start = static_cast<int64_t>(indptr[r]);
  const int64_t deg = static_cast<int64_t>(indptr[r + 1]) - start;
  const int64_t o = static_cast<int64_t>(out_indptr[i]);
  // Positions first (arithmetic only), then the picked neighbours in batches of 8 independent loads:
  // a load -> store pair per pick made every pick wait a full memory round trip (15 picks: 34 us for
  // a kernel whose traffic is a few hundred kilobytes).
  constexpr bool REG = FM <= 32;
  int64_t chosen[FM];
  int64_t n = 0;
  bool identity = false;  // chosen[k] == k: the whole neighbourhood, in CSR order
  if (fanout < 0 || (!replace && deg <= fanout)) {
    n = deg;
    identity = true;
  } else if (deg == 0) {
    return;
  } else if (replace) {
    if constexpr (REG) {
#pragma unroll
      for (int k = 0; k < FM; ++k)
        if (k < fanout) chosen[k] = static_cast<int64_t>(draw(rng_seed, r, k, deg));
    } else {
      for (int k = 0; k < fanout; ++k) chosen[k] = static_cast<int64_t>(draw(rng_seed, r, k, deg));
    }
    n = fanout;
  } else {
    // Floyd: for j = deg - fanout .. deg - 1: t = U[0, j]; take t unless already taken, else j
    if constexpr (REG) {
#pragma unroll
      for (int m = 0; m < FM; ++m) {
        if (m < fanout) {
          const int64_t j = deg - fanout + m;
          int64_t t = static_cast<int64_t>(draw(rng_seed, r, static_cast<uint32_t>(m), j + 1));
          bool dup = false;
#pragma unroll
          for (int q = 0; q < FM; ++q)
            if (q < m) dup = dup || chosen[q] == t;
          chosen[m] = dup ? j : t;
        }
      }
    } else {
      int m = 0;
      for (int64_t j = deg - fanout; j < deg; ++j) {
        int64_t t = static_cast<int64_t>(draw(rng_seed, r, static_cast<uint32_t>(m), j + 1));
        bool dup = false;
        for (int q = 0; q < m; ++q) dup = dup || chosen[q] == t;
        if (dup) t = j;
        chosen[m] = t;
        ++m;
      }
    }
    n = fanout;
  }
  if (identity) {  // (any length: never touches `chosen`)
    for (int64_t k = 0; k < n; k += 8) {
      Idx sv[8], ev[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t pos = k + u < n ? k + u : n - 1;
        sv[u] = indices[start + pos];
        ev[u] = eids ? eids[start + pos] : static_cast<Idx>(start + pos);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k + u < n) {
          out_src[o + k + u] = sv[u];
          out_eids[o + k + u] = ev[u];
        }
      }
    }
    return;
  }
  if constexpr (REG) {
    // all picks' loads are independent: issue them, then store
    Idx sv[FM], ev[FM];
#pragma unroll
    for (int k = 0; k < FM; ++k) {
      if (k < n) {
        sv[k] = indices[start + chosen[k]];
        ev[k] = eids ? eids[start + chosen[k]] : static_cast<Idx>(start + chosen[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < FM; ++k) {
      if (k < n) {
        out_src[o + k] = sv[k];
        out_eids[o + k] = ev[k];
      }
    }
  } else {
    for (int64_t k = 0; k < n; k += 8) {
      Idx sv[8], ev[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t kk = k + u < n ? k + u : n - 1;
        sv[u] = indices[start + chosen[kk]];
        ev[u] = eids ? eids[start + chosen[kk]] : static_cast<Idx>(start + chosen[kk]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k + u < n) {
          out_src[o + k + u] = sv[u];
          out_eids[o + k + u] = ev[u];
        }
      }
    }
  }
}

// ---- weighted sampling ------------------------------------------------------------------------
// Reference: CSRRowWiseSampling<kDGLCUDA> with probabilities (src/array/cuda/rowwise_sampling_prob.cu:
// without replacement A-Res keys u^(1/p) for every edge of a row, a segmented sort, the top
// `num_picks` — :150-300; with replacement a per-row CDF + binary search — :305-380; picks whose
// probability is zero are removed afterwards — :395-460).
//
// Here: ONE WAVEFRONT per seed row, no key array, no sort, no CDF array.
//  * without replacement: the equivalent exponential-clock form of A-Res — key = -ln(u) / p, the
//    `fanout` SMALLEST keys win (u^(1/p) largest <=> -ln(u)/p smallest).  u comes from the
//    counter-based generator keyed by (seed, row, position), so a key can be RE-computed instead
//    of stored: round t makes every lane scan its strided share of the row for the smallest key
//    above the previous round's (key, position), an xor-shuffle argmin elects the pick.
//    O(fanout * deg / 64) key evaluations per lane, zero scratch, picks independent of the
//    launch geometry.  Edges with p <= 0 (or NaN) are never picked: a row yields
//    min(fanout, #positive edges) picks, the reference's result after its removal pass.
//  * with replacement: the row's total weight by a wave reduction, then per pick a walk over the
//    row in chunks of 64 with a wave prefix sum until the chunk holding u * total is found.
template <typename F>
__device__ __forceinline__ double clock_key(uint64_t seed, uint64_t row, uint64_t pos, F p) {
  // u in (0, 1]: 53 random bits, never 0, so that -ln(u) is finite
  const uint64_t r = mix64(mix64(seed ^ (row * 0xD1B54A32D192ED03ull)) + 0x5851F42D4C957F2Dull * (pos + 1));
  const double u = (static_cast<double>(r >> 11) + 1.0) * (1.0 / 9007199254740992.0);
  return -log(u) / static_cast<double>(p);
}

template <typename Idx, typename F>
__global__ __launch_bounds__(256) void weighted_count_kernel(const Idx* __restrict__ indptr,
                                                             const Idx* __restrict__ eids,
                                                             const F* __restrict__ prob,
                                                             const Idx* __restrict__ seeds,
                                                             int64_t num_seeds, int fanout, int replace,
                                                             Idx* __restrict__ counts,
                                                             const int64_t* __restrict__ num_valid = nullptr) {
  const int64_t i = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i > num_seeds) return;
  if (i == num_seeds || (num_valid != nullptr && i >= *num_valid)) {  // (padding slots pick nothing)
    if (lane == 0) counts[i] = 0;
    return;
  }
  const int64_t r = static_cast<int64_t>(seeds[i]);
  const int64_t start = static_cast<int64_t>(indptr[r]);
  const int64_t deg = static_cast<int64_t>(indptr[r + 1]) - start;
  int pos = 0;
  for (int64_t j = lane; j < deg; j += 64) {
    const F p = prob[eids ? static_cast<int64_t>(eids[start + j]) : start + j];
    pos += p > F(0) ? 1 : 0;
  }
  for (int m = 32; m >= 1; m >>= 1) pos += __shfl_xor(pos, m, 64);
  int64_t c = pos < fanout ? pos : fanout;
  if (replace) c = pos == 0 ? 0 : fanout;
  if (lane == 0) counts[i] = static_cast<Idx>(c);
}

template <typename Idx, typename F>
__global__ __launch_bounds__(256) void weighted_pick_kernel(
    const Idx* __restrict__ indptr, const Idx* __restrict__ indices, const Idx* __restrict__ eids,
    const F* __restrict__ prob, const Idx* __restrict__ seeds, int64_t num_seeds, int fanout,
    int replace, uint64_t rng_seed, const Idx* __restrict__ out_indptr, Idx* __restrict__ out_src,
    Idx* __restrict__ out_eids, const int64_t* __restrict__ rng_counter = nullptr) {
  const int64_t i = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= num_seeds) return;
  if (rng_counter != nullptr) rng_seed += static_cast<uint64_t>(*rng_counter) * 0x9E3779B97F4A7C15ull;
  const int64_t r = static_cast<int64_t>(seeds[i]);
  const int64_t start = static_cast<int64_t>(indptr[r]);
  const int64_t deg = static_cast<int64_t>(indptr[r + 1]) - start;
  const int64_t o = static_cast<int64_t>(out_indptr[i]);
  const int picks = static_cast<int>(static_cast<int64_t>(out_indptr[i + 1]) - o);
  if (picks == 0) return;
  auto weight = [&](int64_t j) {
    return prob[eids ? static_cast<int64_t>(eids[start + j]) : start + j];
  };
  auto emit = [&](int slot, int64_t pos) {
    out_src[o + slot] = indices[start + pos];
    out_eids[o + slot] = eids ? eids[start + pos] : static_cast<Idx>(start + pos);
  };
  const double inf = static_cast<double>(__builtin_huge_valf());
  if (!replace) {
    double last_key = -1.0;  // keys are >= 0
    int64_t last_pos = -1;
    for (int t = 0; t < picks; ++t) {
      double best = inf;
      int64_t best_pos = 0x7fffffffffffffffLL;
      for (int64_t j = lane; j < deg; j += 64) {
        const F p = weight(j);
        if (!(p > F(0))) continue;
        const double k = clock_key<F>(rng_seed, static_cast<uint64_t>(r), static_cast<uint64_t>(j), p);
        const bool after = k > last_key || (k == last_key && j > last_pos);  // not picked yet
        if (after && (k < best || (k == best && j < best_pos))) {
          best = k;
          best_pos = j;
        }
      }
      for (int m = 32; m >= 1; m >>= 1) {
        const double ok = __shfl_xor(best, m, 64);
        const int64_t op = __shfl_xor(best_pos, m, 64);
        if (ok < best || (ok == best && op < best_pos)) {
          best = ok;
          best_pos = op;
        }
      }
      if (lane == 0) emit(t, best_pos);
      last_key = best;
      last_pos = best_pos;
    }
    return;
  }
  // with replacement
  double total = 0.0;
  for (int64_t j = lane; j < deg; j += 64) {
    const F p = weight(j);
    total += p > F(0) ? static_cast<double>(p) : 0.0;
  }
  for (int m = 32; m >= 1; m >>= 1) total += __shfl_xor(total, m, 64);
  for (int t = 0; t < picks; ++t) {
    const uint64_t rr = mix64(mix64(rng_seed ^ (static_cast<uint64_t>(r) * 0xD1B54A32D192ED03ull)) + 0x2545F4914F6CDD1Dull * (t + 1));
    const double target = (static_cast<double>(rr >> 11) + 0.5) * (1.0 / 9007199254740992.0) * total;
    double run = 0.0;       // weight of the chunks already passed
    int64_t chosen = -1, last_positive = -1;
    for (int64_t base = 0; base < deg && chosen < 0; base += 64) {
      const int64_t j = base + lane;
      const F p = j < deg ? weight(j) : F(0);
      const double w = p > F(0) ? static_cast<double>(p) : 0.0;
      double pre = w;  // inclusive prefix over the lanes
      for (int m = 1; m < 64; m <<= 1) {
        const double up = __shfl_up(pre, m, 64);
        if (lane >= m) pre += up;
      }
      const bool hit = w > 0.0 && run + pre >= target;
      const unsigned long long mask = __ballot(hit);
      const unsigned long long posm = __ballot(w > 0.0);
      if (posm) last_positive = base + 63 - __builtin_clzll(posm);
      if (mask) chosen = base + (__ffsll(static_cast<long long>(mask)) - 1);
      run += __shfl(pre, 63, 64);
    }
    if (chosen < 0) chosen = last_positive;  // rounding left the target past the last prefix
    if (lane == 0) emit(t, chosen);
  }
}

// Padded form of the sampler's output (static shapes, no read-back; dgla_sample_neighbors_padded):
// the picks fill [0, total) of a buffer of `cap` = num_seeds * fanout entries; the rest becomes the
// edges of `sinks` extra SINK rows (rows num_seeds .. num_seeds + sinks - 1, equal shares; the last
// out_indptr entry = cap) pointing at the real seeds in turn — nodes the block holds anyway — so that
// every consumer sees a well-formed CSR of fixed size whose real rows are untouched.  Several sink
// rows and spread-out targets instead of one row aimed at one node: a single 80 k-edge row made the
// SpMM's fix-up walk 160 carry slots serially (38 us) and the backward COO kernel's atomics pile up
// on one feature row (66 us).
template <typename Idx>
__global__ __launch_bounds__(256) void pad_tail_kernel(Idx* __restrict__ out_indptr, int64_t num_seeds,
                                                       int64_t cap, int sinks, const Idx* __restrict__ seeds,
                                                       const int64_t* __restrict__ num_valid,
                                                       Idx* __restrict__ out_src, Idx* __restrict__ out_eids) {
  const int64_t j = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int64_t total = static_cast<int64_t>(out_indptr[num_seeds]);
  const int64_t pad = cap - total;
  if (j >= 1 && j <= sinks) out_indptr[num_seeds + j] = static_cast<Idx>(total + pad * j / sinks);
  if (j >= total && j < cap) {
    int64_t nv = num_valid != nullptr ? *num_valid : num_seeds;
    if (nv < 1) nv = 1;
    out_src[j] = seeds[(j - total) % nv];
    out_eids[j] = Idx(0);
  }
}

// ---- to_block -------------------------------------------------------------------------------
template <typename Idx>
__global__ __launch_bounds__(256) void scatter_seed_ids_kernel(const Idx* __restrict__ seeds,
                                                               int64_t num_seeds,
                                                               int32_t* __restrict__ node_map,
                                                               Idx* __restrict__ src_nodes,
                                                               const int64_t* __restrict__ num_valid = nullptr) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= num_seeds) return;
  // padded form: seeds past the device-side count are padding (copies of some node id): they keep
  // their row in the block but must not claim the node's local id
  if (num_valid == nullptr || i < *num_valid) node_map[seeds[i]] = static_cast<int32_t>(i);
  src_nodes[i] = seeds[i];
}

// flag[i] = sorted[i] starts a run of a node that is not a seed (gets a fresh local id)
template <typename Idx>
__global__ __launch_bounds__(256) void flag_new_nodes_kernel(const Idx* __restrict__ sorted, int64_t n,
                                                             const int32_t* __restrict__ node_map,
                                                             int32_t* __restrict__ flag) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    flag[i] = 0;
    return;
  }
  const bool first = i == 0 || sorted[i] != sorted[i - 1];
  flag[i] = (first && node_map[sorted[i]] < 0) ? 1 : 0;
}

template <typename Idx>
__global__ __launch_bounds__(256) void assign_new_ids_kernel(const Idx* __restrict__ sorted, int64_t n,
                                                             const int32_t* __restrict__ flag,
                                                             const int32_t* __restrict__ rank,
                                                             int64_t num_seeds,
                                                             int32_t* __restrict__ node_map,
                                                             Idx* __restrict__ src_nodes,
                                                             int64_t* __restrict__ num_src_out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    *num_src_out = num_seeds + rank[n];  // rank is the exclusive scan: rank[n] = number of new nodes
    return;
  }
  if (flag[i]) {
    const int64_t id = num_seeds + rank[i];
    node_map[sorted[i]] = static_cast<int32_t>(id);
    src_nodes[id] = sorted[i];
  }
}

template <typename Idx>
__global__ __launch_bounds__(256) void relabel_kernel(const Idx* __restrict__ src, int64_t n,
                                                      const int32_t* __restrict__ node_map,
                                                      Idx* __restrict__ local_src) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n) local_src[i] = static_cast<Idx>(node_map[src[i]]);
}

// put the map back to all -1: only the entries of this block's source nodes were touched
template <typename Idx>
__global__ __launch_bounds__(256) void reset_map_kernel(const Idx* __restrict__ src_nodes,
                                                        const int64_t* __restrict__ num_src,
                                                        int64_t capacity, int32_t* __restrict__ node_map) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < capacity && i < *num_src) node_map[src_nodes[i]] = -1;
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }
unsigned grid1(int64_t n) { return static_cast<unsigned>((n + 255) / 256 < 1 ? 1 : (n + 255) / 256); }

template <typename Idx>
size_t scan_temp_bytes(int64_t n) {
  return msd::scan_temp_bytes(n, sizeof(Idx));
}

// scratch of the keys-only sort (sort.hip.h); sized for the widest keys of the id type, so the same buffer serves
// callers with and without a node count
template <typename Idx>
size_t sort_keys_temp_bytes(int64_t n) { return msd::make_keys_plan(n, 0, sizeof(Idx)).bytes; }

template <typename Idx>
int run_sample(const dgla_csr* csc, const void* seeds, int64_t num_seeds, int fanout, int replace,
               uint64_t rng_seed, void* out_indptr, void* out_src, void* out_eids, char* ws, hipStream_t s,
               const int64_t* num_valid = nullptr, const int64_t* rng_counter = nullptr, int sinks = 0) {
  // counts -> exclusive scan in place of out_indptr (num_seeds + 1 entries)
  Idx* counts = reinterpret_cast<Idx*>(ws);
  void* temp = ws + align256(sizeof(Idx) * (num_seeds + 1));
  size_t temp_bytes = scan_temp_bytes<Idx>(num_seeds + 1);
  hipLaunchKernelGGL(sample_count_kernel<Idx>, dim3(grid1(num_seeds + 1)), dim3(256), 0, s,
                     static_cast<const Idx*>(csc->indptr), static_cast<const Idx*>(seeds), num_seeds,
                     fanout, replace, counts, num_valid);
  if (msd::exclusive_scan<Idx, Idx>(counts, static_cast<Idx*>(out_indptr), num_seeds + 1, temp, s)) return -1;
  if (out_src) {
#define DGLA_PICK(FM)                                                                                      \
  hipLaunchKernelGGL((sample_pick_kernel<Idx, FM>), dim3(grid1(num_seeds)), dim3(256), 0, s,                \
                     static_cast<const Idx*>(csc->indptr), static_cast<const Idx*>(csc->indices),          \
                     static_cast<const Idx*>(csc->data), static_cast<const Idx*>(seeds), num_seeds, fanout, \
                     replace, rng_seed, static_cast<const Idx*>(out_indptr), static_cast<Idx*>(out_src),    \
                     static_cast<Idx*>(out_eids), num_valid, rng_counter)
    if (fanout > 0 && fanout <= 16)
      DGLA_PICK(16);
    else if (fanout > 0 && fanout <= 32)
      DGLA_PICK(32);
    else
      DGLA_PICK(kMaxFanout);
#undef DGLA_PICK
  }
  if (sinks > 0) {
    const int64_t cap = num_seeds * fanout;
    hipLaunchKernelGGL(pad_tail_kernel<Idx>, dim3(grid1(cap > sinks ? cap : sinks + 1)), dim3(256), 0, s,
                       static_cast<Idx*>(out_indptr), num_seeds, cap, sinks, static_cast<const Idx*>(seeds),
                       num_valid, static_cast<Idx*>(out_src), static_cast<Idx*>(out_eids));
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx, typename F>
int run_sample_weighted(const dgla_csr* csc, const void* prob, const void* seeds, int64_t num_seeds,
                        int fanout, int replace, uint64_t rng_seed, void* out_indptr, void* out_src,
                        void* out_eids, char* ws, hipStream_t s, const int64_t* num_valid = nullptr,
                        const int64_t* rng_counter = nullptr, int sinks = 0) {
  Idx* counts = reinterpret_cast<Idx*>(ws);
  void* temp = ws + align256(sizeof(Idx) * (num_seeds + 1));
  size_t temp_bytes = scan_temp_bytes<Idx>(num_seeds + 1);
  const unsigned blocks = static_cast<unsigned>((num_seeds + 1 + 3) / 4);  // 4 waves = 4 rows per block
  hipLaunchKernelGGL((weighted_count_kernel<Idx, F>), dim3(blocks), dim3(256), 0, s,
                     static_cast<const Idx*>(csc->indptr), static_cast<const Idx*>(csc->data),
                     static_cast<const F*>(prob), static_cast<const Idx*>(seeds), num_seeds, fanout, replace,
                     counts, num_valid);
  if (msd::exclusive_scan<Idx, Idx>(counts, static_cast<Idx*>(out_indptr), num_seeds + 1, temp, s)) return -1;
  if (out_src && num_seeds > 0)
    hipLaunchKernelGGL((weighted_pick_kernel<Idx, F>), dim3(static_cast<unsigned>((num_seeds + 3) / 4)),
                       dim3(256), 0, s, static_cast<const Idx*>(csc->indptr),
                       static_cast<const Idx*>(csc->indices), static_cast<const Idx*>(csc->data),
                       static_cast<const F*>(prob), static_cast<const Idx*>(seeds), num_seeds, fanout, replace,
                       rng_seed, static_cast<const Idx*>(out_indptr), static_cast<Idx*>(out_src),
                       static_cast<Idx*>(out_eids), rng_counter);
  if (sinks > 0) {
    const int64_t cap = num_seeds * fanout;
    hipLaunchKernelGGL(pad_tail_kernel<Idx>, dim3(grid1(cap > sinks ? cap : sinks + 1)), dim3(256), 0, s,
                       static_cast<Idx*>(out_indptr), num_seeds, cap, sinks, static_cast<const Idx*>(seeds),
                       num_valid, static_cast<Idx*>(out_src), static_cast<Idx*>(out_eids));
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx>
int run_to_block(const void* seeds, int64_t num_seeds, const void* src, int64_t nnz, int32_t* node_map,
                 void* local_src, void* src_nodes, int64_t* num_src_out, char* ws, hipStream_t s,
                 const int64_t* num_valid = nullptr, int key_bits = 0) {
  const size_t a_sorted = align256(sizeof(Idx) * (nnz + 1));
  const size_t a_flag = align256(sizeof(int32_t) * (nnz + 1));
  Idx* sorted = reinterpret_cast<Idx*>(ws);
  int32_t* flag = reinterpret_cast<int32_t*>(ws + a_sorted);
  int32_t* rank = reinterpret_cast<int32_t*>(ws + a_sorted + a_flag);
  void* temp = ws + a_sorted + 2 * a_flag;
  hipLaunchKernelGGL(scatter_seed_ids_kernel<Idx>, dim3(grid1(num_seeds)), dim3(256), 0, s,
                     static_cast<const Idx*>(seeds), num_seeds, node_map, static_cast<Idx*>(src_nodes), num_valid);
  if (nnz > 0) {
    // (with a node count the sort looks at the bits the ids use: 3 passes at ogbn-products size; without one, at all
    // 31 / 63 value bits of the id type: 4 / 7 passes — no read-back either way)
    if (msd::sort_keys<Idx>(static_cast<const Idx*>(src), sorted, nnz, key_bits, static_cast<char*>(temp), s)) return -1;
  }
  hipLaunchKernelGGL(flag_new_nodes_kernel<Idx>, dim3(grid1(nnz + 1)), dim3(256), 0, s, sorted, nnz, node_map,
                     flag);
  if (msd::exclusive_scan<int32_t, int32_t>(flag, rank, nnz + 1, temp, s)) return -1;
  hipLaunchKernelGGL(assign_new_ids_kernel<Idx>, dim3(grid1(nnz + 1)), dim3(256), 0, s, sorted, nnz, flag, rank,
                     num_seeds, node_map, static_cast<Idx*>(src_nodes), num_src_out);
  if (nnz > 0)
    hipLaunchKernelGGL(relabel_kernel<Idx>, dim3(grid1(nnz)), dim3(256), 0, s, static_cast<const Idx*>(src), nnz,
                       node_map, static_cast<Idx*>(local_src));
  hipLaunchKernelGGL(reset_map_kernel<Idx>, dim3(grid1(num_seeds + nnz)), dim3(256), 0, s,
                     static_cast<const Idx*>(src_nodes), num_src_out, num_seeds + nnz, node_map);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx>
size_t to_block_ws(int64_t nnz) {
  const size_t scan_b = msd::scan_temp_bytes(nnz + 1, sizeof(int32_t));
  const size_t sort_b = nnz > 0 ? sort_keys_temp_bytes<Idx>(nnz) : 0;
  return align256(sizeof(Idx) * (nnz + 1)) + 2 * align256(sizeof(int32_t) * (nnz + 1)) +
         align256(scan_b > sort_b ? scan_b : sort_b);
}

}  // namespace
}  // namespace dgla

using namespace dgla;

extern "C" {

size_t dgla_sample_neighbors_workspace_bytes(int idtype_bits, int64_t num_seeds) {
  const size_t ib = idtype_bits / 8;
  const size_t scan = idtype_bits == 32 ? scan_temp_bytes<int32_t>(num_seeds + 1)
                                        : scan_temp_bytes<int64_t>(num_seeds + 1);
  return align256(ib * (num_seeds + 1)) + align256(scan);
}

int dgla_sample_neighbors(const dgla_csr* csc, const void* seeds, int64_t num_seeds, int fanout,
                          int replace, uint64_t rng_seed, void* out_indptr, void* out_src,
                          void* out_eids, void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!csc || !csc->indptr) return sfail("csc is null");
  if (csc->idtype_bits != 32 && csc->idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (num_seeds < 0) return sfail("negative number of seeds");
  if (fanout > kMaxFanout) return sfail("fanout larger than " + std::to_string(kMaxFanout) + " is not supported");
  if (fanout == 0 || fanout < -1) return sfail("fanout must be positive, or -1 for all neighbours");
  if (!out_indptr) return sfail("out_indptr is null");
  if (num_seeds > 0 && !seeds) return sfail("seeds is null");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out_indptr);
  const size_t need = dgla_sample_neighbors_workspace_bytes(csc->idtype_bits, num_seeds);
  void* owned = nullptr;
  if (!workspace || workspace_bytes < need) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, need, s));
    workspace = owned;
  }
  const int rc = csc->idtype_bits == 32
                     ? run_sample<int32_t>(csc, seeds, num_seeds, fanout, replace, rng_seed, out_indptr,
                                           out_src, out_eids, static_cast<char*>(workspace), s)
                     : run_sample<int64_t>(csc, seeds, num_seeds, fanout, replace, rng_seed, out_indptr,
                                           out_src, out_eids, static_cast<char*>(workspace), s);
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

int dgla_sample_neighbors_weighted(const dgla_csr* csc, const void* prob, dgla_dtype prob_dtype,
                                   const void* seeds, int64_t num_seeds, int fanout, int replace,
                                   uint64_t rng_seed, void* out_indptr, void* out_src, void* out_eids,
                                   void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (!csc || !csc->indptr) return sfail("csc is null");
  if (csc->idtype_bits != 32 && csc->idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (prob_dtype != DGLA_F32 && prob_dtype != DGLA_F64) return sfail("probabilities must be float32 or float64");
  if (!prob && csc->nnz > 0) return sfail("prob is null");
  if (num_seeds < 0) return sfail("negative number of seeds");
  if (fanout < 1 || fanout > kMaxFanout)
    return sfail("weighted sampling needs 1 <= fanout <= " + std::to_string(kMaxFanout));
  if (!out_indptr) return sfail("out_indptr is null");
  if (num_seeds > 0 && !seeds) return sfail("seeds is null");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out_indptr);
  const size_t need = dgla_sample_neighbors_workspace_bytes(csc->idtype_bits, num_seeds);
  void* owned = nullptr;
  if (!workspace || workspace_bytes < need) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, need, s));
    workspace = owned;
  }
  char* ws = static_cast<char*>(workspace);
  int rc;
  if (csc->idtype_bits == 32)
    rc = prob_dtype == DGLA_F32
             ? run_sample_weighted<int32_t, float>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed,
                                                   out_indptr, out_src, out_eids, ws, s)
             : run_sample_weighted<int32_t, double>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed,
                                                    out_indptr, out_src, out_eids, ws, s);
  else
    rc = prob_dtype == DGLA_F32
             ? run_sample_weighted<int64_t, float>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed,
                                                   out_indptr, out_src, out_eids, ws, s)
             : run_sample_weighted<int64_t, double>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed,
                                                    out_indptr, out_src, out_eids, ws, s);
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

int dgla_sample_neighbors_padded(const dgla_csr* csc, const void* prob, dgla_dtype prob_dtype, const void* seeds,
                                 int64_t num_seeds, const int64_t* num_valid, int fanout, int replace,
                                 uint64_t rng_seed, const int64_t* rng_counter, int sink_rows, void* out_indptr,
                                 void* out_src, void* out_eids, void* workspace, size_t workspace_bytes,
                                 void* hip_stream) {
  if (sink_rows < 1 || sink_rows > 4096) return sfail("sink_rows must be in [1, 4096]");
  if (prob && prob_dtype != DGLA_F32 && prob_dtype != DGLA_F64) return sfail("probabilities must be float32 or float64");
  if (!csc || !csc->indptr) return sfail("csc is null");
  if (csc->idtype_bits != 32 && csc->idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (num_seeds < 1) return sfail("the padded form needs at least one seed slot");
  if (fanout < 1 || fanout > kMaxFanout)
    return sfail("the padded form needs 1 <= fanout <= " + std::to_string(kMaxFanout));
  if (!out_indptr || !out_src || !out_eids || !seeds) return sfail("seeds / output arrays are null");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, out_indptr);
  const size_t need = dgla_sample_neighbors_workspace_bytes(csc->idtype_bits, num_seeds);
  if (!workspace || workspace_bytes < need)  // no allocation here: the call must be capturable
    return sfail("sample_neighbors_padded: workspace of " + std::to_string(need) + " bytes required");
  char* ws = static_cast<char*>(workspace);
  if (prob) {
    if (csc->idtype_bits == 32)
      return prob_dtype == DGLA_F32
                 ? run_sample_weighted<int32_t, float>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed, out_indptr,
                                                       out_src, out_eids, ws, s, num_valid, rng_counter, sink_rows)
                 : run_sample_weighted<int32_t, double>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed, out_indptr,
                                                        out_src, out_eids, ws, s, num_valid, rng_counter, sink_rows);
    return prob_dtype == DGLA_F32
               ? run_sample_weighted<int64_t, float>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed, out_indptr,
                                                     out_src, out_eids, ws, s, num_valid, rng_counter, sink_rows)
               : run_sample_weighted<int64_t, double>(csc, prob, seeds, num_seeds, fanout, replace, rng_seed, out_indptr,
                                                      out_src, out_eids, ws, s, num_valid, rng_counter, sink_rows);
  }
  return csc->idtype_bits == 32
             ? run_sample<int32_t>(csc, seeds, num_seeds, fanout, replace, rng_seed, out_indptr, out_src,
                                   out_eids, ws, s, num_valid, rng_counter, sink_rows)
             : run_sample<int64_t>(csc, seeds, num_seeds, fanout, replace, rng_seed, out_indptr, out_src,
                                   out_eids, ws, s, num_valid, rng_counter, sink_rows);
}

int dgla_to_block_padded(int idtype_bits, const void* seeds, int64_t num_seeds, const int64_t* num_valid,
                         const void* src, int64_t nnz, int64_t num_nodes, void* node_map, void* local_src,
                         void* src_nodes, int64_t* num_src_out, void* workspace, size_t workspace_bytes,
                         void* hip_stream) {
  int key_bits = 0;
  if (num_nodes > 0) {
    key_bits = 1;
    while ((int64_t(1) << key_bits) < num_nodes) ++key_bits;
  }
  if (idtype_bits != 32 && idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (num_seeds < 1 || nnz < 0) return sfail("bad size");
  if (!node_map || !src_nodes || !num_src_out || !seeds) return sfail("node_map / src_nodes / num_src_out / seeds is null");
  if (nnz > 0 && (!src || !local_src)) return sfail("input arrays are null");
  if (num_seeds + nnz > 0x7fffffffLL) return sfail("a block with more than 2^31-1 source nodes is not supported");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, node_map);
  const size_t need = dgla_to_block_workspace_bytes(idtype_bits, nnz);
  if (!workspace || workspace_bytes < need)
    return sfail("to_block_padded: workspace of " + std::to_string(need) + " bytes required");
  return idtype_bits == 32
             ? run_to_block<int32_t>(seeds, num_seeds, src, nnz, static_cast<int32_t*>(node_map), local_src,
                                     src_nodes, num_src_out, static_cast<char*>(workspace), s, num_valid, key_bits)
             : run_to_block<int64_t>(seeds, num_seeds, src, nnz, static_cast<int32_t*>(node_map), local_src,
                                     src_nodes, num_src_out, static_cast<char*>(workspace), s, num_valid, key_bits);
}

size_t dgla_to_block_workspace_bytes(int idtype_bits, int64_t nnz) {
  return idtype_bits == 32 ? to_block_ws<int32_t>(nnz) : to_block_ws<int64_t>(nnz);
}

int dgla_to_block(int idtype_bits, const void* seeds, int64_t num_seeds, const void* src, int64_t nnz,
                  void* node_map, void* local_src, void* src_nodes, int64_t* num_src_out,
                  void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (idtype_bits != 32 && idtype_bits != 64) return sfail("idtype must be int32 or int64");
  if (num_seeds < 0 || nnz < 0) return sfail("negative size");
  if (!node_map || !src_nodes || !num_src_out) return sfail("node_map / src_nodes / num_src_out is null");
  if ((num_seeds > 0 && !seeds) || (nnz > 0 && (!src || !local_src))) return sfail("input arrays are null");
  if (num_seeds + nnz > 0x7fffffffLL) return sfail("a block with more than 2^31-1 source nodes is not supported");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, node_map);
  const size_t need = dgla_to_block_workspace_bytes(idtype_bits, nnz);
  void* owned = nullptr;
  if (!workspace || workspace_bytes < need) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, need, s));
    workspace = owned;
  }
  const int rc = idtype_bits == 32
                     ? run_to_block<int32_t>(seeds, num_seeds, src, nnz, static_cast<int32_t*>(node_map), local_src,
                                             src_nodes, num_src_out, static_cast<char*>(workspace), s)
                     : run_to_block<int64_t>(seeds, num_seeds, src, nnz, static_cast<int32_t*>(node_map), local_src,
                                             src_nodes, num_src_out, static_cast<char*>(workspace), s);
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

}  // extern "C"
