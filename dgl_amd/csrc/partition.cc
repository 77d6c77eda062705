// Native k-way node-cut partitioner for the multi-GPU row split.
//
// Takes the place of METIS in the reference's `metis_partition_assignment`
// (python/dgl/partition.py:278-397 -> _CAPI_DGLMetisPartition_Hetero; METIS itself is an
// un-vendored submodule, third_party/METIS is empty in the checkout).  Written from the
// published multilevel scheme, in the label-propagation flavour that suits power-law graphs
// (Meyerhenke, Sanders, Schulz: "Partitioning complex networks via size-constrained
// clustering", SEA 2014), not from METIS code:
//
//   1. symmetrise the input (the reference does the same, partition.py:319-327);
//   2. COARSEN: size-constrained label propagation clusters the graph (a cluster may not
//      outgrow max_part_weight / kClusterFactor), clusters are contracted into weighted
//      vertices / edges; repeat until the graph is small or stops shrinking;
//   3. INITIAL PARTITION of the coarsest graph: greedy graph growing from k seeds picked far
//      apart, vertices handed to the lightest part they touch;
//   4. UNCOARSEN: project, then refine each level with size-constrained label propagation
//      (a vertex moves to the neighbouring part it is connected to most strongly if that
//      part has room), plus a rebalancing pass when a part is over the limit.
//
// Vertex weight = 1 (+ in-degree when balance_edges, the reference's flag of the same name):
// the SpMM work of a row partition is its in-edge count.  Deterministic: fixed visiting
// orders and a seeded xorshift, single-threaded host code (this is offline preprocessing,
// exactly as METIS is in the reference).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/dgl_amd.h"

namespace dgla {
std::string& last_error();
}

namespace {

using vid = int32_t;
using wgt = int64_t;

struct Graph {
  vid n = 0;
  std::vector<int64_t> xadj;  // n + 1
  std::vector<vid> adj;
  std::vector<wgt> ewgt;
  std::vector<wgt> vwgt;
  wgt total_vwgt = 0;
};

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
  uint64_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
  }
  uint64_t below(uint64_t n) { return next() % n; }
};

template <typename Idx>
Graph symmetrise(int64_t n, const Idx* indptr, const Idx* indices, bool balance_edges) {
  // undirected multigraph of the CSR (rows = destinations): u -- v for every stored edge,
  // parallel edges merged into weights, self loops dropped
  Graph g;
  g.n = static_cast<vid>(n);
  std::vector<int64_t> deg(n + 1, 0);
  for (int64_t r = 0; r < n; ++r)
    for (Idx j = indptr[r]; j < indptr[r + 1]; ++j) {
      const int64_t c = static_cast<int64_t>(indices[j]);
      if (c == r || c < 0 || c >= n) continue;
      ++deg[r + 1];
      ++deg[c + 1];
    }
  std::partial_sum(deg.begin(), deg.end(), deg.begin());
  std::vector<vid> tmp(deg[n]);
  std::vector<int64_t> fill(deg.begin(), deg.end() - 1);
  for (int64_t r = 0; r < n; ++r)
    for (Idx j = indptr[r]; j < indptr[r + 1]; ++j) {
      const int64_t c = static_cast<int64_t>(indices[j]);
      if (c == r || c < 0 || c >= n) continue;
      tmp[fill[r]++] = static_cast<vid>(c);
      tmp[fill[c]++] = static_cast<vid>(r);
    }
  g.xadj.assign(n + 1, 0);
  g.adj.reserve(tmp.size());
  g.ewgt.reserve(tmp.size());
  for (int64_t v = 0; v < n; ++v) {
    std::sort(tmp.begin() + deg[v], tmp.begin() + deg[v + 1]);
    for (int64_t j = deg[v]; j < deg[v + 1];) {
      int64_t e = j;
      while (e < deg[v + 1] && tmp[e] == tmp[j]) ++e;
      g.adj.push_back(tmp[j]);
      g.ewgt.push_back(e - j);
      j = e;
    }
    g.xadj[v + 1] = static_cast<int64_t>(g.adj.size());
  }
  g.vwgt.assign(n, 1);
  if (balance_edges)
    for (int64_t r = 0; r < n; ++r) g.vwgt[r] += static_cast<wgt>(indptr[r + 1] - indptr[r]);
  g.total_vwgt = std::accumulate(g.vwgt.begin(), g.vwgt.end(), wgt(0));
  return g;
}

// Size-constrained label propagation.  `label` holds the current cluster / part of every
// vertex and `load` the weight of every label.  A vertex adopts the label it is connected to
// with the largest edge weight among those with room (ties: keep the current one, else the
// lighter label).  Returns the number of moves of the last sweep.
int64_t label_propagation(const Graph& g, std::vector<vid>& label, std::vector<wgt>& load,
                          wgt max_load, int sweeps, Rng& rng, bool random_order, int64_t min_moves = 0) {
  const vid n = g.n;
  std::vector<vid> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::vector<wgt> conn(load.size(), 0);
  std::vector<vid> touched;
  int64_t moved = 0;
  for (int it = 0; it < sweeps; ++it) {
    if (random_order)
      for (vid i = n - 1; i > 0; --i) std::swap(order[i], order[rng.below(i + 1)]);
    moved = 0;
    for (vid oi = 0; oi < n; ++oi) {
      const vid v = order[oi];
      const vid cur = label[v];
      touched.clear();
      for (int64_t j = g.xadj[v]; j < g.xadj[v + 1]; ++j) {
        const vid l = label[g.adj[j]];
        if (conn[l] == 0) touched.push_back(l);
        conn[l] += g.ewgt[j];
      }
      vid best = cur;
      wgt best_conn = conn[cur];
      for (vid l : touched) {
        if (l == cur) continue;
        if (load[l] + g.vwgt[v] > max_load) continue;
        if (conn[l] > best_conn || (conn[l] == best_conn && best != cur && load[l] < load[best])) {
          best = l;
          best_conn = conn[l];
        }
      }
      for (vid l : touched) conn[l] = 0;
      if (best != cur) {
        load[cur] -= g.vwgt[v];
        load[best] += g.vwgt[v];
        label[v] = best;
        ++moved;
      }
    }
    if (moved == 0 || moved < min_moves) break;  // converged (min_moves > 0: or nearly; unused — stopping
                                                 // refinement at 0.5 % moves tripled the cut on planted communities)
  }
  return moved;
}

// Contract the clusters of `label` (values in [0, n), sparse) into a coarse graph.
Graph contract(const Graph& g, std::vector<vid>& label, vid* num_coarse) {
  const vid n = g.n;
  std::vector<vid> remap(n, -1);
  vid nc = 0;
  for (vid v = 0; v < n; ++v) {
    if (remap[label[v]] < 0) remap[label[v]] = nc++;
    label[v] = remap[label[v]];
  }
  *num_coarse = nc;
  Graph c;
  c.n = nc;
  c.vwgt.assign(nc, 0);
  for (vid v = 0; v < n; ++v) c.vwgt[label[v]] += g.vwgt[v];
  c.total_vwgt = g.total_vwgt;
  // members of each coarse vertex
  std::vector<int64_t> start(nc + 1, 0);
  for (vid v = 0; v < n; ++v) ++start[label[v] + 1];
  std::partial_sum(start.begin(), start.end(), start.begin());
  std::vector<vid> members(n);
  {
    std::vector<int64_t> pos(start.begin(), start.end() - 1);
    for (vid v = 0; v < n; ++v) members[pos[label[v]]++] = v;
  }
  c.xadj.assign(nc + 1, 0);
  std::vector<wgt> acc(nc, 0);
  std::vector<vid> touched;
  for (vid cv = 0; cv < nc; ++cv) {
    touched.clear();
    for (int64_t m = start[cv]; m < start[cv + 1]; ++m) {
      const vid v = members[m];
      for (int64_t j = g.xadj[v]; j < g.xadj[v + 1]; ++j) {
        const vid cu = label[g.adj[j]];
        if (cu == cv) continue;
        if (acc[cu] == 0) touched.push_back(cu);
        acc[cu] += g.ewgt[j];
      }
    }
    std::sort(touched.begin(), touched.end());
    for (vid cu : touched) {
      c.adj.push_back(cu);
      c.ewgt.push_back(acc[cu]);
      acc[cu] = 0;
    }
    c.xadj[cv + 1] = static_cast<int64_t>(c.adj.size());
  }
  return c;
}

// Greedy graph growing on the (small) coarsest graph.
void initial_partition(const Graph& g, int k, wgt max_load, std::vector<vid>& part,
                       std::vector<wgt>& load, Rng& rng) {
  const vid n = g.n;
  part.assign(n, -1);
  load.assign(k, 0);
  // seeds: the heaviest unassigned vertex that is not adjacent to an earlier seed
  std::vector<vid> by_weight(n);
  std::iota(by_weight.begin(), by_weight.end(), 0);
  std::stable_sort(by_weight.begin(), by_weight.end(),
                   [&](vid a, vid b) { return g.vwgt[a] > g.vwgt[b]; });
  std::vector<char> near_seed(n, 0);
  std::vector<std::vector<vid>> frontier(k);
  int p = 0;
  for (vid v : by_weight) {
    if (p == k) break;
    if (near_seed[v]) continue;
    part[v] = p;
    load[p] += g.vwgt[v];
    frontier[p].push_back(v);
    near_seed[v] = 1;
    for (int64_t j = g.xadj[v]; j < g.xadj[v + 1]; ++j) near_seed[g.adj[j]] = 1;
    ++p;
  }
  for (vid v : by_weight) {  // fewer than k independent seeds: take anything unassigned
    if (p == k) break;
    if (part[v] >= 0) continue;
    part[v] = p;
    load[p] += g.vwgt[v];
    frontier[p].push_back(v);
    ++p;
  }
  // grow: always extend the lightest part that still has a frontier
  std::vector<size_t> head(k, 0);
  vid assigned = 0;
  for (vid v = 0; v < n; ++v) assigned += part[v] >= 0;
  while (assigned < n) {
    int best = -1;
    for (int q = 0; q < k; ++q)
      if (head[q] < frontier[q].size() && (best < 0 || load[q] < load[best])) best = q;
    if (best < 0) {  // disconnected remainder: give the next free vertex to the lightest part
      int light = 0;
      for (int q = 1; q < k; ++q)
        if (load[q] < load[light]) light = q;
      for (vid v : by_weight)
        if (part[v] < 0) {
          part[v] = light;
          load[light] += g.vwgt[v];
          frontier[light].push_back(v);
          ++assigned;
          break;
        }
      continue;
    }
    const vid v = frontier[best][head[best]++];
    for (int64_t j = g.xadj[v]; j < g.xadj[v + 1]; ++j) {
      const vid u = g.adj[j];
      if (part[u] >= 0) continue;
      if (load[best] + g.vwgt[u] > max_load) continue;
      part[u] = best;
      load[best] += g.vwgt[u];
      frontier[best].push_back(u);
      ++assigned;
    }
    if (head[best] == frontier[best].size()) {
      // exhausted without room for its neighbours: they will be picked up by other parts or
      // by the disconnected-remainder rule above
      bool any = false;
      for (int q = 0; q < k; ++q) any = any || head[q] < frontier[q].size();
      if (!any) {
        int light = 0;
        for (int q = 1; q < k; ++q)
          if (load[q] < load[light]) light = q;
        for (vid u : by_weight)
          if (part[u] < 0) {
            part[u] = light;
            load[light] += g.vwgt[u];
            frontier[light].push_back(u);
            ++assigned;
            break;
          }
      }
    }
  }
  (void)rng;
}

// Move boundary vertices out of overloaded parts into the neighbouring (else lightest) part
// with room, cheapest cut increase first.
void rebalance(const Graph& g, int k, wgt max_load, std::vector<vid>& part, std::vector<wgt>& load) {
  for (int round = 0; round < 8; ++round) {
    bool over = false;
    for (int q = 0; q < k; ++q) over = over || load[q] > max_load;
    if (!over) return;
    std::vector<wgt> conn(k, 0);
    for (vid v = 0; v < g.n; ++v) {
      const vid cur = part[v];
      if (load[cur] <= max_load) continue;
      std::fill(conn.begin(), conn.end(), 0);
      for (int64_t j = g.xadj[v]; j < g.xadj[v + 1]; ++j) conn[part[g.adj[j]]] += g.ewgt[j];
      int best = -1;
      for (int q = 0; q < k; ++q) {
        if (q == cur || load[q] + g.vwgt[v] > max_load) continue;
        if (best < 0 || conn[q] > conn[best] || (conn[q] == conn[best] && load[q] < load[best])) best = q;
      }
      if (best < 0) continue;
      load[cur] -= g.vwgt[v];
      load[best] += g.vwgt[v];
      part[v] = best;
    }
  }
}

wgt edge_cut(const Graph& g, const std::vector<vid>& part) {
  wgt cut = 0;
  for (vid v = 0; v < g.n; ++v)
    for (int64_t j = g.xadj[v]; j < g.xadj[v + 1]; ++j)
      if (part[g.adj[j]] != part[v]) cut += g.ewgt[j];
  return cut / 2;
}

constexpr int kClusterFactor = 18;  // coarse vertices per part at most this heavy: max_load / 18

template <typename Idx>
int partition_impl(int64_t n, const Idx* indptr, const Idx* indices, int k, double imbalance,
                   bool balance_edges, uint64_t seed, int64_t* out_part, int64_t* stats) {
  const bool trace = std::getenv("DGLA_PARTITION_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = now();
  auto lap = [&](const char* what) {
    if (trace) { const double t = now(); std::fprintf(stderr, "[partition] %-28s %.2f s\n", what, t - t_last); t_last = t; }
  };
  Graph g0 = symmetrise(n, indptr, indices, balance_edges);
  lap("symmetrise");
  const wgt avg = (g0.total_vwgt + k - 1) / k;
  wgt max_vw = 0;
  for (wgt w : g0.vwgt) max_vw = std::max(max_vw, w);
  // a part must at least be able to hold the heaviest vertex
  const wgt max_load = std::max<wgt>(static_cast<wgt>(avg * (1.0 + imbalance)) + 1, max_vw);
  Rng rng(seed);

  // ---- coarsen ---------------------------------------------------------------------
  std::vector<Graph> levels;
  std::vector<std::vector<vid>> maps;  // maps[l][v] = coarse vertex of v at level l + 1
  levels.push_back(std::move(g0));
  const vid stop_at = std::max<vid>(static_cast<vid>(k) * 64, 2048);
  while (levels.back().n > stop_at && levels.size() < 24) {
    const Graph& g = levels.back();
    std::vector<vid> label(g.n);
    std::iota(label.begin(), label.end(), 0);
    std::vector<wgt> load(g.vwgt);
    const wgt cluster_cap = std::max<wgt>(max_load / kClusterFactor, 1);
    label_propagation(g, label, load, cluster_cap, 3, rng, true);
    lap("coarsen: label propagation");
    vid nc = 0;
    Graph c = contract(g, label, &nc);
    lap("coarsen: contract");
    if (nc > g.n * 0.95) break;  // no longer shrinking (e.g. isolated vertices only)
    maps.push_back(std::move(label));
    levels.push_back(std::move(c));
  }

  // ---- initial partition -------------------------------------------------------------
  std::vector<vid> part;
  std::vector<wgt> load;
  initial_partition(levels.back(), k, max_load, part, load, rng);
  label_propagation(levels.back(), part, load, max_load, 12, rng, true);
  rebalance(levels.back(), k, max_load, part, load);

  // ---- uncoarsen + refine ---------------------------------------------------------------
  for (int l = static_cast<int>(levels.size()) - 2; l >= 0; --l) {
    const Graph& g = levels[l];
    std::vector<vid> fine(g.n);
    for (vid v = 0; v < g.n; ++v) fine[v] = part[maps[l][v]];
    part.swap(fine);
    label_propagation(g, part, load, max_load, l == 0 ? 6 : 8, rng, true);
    rebalance(g, k, max_load, part, load);
    lap("refine level");
  }
  for (int64_t v = 0; v < n; ++v) out_part[v] = part[v];
  if (stats) {
    stats[0] = edge_cut(levels[0], part);  // weight of cut undirected edges (= stored edges cut)
    stats[1] = *std::max_element(load.begin(), load.end());
    stats[2] = avg;
    stats[3] = static_cast<int64_t>(levels.size());
  }
  return 0;
}

}  // namespace

extern "C" int dgla_partition_kway(int idtype_bits, int64_t num_nodes, const void* indptr,
                                   const void* indices, int num_parts, double imbalance,
                                   int balance_edges, uint64_t seed, int64_t* out_part,
                                   int64_t* stats) {
  auto fail = [](const char* m) {
    dgla::last_error() = m;
    return -1;
  };
  if (idtype_bits != 32 && idtype_bits != 64) return fail("idtype must be int32 or int64");
  if (num_nodes < 0 || num_nodes > 0x7fffffffLL) return fail("num_nodes out of range");
  if (num_parts < 1) return fail("num_parts must be >= 1");
  if (num_nodes > 0 && (!indptr || !out_part)) return fail("indptr / out_part is null");
  if (imbalance < 0) return fail("imbalance must be >= 0");
  if (num_nodes == 0) return 0;
  if (num_parts == 1) {
    std::memset(out_part, 0, sizeof(int64_t) * num_nodes);
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    return 0;
  }
  try {
    if (idtype_bits == 32)
      return partition_impl<int32_t>(num_nodes, static_cast<const int32_t*>(indptr),
                                     static_cast<const int32_t*>(indices), num_parts, imbalance,
                                     balance_edges != 0, seed, out_part, stats);
    return partition_impl<int64_t>(num_nodes, static_cast<const int64_t*>(indptr),
                                   static_cast<const int64_t*>(indices), num_parts, imbalance,
                                   balance_edges != 0, seed, out_part, stats);
  } catch (const std::exception& e) {
    dgla::last_error() = e.what();
    return -1;
  }
}
