// Micro-benchmark (gfx950): why does a kernel that reads 400-byte rows IN ORDER (segment reduce,
// docs/DESIGN_detail_r1_r5.md §3.6) move fewer lines per second than the same instruction stream reading them at
// RANDOM (the g-SpMM gather, 55 G lines/s)?  Every pattern below issues the merge kernel's load
// shape — 50 of 64 lanes, two 400-byte rows per instruction, U = 4 instructions in flight — and
// differs only in WHICH rows a wave has in flight at a time:
//   chunk        wave w streams rows [512 w, 512 w + 512), one unit per wave, one-pass grid
//   chunk64      the same with 64-row units
//   persistent   chunk, but 16 waves per CU loop over the units (stride = waves in the grid)
//   spread       chunk, the four loads in flight 128 rows (51 KB) apart inside the unit
//   lockstep     all waves of a persistent grid walk one contiguous window
//   flat         chunk read as flat bytes, 64 lanes x 16 B (no row structure)
//   random       rows through a hash (the gather's pattern)
//   *_nt         the same with non-temporal loads
// Build: hipcc --offload-arch=gfx950 -O3 seq_rows.hip -o seq_rows.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int kRowPieces = 25;  // 400-byte rows
constexpr int kU = 4;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

enum { CHUNK, CHUNK64, PERSISTENT, SPREAD, LOCKSTEP, FLAT, RANDOM };

template <int P, bool NT = false>
__global__ __launch_bounds__(256) void rows_kernel(const u4* __restrict__ a, uint32_t n_rows,
                                                   uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t waves = gridDim.x * 4u;
  const uint32_t g = lane / kRowPieces, piece = lane % kRowPieces;
  const bool on = lane < 2 * kRowPieces;
  u4 acc = {0, 0, 0, 0};
  auto row_ptr = [&](uint32_t r) { return a + size_t(r) * kRowPieces + piece; };
  auto ld = [&](const u4* p) { return NT ? __builtin_nontemporal_load(p) : *p; };
  if constexpr (P == CHUNK || P == CHUNK64 || P == PERSISTENT || P == SPREAD || P == RANDOM) {
    constexpr uint32_t unit = P == CHUNK64 ? 64 : 512;
    const uint32_t n_units = n_rows / unit;
    for (uint32_t un = wave; un < n_units; un += waves) {
      for (uint32_t s = 0; s < unit; s += 2 * kU) {
        u4 v[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) {
          uint32_t r;
          if constexpr (P == SPREAD)
            r = un * unit + k * (unit / kU) + s / kU + g;
          else
            r = un * unit + s + 2 * k + g;
          if constexpr (P == RANDOM) r = uint32_t(uint64_t(mix(r * 2654435761u + 7u)) * n_rows >> 32);
          v[k] = on ? ld(row_ptr(r)) : acc;
        }
#pragma unroll
        for (int k = 0; k < kU; ++k) acc += v[k];
      }
    }
  } else if constexpr (P == LOCKSTEP) {
    const uint32_t n_steps = n_rows / (2 * kU);
    for (uint32_t t = wave; t < n_steps; t += waves) {
      u4 v[kU];
#pragma unroll
      for (int k = 0; k < kU; ++k) v[k] = on ? ld(row_ptr(t * 2 * kU + 2 * k + g)) : acc;
#pragma unroll
      for (int k = 0; k < kU; ++k) acc += v[k];
    }
  } else {  // FLAT: the wave's 512-row chunk as 12 800 16-byte pieces, 64 per instruction
    const uint32_t n_units = n_rows / 512;
    for (uint32_t un = wave; un < n_units; un += waves) {
      const u4* base = a + size_t(un) * 512 * kRowPieces;
      for (uint32_t s = 0; s < 512 * kRowPieces; s += 64 * kU) {
        u4 v[kU];
#pragma unroll
        for (int k = 0; k < kU; ++k) v[k] = ld(base + s + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < kU; ++k) acc += v[k];
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345u) sink[blockIdx.x * 256u + threadIdx.x] = acc.x;
}

template <int P, bool NT = false>
void run(const char* name, const u4* a, uint32_t n_rows, uint32_t blocks, uint32_t* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rows_kernel<P, NT>), dim3(blocks), dim3(256), 0, 0, a, n_rows, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = double(n_rows) * 400.0;
  printf("{\"pattern\": \"%s\", \"blocks\": %u, \"ms\": %.3f, \"TB_per_s\": %.2f, \"G_lines_per_s\": %.1f}\n",
         name, blocks, best, bytes / best / 1e9, bytes / 128.0 / best / 1e6);
  fflush(stdout);
}

int main() {
  const uint32_t n_rows = 15u << 20;  // 15.7 M rows x 400 B = 6.3 GB
  u4* a;
  uint32_t* sink;
  hipMalloc(&a, size_t(n_rows) * 400);
  hipMalloc(&sink, size_t(n_rows / 512 / 4 * 8 + 8192) * 256 * 4);
  hipMemset(a, 1, size_t(n_rows) * 400);
  const uint32_t one_pass = n_rows / 512 / 4;
  run<CHUNK>("chunk", a, n_rows, one_pass, sink);
  run<CHUNK64>("chunk64", a, n_rows, n_rows / 64 / 4, sink);
  run<PERSISTENT>("persistent_16w", a, n_rows, 256 * 4, sink);
  run<PERSISTENT>("persistent_8w", a, n_rows, 256 * 2, sink);
  run<PERSISTENT>("persistent_32w", a, n_rows, 256 * 8, sink);
  run<SPREAD>("spread", a, n_rows, one_pass, sink);
  run<LOCKSTEP>("lockstep_16w", a, n_rows, 256 * 4, sink);
  run<LOCKSTEP>("lockstep_32w", a, n_rows, 256 * 8, sink);
  run<FLAT>("flat", a, n_rows, one_pass, sink);
  run<FLAT>("flat_persistent_16w", a, n_rows, 256 * 4, sink);
  run<CHUNK, true>("chunk_nt", a, n_rows, one_pass, sink);
  run<PERSISTENT, true>("persistent_16w_nt", a, n_rows, 256 * 4, sink);
  run<LOCKSTEP, true>("lockstep_16w_nt", a, n_rows, 256 * 4, sink);
  run<FLAT, true>("flat_nt", a, n_rows, one_pass, sink);
  run<FLAT, true>("flat_persistent_16w_nt", a, n_rows, 256 * 4, sink);
  run<RANDOM>("random", a, n_rows, one_pass, sink);
  run<RANDOM, true>("random_nt", a, n_rows, one_pass, sink);
  run<RANDOM>("random_persistent_16w", a, n_rows, 256 * 4, sink);
  return 0;
}
