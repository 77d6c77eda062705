// Micro-benchmark (gfx950): what does one random 16-byte gather cost at the L2 <-> fabric interface,
// by cache policy of the load and by where the array lives (39 MB = Infinity Cache resident, 4 GB =
// HBM)?  Question behind it (docs/DESIGN_detail_r1_r5.md §3.1): the merge kernel's fourth request per edge fetches a
// whole 128-byte line for a 16-byte row tail; is there a load flavour whose miss is a narrower
// request?  Build: hipcc --offload-arch=gfx950 -O3 gather_req.hip -o gather_req.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__device__ __forceinline__ u4 ld(const u4* p) {
  u4 v;
  if constexpr (MODE == 0) {
    v = *p;
  } else if constexpr (MODE == 1) {
    v = __builtin_nontemporal_load(p);
  } else if constexpr (MODE == 2) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  } else if constexpr (MODE == 3) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  } else if constexpr (MODE == 4) {
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  } else {
    asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  }
  return v;
}

// every lane: `iters` dependent-free random 16-byte loads, 8 in flight
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const u4* __restrict__ a, uint32_t n_pieces,
                                                     int iters, uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  uint32_t s = mix(tid * 2654435761u + 12345u);
  u4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it += 8) {
    u4 v[8];
    if constexpr (MODE <= 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s = mix(s + 0x9e3779b9u);
        v[k] = ld<MODE>(a + (uint64_t(s) * n_pieces >> 32));
      }
    } else {
      // inline-asm loads: issue 8, one wait at the end
      const u4* p[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s = mix(s + 0x9e3779b9u);
        p[k] = a + (uint64_t(s) * n_pieces >> 32);
      }
#define LD8(POL)                                                                          \
  asm volatile("global_load_dwordx4 %0, %8, off " POL "\n global_load_dwordx4 %1, %9, off " POL  \
               "\n global_load_dwordx4 %2, %10, off " POL "\n global_load_dwordx4 %3, %11, off " POL \
               "\n global_load_dwordx4 %4, %12, off " POL "\n global_load_dwordx4 %5, %13, off " POL \
               "\n global_load_dwordx4 %6, %14, off " POL "\n global_load_dwordx4 %7, %15, off " POL \
               "\n s_waitcnt vmcnt(0)"                                                     \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]),    \
                 "=&v"(v[6]), "=&v"(v[7])                                                  \
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]),   \
                 "v"(p[7])                                                                 \
               : "memory")
      if constexpr (MODE == 2) LD8("sc1");
      if constexpr (MODE == 3) LD8("sc0 sc1");
      if constexpr (MODE == 4) LD8("sc0 sc1 nt");
      if constexpr (MODE == 5) LD8("sc0");
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 0x12345u) sink[tid] = acc.x;
}

template <int MODE>
void run(const char* name, const u4* a, uint32_t n_pieces, const char* where, uint32_t* sink) {
  const int blocks = 256 * 32, iters = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, a, n_pieces, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double loads = double(blocks) * 256 * iters;
  printf("{\"array\": \"%s\", \"load\": \"%s\", \"ms\": %.3f, \"G_loads_per_s\": %.2f}\n", where, name, best,
         loads / best / 1e6);
}

int main() {
  const size_t small = size_t(39) << 20, big = size_t(4) << 30;
  u4 *a, *b;
  uint32_t* sink;
  hipMalloc(&a, small);
  hipMalloc(&b, big);
  hipMalloc(&sink, 256 * 32 * 256 * 4);
  hipMemset(a, 1, small);
  hipMemset(b, 1, big);
  for (int w = 0; w < 2; ++w) {
    const u4* p = w ? b : a;
    const uint32_t n = uint32_t((w ? big : small) / 16);
    const char* where = w ? "4GB_hbm" : "39MB_infinity_cache";
    run<0>("plain", p, n, where, sink);
    run<1>("nt", p, n, where, sink);
    run<2>("sc1", p, n, where, sink);
    run<3>("sc0_sc1", p, n, where, sink);
    run<4>("sc0_sc1_nt", p, n, where, sink);
    run<5>("sc0", p, n, where, sink);
  }
  return 0;
}
