// Auxiliary gfx950 kernels: COO g-SpMM (atomics), streaming copy.
#include "common.h"

namespace dgla {

// ---------------------------------------------------------------------------------------
// Streaming copy: 16 bytes per lane, grid-stride.  bench.py uses it to measure the HBM
// peak the roofline fraction is quoted against (MI355X_MICROARCH.md: 6.29 TB/s float4 copy).
// ---------------------------------------------------------------------------------------
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

// MODE 0: copy with non-temporal loads and stores; 1: copy with default cache policy;
// 2: read only (xor-reduce, one 16-byte store per lane at the end) — the read-side peak, which
// is what a gather-dominated kernel (97 % reads) is really up against.
template <int U, int MODE>
__global__ __launch_bounds__(256) void stream_copy_kernel(const u32x4* __restrict__ src,
                                                          u32x4* __restrict__ dst, size_t n) {
  // U independent 16-byte loads per lane in flight, then U stores; blocks walk the array in
  // 256 * U * 16-byte tiles, grid-strided.
  const size_t tile = static_cast<size_t>(blockDim.x) * U;
  const size_t stride = static_cast<size_t>(gridDim.x) * tile;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (size_t base = blockIdx.x * tile + threadIdx.x; base < n; base += stride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * blockDim.x;
      if (i < n) {
        if constexpr (MODE == 1)
          v[u] = src[i];
        else
          v[u] = __builtin_nontemporal_load(src + i);
      } else {
        v[u] = acc;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + static_cast<size_t>(u) * blockDim.x;
      if constexpr (MODE == 2) {
        acc ^= v[u];
      } else if (i < n) {
        if constexpr (MODE == 1)
          dst[i] = v[u];
        else
          __builtin_nontemporal_store(v[u], dst + i);
      }
    }
  }
  if constexpr (MODE == 2) dst[blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x] = acc;
}

// variant = MODE + 4 * (blocks-per-CU selector) + 16 * (unroll selector) + 32 * (one tile per
// workgroup); 0 is the default
// used for the roofline denominator, the others exist for benchmarks/bench_peak.py.
int launch_stream_copy(void* dst, const void* src, size_t bytes, int variant, hipStream_t stream) {
  const size_t n = bytes / 16;
  if (n == 0) return 0;
  const int mode = variant & 3;
  const int bpc = 1 << (((variant >> 2) & 3) + 2);  // 4, 8, 16, 32 blocks per CU
  const int usel = (variant >> 4) & 1;               // 0: U = 4, 1: U = 8
  const size_t tile = 256 * static_cast<size_t>(usel ? 8 : 4);
  size_t blocks = (n + tile - 1) / tile;
  // bit 5: ONE tile per workgroup, no grid-stride loop (a grid of millions of short workgroups —
  // the shape of the split-row copy, which moves 5.9 TB/s where the persistent grid moves 5.0-5.2)
  if (!(variant & 32) && blocks > static_cast<size_t>(256 * bpc)) blocks = 256 * bpc;
  if (blocks > 0x7fffffffull) blocks = 0x7fffffffull;
  const dim3 g(static_cast<unsigned>(blocks)), b(256);
  const u32x4* s = static_cast<const u32x4*>(src);
  u32x4* d = static_cast<u32x4*>(dst);
#define DGLA_COPY(UU, MM) hipLaunchKernelGGL((stream_copy_kernel<UU, MM>), g, b, 0, stream, s, d, n)
  if (usel) {
    if (mode == 0) DGLA_COPY(8, 0); else if (mode == 1) DGLA_COPY(8, 1); else DGLA_COPY(8, 2);
  } else {
    if (mode == 0) DGLA_COPY(4, 0); else if (mode == 1) DGLA_COPY(4, 1); else DGLA_COPY(4, 2);
  }
#undef DGLA_COPY
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// COO g-SpMM.  Replaces SpMMCooKernel / ArgSpMMCooKernel (src/array/cuda/spmm.cuh:410-487,
// host :624-682): fill with the reducer identity, edge-parallel atomics into out[dst].
// Unlike the reference's racy second pass (ties -> last writer), arg results are made
// deterministic: among the edges that attain the extremum the LOWEST position wins, the
// same answer the CSR path and the sequential CPU loop give.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void fill_kernel(T* p, int64_t n, T v) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride)
    p[i] = v;
}

__device__ __forceinline__ void atomic_max_f(float* a, float v) {
  int old = __float_as_int(*a);
  while (__int_as_float(old) < v) {
    const int assumed = old;
    old = atomicCAS(reinterpret_cast<int*>(a), assumed, __float_as_int(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_min_f(float* a, float v) {
  int old = __float_as_int(*a);
  while (__int_as_float(old) > v) {
    const int assumed = old;
    old = atomicCAS(reinterpret_cast<int*>(a), assumed, __float_as_int(v));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_max_f(double* a, double v) {
  unsigned long long old = __double_as_longlong(*a);
  while (__longlong_as_double(old) < v) {
    const unsigned long long assumed = old;
    old = atomicCAS(reinterpret_cast<unsigned long long*>(a), assumed,
                    static_cast<unsigned long long>(__double_as_longlong(v)));
    if (old == assumed) break;
  }
}
__device__ __forceinline__ void atomic_min_f(double* a, double v) {
  unsigned long long old = __double_as_longlong(*a);
  while (__longlong_as_double(old) > v) {
    const unsigned long long assumed = old;
    old = atomicCAS(reinterpret_cast<unsigned long long*>(a), assumed,
                    static_cast<unsigned long long>(__double_as_longlong(v)));
    if (old == assumed) break;
  }
}

template <typename Idx>
__device__ __forceinline__ void atomic_min_idx(Idx* a, Idx v);
template <>
__device__ __forceinline__ void atomic_min_idx<int32_t>(int32_t* a, int32_t v) {
  atomicMin(a, v);
}
template <>
__device__ __forceinline__ void atomic_min_idx<int64_t>(int64_t* a, int64_t v) {
  atomicMin(reinterpret_cast<long long*>(a), static_cast<long long>(v));
}

struct CooSpmmParams {
  const void* row;
  const void* col;
  const void* eids;
  int64_t nnz, num_dst;
  const void* ufeat;
  const void* efeat;
  void* out;
  void* arg_u;
  void* arg_e;
  int out_len, lhs_len, rhs_len;
  int use_bcast;
  BcastDims bd;
};

template <int OP, typename T>
__device__ __forceinline__ T coo_apply(T l, T r) {
  if constexpr (OP == kAdd) return l + r;
  if constexpr (OP == kSub) return l - r;
  if constexpr (OP == kMul) return l * r;
  if constexpr (OP == kDiv) return l / r;
  if constexpr (OP == kCopyLhs) return l;
  return r;
}

// PASS: 0 = reduce values, 1 = record the lowest position attaining the extremum.
template <typename Idx, typename DT, int OP, int RED, int PASS>
__global__ __launch_bounds__(256) void spmm_coo_kernel(const CooSpmmParams p) {
  constexpr bool UL = op_uses_lhs(OP), UR = op_uses_rhs(OP);
  const Idx* __restrict__ row = static_cast<const Idx*>(p.row);
  const Idx* __restrict__ col = static_cast<const Idx*>(p.col);
  const Idx* __restrict__ eids = static_cast<const Idx*>(p.eids);
  const DT* __restrict__ X = static_cast<const DT*>(p.ufeat);
  const DT* __restrict__ W = static_cast<const DT*>(p.efeat);
  DT* out = static_cast<DT*>(p.out);
  Idx* posbuf = static_cast<Idx*>(UR ? p.arg_e : p.arg_u);
  const int64_t total = p.nnz * p.out_len;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; idx < total;
       idx += stride) {
    const int64_t i = idx / p.out_len;
    const int k = static_cast<int>(idx - i * p.out_len);
    const int64_t src = row[i], dst = col[i];
    const int64_t eid = eids ? static_cast<int64_t>(eids[i]) : i;
    int lo = k, ro = k;
    if (p.use_bcast) bcast_offsets(p.bd, k, &lo, &ro);
    DT l = DT(0), r = DT(0);
    if constexpr (UL) l = X[src * p.lhs_len + lo];
    if constexpr (UR) r = W[eid * p.rhs_len + ro];
    const DT val = coo_apply<OP, DT>(l, r);
    DT* o = out + dst * p.out_len + k;
    if constexpr (PASS == 0) {
      if constexpr (RED == kSum) {
        atomicAdd(o, val);
      } else if constexpr (RED == kMax) {
        atomic_max_f(o, val);
      } else {
        atomic_min_f(o, val);
      }
    } else {
      if (val == *o) atomic_min_idx<Idx>(posbuf + dst * p.out_len + k, static_cast<Idx>(i));
    }
  }
}

// pass 2: translate the winning position into (source id, edge id); untouched -> 0.
template <typename Idx, int OP>
__global__ __launch_bounds__(256) void spmm_coo_arg_finish_kernel(const CooSpmmParams p, Idx sentinel) {
  constexpr bool UL = op_uses_lhs(OP), UR = op_uses_rhs(OP);
  const Idx* __restrict__ row = static_cast<const Idx*>(p.row);
  const Idx* __restrict__ eids = static_cast<const Idx*>(p.eids);
  Idx* argu = static_cast<Idx*>(p.arg_u);
  Idx* arge = static_cast<Idx*>(p.arg_e);
  Idx* posbuf = UR ? arge : argu;
  const int64_t total = p.num_dst * p.out_len;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; idx < total;
       idx += stride) {
    const Idx pos = posbuf[idx];
    const bool hit = pos != sentinel;
    if constexpr (UL) argu[idx] = hit ? row[pos] : Idx(0);
    if constexpr (UR) arge[idx] = hit ? (eids ? eids[pos] : pos) : Idx(0);
  }
}

static unsigned grid_for(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 256 * 32) b = 256 * 32;
  if (b < 1) b = 1;
  return static_cast<unsigned>(b);
}

template <typename Idx, typename DT, int OP, int RED>
static int spmm_coo_run(const CooSpmmParams& p, hipStream_t s) {
  const int64_t nout = p.num_dst * p.out_len;
  DT ident = DT(0);
  if (RED == kMax) ident = -static_cast<DT>(__builtin_huge_val());
  if (RED == kMin) ident = static_cast<DT>(__builtin_huge_val());
  hipLaunchKernelGGL((fill_kernel<DT>), dim3(grid_for(nout)), dim3(256), 0, s,
                     static_cast<DT*>(p.out), nout, ident);
  const unsigned g = grid_for(p.nnz * p.out_len);
  hipLaunchKernelGGL((spmm_coo_kernel<Idx, DT, OP, RED, 0>), dim3(g), dim3(256), 0, s, p);
  if (RED != kSum) {
    const Idx sentinel = static_cast<Idx>((~static_cast<uint64_t>(0)) >> (65 - 8 * sizeof(Idx)));
    Idx* posbuf = static_cast<Idx*>(op_uses_rhs(OP) ? p.arg_e : p.arg_u);
    hipLaunchKernelGGL((fill_kernel<Idx>), dim3(grid_for(nout)), dim3(256), 0, s, posbuf, nout,
                       sentinel);
    hipLaunchKernelGGL((spmm_coo_kernel<Idx, DT, OP, RED, 1>), dim3(g), dim3(256), 0, s, p);
    hipLaunchKernelGGL((spmm_coo_arg_finish_kernel<Idx, OP>), dim3(grid_for(nout)), dim3(256), 0,
                       s, p, sentinel);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx, typename DT, int OP>
static int spmm_coo_red(const CooSpmmParams& p, int red, hipStream_t s) {
  switch (red) {
    case kSum: return spmm_coo_run<Idx, DT, OP, kSum>(p, s);
    case kMax: return spmm_coo_run<Idx, DT, OP, kMax>(p, s);
    case kMin: return spmm_coo_run<Idx, DT, OP, kMin>(p, s);
  }
  last_error() = "unsupported SpMM reducer";
  return -1;
}

template <typename Idx, typename DT>
static int spmm_coo_op(const CooSpmmParams& p, int op, int red, hipStream_t s) {
  switch (op) {
    case kAdd: return spmm_coo_red<Idx, DT, kAdd>(p, red, s);
    case kSub: return spmm_coo_red<Idx, DT, kSub>(p, red, s);
    case kMul: return spmm_coo_red<Idx, DT, kMul>(p, red, s);
    case kDiv: return spmm_coo_red<Idx, DT, kDiv>(p, red, s);
    case kCopyLhs: return spmm_coo_red<Idx, DT, kCopyLhs>(p, red, s);
    case kCopyRhs: return spmm_coo_red<Idx, DT, kCopyRhs>(p, red, s);
  }
  last_error() = "unsupported SpMM binary operator";
  return -1;
}

int launch_spmm_coo(const CooView& coo, int op, int red, int dtype, const void* ufeat,
                    const void* efeat, void* out, void* arg_u, void* arg_e, int64_t out_len,
                    int64_t lhs_len, int64_t rhs_len, bool use_bcast, const BcastDims& bd,
                    hipStream_t stream) {
  CooSpmmParams p;
  p.row = coo.row;
  p.col = coo.col;
  p.eids = coo.eids;
  p.nnz = coo.nnz;
  p.num_dst = coo.num_cols;
  p.ufeat = ufeat;
  p.efeat = efeat;
  p.out = out;
  p.arg_u = arg_u;
  p.arg_e = arg_e;
  p.out_len = static_cast<int>(out_len);
  p.lhs_len = static_cast<int>(lhs_len);
  p.rhs_len = static_cast<int>(rhs_len);
  p.use_bcast = use_bcast ? 1 : 0;
  p.bd = bd;
  if (dtype == kF32)
    return coo.idbits == 32 ? spmm_coo_op<int32_t, float>(p, op, red, stream)
                            : spmm_coo_op<int64_t, float>(p, op, red, stream);
  if (dtype == kF64)
    return coo.idbits == 32 ? spmm_coo_op<int32_t, double>(p, op, red, stream)
                            : spmm_coo_op<int64_t, double>(p, op, red, stream);
  // the reference refuses half types on this path as well (spmm.cuh:633-641)
  last_error() = "SpMM on COO does not support fp16/bf16 features; use the CSR format";
  return -1;
}

}  // namespace dgla
