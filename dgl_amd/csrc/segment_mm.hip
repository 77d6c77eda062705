// Segment / gather matrix multiply for gfx950: the per-relation dense
// transform next to the g-SpMM in R-GCN / HGT (nn/pytorch/linear.py:208-210 TypedLinear).
//
// Reference: src/array/cuda/gather_mm.cu — SegmentMM :201-246 (a host loop issuing one cuBLAS
// GEMM per relation), SegmentMMBackwardB :248-291 (same), GatherMM / GatherMMScatter :293-360
// (one warp per row, scalar FMAs); registered at src/array/kernel.cc:501-540.
//
// MI355X-first design: ONE grouped-GEMM launch for all relations.
//  * A tiny plan kernel turns the segment lengths into a table of 128-row tiles
//    (tile -> relation, first row); every workgroup of the main kernel looks its tile up with
//    a binary search, so a relation with 3 rows and one with 30 M rows share the launch and
//    nothing is serialised on the host.
//  * 256 threads = 4 wavefronts in a 2 x 2 grid; forward: 128 x 256 output tile (128 x 128 when
//    N <= 128), each wave 64 x 128 = 2 x 4 MFMA tiles of 32 x 32; weight gradient: 128 x 128: v_mfma_f32_32x32x16_{bf16,f16} for 16-bit storage (fp32
//    accumulate, as cuBLAS does for the reference: CUBLAS_COMPUTE_32F), v_mfma_f32_32x32x2_f32
//    for fp32 (exact fp32 FMA chain, no TF32-like rounding).  fp64 takes a plain FMA kernel.
//  * Both operands are staged through LDS K-contiguous, so every MFMA fragment is one
//    ds_read_b128.  The weight operand is needed K-contiguous per output column: for
//    C = A . B[r] the (small) weights are transposed once per call into scratch, for
//    C = A . B[r]^T (the backward w.r.t. A) they already are.
//  * Default data path (operands in whole aligned 16-byte pieces): LDS-direct — global_load_lds
//    DMA into slab rings with counted vmcnt, no VGPR staging (segment_mm_glds_kernel,
//    segment_mm_bwd_b_glds_kernel with transposing LDS reads, …_f32_kernel).  General path (odd
//    widths, unaligned or row-indexed weight gradient): register-staged kernels, 16-byte global
//    pieces, next K-slab fetched into registers while the current one is multiplied.
//  * Weight-gradient dB[r] = A_r^T . dC_r contracts over the rows of a segment: split over row
//    slabs sized to the chip whose fp32 partial tiles are added with hardware float atomics.
#include "../../include/dgl_amd.h"

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace dgla {
namespace {

int mfail(const std::string& m) {
  last_error() = m;
  return -1;
}

constexpr int BM = 128;
constexpr int kFwdSlab = 128;   // forward: bytes of K per tile row and slab (pitch + 16: conflict-free ds_read_b128)
constexpr int kBwdSlab = 64;    // weight gradient: 64-byte slabs
constexpr int kMinSlabRows = 2048;  // rows per split-K slab of the weight-gradient kernel (lower bound)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- plan: tiles per relation ---------------------------------------------------------
// plan[0 .. R]        exclusive prefix of ceil(len / rows_per_tile)
// plan[R+1 .. 2R+1]   exclusive prefix of len (row offsets)
// Lengths are clamped to the rows that exist (a device-side seglen cannot be checked on the host
// without a synchronisation): segments never reach past row num_rows, whatever seglen holds.
template <typename Idx>
__global__ void segment_plan_kernel(const Idx* __restrict__ seglen, int64_t num_rel,
                                    int rows_per_tile, int64_t num_rows,
                                    int64_t* __restrict__ plan, int64_t* __restrict__ tile32) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int64_t t = 0, r0 = 0, t32 = 0;
  for (int64_t r = 0; r < num_rel; ++r) {
    plan[r] = t;
    plan[num_rel + 1 + r] = r0;
    if (tile32) tile32[r] = t32;  // 32-row tiles of the weights-stationary forward kernel
    int64_t len = static_cast<int64_t>(seglen[r]);
    if (len < 0) len = 0;
    if (len > num_rows - r0) len = num_rows - r0;
    t += (len + rows_per_tile - 1) / rows_per_tile;
    t32 += (len + 31) / 32;
    r0 += len;
  }
  if (tile32) tile32[num_rel] = t32;
  plan[num_rel] = t;
  plan[2 * num_rel + 1] = r0;
  plan[3 * num_rel + 2] = 0;  // NaN flag of the X3 (fp32 as 3 x bf16) kernels, behind the staged seglen
}

// Values that are the same in every lane (derived from blockIdx and the plan): tell the compiler,
// so they live in scalar registers instead of costing two VGPRs each.
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

__device__ __forceinline__ int64_t find_segment(const int64_t* __restrict__ tile_off,
                                                int64_t num_rel, int64_t t) {
  // largest r with tile_off[r] <= t  (tile_off is non-decreasing; empty relations repeat)
  int64_t lo = 0, hi = num_rel - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (tile_off[mid] <= t)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// ---- weight transpose: Bt[r][n][k] = B[r][k][n] -----------------------------------------
template <typename DT>
__global__ __launch_bounds__(256) void transpose_weights_kernel(const DT* __restrict__ b,
                                                               DT* __restrict__ bt, int K, int N) {
  __shared__ DT tile[32][33];
  const int64_t base = static_cast<int64_t>(blockIdx.z) * K * N;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8)
    if (k0 + i < K && n0 + tx < N) tile[i][tx] = b[base + static_cast<int64_t>(k0 + i) * N + n0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (n0 + i < N && k0 + tx < K) bt[base + static_cast<int64_t>(n0 + i) * K + k0 + tx] = tile[tx][i];
}

// ---- MFMA fragment helpers ----------------------------------------------------------------
template <typename DT>
struct Mma;  // KSTEP: K elements consumed by one MFMA

template <>
struct Mma<float> {
  static constexpr int KSTEP = 2;
  __device__ static __forceinline__ void step(const char* sa, const char* sb, int s, int khalf,
                                              f32x16& acc) {
    const float a = *reinterpret_cast<const float*>(sa + (s * 2 + khalf) * 4);
    const float b = *reinterpret_cast<const float*>(sb + (s * 2 + khalf) * 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma<f16_t> {
  static constexpr int KSTEP = 16;
  __device__ static __forceinline__ void step(const char* sa, const char* sb, int s, int khalf,
                                              f32x16& acc) {
    const h16x8 a = *reinterpret_cast<const h16x8*>(sa + (s * 16 + khalf * 8) * 2);
    const h16x8 b = *reinterpret_cast<const h16x8*>(sb + (s * 16 + khalf * 8) * 2);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma<bf16_t> {
  static constexpr int KSTEP = 16;
  __device__ static __forceinline__ void step(const char* sa, const char* sb, int s, int khalf,
                                              f32x16& acc) {
    const b16x8 a = *reinterpret_cast<const b16x8*>(sa + (s * 16 + khalf * 8) * 2);
    const b16x8 b = *reinterpret_cast<const b16x8*>(sb + (s * 16 + khalf * 8) * 2);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};

// One 16-byte piece of a K-contiguous operand row: `valid` elements starting at `src`
// (0 <= valid <= E); vector load when the piece is whole and aligned, else element-wise.
template <typename DT>
__device__ __forceinline__ u32x4 load_piece(const DT* src, int valid, bool vec_ok) {
  constexpr int E = 16 / sizeof(DT);
  u32x4 v = {0u, 0u, 0u, 0u};
  if (valid >= E && vec_ok) {
    v = *reinterpret_cast<const u32x4*>(src);
  } else if (valid > 0) {
    DT tmp[E];
#pragma unroll
    for (int j = 0; j < E; ++j) tmp[j] = j < valid ? src[j] : DT{};
    __builtin_memcpy(&v, tmp, 16);
  }
  return v;
}

// Direct stores of a wave's accumulators (C/D layout of the 32x32 MFMAs: col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); a half-wave writes 32 consecutive
// columns of one row.  Row-indexed output (INDEXED): the 16 physical rows of a 32-row block
// are looked up TOGETHER before its stores.  A lookup per store costs a s_waitcnt vmcnt(0) in
// front of every store — loads and stores share the counter, so each store would wait for the
// one before it (measured on the fp32 forward: 15.9 -> 12.2 ms came from the K-loop, the
// remaining 1.4 ms to the vendor GEMM from this).
template <typename DT, int NJ, bool INDEXED>
__device__ __forceinline__ void store_acc_direct(DT* __restrict__ C, const f32x16 (&acc)[2][NJ],
                                                 const int64_t* __restrict__ row_index, int64_t row_base,
                                                 int64_t row_end, int col_base, int N, int rbase) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int64_t pr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row_base + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
      if constexpr (INDEXED)
        pr[r] = row < row_end ? row_index[row] : -1;
      else
        pr[r] = row < row_end ? row : -1;
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int col = col_base + jj * 32;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (pr[r] >= 0) C[pr[r] * N + col] = from_acc<DT>(acc[i][jj][r]);
    }
  }
}

struct MmParams {
  const void* a;    // [M, K] rows grouped by relation
  const void* bt;   // [R, N, K]  (K contiguous)
  void* c;          // [M, N]
  const int64_t* plan;
  int64_t num_rel;
  int K, N;
  int vec_a, vec_b;  // 16-byte loads allowed (alignment + K % E == 0)
  int vec_c;         // 16-byte stores of C rows allowed (alignment + N % E == 0)
  const int64_t* row_index;  // optional: logical row r of A and C lives at physical row row_index[r]
  int64_t n_tiles;   // ceil(N / BN)
  uint32_t tune;     // kTune* bits
  uint32_t* nan_flag;  // X3 kernels: raised when an accumulator came out NaN (zeroed by the plan kernel)
};

// ---- forward: C_r = A_r . Bt_r^T ----------------------------------------------------------
// TBN = 256 (used when N > 128): one workgroup owns 128 rows x 256 columns, so for N <= 256
// every element of A — the operand that comes from HBM; the weights sit in L2 — is read
// exactly once.  TBN = 128 keeps narrow outputs from wasting half of the MFMA work.
// XCD-aware tile order for N > TBN: workgroup L runs on XCD L % 8 (round-robin dispatch); the
// n-tiles of one row tile go to workgroups L, L + 8, ... of the SAME XCD, which re-read the A
// tile from that XCD's L2.  Index math only: any placement is correct.
template <typename DT, int TBN, bool VEC>  // VEC: A, Bt and C rows are 16-byte aligned and whole pieces
__global__ __launch_bounds__(256, 2) void segment_mm_kernel(const MmParams p) {
  using M = Mma<DT>;
  constexpr int KE = kFwdSlab / sizeof(DT);  // K elements per slab
  constexpr int E = 16 / sizeof(DT);         // elements per 16-byte piece
  constexpr int kPitch = kFwdSlab + 16;
  constexpr int PPR = kFwdSlab / 16;         // 16-byte pieces per tile row
  constexpr int RSTEP = 256 / PPR;           // tile rows covered by one pass of the 256 threads
  constexpr int PA = BM / RSTEP, PB = TBN / RSTEP;  // pieces per thread: A tile, weight tile
  constexpr int NJ = TBN / 64;               // 32-column MFMA tiles per wave
  constexpr int kCPitch = TBN * 2 + 16;      // 16-bit epilogue staging: pitch of a C row
  static_assert(sizeof(DT) != 2 || 64 * kCPitch <= (BM + TBN) * kPitch, "half a C tile must fit");
  __shared__ __attribute__((aligned(16))) char smem[(BM + TBN) * kPitch];
  char* sA = smem;
  char* sB = smem + BM * kPitch;

  const int64_t* tile_off = p.plan;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const int K = p.K, N = p.N;
  const int64_t L = blockIdx.x;
  const int64_t j = L >> 3;
  const int64_t tile = (j / p.n_tiles) * 8 + (L & 7);
  const int n0 = static_cast<int>(j % p.n_tiles) * TBN;
  if (tile >= tile_off[p.num_rel]) return;
  const int64_t rel = find_segment(tile_off, p.num_rel, tile);
  const int64_t row0 = row_off[rel] + (tile - tile_off[rel]) * BM;
  const int64_t row_end = row_off[rel + 1];

  const DT* __restrict__ A = static_cast<const DT*>(p.a);
  const DT* __restrict__ Bt = static_cast<const DT*>(p.bt) + rel * static_cast<int64_t>(N) * K;
  DT* __restrict__ C = static_cast<DT*>(p.c);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;      // wave tile: rows wm * 64 .., cols wn * (TBN / 2) ..
  const int pr0 = tid / PPR, pc = tid % PPR;    // piece h: tile row pr0 + h * RSTEP, chunk pc
  const int lrow = lane & 31, khalf = lane >> 5;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

  auto fetch = [&](int k0, u32x4 (&ra)[PA], u32x4 (&rb)[PB]) {
    const int kk = k0 + pc * E;
    const int valid = K - kk;
#pragma unroll
    for (int h = 0; h < PA; ++h) {
      const int64_t ar = row0 + pr0 + RSTEP * h;
      const bool in = ar < row_end;
      const int64_t pr = (in && p.row_index) ? p.row_index[ar] : ar;
      ra[h] = load_piece<DT>(A + pr * K + kk, in ? valid : 0, VEC);
    }
#pragma unroll
    for (int h = 0; h < PB; ++h) {
      const int bn = n0 + pr0 + RSTEP * h;
      rb[h] = load_piece<DT>(Bt + static_cast<int64_t>(bn) * K + kk, bn < N ? valid : 0, VEC);
    }
  };
  auto stash = [&](const u32x4 (&ra)[PA], const u32x4 (&rb)[PB]) {
#pragma unroll
    for (int h = 0; h < PA; ++h)
      *reinterpret_cast<u32x4*>(sA + (pr0 + RSTEP * h) * kPitch + pc * 16) = ra[h];
#pragma unroll
    for (int h = 0; h < PB; ++h)
      *reinterpret_cast<u32x4*>(sB + (pr0 + RSTEP * h) * kPitch + pc * 16) = rb[h];
  };

  u32x4 ra[PA], rb[PB];
  fetch(0, ra, rb);
  for (int k0 = 0; k0 < K; k0 += KE) {
    __syncthreads();  // previous slab fully consumed
    stash(ra, rb);
    __syncthreads();
    if (k0 + KE < K) fetch(k0 + KE, ra, rb);  // in flight while this slab is multiplied
    auto kstep = [&](int s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* sa = sA + (wm * 64 + i * 32 + lrow) * kPitch;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          const char* sb = sB + (wn * (TBN / 2) + jj * 32 + lrow) * kPitch;
          M::step(sa, sb, s, khalf, acc[i][jj]);
        }
      }
    };
    // 16-bit fragments are 4 VGPRs each: keep one k-step of them live (the fully unrolled
    // form hoists every LDS read of the slab and costs a wave of occupancy); fp32 fragments
    // are one VGPR, so four k-steps are unrolled to keep LDS reads ahead of the MFMAs.
    if constexpr (sizeof(DT) == 2) {
#pragma unroll 1
      for (int s = 0; s < KE / M::KSTEP; ++s) kstep(s);
    } else {
#pragma unroll 4
      for (int s = 0; s < KE / M::KSTEP; ++s) kstep(s);
    }
  }

  // C/D layout of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int rbase = 4 * khalf;
  if constexpr (sizeof(DT) == 2 && VEC) {
    {
      // 16-bit results: a lane owns ONE column, so direct stores would be 2 bytes wide.  Stage
      // the tile in LDS, 64 rows at a time (the waves of one wm), and write 16-byte row pieces.
      constexpr int CPR = TBN * 2 / 16;  // 16-byte pieces per C row
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        __syncthreads();  // operands (or the previous half) no longer needed
        if (wm == half) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
                const int col = wn * (TBN / 2) + jj * 32 + lrow;
                *reinterpret_cast<DT*>(smem + row * kCPitch + col * 2) = from_acc<DT>(acc[i][jj][r]);
              }
        }
        __syncthreads();
        int64_t prow[64 * CPR / 256];  // physical rows first (one wait), then the stores
#pragma unroll
        for (int h = 0; h < 64 * CPR / 256; ++h) {
          const int64_t grow = row0 + half * 64 + (tid + 256 * h) / CPR;
          prow[h] = grow < row_end ? (p.row_index ? p.row_index[grow] : grow) : -1;
        }
#pragma unroll
        for (int h = 0; h < 64 * CPR / 256; ++h) {
          const int pidx = tid + 256 * h;
          const int row = pidx / CPR, chunk = pidx % CPR;
          const int col = n0 + chunk * 8;
          if (prow[h] >= 0 && col < N)  // N % 8 == 0 here: a piece is all in or all out
            *reinterpret_cast<u32x4*>(C + prow[h] * N + col) =
                *reinterpret_cast<const u32x4*>(smem + row * kCPitch + chunk * 16);
        }
      }
      return;
    }
  }
  if (p.row_index)
    store_acc_direct<DT, NJ, true>(C, acc, p.row_index, row0 + wm * 64, row_end, n0 + wn * (TBN / 2) + lrow, N, rbase);
  else
    store_acc_direct<DT, NJ, false>(C, acc, nullptr, row0 + wm * 64, row_end, n0 + wn * (TBN / 2) + lrow, N, rbase);
}

// ---- forward, LDS-direct variant (16-bit and fp32 storage, operands in whole aligned 16-byte
// pieces): the default for those shapes.  Same tile and MFMAs as segment_mm_kernel<DT, TBN, true>, but
// the operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, 180 instead of
// 238 VGPRs) into two rings of 64-byte-per-row slabs: NA = 6 slots for A (the operand that
// comes from HBM: 5 slabs = 40 KB per workgroup always in flight) and NB = 2 for the weights
// (L2 hits: one slab of lead is enough).  80 KB of LDS per workgroup = 2 workgroups per CU.
// The K-loop never drains the memory pipe: one raw s_barrier per slab with a counted
// s_waitcnt vmcnt(n) that lets every load issued after the needed weight slab stay outstanding.
// The register-staged kernel is latency-bound (69 % of wave time parked, profiles/r1/
// pmc_segment_mm_bf16.json); measured on MI355X (profiles/r1/glds_ab.jsonl), 8 relations:
//   10 M x 256 x 256 bf16   3.79 -> 2.58 ms (509 TFLOP/s; per-relation vendor GEMM loop 2.93)
//   10 M x 128 x 256 bf16   2.76 -> 1.81 ms      2 M x 1024 x 1024 bf16  7.06 -> 4.99 ms (840 TF)
//   10 M x 256 x 256 fp32  15.85 -> 12.2 ms      16-bit results bit-identical to the other kernel
// Ring depths tried: one 3-deep ring for both operands 2.90 ms, (NA, NB) = (4, 3) 2.65 ms,
// (6, 2) 2.58 ms; 256-row tiles with 512 threads (half the weight traffic, 1 workgroup per CU)
// 3.07 ms; non-temporal A loads +13 % time; delaying every other workgroup of the first wave by
// 16-160 k cycles (to de-phase the two workgroups of a CU) +-0.
//  * LDS layout: rows x 64 B, no padding — the DMA writes lane l at base + 16 l, i.e. 16 rows
//    x 4 chunks per instruction.  Bank conflicts are avoided by a chunk swizzle applied on the
//    SOURCE side: physical chunk c of row r holds logical chunk c ^ ((r >> 2) & 3), so the 16
//    lanes a ds_read_b128 serves together (rows 4a + b) hit 16 distinct 16-byte bank groups.
//  * Rows past the end of a segment / of the weight matrix are clamped to the last valid row:
//    their products land in accumulator rows / columns that are never stored.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

// Source of DMA lanes that must contribute zeros (K tail of the last slab, rows past a slab end).
__device__ __attribute__((aligned(256))) char g_mm_zero_page[256];

constexpr int kGldsSlabBytes = 64;  // bytes of K per tile row and slab (32 16-bit / 16 fp32 elements)

// fp32 operands as three bf16 terms (X3).  x = h + m + l with h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m), all round-to-nearest on gfx950's v_cvt_pk_bf16_f32: the differences are exact
// in fp32 (bf16 has fp32's exponent range), |m| <= 2^-9 |x|, |l| <= 2^-18 |x|, and what is left of
// x after the three terms is below 2^-27 |x|.  a * b = (ah + am + al)(bh + bm + bl) is evaluated as
// the six products of order <= 2 (hh, hm, mh, mm, hl, lh) on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, small terms first; the dropped ml, lm, ll terms are < 2^-26 |a b| and of either
// sign.  Six 32-cycle MFMAs replace eight 64-cycle v_mfma_f32_32x32x2_f32 per 16 k (gfx950 has no
// xf32 MFMA): 2.7x less matrix-pipe time for fp32-level accuracy (tests/test_mm.py's bound:
// 4 sqrt(k) 2^-24 sum |a||b| against the exact fp64 product, down to k = 1).
// Non-finite operands (round 3): x = +-inf gives r = inf - inf = NaN, and |x| >= 0x7f7f8000 rounds to
// a bf16 infinity with r = -inf; either way the m term is NaN and EVERY output the value enters comes
// out NaN, where IEEE fp32 arithmetic may give +-inf or a finite number.  The K loop is left alone
// (a check inside it cost 2.7x: 8.3 -> 23.5 ms; an in-epilogue repair cost registers: 228 -> 256 VGPRs
// + scratch); instead the epilogue looks at its accumulators — a NaN there is the only symptom — and
// raises a flag in the plan scratch; x3_repair_{fwd,bwd}_kernel, launched behind every X3 kernel, exit
// on their first load unless it is set and otherwise recompute every NaN element of the result as a
// plain fp32 dot product from global memory.  inf * b = inf, inf * 0 = NaN, inf - inf = NaN, NaN * b =
// NaN then come out as an fp32 GEMM gives them
// (tests/test_mm.py::test_segment_mm_nonfinite).  kTuneMmF32 selects the plain fp32 MFMA path.
__device__ __forceinline__ void split3(const f32x4 lo4, const f32x4 hi4, b16x8& h, b16x8& m, b16x8& l) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  uint32_t hp[4], mp[4], lp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // elements 2q, 2q + 1 of the 8
    const f32x2 x = q < 2 ? f32x2{lo4[2 * q], lo4[2 * q + 1]} : f32x2{hi4[2 * q - 4], hi4[2 * q - 3]};
    const b16x2 hb = __builtin_convertvector(x, b16x2);
    const f32x2 r = x - __builtin_convertvector(hb, f32x2);
    const b16x2 mb = __builtin_convertvector(r, b16x2);
    const b16x2 lb = __builtin_convertvector(r - __builtin_convertvector(mb, f32x2), b16x2);
    hp[q] = __builtin_bit_cast(uint32_t, hb);
    mp[q] = __builtin_bit_cast(uint32_t, mb);
    lp[q] = __builtin_bit_cast(uint32_t, lb);
  }
  h = __builtin_bit_cast(b16x8, (u32x4_t{hp[0], hp[1], hp[2], hp[3]}));
  m = __builtin_bit_cast(b16x8, (u32x4_t{mp[0], mp[1], mp[2], mp[3]}));
  l = __builtin_bit_cast(b16x8, (u32x4_t{lp[0], lp[1], lp[2], lp[3]}));
}

template <int NI, int NJ>
__device__ __forceinline__ bool any_nan(const f32x16 (&acc)[NI][NJ]) {
  bool bad = false;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) bad |= acc[i][j][r] != acc[i][j][r];
  return __builtin_amdgcn_ballot_w64(bad) != 0;
}

template <typename DT, int TBN, int BMT, int NA, int NB, bool X3 = false>  // BMT x TBN output tile, 2 * BMT threads
__global__ __launch_bounds__(2 * BMT, 256 / BMT) void segment_mm_glds_kernel(const MmParams p) {
  using M = Mma<DT>;
  constexpr int ES = sizeof(DT);  // 2: 32x32x16 MFMAs, LDS-staged epilogue; 4: 32x32x2 MFMAs, direct stores
  static_assert(ES == 2 || ES == 4, "16-bit and fp32 storage");
  // NA > NB strictly: iteration i issues [weights slab i + NB - 1, A slab i + NA - 1]; with NA == NB the A load
  // that follows weights slab t in issue order is A slab t ITSELF, which wait_for_slab would leave in flight
  // (tried as (3, 3) in round 3: wrong sums)
  static_assert(NA > NB && NB >= 2, "A ring deeper than the weight ring");
  constexpr int NJ = TBN / 64;
  constexpr int NT = 2 * BMT, NW = NT / 64;   // threads, waves (one wave per 64 x TBN/2 of C)
  constexpr int kASlab = BMT * 64, kBSlab = TBN * 64;  // bytes of one slab of each operand
  constexpr int nA = kASlab / 1024 / NW, nB = kBSlab / 1024 / NW;  // DMA instructions per wave and slab
  static_assert(nA * NW * 1024 == kASlab && nB * NW * 1024 == kBSlab, "whole DMA instructions per wave");
  constexpr int kBBase = NA * kASlab;
  constexpr int kCPitch = TBN * 2 + 16;
  __shared__ __attribute__((aligned(1024))) char smem[NA * kASlab + NB * kBSlab];

  const int64_t* tile_off = p.plan;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const int K = p.K, N = p.N;
  const int64_t L = blockIdx.x;
  const int64_t j = L >> 3;
  const int64_t tile = (j / p.n_tiles) * 8 + (L & 7);
  const int n0 = static_cast<int>(j % p.n_tiles) * TBN;
  if (tile >= tile_off[p.num_rel]) return;
  const int64_t rel = find_segment(tile_off, p.num_rel, tile);
  const int64_t row0 = row_off[rel] + (tile - tile_off[rel]) * BMT;
  const int64_t row_end = row_off[rel + 1];

  const char* __restrict__ A = static_cast<const char*>(p.a);
  const char* __restrict__ Bt = static_cast<const char*>(p.bt) + rel * static_cast<int64_t>(N) * K * ES;
  DT* __restrict__ C = static_cast<DT*>(p.c);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, khalf = lane >> 5;

  // Source of this lane for each of the wave's DMA instructions per slab: instruction q of an
  // operand fills its tile rows 16 q .. 16 q + 15; the lane owns row 16 q + (lane >> 2),
  // physical chunk lane & 3.
  const int src_chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const char* srcA[nA];
  const char* srcB[nB];
#pragma unroll
  for (int i = 0; i < nA; ++i) {
    int64_t ar = row0 + (wave * nA + i) * 16 + (lane >> 2);
    if (ar >= row_end) ar = row_end - 1;
    if (p.row_index) ar = p.row_index[ar];
    srcA[i] = A + ar * K * ES + src_chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < nB; ++i) {
    int bn = n0 + (wave * nB + i) * 16 + (lane >> 2);
    if (bn >= N) bn = N - 1;
    srcB[i] = Bt + static_cast<int64_t>(bn) * K * ES + src_chunk * 16;
  }
  // K need not be a whole number of slabs: in the last slab, the lanes whose 16-byte chunk starts
  // at or past the end of the row read the zero page instead (both operands).
  const int kbytes = K * ES;
  const int nslab = (kbytes + kGldsSlabBytes - 1) / kGldsSlabBytes;
  const bool tail_zero = (nslab - 1) * kGldsSlabBytes + src_chunk * 16 >= kbytes;
  auto issue_a = [&](int t) {  // slab t of A -> ring slot t % NA
    if (t >= nslab) return;
    char* dst = smem + (t % NA) * kASlab + wave * (nA * 1024);
    const bool zero = tail_zero && t == nslab - 1;
#pragma unroll
    for (int i = 0; i < nA; ++i)
      __builtin_amdgcn_global_load_lds(
          (gbl_ptr_t)(zero ? g_mm_zero_page : srcA[i] + static_cast<int64_t>(t) * kGldsSlabBytes),
          (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int t) {  // slab t of the weights -> ring slot t % NB
    if (t >= nslab) return;
    char* dst = smem + kBBase + (t % NB) * kBSlab + wave * (nB * 1024);
    const bool zero = tail_zero && t == nslab - 1;
#pragma unroll
    for (int i = 0; i < nB; ++i)
      __builtin_amdgcn_global_load_lds(
          (gbl_ptr_t)(zero ? g_mm_zero_page : srcB[i] + static_cast<int64_t>(t) * kGldsSlabBytes),
          (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
  };
  // Iteration i (also the prologue iterations i < 0) issues [weights slab i + NB - 1, A slab
  // i + NA - 1], so the loads of this wave still allowed in flight when slab t is needed are
  // those issued after weights slab t: the A slab of that iteration and all of the NB - 2
  // iterations after it.  vmcnt retires in order, so A slab t (issued earlier) is done too.
  auto issued_a = [&](int i) { return (i + NA - 1 >= 0 && i + NA - 1 < nslab) ? nA : 0; };
  auto issued_b = [&](int i) { return (i + NB - 1 >= 0 && i + NB - 1 < nslab) ? nB : 0; };
  auto wait_for_slab = [&](int t) {
    int allowed = issued_a(t - NB + 1);
#pragma unroll
    for (int i = 2; i < NB; ++i) allowed += issued_a(t - NB + i) + issued_b(t - NB + i);
    switch (allowed) {  // s_waitcnt takes an immediate: vmcnt in bits 3:0 (values < 16)
      case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
      case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;
      case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
      case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
      case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
      case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
      case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
      case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
      case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
      case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
      case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
      case 11: __builtin_amdgcn_s_waitcnt(0x0F7B); break;
      case 12: __builtin_amdgcn_s_waitcnt(0x0F7C); break;
      default: __builtin_amdgcn_s_waitcnt(0x0F70); break;  // never looser than needed
    }
  };
  static_assert(nA + (NB - 2) * (nA + nB) <= 12, "extend the vmcnt switch");

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

  // fragment addresses inside a slab (swizzled chunk per k-step)
  const int swz = (lrow >> 2) & 3;
  const int a_off = (wm * 64 + lrow) * 64;
  const int b_off = kBBase + (wn * (TBN / 2) + lrow) * 64;

#pragma unroll
  for (int i = 1 - NA; i < 0; ++i) {  // prologue iterations
    issue_b(i + NB - 1 >= 0 ? i + NB - 1 : nslab);
    issue_a(i + NA - 1);
  }
  for (int t = 0; t < nslab; ++t) {
    wait_for_slab(t);  // this wave's pieces of slab t have landed ...
    __builtin_amdgcn_s_barrier();  // ... and everybody's; slab t - 1 has been consumed by all
    issue_b(t + NB - 1);  // into the slots slab t - 1 was multiplied from
    issue_a(t + NA - 1);
    const char* bufa = smem + (t % NA) * kASlab;
    const char* bufb = smem + (t % NB) * kBSlab;
    if constexpr (ES == 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // two k-steps of 16 elements
        const int chunk = ((s * 2 + khalf) ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const char* sa = bufa + a_off + i * 32 * 64 + chunk;
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) {
            const char* sb = bufb + b_off + jj * 32 * 64 + chunk;
            // operands swapped: the MFMA produces the TRANSPOSED 32 x 32 block, so a lane ends up with
            // four CONSECUTIVE COLUMNS of one C row per register quad (row = lane & 31,
            // col = 8 q + 4 (lane >> 5) + j) and the epilogue stages 8-byte pieces, not 2-byte ones
            M::step(sb, sa, 0, 0, acc[i][jj]);
          }
        }
      }
    } else if constexpr (X3) {
      // fp32 as 3 x bf16: the lane's 8 k of the slab (chunks khalf and 2 + khalf — the same
      // permutation of the contraction index for both operands) split into (h, m, l)
      const int c0 = ((0 * 2 + khalf) ^ swz) * 16, c1 = ((1 * 2 + khalf) ^ swz) * 16;
      b16x8 ah[2], am[2], al[2], bh[NJ], bm[NJ], bl[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* sa = bufa + a_off + i * 32 * 64;
        split3(*reinterpret_cast<const f32x4*>(sa + c0), *reinterpret_cast<const f32x4*>(sa + c1), ah[i], am[i], al[i]);
      }
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const char* sb = bufb + b_off + jj * 32 * 64;
        split3(*reinterpret_cast<const f32x4*>(sb + c0), *reinterpret_cast<const f32x4*>(sb + c1), bh[jj], bm[jj], bl[jj]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {  // small terms first
          f32x16 c = acc[i][jj];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[jj], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[jj], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[jj], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[jj], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[jj], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[jj], c, 0, 0, 0);
          acc[i][jj] = c;
        }
    } else {
      // fp32: a lane reads 4 consecutive k of its row in one ds_read_b128 (chunk 2u + khalf) and
      // feeds them to 4 MFMAs; the MFMA's k = 0 / 1 halves are then k = 8u + j and 8u + 4 + j —
      // the same permutation of the contraction index for both operands.
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int chunk = ((u * 2 + khalf) ^ swz) * 16;
        f32x4 fa[2], fb[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(bufa + a_off + i * 32 * 64 + chunk);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) fb[jj] = *reinterpret_cast<const f32x4*>(bufb + b_off + jj * 32 * 64 + chunk);
#pragma unroll
        for (int jk = 0; jk < 4; ++jk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][jk], fb[jj][jk], acc[i][jj], 0, 0, 0);
      }
    }
  }

  if constexpr (ES == 4) {
    if constexpr (X3) {  // non-finite operands (see split3): tell x3_repair_fwd_kernel to look
      if (__builtin_expect(any_nan(acc), 0) && lane == 0) atomicOr(p.nan_flag, 1u);
    }
    // fp32: a half-wave owns 32 consecutive columns of one row = one 128-byte line per store
    if (p.row_index)
      store_acc_direct<DT, NJ, true>(C, acc, p.row_index, row0 + wm * 64, row_end, n0 + wn * (TBN / 2) + lrow, N, 4 * khalf);
    else
      store_acc_direct<DT, NJ, false>(C, acc, nullptr, row0 + wm * 64, row_end, n0 + wn * (TBN / 2) + lrow, N, 4 * khalf);
    return;
  }

  // epilogue: the whole C tile is staged in the (now idle) slab rings — every wave writes its 64 x TBN/2
  // part as 8-byte pieces (see the operand swap above), one barrier, then 16-byte row pieces go out.
  // (Round 2 staged 64 rows at a time with 2-byte LDS writes: 4 serial phases of 128 ds_write_b16 per
  // wave with one workgroup per CU — about 40 % of the tile's time.)
  static_assert(BMT * kCPitch <= NA * kASlab + NB * kBSlab, "the C tile must fit in the rings");
  constexpr int CPR = TBN * 2 / 16;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  __syncthreads();  // everybody is done reading the last slab
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wm * 64 + i * 32 + lrow;
        const int col = wn * (TBN / 2) + jj * 32 + 8 * q + 4 * khalf;
        DT t4[4];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) t4[j4] = from_acc<DT>(acc[i][jj][4 * q + j4]);
        u32x2 w;
        __builtin_memcpy(&w, t4, 8);
        *reinterpret_cast<u32x2*>(smem + row * kCPitch + col * ES) = w;
      }
  __syncthreads();
  constexpr int NP = BMT * CPR / NT;  // 16-byte pieces per thread
  int64_t prow[NP];  // physical rows first (one wait), then the stores
#pragma unroll
  for (int h = 0; h < NP; ++h) {
    const int64_t grow = row0 + (tid + NT * h) / CPR;
    prow[h] = grow < row_end ? (p.row_index ? p.row_index[grow] : grow) : -1;
  }
#pragma unroll
  for (int h = 0; h < NP; ++h) {
    const int pidx = tid + NT * h;
    const int row = pidx / CPR, chunk = pidx % CPR;
    const int col = n0 + chunk * 8;
    if (prow[h] >= 0 && col < N) {
      u32x4* dstp = reinterpret_cast<u32x4*>(C + prow[h] * N + col);
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * kCPitch + chunk * 16);
      *dstp = v;
    }
  }
}


// ---- forward, WEIGHTS-STATIONARY persistent kernel (round 4; VERDICT r3 Next #3) -----------------------
// The LDS-direct kernel above streams BOTH operands through LDS for every 128-row tile: at 10 M x 256 x 256
// bf16 the weights (128 KB per tile, from L2) are two thirds of its LDS-DMA traffic and the kernel runs at
// the DMA rate (15.4 GB / 2.33 ms = 6.6 TB/s), not at the HBM rate of A + C (10.2 GB).  This kernel keeps
// the weights where they are re-used and gives A the path that needs no LDS at all:
//   * ONE workgroup of 8 waves per CU for the whole launch; it owns a contiguous chunk of 32-row tiles.
//   * The relation's weights are loaded into LDS ONCE per (workgroup, relation) by global_load_lds —
//     column n of the K-contiguous weights at LDS row (n % NJ) * 32 + n / NJ (see the epilogue), rows
//     padded to an ODD number of 16-byte pieces so the 16 lanes of a ds_read_b128 group fall into
//     distinct bank groups.  256 x 256 x 2 B = 132 KB with the pad: the reason A cannot live in LDS too.
//   * Every WAVE runs its own 32-row tiles with no barrier in between: the A operand goes from global
//     memory straight into the MFMA fragment layout (lane = row, 16 bytes of K per lane: one
//     global_load_dwordx4 per k-step), through a ring of 8 fragments that is refilled in place — k-step
//     u of the NEXT round (or of the wave's next tile) is requested the moment k-step u has been
//     multiplied, so 8 KB per wave = 64 KB per CU are in flight at all times, counted by the compiler's
//     own vmcnt bookkeeping.
//   * Epilogue from registers: MFMA block jj multiplies weight columns {NJ * l + jj}, so lane l holds NJ
//     CONSECUTIVE output columns of a row; they leave as one 16-byte (NJ = 8) store per row and lane —
//     a half-wave writes 512 contiguous bytes.  No LDS staging, no barrier.
//   * fp32 (X3: operands as three bf16 terms, see split3): the weights are split ONCE per call into three
//     bf16 planes (split_weights_kernel) and a workgroup holds 64 columns x 3 planes; only the A fragment
//     is split in the loop (44 VALU operations against 12 MFMAs of 32 cycles; the LDS-direct kernel split
//     the weight fragments too, in every wave and tile: VALU 49 % beside MFMA 50 %, not overlapped).
//     N / 64 workgroups of the SAME XCD walk the same row chunk, one column group each, so the A rows come
//     from HBM once and from that XCD's L2 afterwards.
// Contraction order: k-steps of 16 in increasing k, inside a k-step the MFMA's own order — results equal the
// LDS-direct kernel's to fp32 rounding, not bit for bit (same contract as every GEMM here: tolerance).
template <typename DT, int V>
__device__ __forceinline__ void store_nt_vec(DT* dst, const VecT<DT, V>& v) {
  constexpr int B = sizeof(DT) * V;
  if constexpr (B == 16) {
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&v), reinterpret_cast<u32x4*>(dst));
  } else if constexpr (B == 8) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(*reinterpret_cast<const u32x2*>(&v), reinterpret_cast<u32x2*>(dst));
  } else if constexpr (B == 4) {
    __builtin_nontemporal_store(*reinterpret_cast<const uint32_t*>(&v), reinterpret_cast<uint32_t*>(dst));
  } else {
    *reinterpret_cast<VecT<DT, V>*>(dst) = v;
  }
}

template <typename DT>
struct WsFrag;
template <>
struct WsFrag<bf16_t> {
  typedef b16x8 type;
  __device__ static __forceinline__ f32x16 mma(type a, type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct WsFrag<f16_t> {
  typedef h16x8 type;
  __device__ static __forceinline__ f32x16 mma(type a, type b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kWsWaves = 8;        // waves per workgroup (one workgroup per CU)
constexpr int kWsMaxK = 256;       // contraction length the LDS image is sized for
constexpr int kWsPitchPieces = kWsMaxK / 8 + 1;  // 16-byte pieces per LDS row at K = 256 (odd)

struct WsParams {
  MmParams m;
  const void* planes;        // X3: [3][R, N, Kp] bf16 (h, m, l), Kp = K rounded up to 8, zero padded
  const int64_t* tile32_off; // [R + 1] exclusive prefix of ceil(len / 32)
  int ncg;                   // column groups (workgroups sharing a row chunk)
  int kp;                    // X3: row pitch of the planes in elements
  char* dummy;               // 512 bytes nobody reads (the held stores of a wave that holds nothing)
  int* queue;                // one column group only: ticket of the chunk queue (zeroed per launch); NULL = static chunks
  int qsplit;                // dynamic chunks per static one
  int* prog;                 // PIPE kernels: [chunk][wave][4] tiles STARTED by each sibling wave (zeroed per launch)
  int rot;                   // ncg > 1: 1 = sibling c starts every row at line c * nss / ncg, 0 = paced only, -1 = neither
};

// Three bf16 planes of the K-contiguous fp32 weights: out[pl][row][k], row pitch kp, zero padded.
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ bt, bf16_t* __restrict__ out,
                                                            int64_t rows, int K, int kp) {
  const int64_t total = rows * kp;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = i / kp;
    const int k = static_cast<int>(i - r * kp);
    float x = k < K ? bt[r * K + k] : 0.f;
    const bf16_t h = from_acc<bf16_t>(x);
    const float r1 = x - to_acc<bf16_t>(h);
    const bf16_t m = from_acc<bf16_t>(r1);
    const bf16_t l = from_acc<bf16_t>(r1 - to_acc<bf16_t>(m));
    out[i] = h;
    out[total + i] = m;
    out[2 * total + i] = l;
  }
}

template <typename DT, int NJ, bool X3, bool INDEXED, int RS, bool PIPE>
__global__ __launch_bounds__(64 * kWsWaves, 2) void segment_mm_ws_kernel(const WsParams wp) {
  const MmParams& p = wp.m;
  constexpr int ES = sizeof(DT);
  constexpr int PL = X3 ? 3 : 1;
  constexpr int ROWS = PL * NJ * 32;
  constexpr int KSS = X3 ? 2 : 4;  // k-steps fed by one ring slot (one 128-byte line of the A row)
  static_assert(X3 == (ES == 4), "X3 is the fp32 path");
  __shared__ __attribute__((aligned(1024))) char smem[ROWS * kWsPitchPieces * 16 + 1024];  // (+ the last DMA instruction's overhang)

  const int K = p.K, N = p.N;
  const int nss = (K * ES + 127) >> 7;          // ring slots (128-byte lines) per A row
  const int nround = (nss + RS - 1) / RS;
  const int pp = (X3 ? 4 : 8) * nss + 1;        // LDS row pitch in 16-byte pieces (odd)
  const int kpieces = X3 ? (wp.kp >> 3) : (K >> 3);  // 16-byte pieces with data per weight row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform by construction: tell the compiler, it then keeps the
                                                              // tile loop and the epilogue's whole / ragged choice SCALAR)
  const int l = lane & 31, khalf = lane >> 5;

  const int64_t* __restrict__ tile_off = wp.tile32_off;
  const int64_t* __restrict__ row_off = p.plan + p.num_rel + 1;
  // workgroup -> (row chunk, column group): the ncg groups of a chunk are workgroups of ONE XCD
  const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
  // Column groups > 1 (fp32: 64 columns per workgroup; N > 256 for 16-bit): the ncg sibling workgroups of a row
  // chunk sit on ONE XCD and read the same A rows; HBM sees A about once (PMC: 1.16 x).  Two measures keep the
  // siblings' misses apart, together worth 3-4 % at 10 M x 256 x 256 (fp32, pipelined kernel only): (a) sibling c
  // walks the lines of a row in a ROTATED order (starts at line c * nss / ncg), so in step the siblings miss on
  // DIFFERENT lines and hit in their XCD's L2 on the lines the others fetched; (b) a soft barrier keeps them in
  // step: every wave publishes the tiles it has started, asks for its siblings' counts at the START of a tile and
  // looks at them before the LAST slot refill of the tile's last round — by then they have long arrived, so the check
  // costs no drain of the A ring (checking at once did: 9.3-9.9 ms instead of 7.6-8.0); a wave more than one
  // tile ahead of the slowest waits (bounded).  What bounds this kernel is not the A stream: with A always hitting
  // the cache and no stores it still runs 6.1 ms (MFMA pipe time of its 240 M instructions: 3.1 ms at 2.4 GHz) —
  // docs/DESIGN_detail_r1_r5.md §3.7 has the breakdown.
  const int ncg = wp.ncg;
  const int cg = qq % ncg;
  const bool pace = PIPE && wp.rot >= 0 && ncg > 1 && ncg <= 4;   // (rot < 0: pacing off, the words are still read)
  const int rot = (wp.rot > 0 && ncg > 1) ? cg * (((K * ES + 127) >> 7) / ncg) : 0;
  // Row chunks.  Static: one contiguous share of the tiles per group of ncg sibling workgroups.  Dynamic (wp.queue, one
  // column group only): qsplit times as many, smaller chunks handed out by a ticket — the workgroups do not run at one
  // speed, and the HBM-bound 16-bit kernels end 2.7 % sooner (2.09 -> 2.03 ms at 10 M x 256 x 256).  With sibling
  // column groups (fp32, wide N) a shared queue was built too (the first sibling drew, the others read a write-once
  // log) and changed nothing (8.07-8.20 against 8.04-8.15 ms): those keep the static split, and no spin.
  const int ngroups = (static_cast<int>(gridDim.x) >> 3) / ncg * 8;
  const int group = (qq / ncg) * 8 + xcd;
  if (group >= ngroups) return;
  const int nchunks = wp.queue ? ngroups * wp.qsplit : ngroups;
  __shared__ int s_chunk;
  const int64_t T = tile_off[p.num_rel];
  const int n0 = cg * 32 * NJ;

  const char* __restrict__ A = static_cast<const char*>(p.a);
  DT* __restrict__ C = static_cast<DT*>(p.c);
  typedef typename WsFrag<typename std::conditional<X3, bf16_t, DT>::type>::type frag_t;

  int started = 0;       // tiles this wave has started, across all of its group's chunks
  bool gave_up = false;
  int* const pw = wp.prog + (static_cast<int64_t>(group) * kWsWaves + wave) * 4;
  for (int gen = 0;; ++gen) {
  int chunk = group;
  if (wp.queue) {
    __syncthreads();  // everybody has read the previous s_chunk
    if (tid == 0) s_chunk = __hip_atomic_fetch_add(wp.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    chunk = s_chunk;
  } else if (gen > 0) {
    break;
  }
  if (chunk >= nchunks) break;
  const int64_t t0 = uniform64(T * chunk / nchunks), t1 = uniform64(T * (chunk + 1) / nchunks);
  int64_t t = t0;
  while (t < t1) {
    const int64_t rel = uniform64(find_segment(tile_off, p.num_rel, t));
    const int64_t rel_t0 = uniform64(tile_off[rel]);
    int64_t rel_end = uniform64(tile_off[rel + 1]);
    if (rel_end > t1) rel_end = t1;
    const int64_t rel_row0 = uniform64(row_off[rel]), row_end = uniform64(row_off[rel + 1]);

    // ---- this relation's weights -> LDS ----------------------------------------------------------------
    __syncthreads();  // nobody still multiplies from the previous relation's image
    {
      const int total = ROWS * pp, ninst = (total + 63) >> 6;
      const int64_t rpitch = X3 ? static_cast<int64_t>(wp.kp) * 2 : static_cast<int64_t>(K) * ES;
      const int64_t plane_bytes = X3 ? static_cast<int64_t>(p.num_rel) * N * rpitch : 0;
      const char* __restrict__ W = X3 ? static_cast<const char*>(wp.planes) : static_cast<const char*>(p.bt);
      for (int i = wave; i < ninst; i += kWsWaves) {
        const int j = i * 64 + lane;
        const int rho = j / pp, pc = j - rho * pp;
        const int pl = rho / (NJ * 32), jj = (rho >> 5) % NJ, ll = rho & 31;
        const int n = n0 + NJ * ll + jj;
        const bool valid = j < total && pc < kpieces && n < N;
        const char* src = valid ? W + pl * plane_bytes + (rel * N + n) * rpitch + pc * 16 : g_mm_zero_page;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + i * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces have landed
    }
    __syncthreads();

    // ---- this wave's tiles of the relation: t + wave, t + wave + 8, ... ------------------------------------
    const int64_t first = t + wave;
    auto lane_ptr = [&](int64_t tt) -> const char* {
      int64_t row = rel_row0 + (tt - rel_t0) * 32 + l;
      if (row >= row_end) row = row_end - 1;
      if constexpr (INDEXED) row = p.row_index[row];
      return A + row * (static_cast<int64_t>(K) * ES) + khalf * 64;
    };
    // One ring slot = the lane's half (64 bytes) of one 128-byte LINE of its A row: four 16-byte loads
    // issued back to back, so the two lanes of a row ask for the whole line within a few cycles and the
    // vector cache sees ONE miss per line (k-steps of 32 bytes per row, requested a k-step apart, made every
    // line four separate requests: 100 G requests/s at 5 TB/s).  16-bit: the slot feeds 4 k-steps (k-step i
    // multiplies elements 64 ss + 32 h + 8 i ..+8); fp32: 2 k-steps of 8 floats per lane.  The weights'
    // piece for (slot ss, half h, k-step i) is 8 ss + 4 h + i (16-bit) / 4 ss + 2 h + i (bf16 planes): the
    // contraction index is permuted identically on both operands.
    struct ASlot {
      u32x4 v[4];
    };
    // Branch-free: a piece past the end of the row (K not a multiple of the line), and every lane of a refill
    // that has nothing left to fetch, reads the zero page instead — a predicated load would sit in a divergent
    // block and the compiler then waits for vmcnt(0) at every use (seen in the first build).
    auto phys = [&](int ss) {  // (uniform) line visited at ring position ss
      const int q = ss + rot;
      return q >= nss ? q - nss : q;
    };
    auto load_slot = [&](const char* base, int ss_ring, bool on) -> ASlot {
      const int ss = phys(ss_ring < nss ? ss_ring : 0);
      ASlot f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = (ss * 128 + khalf * 64 + i * 16) / ES;  // first element of the piece
        // PIPE: every piece of every line exists (K a whole number of lines) and a refill with nothing left to
        // fetch just re-reads the current tile — no select, no compare: the loop body stays free of the
        // branches and copies that made the compiler keep TWO copies of the accumulators (64 v_mov per k-step)
        const char* src = PIPE ? base + ss * 128 + i * 16
                               : ((on && e < K) ? base + ss * 128 + i * 16 : reinterpret_cast<const char*>(g_mm_zero_page));
        f.v[i] = *reinterpret_cast<const u32x4*>(src);  // (non-temporal loads measured: 2.06 -> 2.55 ms, bf16)
      }
      return f;
    };

    if (first < rel_end) {
      const char* cur = lane_ptr(first);
      ASlot ar[RS];
#pragma unroll
      for (int u = 0; u < RS; ++u) {
        ar[u] = load_slot(cur, u, u < nss);
        __builtin_amdgcn_sched_barrier(0);  // same issue order as the refills below (else: vmcnt(0) in the fp32 kernel)
      }
      // PIPE (fp32, K a whole number of lines and of rounds): the three planes' weight fragments of k-step e + 1
      // are read from LDS while k-step e multiplies (double buffer bq[e & 1]); without it every fragment read was
      // followed by its own lgkmcnt(0) wait — five exposed LDS latencies per k-step, more than its 12 MFMAs
      [[maybe_unused]] b16x8 bq[2][3][NJ];
      [[maybe_unused]] auto load_b = [&](int piece0, b16x8 (&bf)[3][NJ]) {
        const char* brow = smem + (l * pp + piece0 + 2 * khalf) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) bf[pl][jj] = *reinterpret_cast<const b16x8*>(brow + (pl * NJ + jj) * 32 * pp * 16);
      };
      // ... and so is the A side: k-step e + 1's fp32 -> (h, m, l) split (36 VALU) is computed BETWEEN the MFMAs of
      // k-step e (at[e & 1]), four VALU per MFMA gap: an in-order wave hides about five single-issue instructions
      // behind one 32-cycle MFMA, and a split sitting in one clump in front of its own 12 MFMAs had no cover
      [[maybe_unused]] b16x8 at[2][3];
      if constexpr (PIPE) {
        load_b(4 * phys(0), bq[0]);
        split3(__builtin_bit_cast(f32x4, ar[0].v[0]), __builtin_bit_cast(f32x4, ar[0].v[1]), at[0][0], at[0][1], at[0][2]);
      }
      // (once per relation: drain here, so that the tile loop's waits are counted against the loop's own back edge —
      // loads, then the 16 stores of the epilogue — and not against this store-free entry)
      // (a USE of the youngest load: the compiler places — and, unlike a hand-written s_waitcnt, models — the drain)
      asm volatile("" ::"v"(ar[RS - 1].v[0]), "v"(ar[RS - 1].v[1]), "v"(ar[RS - 1].v[2]), "v"(ar[RS - 1].v[3]));
      // HELD stores (fp32 pipelined kernel, rows not indexed): a whole tile's accumulators are not stored in its own
      // epilogue but copied aside and stored four rows per slot during the NEXT tile's first round.  gfx9 counts stores
      // in vmcnt, in order with the loads: a burst of 16 stores right behind the ring's loads made the first round's
      // counted waits ("at most 12 operations in flight") wait for the stores' acknowledgements as well — once per
      // tile, 1.1 ms of 8.1 (the kernel with its stores removed ran 7.0 ms).  Spread over the round the stores are
      // long acknowledged when a wait reaches back to them.  The held stores are UNCONDITIONAL (nothing held: they go
      // to a dummy line) — a conditional store would count as zero stores in the compiler's waits.
      constexpr bool HOLD = PIPE && !INDEXED;
      [[maybe_unused]] f32x16 held[NJ];
      [[maybe_unused]] char* hbase = wp.dummy + lane * 8;   // lane's address of (row 4 khalf, column col) of the held tile
      [[maybe_unused]] int64_t hstep = 0;                   // bytes between its rows; 0 = nothing held
      if constexpr (HOLD) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) held[jj][r] = 0.f;
      }
      [[maybe_unused]] auto store_held = [&](const int r) {  // accumulator register r = row (r & 3) + 8 (r >> 2) + 4 khalf
        static_assert(!HOLD || NJ == 2, "held stores: two fp32 columns per lane");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 w = {held[0][r], held[NJ - 1][r]};
        __builtin_nontemporal_store(w, reinterpret_cast<f32x2*>(hbase + ((r & 3) + 8 * (r >> 2)) * hstep));
      };
      for (int64_t tt = first; tt < rel_end; tt += kWsWaves) {
        ++started;
        // PIPE kernels: publish my count, ask for the siblings' (four words, always — unconditional loads with an
        // unconditional position in the instruction stream are what lets the compiler wait for them with a COUNTED
        // vmcnt later instead of draining the A ring; see the check in the last round)
        [[maybe_unused]] int sib[4];
        if constexpr (PIPE) {
          if (lane == 0) __hip_atomic_store(pw + cg, started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int j = 0; j < 4; ++j) sib[j] = __hip_atomic_load(pw + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool more = tt + kWsWaves < rel_end;
        const char* nxt = more ? lane_ptr(tt + kWsWaves) : cur;
        f32x16 acc[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;
        // One round = RS slots.  The body exists TWICE (the tag only forces a second instantiation): the first round
        // of a tile runs right behind the previous tile's 16 stores, and gfx9 counts stores in vmcnt — with one
        // shared body the compiler's counted waits ("at most 12 younger loads in flight") also waited for those
        // stores to be acknowledged, once per tile (measured: the kernel without its stores 7.0 ms, with 8.1).
        // In its own copy the first round's waits count the stores as younger operations.
        auto do_round = [&](const int q, [[maybe_unused]] auto first_tag) __attribute__((always_inline)) {
          const bool last = q + 1 == nround;
          const char* rp = last ? nxt : cur;
          const int rk0 = last ? 0 : (q + 1) * RS;
          const bool refill = !last || more;
if constexpr (PIPE) {
            static_assert(!PIPE || (X3 && KSS == 2), "the pipelined body is the fp32 one");
#pragma unroll
            for (int u = 0; u < RS; ++u) {
              const int ss = q * RS + u;
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                const int e = 2 * u + i;  // (compile-time after unrolling) k-step of the round
                if (i == 1) {
                  // both halves of slot u have been split (one and two k-steps ago): refill it now, in place
                  if (u == RS - 1 && last && pace && !gave_up) {
                    // the siblings' counts asked for at the start of this tile: 4 (RS - 1) younger loads are in flight
                    // behind them, so the wait is vmcnt(12), not a drain.  More than a tile ahead of the slowest: wait.
                    // (the empty asm pins the first USE here: left alone the compiler combines the four words right
                    // behind their loads, and waits for them there)
                    int slowest = started;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                      asm volatile("" : "+v"(sib[j]));
                      slowest = (j < ncg && sib[j] < slowest) ? sib[j] : slowest;
                    }
                    slowest = __builtin_amdgcn_readfirstlane(slowest);
                    for (int spins = 0; started - slowest > 1; ++spins) {
                      if (spins > (1 << 14)) {  // a sibling that is not resident (CU masks, co-running kernels): stop pacing
                        gave_up = true;
                        break;
                      }
                      __builtin_amdgcn_s_sleep(4);
                      slowest = started;
#pragma unroll
                      for (int j = 0; j < 4; ++j) {
                        const int v = __hip_atomic_load(pw + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        slowest = (j < ncg && v < slowest) ? v : slowest;
                      }
                      slowest = __builtin_amdgcn_readfirstlane(slowest);
                    }
                  }
                  ar[u] = load_slot(rp, rk0 + u, true);
                }
                const int next_piece = (i == 0) ? 4 * phys(ss) + 1
                                                : 4 * phys((u + 1 < RS) ? ss + 1 : (last ? 0 : (q + 1) * RS));
                load_b(next_piece, bq[(e + 1) & 1]);
                // A of k-step e + 1: the second half of this slot, or the first half of the next slot (the next
                // round's / tile's slot 0 after the last one — refilled earlier in this round)
                const ASlot& an = ar[i == 0 ? u : (u + 1) % RS];
                const f32x4 xlo = __builtin_bit_cast(f32x4, an.v[i == 0 ? 2 : 0]);
                const f32x4 xhi = __builtin_bit_cast(f32x4, an.v[i == 0 ? 3 : 1]);
                __builtin_amdgcn_sched_barrier(0);
                // MFMA g, then a third of one element pair's split (2-4 VALU), nothing allowed across: the compiler's
                // own order was [36 VALU][12 MFMA] (sched_group_barrier patterns were not honoured either)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
                constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};  // small terms first
                f32x2 xr[4];
                uint32_t hp[4], mp[4], lp[4];
#pragma unroll
                for (int g = 0; g < 6 * NJ; ++g) {
                  const int term = g / NJ, jj = g % NJ;  // consecutive MFMAs go to DIFFERENT accumulators
                  acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at[e & 1][ta[term]], bq[e & 1][tb[term]][jj], acc[jj], 0, 0, 0);
                  constexpr int kStages = 12;
                  // stage st of the split rides behind MFMA st * (6 NJ) / 12
                  if ((g * kStages) % (6 * NJ) == 0) {
                    const int st = g * kStages / (6 * NJ), pr = st / 3;  // element pair pr: elements 2 pr, 2 pr + 1 of the 8
                    if (st % 3 == 0) {
                      const f32x2 x = pr < 2 ? f32x2{xlo[2 * pr], xlo[2 * pr + 1]} : f32x2{xhi[2 * pr - 4], xhi[2 * pr - 3]};
                      const b16x2 hb = __builtin_convertvector(x, b16x2);
                      hp[pr] = __builtin_bit_cast(uint32_t, hb);
                      xr[pr] = x - __builtin_convertvector(hb, f32x2);
                    } else if (st % 3 == 1) {
                      const b16x2 mb = __builtin_convertvector(xr[pr], b16x2);
                      mp[pr] = __builtin_bit_cast(uint32_t, mb);
                      xr[pr] = xr[pr] - __builtin_convertvector(mb, f32x2);
                    } else {
                      lp[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(xr[pr], b16x2));
                    }
                  }
                  __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (HOLD && decltype(first_tag)::value) {
                  if (u < 2) {  // eight rows behind each of the first four k-steps' MFMAs... early, so that the second
                                // round's waits (counted against its own loop, without the stores) find them done
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) store_held(8 * u + 4 * i + rr);
                  }
                }
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                at[(e + 1) & 1][0] = __builtin_bit_cast(b16x8, (u32x4_t{hp[0], hp[1], hp[2], hp[3]}));
                at[(e + 1) & 1][1] = __builtin_bit_cast(b16x8, (u32x4_t{mp[0], mp[1], mp[2], mp[3]}));
                at[(e + 1) & 1][2] = __builtin_bit_cast(b16x8, (u32x4_t{lp[0], lp[1], lp[2], lp[3]}));
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          } else {
#pragma unroll
            for (int u = 0; u < RS; ++u) {
              const int ss = q * RS + u;
              if (ss < nss) {  // (uniform)
#pragma unroll
                for (int i = 0; i < KSS; ++i) {
                  const int piece0 = (X3 ? 4 : 8) * phys(ss) + i;  // the h = 0 lanes' piece of this k-step
                  if (piece0 >= kpieces) continue;   // (uniform) a k-step wholly past the end of the row
                  const char* brow = smem + (l * pp + piece0 + (X3 ? 2 : 4) * khalf) * 16;
                  if constexpr (!X3) {
                    frag_t bf[NJ];
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj)
                      bf[jj] = *reinterpret_cast<const frag_t*>(brow + jj * 32 * pp * 16);
                    const frag_t af = __builtin_bit_cast(frag_t, ar[u].v[i]);
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) acc[jj] = WsFrag<DT>::mma(af, bf[jj], acc[jj]);
                  } else {
                    b16x8 ah, am, al;
                    split3(__builtin_bit_cast(f32x4, ar[u].v[2 * i]), __builtin_bit_cast(f32x4, ar[u].v[2 * i + 1]), ah, am, al);
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) {
                      const b16x8 bh = *reinterpret_cast<const b16x8*>(brow + (0 * NJ + jj) * 32 * pp * 16);
                      const b16x8 bm = *reinterpret_cast<const b16x8*>(brow + (1 * NJ + jj) * 32 * pp * 16);
                      const b16x8 bl = *reinterpret_cast<const b16x8*>(brow + (2 * NJ + jj) * 32 * pp * 16);
                      f32x16 c = acc[jj];  // small terms first
                      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
                      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
                      acc[jj] = c;
                    }
                  }
                }
              }
              ar[u] = load_slot(rp, rk0 + u, refill && rk0 + u < nss);  // in place: the ring stays in flight
              // keep program order = source order: hoisting the refill above the slot's own multiply makes the
              // compiler rotate the ring's registers and (fp32 kernel) drain the ring with vmcnt(0) once per round
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        };
        do_round(0, std::true_type{});
        for (int q = 1; q < nround; ++q) do_round(q, std::false_type{});
        // ---- epilogue: registers -> global, NJ consecutive columns per lane --------------------------------
        if constexpr (X3) {
          bool bad = false;
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) bad |= acc[jj][r] != acc[jj][r];
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0) && lane == 0) atomicOr(p.nan_flag, 1u);
        }
        const int64_t trow0 = rel_row0 + (tt - rel_t0) * 32;
        const int col = n0 + NJ * l;
        constexpr bool nt_c = true;  // C is written once and never re-read here (1.94 -> 1.90 ms at the R-GCN shape)
        // the usual tile — all 32 rows inside the segment, all of the group's columns inside N — stores without a
        // predicate (wave-uniform test); ragged tiles take the element-wise path
        const bool whole = uniform64(trow0) + 32 <= row_end && n0 + 32 * NJ <= N;
        if (HOLD && whole) {
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) held[jj] = acc[jj];
          hstep = static_cast<int64_t>(N) * ES;
          hbase = reinterpret_cast<char*>(C + (trow0 + 4 * khalf) * N + col);
        } else if (whole) {
          int64_t prow[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = trow0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if constexpr (INDEXED)
              prow[r] = p.row_index[row];
            else
              prow[r] = row;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            DT* dst = C + prow[r] * N + col;
            if constexpr (ES == 2) {
              typedef uint32_t u32xh __attribute__((ext_vector_type(NJ >= 2 ? NJ / 2 : 1)));
              if constexpr (NJ == 1) {
                *dst = from_acc<DT>(acc[0][r]);
              } else {
                u32xh w;
#pragma unroll
                for (int h = 0; h < NJ / 2; ++h) {
                  DT t2[2] = {from_acc<DT>(acc[2 * h][r]), from_acc<DT>(acc[2 * h + 1][r])};
                  uint32_t bits;
                  __builtin_memcpy(&bits, t2, 4);
                  if constexpr (NJ == 2)
                    w = bits;
                  else
                    w[h] = bits;
                }
                if (nt_c)
                  __builtin_nontemporal_store(w, reinterpret_cast<u32xh*>(dst));
                else
                  *reinterpret_cast<u32xh*>(dst) = w;
              }
            } else {
              if constexpr ((NJ & (NJ - 1)) == 0) {
                typedef float f32xn __attribute__((ext_vector_type(NJ)));
                f32xn w;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) w[jj] = acc[jj][r];
                if (nt_c)
                  __builtin_nontemporal_store(w, reinterpret_cast<f32xn*>(dst));
                else
                  *reinterpret_cast<f32xn*>(dst) = w;
              } else {  // NJ = 3: three consecutive dwords per lane (a half-wave still writes 384 contiguous bytes)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) __builtin_nontemporal_store(acc[jj][r], reinterpret_cast<float*>(dst) + jj);
              }
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = trow0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (row >= row_end) continue;
            int64_t pr = row;
            if constexpr (INDEXED) pr = p.row_index[row];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
              if (col + jj < N) C[pr * N + col + jj] = from_acc<DT>(acc[jj][r]);
          }
          // (rare path: wait for the whole ring here — through a USE, which the compiler models — so that behind the
          // join the counts are those of the usual path above, 16 stores younger than the ring)
          asm volatile("" ::"v"(ar[RS - 1].v[0]), "v"(ar[RS - 1].v[1]), "v"(ar[RS - 1].v[2]), "v"(ar[RS - 1].v[3]));
        }
        cur = nxt;
      }
      if constexpr (HOLD) {
        if (hstep != 0) {  // the wave's last whole tile of this relation
#pragma unroll
          for (int r = 0; r < 16; ++r) store_held(r);
        }
      }
    }
    t = rel_end;
  }
  }  // chunks
}

// ---- fp32 as TWO fp16 terms under per-row / per-column power-of-two scales (round 5; VERDICT r4 Next #4) -----------
// The X3 kernels above pay six bf16 MFMAs per 16 k of fp32 work and sit on the matrix pipe at the clock the chip gives
// a kernel that dense (1.59 GHz): 8.0 ms at 10 M x 256 x 256.  fp16 has three more significand bits than bf16: TWO
// terms, x s = h + l with h = fp16(x s), l = fp16(x s - h) (round to nearest; the difference is exact in fp32), carry
// 22 bits of x, and a b = (ah + al)(bh + bl) needs THREE products (hh, hl, lh; the dropped ll term is < 2^-22 |a b|):
// half the matrix-pipe work.  What fp16 lacks is bf16's exponent range, so every row of A and every column of the
// weights is multiplied by a power of two that puts its largest magnitude into [2^14, 2^15) — exact, and undone exactly
// on the accumulator (two power-of-two factors) in the epilogue:
//   * weights: h2_prep_weights_kernel, once per call — column maximum / minimum over k, the two planes, 1 / scale;
//   * A: a row's scale needs the whole row BEFORE its first element is split.  A first version read every tile twice
//     (range pass, then the multiply pass through a 4-slot ring): the second read missed L2 — 256 waves x 32 KB in
//     flight per XCD against 4 MB — and the kernel ran at the memory's rate for 4 x A + C (8.96 ms, slower than X3).
//     Now the WHOLE row (K <= 256: the lane's 64-byte half of 8 lines = 128 VGPRs) sits in the register ring; every
//     slot is refilled in place with the next tile's line behind its own k-steps, so the next row has been under way
//     for a whole tile when its range pass (|x| maximum as a float max, smallest non-zero magnitude as an integer
//     minimum of (bits << 1) - 1: 2 operations per element) asks for it.  A is read once per column group.
// The LOW term is stored pre-scaled, l = fp16((x s - h) 2^11): |x s - h| <= 2^-11 |x s| would be subnormal in fp16 for
// every element below 2^-3 of the row's maximum (and cost those elements their low bits: a first version accepted 16 +
// log2(K) / 2 binades only, and any weight tensor of 0.5 M uniform elements holds an element smaller than that with
// probability 0.4 — one such element sent EVERY row down the exact path: 153 ms instead of 4.3).  Pre-scaled, l is
// normal whenever h is, and when it is not its spacing is 2^-25 against a residual that is itself < 2^-25: an element
// keeps a relative error <= 2^-21 down to 2^28 below its row's (column's) maximum.  The two cross products then carry
// 2^11 and get an accumulator of their own: acc0 += ah bh; acc1 += ah bl' + al' bh; result = acc0 + 2^-11 acc1 (exact
// scaling, one fma per output element).  Three MFMAs per 16 k as before.
// What the split is still not trusted with (kH2Spread = 28 binades, any K):
//   * a ROW of A whose non-zero magnitudes span more, or whose maximum is Inf or outside 2^+-60 (the unscaling could
//     over- or underflow on the way): the main kernel writes it down in a short list and h2_exact_rows_kernel, launched
//     behind it and idle unless the list is non-empty, overwrites the row with the plain fp32 dot product (lanes split
//     K, coalesced weight rows, butterfly sum; Inf / NaN as IEEE arithmetic gives them) — for continuous data one row
//     in 10^6 (2.5 G elements x 2^-28).  A list that overflows makes that kernel go over every row and apply the same
//     test itself.  (Through round 5's first form the wave recomputed such rows in place, right after the tile: the
//     rare path's loops then shared the register allocation of the main loop, which parked 36 registers of the ring in
//     AGPRs — every such copy waits for its load.)
//   * single weight ELEMENTS more than 2^28 below their column's maximum: zeroed in the planes and kept in a short
//     list (relation, k, column, value); h2_fix_weights_kernel, launched behind the main kernel and idle unless the
//     list is non-empty, adds a[row][k] * value to the outputs of that relation's rows in fp32.  A list that overflows,
//     or a column whose maximum is Inf / outside 2^+-60, raises the flag that makes h2_exact_rows_kernel redo every row.
// 128 columns per workgroup (2 planes x 128 x 528 B = 132 KB of LDS), so N = 256 reads A through two column groups
// instead of X3's four.  DGLA_TUNE_MM_X3 selects the three-term kernels instead.
constexpr int kH2Spread = 28;          // binades between a row's / column's maximum and the smallest magnitude the split keeps exact to 2^-21
constexpr int kH2FixCap = 4096;        // weight elements below that the correction list holds
constexpr int kH2RowCap = 16384;       // rows of A the exact-row list holds
constexpr int kH2MinExp = 127 - 60, kH2MaxExp = 127 + 60;   // biased exponent range of a row / column maximum

struct H2Params {
  MmParams m;
  const _Float16* planes;    // [2][R * N][kp] fp16 (h, l) of the scaled, K-contiguous weights; zero padded
  const float* colinv;       // [R * N] 1 / column scale
  uint32_t* flags;           // [0] != 0: the weights cannot be split -> every row is recomputed exactly; [1]: rows listed in `rows` (more than kH2RowCap: overflow); [2]: entries in `fix`
  struct H2Fix* fix;         // [kH2FixCap] weight elements too small for their column's scale (zeroed in the planes)
  struct H2Row* rows;        // [kH2RowCap] rows of A the split is not trusted with
  const int64_t* tile32_off; // [R + 1] exclusive prefix of ceil(len / 32)
  int ncg;                   // column groups (workgroups sharing a row chunk)
  int kp;                    // row pitch of the planes in elements (K rounded up to 8)
};

struct H2Fix {
  int rel, k, n;
  float value;
};
struct H2Row {
  int64_t row;   // physical row of A and C
  int64_t rel;
};

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (largest magnitude, smallest non-zero magnitude) of a set of fp32 values, on the raw bits: zero contributes
// 0 to the maximum and 0xffffffff to the minimum key (bits << 1) - 1
struct H2Range {
  uint32_t amax;   // bits of the largest |x| (sign cleared)
  uint32_t kmin;   // min over x != 0 of (bits(x) << 1) - 1; 0xffffffff if all zero
};
__device__ __forceinline__ void h2_range_add(H2Range& r, uint32_t bits) {
  const uint32_t a = bits & 0x7fffffffu;
  r.amax = a > r.amax ? a : r.amax;
  const uint32_t k = (bits << 1) - 1u;
  r.kmin = k < r.kmin ? k : r.kmin;
}
// scale (as float bits), 1 / scale and "do not split this one" from a range
__device__ __forceinline__ bool h2_scale(const H2Range& r, float& scale, float& inv, bool check_spread = true) {
  const int e = static_cast<int>(r.amax >> 23);             // biased exponent of the maximum (0 for zero / denormal)
  if (r.amax == 0u) {                                        // all zero: any scale does
    scale = 1.f;
    inv = 1.f;
    return false;
  }
  const int emin = static_cast<int>(((r.kmin + 1u) >> 1) >> 23);
  const bool bad = e < kH2MinExp || e > kH2MaxExp || (check_spread && e - emin > kH2Spread);
  const int ec = e < kH2MinExp ? kH2MinExp : (e > kH2MaxExp ? kH2MaxExp : e);   // (keeps the bit patterns below valid)
  scale = __builtin_bit_cast(float, static_cast<uint32_t>(127 + 14 + 127 - ec) << 23);   // 2^(14 - (e - 127))
  inv = __builtin_bit_cast(float, static_cast<uint32_t>(ec - 14) << 23);                   // 2^((e - 127) - 14)
  return bad;
}

// Weights: one wave per K-contiguous row (relation, column) of Bt.
__global__ __launch_bounds__(256) void h2_prep_weights_kernel(const float* __restrict__ bt, _Float16* __restrict__ planes,
                                                              float* __restrict__ colinv, uint32_t* __restrict__ flags,
                                                              H2Fix* __restrict__ fix, int64_t rows, int N, int K, int kp) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  const int64_t total = rows * kp;
  for (int64_t row = blockIdx.x * static_cast<int64_t>(blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += nw) {
    const float* src = bt + row * K;
    H2Range r{0u, 0xffffffffu};
    for (int k = lane; k < K; k += 64) h2_range_add(r, __builtin_bit_cast(uint32_t, src[k]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const uint32_t om = __shfl_xor(r.amax, d, 64), ok = __shfl_xor(r.kmin, d, 64);
      r.amax = om > r.amax ? om : r.amax;
      r.kmin = ok < r.kmin ? ok : r.kmin;
    }
    float sc, inv;
    const bool bad = h2_scale(r, sc, inv, /*check_spread=*/false);   // (too-small ELEMENTS are handled one by one below)
    const float tiny = 6.103515625e-05f;                               // 2^-14: the smallest normal fp16
    for (int k = lane; k < kp; k += 64) {
      float x = k < K ? src[k] * sc : 0.f;
      if (x != 0.f && __builtin_fabsf(x) < tiny && !bad) {
        // more than 2^28 below the column's maximum: not in the planes; added in fp32 by h2_fix_weights_kernel
        const uint32_t at = atomicAdd(flags + 2, 1u);
        if (at < static_cast<uint32_t>(kH2FixCap))
          fix[at] = H2Fix{static_cast<int>(row / N), k, static_cast<int>(row % N), src[k]};
        else
          atomicOr(flags, 1u);
        x = 0.f;
      }
      const _Float16 h = static_cast<_Float16>(x);
      const _Float16 l = static_cast<_Float16>((x - static_cast<float>(h)) * 2048.f);
      planes[row * kp + k] = h;
      planes[total + row * kp + k] = l;
    }
    if (lane == 0) {
      colinv[row] = inv;
      if (bad) atomicOr(flags, 1u);
    }
  }
}

// out[row][n] += a[row][k] * value for every listed weight element and every row of its relation (fp32; the main kernel
// multiplied a zero there).  Idle — one load per thread — unless the list is non-empty.
__global__ __launch_bounds__(256) void h2_fix_weights_kernel(const MmParams p, const uint32_t* __restrict__ flags,
                                                             const H2Fix* __restrict__ fix) {
  if (flags[0] != 0u) return;                       // every row went down the exact path with the true weights
  const uint32_t cnt = flags[2] < static_cast<uint32_t>(kH2FixCap) ? flags[2] : static_cast<uint32_t>(kH2FixCap);
  if (cnt == 0u) return;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const float* A = static_cast<const float*>(p.a);
  float* C = static_cast<float*>(p.c);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (uint32_t i = 0; i < cnt; ++i) {              // (entries of one output element are added one after the other: no race)
    const H2Fix f = fix[i];
    for (int64_t r = row_off[f.rel] + blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < row_off[f.rel + 1]; r += stride) {
      const int64_t pr = p.row_index ? p.row_index[r] : r;
      C[pr * p.N + f.n] = __builtin_fmaf(A[pr * p.K + f.k], f.value, C[pr * p.N + f.n]);
    }
  }
}

// Rows of A the split is not trusted with: out[row][:] = the plain fp32 dot products.  One wave per row; lanes split K
// — every weight row is one coalesced read of up to 1 KB — and a butterfly adds the 64 partial sums (a first version
// gave every lane its own column and walked k: 64 cache lines per load, 50 us per row).  Idle — two loads per thread —
// unless the main kernel listed a row.  With flags[0] set (weights that cannot be split) or a list that overflowed it
// goes over EVERY row instead; after an overflow it applies the main kernel's own test to each.
__global__ __launch_bounds__(256) void h2_exact_rows_kernel(const MmParams p, const uint32_t* __restrict__ flags,
                                                            const H2Row* __restrict__ rows) {
  const bool all = flags[0] != 0u;
  const uint32_t listed = flags[1];
  if (!all && listed == 0u) return;
  const bool scan = all || listed > static_cast<uint32_t>(kH2RowCap);
  const int lane = threadIdx.x & 63;
  const int K = p.K, N = p.N;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const float* A = static_cast<const float*>(p.a);
  const float* Bt = static_cast<const float*>(p.bt);
  float* C = static_cast<float*>(p.c);
  const int64_t nw = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  const int64_t count = scan ? row_off[p.num_rel] : static_cast<int64_t>(listed);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x >> 6) + (threadIdx.x >> 6); i < count; i += nw) {
    int64_t pr, rel;
    if (scan) {
      rel = find_segment(row_off, p.num_rel, i);
      pr = p.row_index ? p.row_index[i] : i;
    } else {
      pr = rows[i].row;
      rel = rows[i].rel;
    }
    const float* __restrict__ arow = A + pr * K;
    float av[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = 4 * lane + j < K ? arow[4 * lane + j] : 0.f;
    if (scan && !all) {  // the main kernel's test (its list overflowed)
      float fmax_ = 0.f;
      uint32_t kmin = 0xffffffffu;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fmax_ = __builtin_fmaxf(fmax_, __builtin_fabsf(av[j]));
        const uint32_t k = (__builtin_bit_cast(uint32_t, av[j]) << 1) - 1u;
        kmin = k < kmin ? k : kmin;
      }
      H2Range rg{__builtin_bit_cast(uint32_t, fmax_), kmin};
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const uint32_t om = __shfl_xor(rg.amax, d, 64), ok = __shfl_xor(rg.kmin, d, 64);
        rg.amax = om > rg.amax ? om : rg.amax;
        rg.kmin = ok < rg.kmin ? ok : rg.kmin;
      }
      float sc, inv;
      if (!h2_scale(rg, sc, inv)) continue;
    }
    for (int c0 = 0; c0 < N; c0 += 4) {
      float part[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = c0 + q < N ? c0 + q : N - 1;
        const float* __restrict__ brow = Bt + (rel * N + n) * static_cast<int64_t>(K);
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sacc = __builtin_fmaf(av[j], 4 * lane + j < K ? brow[4 * lane + j] : 0.f, sacc);
        part[q] = sacc;
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) part[q] += __shfl_xor(part[q], d, 64);
      if (lane < 4 && c0 + lane < N) C[pr * N + c0 + lane] = part[lane & 3];
    }
  }
}

// 8 floats of a row (two 16-byte pieces) -> the MFMA fragments of their two fp16 terms
__device__ __forceinline__ void split2(const f32x4 lo4, const f32x4 hi4, const float s, h16x8& h, h16x8& l) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  uint32_t hp[4], lp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // elements 2q, 2q + 1 of the 8
    const f32x2 x = q < 2 ? f32x2{lo4[2 * q], lo4[2 * q + 1]} : f32x2{hi4[2 * q - 4], hi4[2 * q - 3]};
    const f32x2 xs = x * s;                                   // exact: s is a power of two
    const h16x2 hb = __builtin_convertvector(xs, h16x2);       // round to nearest
    const f32x2 r = (xs - __builtin_convertvector(hb, f32x2)) * 2048.f;   // exact difference, exact scaling
    const h16x2 lb = __builtin_convertvector(r, h16x2);
    hp[q] = __builtin_bit_cast(uint32_t, hb);
    lp[q] = __builtin_bit_cast(uint32_t, lb);
  }
  h = __builtin_bit_cast(h16x8, (u32x4_t{hp[0], hp[1], hp[2], hp[3]}));
  l = __builtin_bit_cast(h16x8, (u32x4_t{lp[0], lp[1], lp[2], lp[3]}));
}

// FOUR waves per workgroup, one per SIMD: the ring (128) + the accumulators (64) + one plane's fragments (16) + the split
// leave no room inside the 256 registers two waves per SIMD would get (first build: 55 VGPRs spilled); with 512 the
// kernel has no spill, and one wave keeps its SIMD's matrix pipe busy on its own — twelve independent MFMAs per k-step.
constexpr int kH2Waves = 4;

// NSS > 0: K is exactly NSS whole 128-byte lines (K = 32 NSS) — every guard below is a compile-time constant.  The generic
// form (NSS = 0: K, the number of lines and of weight pieces at run time) keeps uniform branches around every MFMA, LDS
// read and refill; the compiler then neither interleaves the split with the MFMAs nor keeps the guards in scalar
// registers (200 SGPRs spilled through v_writelane inside the loop): 6.3 ms against the specialised form's time.
// COLW: N is a multiple of the workgroup's 32 NJ columns — the store loop then has no predicate at all (a per-lane
// `column < N` around the stores was enough for the compiler to park 40 ring registers in AGPRs again).
template <int NJ, bool INDEXED, int NSS, bool COLW>
__global__ __launch_bounds__(64 * kH2Waves) void segment_mm_h2_kernel(const H2Params hp) {
  constexpr bool FULL = NSS > 0;
  const MmParams& p = hp.m;
  constexpr int ES = 4;
  constexpr int ROWS = 2 * NJ * 32;
  __shared__ __attribute__((aligned(1024))) char smem[ROWS * kWsPitchPieces * 16 + 1024];  // (+ the last DMA instruction's overhang)

  const int K = FULL ? 32 * NSS : p.K, N = p.N;
  const int nss = FULL ? NSS : (K * ES + 127) >> 7;          // 128-byte lines per A row
  const int pp = 4 * nss + 1;                   // LDS row pitch in 16-byte pieces (odd): 8 fp16 per piece, 32 k per line
  const int kpieces = FULL ? 4 * NSS : hp.kp >> 3;   // 16-byte pieces with data per weight row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & 31, khalf = lane >> 5;

  const int64_t* __restrict__ tile_off = hp.tile32_off;
  const int64_t* __restrict__ row_off = p.plan + p.num_rel + 1;
  const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
  const int ncg = hp.ncg;
  const int cg = qq % ncg;
  const int ngroups = (static_cast<int>(gridDim.x) >> 3) / ncg * 8;
  const int group = (qq / ncg) * 8 + xcd;
  if (group >= ngroups) return;
  const int64_t T = tile_off[p.num_rel];
  const int n0 = cg * 32 * NJ;

  const char* __restrict__ A = static_cast<const char*>(p.a);
  const float* __restrict__ Bt = static_cast<const float*>(p.bt);
  float* __restrict__ C = static_cast<float*>(p.c);

  const int64_t t0 = uniform64(T * group / ngroups), t1 = uniform64(T * (group + 1) / ngroups);
  int64_t t = t0;
  while (t < t1) {
    const int64_t rel = uniform64(find_segment(tile_off, p.num_rel, t));
    const int64_t rel_t0 = uniform64(tile_off[rel]);
    int64_t rel_end = uniform64(tile_off[rel + 1]);
    if (rel_end > t1) rel_end = t1;
    const int64_t rel_row0 = uniform64(row_off[rel]), row_end = uniform64(row_off[rel + 1]);

    // ---- this relation's two weight planes -> LDS (row rho = plane * NJ * 32 + jj * 32 + ll <-> column n0 + NJ ll + jj) ----
    __syncthreads();  // nobody still multiplies from the previous relation's image
    {
      const int total = ROWS * pp, ninst = (total + 63) >> 6;
      const int64_t rpitch = static_cast<int64_t>(hp.kp) * 2;
      const int64_t plane_bytes = static_cast<int64_t>(p.num_rel) * N * rpitch;
      const char* __restrict__ W = reinterpret_cast<const char*>(hp.planes);
      for (int i = wave; i < ninst; i += kH2Waves) {
        const int j = i * 64 + lane;
        const int rho = j / pp, pc = j - rho * pp;
        const int pl = rho / (NJ * 32), jj = (rho >> 5) % NJ, ll = rho & 31;
        const int n = n0 + NJ * ll + jj;
        const bool valid = j < total && pc < kpieces && n < N;
        const char* src = valid ? W + pl * plane_bytes + (rel * N + n) * rpitch + pc * 16 : g_mm_zero_page;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + i * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces have landed
    }
    // 1 / scale of this lane's NJ output columns
    float cinv[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const int n = n0 + NJ * l + jj;
      cinv[jj] = n < N ? hp.colinv[rel * N + n] : 0.f;
    }
    __syncthreads();

    auto row_ptr = [&](int64_t tt) -> const char* {   // this lane's row of tile tt (clamped to the segment)
      int64_t row = rel_row0 + (tt - rel_t0) * 32 + l;
      if (row >= row_end) row = row_end - 1;
      if constexpr (INDEXED) row = p.row_index[row];
      return A + row * (static_cast<int64_t>(K) * ES);
    };
    struct ASlot {
      u32x4 v[4];
    };
    auto load_slot = [&](const char* base, int ss, bool on) -> ASlot {   // the lane's 64-byte half of line ss
      ASlot f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = (ss * 128 + khalf * 64 + i * 16) / ES;
        // FULL: every piece of every line exists and a refill with nothing left to fetch re-reads the current row — no
        // select, no compare in the loop
        const char* src = FULL ? base + ss * 128 + khalf * 64 + i * 16
                               : ((on && e < K) ? base + ss * 128 + khalf * 64 + i * 16 : reinterpret_cast<const char*>(g_mm_zero_page));
        f.v[i] = *reinterpret_cast<const u32x4*>(src);
      }
      return f;
    };

    // The whole row lives in the register ring: 8 slots = the lane's 64-byte half of every 128-byte line of a row of up
    // to 256 floats (32 loads in flight per lane, 16 KB per wave).  A slot is refilled IN PLACE with the next tile's
    // line the moment it has been multiplied, so when a tile's last k-step is done the next tile's row has been under
    // way for a whole tile: its range pass (which needs every element before the first split) finds it there.  A is
    // read ONCE per column group; the two groups of a row chunk run on one XCD and share its L2.
    constexpr int NS = FULL ? NSS : kWsMaxK * ES / 128;   // ring slots: the row's lines (8 when K is only known at run time)
    const int64_t first = t + wave;
    if (first < rel_end) {
    const char* cur = row_ptr(first);
    ASlot ar[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) ar[u] = load_slot(cur, u, u < nss);
    // (the first row is waited for HERE, through a use of every load: at the loop header the compiler merges what
    // is outstanding on the way in with what is on the back edge — 16 stores younger than the ring — and with 32 fresh
    // loads on the way in it drained the counter at the top of EVERY tile, stores included)
#pragma unroll
    for (int u = 0; u < NS; ++u) asm volatile("" ::"v"(ar[u].v[0]), "v"(ar[u].v[1]), "v"(ar[u].v[2]), "v"(ar[u].v[3]));
    for (int64_t tt = first; tt < rel_end; tt += kH2Waves) {
      const bool more = tt + kH2Waves < rel_end;
      const char* nxt = more ? row_ptr(tt + kH2Waves) : cur;
      // ---- the row's range: largest magnitude as a float maximum (|x| is a source modifier, NaN is ignored, Inf wins),
      // smallest non-zero magnitude as an integer minimum of (bits << 1) - 1; this lane's half, then both lanes of the row.
      // (Two scales per row — one per half of the contraction, each with its own accumulators, so that both range passes
      // could wait for their slots with a COUNT — were built and measured: 5.56 -> 5.75 ms.  The compiler keeps draining
      // at the loop header: it parks part of the 128-register ring in AGPRs across the back edge, and every such copy
      // needs its load.  One scale per row it is.)
      float fmax_ = 0.f;
      uint32_t kmin = 0xffffffffu;
#pragma unroll
      for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 w = ar[u].v[i];
          const f32x4 x = __builtin_bit_cast(f32x4, w);
          fmax_ = __builtin_fmaxf(__builtin_fmaxf(fmax_, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));
          fmax_ = __builtin_fmaxf(__builtin_fmaxf(fmax_, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
          const uint32_t k0 = (w[0] << 1) - 1u, k1 = (w[1] << 1) - 1u, k2 = (w[2] << 1) - 1u, k3 = (w[3] << 1) - 1u;
          const uint32_t m01 = k0 < k1 ? k0 : k1;
          kmin = kmin < m01 ? kmin : m01;
          const uint32_t m23 = k2 < k3 ? k2 : k3;
          kmin = kmin < m23 ? kmin : m23;
        }
      H2Range rg{__builtin_bit_cast(uint32_t, fmax_), kmin};
      {
        const uint32_t om = __shfl_xor(rg.amax, 32, 64), ok = __shfl_xor(rg.kmin, 32, 64);
        rg.amax = om > rg.amax ? om : rg.amax;
        rg.kmin = ok < rg.kmin ? ok : rg.kmin;
      }
      float sc0, rinv0;
      const bool row_bad = h2_scale(rg, sc0, rinv0);
      const int64_t trow0 = rel_row0 + (tt - rel_t0) * 32;

      // ---- scaled, split, multiplied: slot by slot, each refilled with the next tile's line behind its k-steps ------
      // One wave per SIMD and an in-order issue: what keeps the matrix pipe busy is the ORDER of the instruction stream.
      // The split of k-step e + 1 (four element pairs x three stages, 4-5 VALU each) and the LDS reads of its weight
      // fragments are issued BETWEEN the twelve MFMAs of k-step e, one stage behind each MFMA (32 pipe cycles cover
      // them), pinned with sched_barrier — left to itself the compiler emits [split][reads][12 MFMAs] and the pipe
      // idles through the first two (7.7 ms instead of the two-read version's 9.0, but no better than X3).
      f32x16 acc[2][NJ];           // [0: ah bh | 1: ah bl' + al' bh, both carrying the low terms' 2^11][column block]
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[h][jj][r] = 0.f;
      h16x8 sh[2], sl[2];          // the two terms of k-step e (index e & 1)
      // weight fragments [plane: 0 = h, 1 = l][column block], ONE buffer: a fragment's register is re-loaded with the next
      // k-step's right behind the last MFMA that reads it (plane l: term 0; plane h: term 2) — seven MFMAs before its
      // next use.  (A double buffer put the kernel over 256 architectural VGPRs: the compiler parked part of the ring
      // in AGPRs, and every copy there needs its load to have landed: s_waitcnt vmcnt(3) behind the stores.)
      h16x8 bq[2][NJ];
      auto load_b1 = [&](int piece0, int pl, int jj) -> h16x8 {
        return *reinterpret_cast<const h16x8*>(smem + ((pl * NJ + jj) * 32 + l) * pp * 16 + (piece0 + 2 * khalf) * 16);
      };
      split2(__builtin_bit_cast(f32x4, ar[0].v[0]), __builtin_bit_cast(f32x4, ar[0].v[1]), sc0, sh[0], sl[0]);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) bq[pl][jj] = load_b1(0, pl, jj);
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        if (FULL || u < nss) {  // (uniform)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int e = 2 * u + i;                     // (compile-time after unrolling)
            const int piece0 = 4 * u + i;                // the khalf = 0 lanes' piece of this k-step
            // k-step e + 1: the second half of this slot, or the first half of the next one (past the row's end: zeros)
            const ASlot& an = ar[i == 0 ? u : (u + 1 < NS ? u + 1 : u)];
            const f32x4 xlo = __builtin_bit_cast(f32x4, an.v[i == 0 ? 2 : 0]);
            const f32x4 xhi = __builtin_bit_cast(f32x4, an.v[i == 0 ? 3 : 1]);
            const int npiece = i == 0 ? piece0 + 1 : 4 * (u + 1);
            const bool nread = FULL ? (4 * u + i + 1 < 4 * NSS) : npiece < 4 * nss;   // (uniform) fragments of a k-step past the last line are not read
            f32x2 xs[4];
            uint32_t hw[4], lw[4];
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NM = 3 * NJ;                   // MFMAs of a k-step
#pragma unroll
            for (int g = 0; g < NM; ++g) {
              const int term = g / NJ, jj = g % NJ;      // consecutive MFMAs go to DIFFERENT accumulators; small terms first
              if (FULL || piece0 < kpieces) {            // (uniform) a k-step wholly past the end of the row multiplies nothing
                if (term == 0)
                  acc[1][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[e & 1], bq[1][jj], acc[1][jj], 0, 0, 0);
                else if (term == 1)
                  acc[1][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl[e & 1], bq[0][jj], acc[1][jj], 0, 0, 0);
                else
                  acc[0][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh[e & 1], bq[0][jj], acc[0][jj], 0, 0, 0);
              }
              // behind MFMA g: stages [g * 12 / NM, (g + 1) * 12 / NM) of the next k-step's split, and one or two of
              // its 2 NJ fragment reads
#pragma unroll
              for (int st = g * 12 / NM; st < (g + 1) * 12 / NM; ++st) {
                const int pr = st / 3;                   // element pair pr: elements 2 pr, 2 pr + 1 of the 8
                if (st % 3 == 0) {
                  const f32x2 x = pr < 2 ? f32x2{xlo[2 * pr], xlo[2 * pr + 1]} : f32x2{xhi[2 * pr - 4], xhi[2 * pr - 3]};
                  xs[pr] = x * sc0;                                       // exact: a power of two
                } else if (st % 3 == 1) {
                  const h16x2 hb = __builtin_convertvector(xs[pr], h16x2);  // round to nearest
                  hw[pr] = __builtin_bit_cast(uint32_t, hb);
                  xs[pr] = (xs[pr] - __builtin_convertvector(hb, f32x2)) * 2048.f;   // exact difference, exact scaling
                } else {
                  lw[pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(xs[pr], h16x2));
                }
              }
              if (nread && term == 0) bq[1][jj] = load_b1(npiece, 1, jj);   // this MFMA was the fragment's last reader
              if (nread && term == 2) bq[0][jj] = load_b1(npiece, 0, jj);
              __builtin_amdgcn_sched_barrier(0);
            }
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            sh[(e + 1) & 1] = __builtin_bit_cast(h16x8, (u32x4_t{hw[0], hw[1], hw[2], hw[3]}));
            sl[(e + 1) & 1] = __builtin_bit_cast(h16x8, (u32x4_t{lw[0], lw[1], lw[2], lw[3]}));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        ar[u] = load_slot(nxt, u, more && u < nss);   // in place: the next tile's row is under way from here on
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue: undo the scales (exact), registers -> global, NJ consecutive columns per lane ----------------
      // rows the split is not trusted with go on the list (lane l < 32 holds row trow0 + l; rows past the segment's
      // end are clamped copies); what is stored for them below is overwritten by h2_exact_rows_kernel
      if (row_bad && khalf == 0 && cg == 0 && trow0 + l < row_end) {
        const uint32_t at = atomicAdd(hp.flags + 1, 1u);
        if (at < static_cast<uint32_t>(kH2RowCap)) {
          int64_t pr = trow0 + l;
          if constexpr (INDEXED) pr = p.row_index[pr];
          hp.rows[at] = H2Row{pr, rel};
        }
      }
      const int col = n0 + NJ * l;
      // one output row-register at a time: 2 NJ accumulator reads, the two exact unscalings, one store — computing the
      // whole tile first held 64 more VGPRs over the ring's 128 and the compiler parked ring registers in AGPRs, each
      // copy waiting for its load (s_waitcnt vmcnt(0) at the top of every tile)
      auto out_row = [&](const int r, float (&o)[NJ]) {
        // lane rho (< 32) holds row rho's 1 / scale; this lane's row is rho_a (lanes 0-31) or rho_a + 4 (lanes 32-63): two
        // scalar lane reads and a select — a ds_bpermute per row was an LDS round trip the store behind it waited for
        // (s_waitcnt lgkmcnt(0) sixteen times per tile)
        const int rho_a = (r & 3) + 8 * (r >> 2);
        const int ibits = __builtin_bit_cast(int, rinv0);
        const float ra = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ibits, rho_a));
        const float rb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ibits, rho_a + 4));
        const float r0 = khalf ? rb : ra;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) o[jj] = __builtin_fmaf(acc[1][jj][r], 0x1p-11f, acc[0][jj][r]) * r0 * cinv[jj];
      };
      // A segment's last tile needs no predicate: its rows past the end were READ as copies of the last row (row_ptr
      // clamps), so their accumulators hold the last row's result bit for bit, and they are STORED to the last row as
      // well — several lanes writing one value to one address.  One store path for whole and ragged tiles: a second,
      // predicated path here (and the exact-row loops behind it, through round 5's first form) shared the main loop's
      // register allocation and cost it 36 ring registers parked in AGPRs.
      const int last = static_cast<int>(uniform64(row_end - trow0 < 32 ? row_end - trow0 : 32)) - 1;
      float* const cbase = C + trow0 * N + col;   // (one 64-bit base per tile; a row costs a 32-bit multiply and one add)
      auto row_ptr_c = [&](const int r) -> float* {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const int rc = rho < last ? rho : last;
        if constexpr (INDEXED)
          return C + p.row_index[trow0 + rc] * N + col;
        else
          return cbase + static_cast<uint32_t>(rc) * static_cast<uint32_t>(N);
      };
      const bool vec = COLW || col + NJ <= N;    // (all lanes but those of the last column group of a ragged N)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o[NJ];
        out_row(r, o);
        float* dst = row_ptr_c(r);
        if (vec) {
          if constexpr (NJ == 4) {
            const f32x4 w = {o[0], o[1], o[2], o[3]};
            __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(dst));
          } else if constexpr (NJ == 2) {
            const f32x2 w = {o[0], o[1]};
            __builtin_nontemporal_store(w, reinterpret_cast<f32x2*>(dst));
          } else {
            __builtin_nontemporal_store(o[0], dst);
          }
        } else if constexpr (NJ > 1) {            // the lane that straddles column N
#pragma unroll
          for (int jj = 0; jj < NJ - 1; ++jj)
            if (col + jj < N) dst[jj] = o[jj];
        }
        __builtin_amdgcn_sched_barrier(0);   // one row at a time: hoisting all 2 x 64 accumulator reads is 128 VGPRs over the ring's 128
      }
      cur = nxt;
    }
    }
    t = rel_end;
  }
}

// fp64 (and any shape the MFMA path does not take): one thread per output element.
template <typename DT>
__global__ __launch_bounds__(256) void segment_mm_plain_kernel(const MmParams p, int64_t M) {
  using A_ = typename Acc<DT>::type;
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= M * p.N) return;
  const int64_t row = idx / p.N;
  const int col = static_cast<int>(idx - row * p.N);
  const int64_t* row_off = p.plan + p.num_rel + 1;
  int64_t lo = 0, hi = p.num_rel - 1;  // largest r with row_off[r] <= row
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (row_off[mid] <= row)
      lo = mid;
    else
      hi = mid - 1;
  }
  if (row >= row_off[p.num_rel]) return;  // rows beyond sum(seglen): zeroed by segment_zero_tail_kernel, not here
  const int64_t prow = p.row_index ? p.row_index[row] : row;
  const DT* a = static_cast<const DT*>(p.a) + prow * p.K;
  const DT* b = static_cast<const DT*>(p.bt) + (lo * p.N + col) * static_cast<int64_t>(p.K);
  A_ acc = 0;
  for (int k = 0; k < p.K; ++k) acc += to_acc<DT>(a[k]) * to_acc<DT>(b[k]);
  static_cast<DT*>(p.c)[prow * p.N + col] = from_acc<DT>(acc);
}

constexpr int BN = 128;  // weight-gradient tile width

// ---- weight gradient: dB_r[i][j] = sum_{rows m of r} A[m][i] * dC[m][j] -------------------
struct MmBwdParams {
  const void* a;   // [M, D1]
  const void* dc;  // [M, D2]
  float* acc;      // [R, D1, D2] fp32, zeroed
  const int64_t* plan;  // slab table: rows_per_tile = slab_rows
  int64_t num_rel;
  int D1, D2;
  int vec_a, vec_dc;  // 16-byte loads allowed (row pitch and base 16-byte aligned)
  int tiles_i, tiles_j;  // ceil(D1 / 128), ceil(D2 / 128)
  int64_t slab_rows;     // rows per split-K slab
  const int64_t* row_index;  // optional: logical row r of A and dC lives at physical row row_index[r]
  uint32_t* nan_flag;  // X3 kernel: raised when an accumulator came out NaN (zeroed by the plan kernel)
  const uint32_t* only_if = nullptr;  // X3 kernel behind the two-term kernel: run only when this word is non-zero
};

// ---- repair passes of the X3 kernels (see split3) ------------------------------------------------
// Launched behind every X3 kernel; every thread returns after one load unless the flag is up.
// Forward: every NaN element of C is recomputed as the fp32 dot product IEEE arithmetic defines.
__global__ __launch_bounds__(256) void x3_repair_fwd_kernel(const MmParams p) {
  if (*p.nan_flag == 0) return;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const int64_t total = row_off[p.num_rel];
  const float* A = static_cast<const float*>(p.a);
  const float* Bt = static_cast<const float*>(p.bt);
  float* C = static_cast<float*>(p.c);
  const int64_t n_el = total * p.N, stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < n_el; e += stride) {
    const int64_t r = e / p.N;
    const int col = static_cast<int>(e % p.N);
    const int64_t pr = p.row_index ? p.row_index[r] : r;
    const float v = C[pr * p.N + col];
    if (v == v) continue;
    const int64_t rel = find_segment(row_off, p.num_rel, r);
    const float* a = A + pr * p.K;
    const float* b = Bt + (rel * p.N + col) * static_cast<int64_t>(p.K);
    float acc = 0.f;
    for (int k = 0; k < p.K; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
    C[pr * p.N + col] = acc;
  }
}

// Weight gradient: one wave per element of acc[R, D1, D2]; a NaN element is recomputed over the
// whole segment (lanes stride the rows, partial sums combined by shuffles).
__global__ __launch_bounds__(256) void x3_repair_bwd_kernel(const MmBwdParams p) {
  if (*p.nan_flag == 0) return;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const float* A = static_cast<const float*>(p.a);
  const float* dC = static_cast<const float*>(p.dc);
  const int lane = threadIdx.x & 63;
  const int64_t n_el = p.num_rel * static_cast<int64_t>(p.D1) * p.D2;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6);
  for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x >> 6) + (threadIdx.x >> 6); e < n_el; e += n_waves) {
    const float v = p.acc[e];
    if (v == v) continue;
    const int64_t rel = e / (static_cast<int64_t>(p.D1) * p.D2);
    const int row = static_cast<int>((e / p.D2) % p.D1), col = static_cast<int>(e % p.D2);
    float acc = 0.f;
    for (int64_t m = row_off[rel] + lane; m < row_off[rel + 1]; m += 64) {
      const int64_t pm = p.row_index ? p.row_index[m] : m;
      acc = __builtin_fmaf(A[pm * p.D1 + row], dC[pm * p.D2 + col], acc);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) p.acc[e] = acc;
  }
}

template <typename DT>
__global__ __launch_bounds__(256) void segment_mm_bwd_b_kernel(const MmBwdParams p) {
  using M = Mma<DT>;
  constexpr int KE = kBwdSlab / sizeof(DT);  // slab rows (contraction index) per step
  constexpr int E = 16 / sizeof(DT);
  constexpr int kPitch = kBwdSlab + 16;
  __shared__ __attribute__((aligned(16))) char sA[BM * kPitch];
  __shared__ __attribute__((aligned(16))) char sB[BN * kPitch];

  // 1-D grid, XCD-aware: the (D1 / 128) x (D2 / 128) output tiles of one row slab go to
  // CONSECUTIVE workgroups of the same XCD (workgroup L runs on XCD L % 8), so the slab's rows
  // of A and dC (2 x 1 MB for 256-wide bf16 features) come from HBM once and from that
  // XCD's L2 for the other tiles.
  const int64_t L = blockIdx.x;
  const int64_t jd = L >> 3;
  const int tiles = p.tiles_i * p.tiles_j;
  const int64_t slab = (jd / tiles) * 8 + (L & 7);
  const int tile = static_cast<int>(jd % tiles);
  const int64_t* slab_off = p.plan;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  if (slab >= slab_off[p.num_rel]) return;
  const int64_t rel = find_segment(slab_off, p.num_rel, slab);
  const int64_t m0 = row_off[rel] + (slab - slab_off[rel]) * p.slab_rows;
  int64_t m1 = m0 + p.slab_rows;
  if (m1 > row_off[rel + 1]) m1 = row_off[rel + 1];
  const int i0 = (tile / p.tiles_j) * BM, j0 = (tile % p.tiles_j) * BN;
  const int D1 = p.D1, D2 = p.D2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Transposing stage: the slab rows (contraction index) become the K-contiguous LDS axis.
  // Threads 0-127 move the A part, 128-255 the dC part.  Inside a part, thread u = (cgrp, kk):
  // it loads 16-byte pieces of slab row kk (E consecutive features) and writes each element to
  // LDS row `feature`, byte kk * sizeof(DT).  Lanes of a wavefront differ in kk first, so one
  // ds_write covers consecutive bytes of ONE LDS row: no bank conflicts; the strided 16-byte
  // global reads of a wave touch each 128-byte line completely over its 128 / 16 pieces.
  const int which = tid >> 7, u = tid & 127;
  const DT* __restrict__ src = static_cast<const DT*>(which ? p.dc : p.a);
  const int width = which ? D2 : D1;
  const int f0 = which ? j0 : i0;
  char* sT = which ? sB : sA;
  const int kk = u % KE, cgrp = u / KE;
  constexpr int CGRPS = 128 / KE;            // chunk groups among the 128 threads of a part
  constexpr int CHUNKS = 128 / E;            // 16-byte chunks per 128 features
  constexpr int CPT = CHUNKS / CGRPS;        // chunks per thread
  const bool vec_ok = (which ? p.vec_dc : p.vec_a) != 0;

  u32x4 regs[CPT];
  auto fetch = [&](int64_t m) {
    const int64_t row = m + kk;
    const int64_t prow = (row < m1 && p.row_index) ? p.row_index[row] : row;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int f = f0 + (cgrp + c * CGRPS) * E;
      regs[c] = load_piece<DT>(src + prow * width + f, row < m1 ? width - f : 0, vec_ok);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int fl = (cgrp + c * CGRPS) * E;
      DT tmp[E];
      __builtin_memcpy(tmp, &regs[c], 16);
#pragma unroll
      for (int e = 0; e < E; ++e)
        *reinterpret_cast<DT*>(sT + (fl + e) * kPitch + kk * sizeof(DT)) = tmp[e];
    }
  };

  fetch(m0);
  const int lrow = lane & 31, khalf = lane >> 5;
  for (int64_t m = m0; m < m1; m += KE) {
    __syncthreads();
    stash();
    __syncthreads();
    if (m + KE < m1) fetch(m + KE);
#pragma unroll
    for (int s = 0; s < KE / M::KSTEP; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* sa = sA + (wm * 64 + i * 32 + lrow) * kPitch;
#pragma unroll
        for (int j = 0; j < 2; ++j)
          M::step(sa, sB + (wn * 64 + j * 32 + lrow) * kPitch, s, khalf, acc[i][j]);
      }
  }
  float* out = p.acc + rel * static_cast<int64_t>(D1) * D2;
  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wn * 64 + j * 32 + col_l;
      if (col >= D2) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (row < D1) atomicAdd(out + static_cast<int64_t>(row) * D2 + col, acc[i][j][r]);
      }
    }
}

template <typename DT>
__global__ void convert_from_f32_kernel(const float* __restrict__ src, DT* __restrict__ dst,
                                        int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += stride)
    dst[i] = from_acc<DT>(src[i]);
}

// fp64 weight gradient: one thread per (relation, i, j), sequential over the rows.
__global__ __launch_bounds__(256) void segment_mm_bwd_b_f64_kernel(const double* __restrict__ a,
                                                                  const double* __restrict__ dc,
                                                                  double* __restrict__ db,
                                                                  const int64_t* __restrict__ plan,
                                                                  int64_t num_rel, int D1, int D2,
                                                                  const int64_t* __restrict__ row_index) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= num_rel * D1 * D2) return;
  const int64_t rel = idx / (static_cast<int64_t>(D1) * D2);
  const int ij = static_cast<int>(idx - rel * D1 * D2);
  const int i = ij / D2, j = ij - i * D2;
  const int64_t* row_off = plan + num_rel + 1;
  double acc = 0;
  for (int64_t m = row_off[rel]; m < row_off[rel + 1]; ++m) {
    const int64_t pm = row_index ? row_index[m] : m;
    acc += a[pm * D1 + i] * dc[pm * D2 + j];
  }
  db[idx] = acc;
}

// ---- gather mm: C[idx_c[i]] (+)= A[idx_a[i]] . B[idx_b[i]]  (GatherMMScatterKernel) --------
// One wavefront per row; used for the small shapes the Python layer keeps on this path
// (python/dgl/ops/gather_mm.py:39-60: D1, D2 <= 8 and N <= 1e6), everything else is sorted and
// sent through segment_mm.
template <typename Idx, typename DT>
__global__ __launch_bounds__(256) void gather_mm_kernel(const DT* __restrict__ a,
                                                        const DT* __restrict__ b,
                                                        DT* __restrict__ c,
                                                        const Idx* __restrict__ idx_a,
                                                        const Idx* __restrict__ idx_b,
                                                        const Idx* __restrict__ idx_c,
                                                        int64_t num_rows, int K, int N,
                                                        int scatter_add) {
  using A_ = typename Acc<DT>::type;
  const int64_t row = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= num_rows) return;
  const int64_t ra = idx_a ? static_cast<int64_t>(idx_a[row]) : row;
  const int64_t rb = idx_b ? static_cast<int64_t>(idx_b[row]) : row;
  const int64_t rc = idx_c ? static_cast<int64_t>(idx_c[row]) : row;
  const DT* ar = a + ra * K;
  const DT* br = b + rb * static_cast<int64_t>(K) * N;
  for (int n = lane; n < N; n += 64) {
    A_ acc = 0;
    for (int k = 0; k < K; ++k) acc += to_acc<DT>(ar[k]) * to_acc<DT>(br[static_cast<int64_t>(k) * N + n]);
    c[rc * N + n] = from_acc<DT>(acc);
  }
  (void)scatter_add;
}

// ---- host side ------------------------------------------------------------------------------
size_t align256(size_t x) { return (x + 255) / 256 * 256; }
constexpr size_t kWsProgWords = 16 * 1024;  // >= row chunks (<= CUs) x 8 waves x 4 words
constexpr int kWsQueueSplit = 8;            // dynamic chunks per group (chunk queue of the weights-stationary kernels)
constexpr size_t kWsQueueWords = 64;         // the queue's ticket, on a line of its own

struct MmScratch {
  size_t off_plan, off_t32, off_prog, off_bt, off_planes, off_acc, off_h2b, total;
};
// fp32 weight gradient on two fp16 terms: sample maxima, scales, 1 / scale, limits ([R][D1 + D2] words each), flags, list
size_t h2b_scratch_bytes(int64_t num_rel, int64_t d1, int64_t d2) {
  const size_t tab = (static_cast<size_t>(num_rel) * (d1 + d2) * 4 + 255) / 256 * 256;
  return 4 * tab + 256 + static_cast<size_t>(16) * 16384;
}

MmScratch mm_scratch(int64_t num_rel, int64_t K, int64_t N, size_t elem, bool need_bt,
                     bool need_acc, bool forward = false) {
  MmScratch s;
  size_t off = 0;
  s.off_plan = off;
  off = align256(off + sizeof(int64_t) * (2 * num_rel + 2) + /* staged seglen */ 8 * num_rel + /* NaN flag */ 8);
  s.off_t32 = off;
  if (forward) off = align256(off + sizeof(int64_t) * (num_rel + 1));
  s.off_prog = off;  // pacing words of the weights-stationary kernel: [row chunk][wave][column group]
  if (forward) off = align256(off + sizeof(int) * (kWsProgWords + kWsQueueWords) + 1024);  // (+ chunk queue, + the dummy line of the held stores)
  s.off_bt = off;
  if (need_bt) off = align256(off + static_cast<size_t>(num_rel) * K * N * elem);
  s.off_planes = off;  // fp32 forward: the weights as three bf16 planes (weights-stationary kernel)
  if (forward && elem == 4) {
    const size_t kp = (K + 7) / 8 * 8, rows = static_cast<size_t>(num_rel) * N;
    const size_t x3 = static_cast<size_t>(3) * rows * kp * 2;                                        // three bf16 planes
    const size_t h2 = align256(static_cast<size_t>(2) * rows * kp * 2) + align256(sizeof(float) * rows) + 256 + 16 * 4096 + 16 * 16384;  // two fp16 planes, 1 / column scale, flags, correction list (kH2FixCap entries), exact-row list (kH2RowCap)
    off = align256(off + std::max(x3, h2));
  }
  s.off_acc = off;
  if (need_acc) off = align256(off + static_cast<size_t>(num_rel) * K * N * sizeof(float));
  s.off_h2b = off;
  if (!forward && elem == 4) off = align256(off + h2b_scratch_bytes(num_rel, K, N));
  s.total = off;
  return s;
}

// Rows [sum(seglen), num_rows) belong to no relation: the reference returns zeros there (its
// output starts as th.zeros, python/dgl/backend/pytorch/sparse.py:975).  Grid-stride over the
// tail's 4-byte words (or bytes); exits after one load in the usual case sum(seglen) == num_rows.
__global__ __launch_bounds__(256) void segment_zero_tail_kernel(
    const int64_t* __restrict__ plan, int64_t num_rel, const int64_t* __restrict__ row_index,
    unsigned char* __restrict__ c, int64_t num_rows, int64_t row_bytes) {
  const int64_t total = plan[2 * num_rel + 1];
  if (total >= num_rows) return;
  const int64_t n = (num_rows - total) * row_bytes;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = total + i / row_bytes, b = i % row_bytes;
    c[(row_index ? row_index[r] : r) * row_bytes + b] = 0;
  }
}

int stage_plan(int idbits, const void* seglen, int seglen_on_host, int64_t num_rel,
               int rows_per_tile, int64_t num_rows, char* ws, const MmScratch& sc, hipStream_t s,
               bool forward = false) {
  int64_t* plan = reinterpret_cast<int64_t*>(ws + sc.off_plan);
  int64_t* t32 = forward ? reinterpret_cast<int64_t*>(ws + sc.off_t32) : nullptr;
  const void* dev_seglen = seglen;
  if (seglen_on_host) {
    // the reference hands seglen over in host memory (it loops over it on the CPU,
    // gather_mm.cu:221-246); stage it behind the plan
    void* stage = plan + 2 * num_rel + 2;
    DGLA_CHECK_HIP(hipMemcpyAsync(stage, seglen, static_cast<size_t>(idbits / 8) * num_rel,
                                  hipMemcpyHostToDevice, s));
    dev_seglen = stage;
  }
  if (idbits == 32)
    hipLaunchKernelGGL(segment_plan_kernel<int32_t>, dim3(1), dim3(64), 0, s,
                       static_cast<const int32_t*>(dev_seglen), num_rel, rows_per_tile, num_rows, plan, t32);
  else
    hipLaunchKernelGGL(segment_plan_kernel<int64_t>, dim3(1), dim3(64), 0, s,
                       static_cast<const int64_t*>(dev_seglen), num_rel, rows_per_tile, num_rows, plan, t32);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// CU count of the CURRENT device (the entry points' DeviceGuard has made the operands' device
// current), cached per device id.
int mm_num_cus() {
  constexpr int kMaxDev = 64;
  static int cus[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;  // MI355X
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// LDS-direct forward kernel: 16-bit storage, whole 16-byte pieces, K a multiple of the slab.
// (whole 16-byte pieces of K in both operands; 16-bit results also leave as 16-byte pieces)
bool glds_eligible(size_t elem, int64_t K, int64_t N, bool vec_ab, bool vec_c) {
  return (elem == 2 || elem == 4) && vec_ab && (vec_c || elem == 4) && (tuning_flags() & kTuneGlds);
}

// Weights-stationary persistent forward kernel: whole 16-byte pieces, K <= 256 (the LDS image), and enough rows
// that loading a relation's weights once per (workgroup, relation) is small beside streaming A:
// M rows >= 8 x (workgroups + relations) weight images' worth of rows (DGLA_MM_WS=0 / 1 forces it off / on).
bool ws_eligible(size_t elem, int64_t M, int64_t K, int64_t N, int64_t num_rel, bool vec_ab, bool vec_c) {
  const char* e = getenv("DGLA_MM_WS");  // read per call: tests switch it inside one process
  const int force = e && *e ? atoi(e) : -1;
  if (force == 0 || !(tuning_flags() & kTuneGlds)) return false;
  if (!(elem == 2 || (elem == 4 && !(tuning_flags() & kTuneMmF32))) || !vec_ab || !vec_c) return false;
  if (K > kWsMaxK || K < 8 || N < 8) return false;
  if (force == 1) return true;
  return M >= 8 * (mm_num_cus() + num_rel) * std::max<int64_t>(N, 32);
}

template <typename DT>
int launch_segment_mm_ws(const MmParams& p, char* ws, const MmScratch& sc, hipStream_t s) {
  WsParams wp;
  wp.m = p;
  wp.planes = nullptr;
  wp.tile32_off = reinterpret_cast<const int64_t*>(ws + sc.off_t32);
  wp.kp = 0;
  wp.prog = nullptr;
  wp.dummy = nullptr;
  wp.queue = nullptr;
  wp.qsplit = kWsQueueSplit;
  wp.rot = 0;
  const int N = p.N;
  int cus = mm_num_cus();
  if (const char* ew = getenv("DGLA_MM_WS_WGS")) {  // experiment: fewer workgroups than CUs
    if (atoi(ew) >= 8) cus = atoi(ew);
  }
  cus = std::min(cus, 512);
  const dim3 block(64 * kWsWaves);
  // experiment switch (round 4 A/B; docs/DESIGN_detail_r1_r5.md §3.7): bit 0 = fp32 without the fragment double buffer, bit 2 = siblings unpaced, bit 3 = paced but every sibling walks the lines in order, bit 4 = static row chunks (no queue)
  const char* ev = getenv("DGLA_MM_WS_VARIANT");
  const int variant = ev && *ev ? atoi(ev) : 0;
  // chunk queue (one column group only; variant bit 4 = static chunks): one ticket word, zeroed per launch
  auto arm_queue = [&](int ncg) -> int {
    if ((variant & 16) || ncg != 1) return 0;
    int* q = reinterpret_cast<int*>(ws + sc.off_prog) + kWsProgWords;
    DGLA_CHECK_HIP(hipMemsetAsync(q, 0, sizeof(int) * 64, s));
    wp.queue = q;
    return 0;
  };
  if constexpr (sizeof(DT) == 4) {
    if (!(p.tune & kTuneMmX3) && !(variant & 32)) {
      // fp32 as two scaled fp16 terms (round 5, default): weights prepared once per call, 128 columns per workgroup
      H2Params hp;
      hp.m = p;
      hp.kp = (p.K + 7) / 8 * 8;
      const int64_t rows = p.num_rel * static_cast<int64_t>(N);
      char* base = ws + sc.off_planes;
      hp.planes = reinterpret_cast<const _Float16*>(base);
      float* colinv = reinterpret_cast<float*>(base + align256(static_cast<size_t>(2) * rows * hp.kp * 2));
      uint32_t* flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(colinv) + align256(sizeof(float) * rows));
      hp.colinv = colinv;
      hp.flags = flags;
      hp.tile32_off = wp.tile32_off;
      DGLA_CHECK_HIP(hipMemsetAsync(flags, 0, 4 * sizeof(uint32_t), s));
      H2Fix* fix = reinterpret_cast<H2Fix*>(reinterpret_cast<char*>(flags) + 256);
      hp.fix = fix;
      H2Row* exact_rows = reinterpret_cast<H2Row*>(reinterpret_cast<char*>(fix) + sizeof(H2Fix) * kH2FixCap);
      hp.rows = exact_rows;
      hipLaunchKernelGGL(h2_prep_weights_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((rows + 3) / 4, 4096))), dim3(256), 0, s,
                         static_cast<const float*>(p.bt), const_cast<_Float16*>(hp.planes), colinv, flags, fix, rows, N, p.K, hp.kp);
      const int full = (p.K == 256) ? 8 : (p.K == 128 ? 4 : (p.K == 64 ? 2 : 0));   // K = a whole number of lines the kernel is specialised for (the usual hidden sizes)
      // (the generic form — K, lines and pieces at run time — keeps 64 columns per workgroup: with 128 its two accumulator
      // sets and the guards' live values spill to scratch; tools/isa_audit.py holds this file to zero scratch instructions)
      const int nj = N <= 32 ? 1 : ((N <= 64 || full == 0) ? 2 : 4);
      hp.ncg = (N + 32 * nj - 1) / (32 * nj);
      const int groups = std::max(1, cus / 8 / hp.ncg);
      const dim3 grid(static_cast<unsigned>(groups * hp.ncg * 8));
      const dim3 block(64 * kH2Waves);
#define DGLA_H2C(NJV, IDX, CW)                                                                        \
  do {                                                                                                \
    if (full == 8)                                                                                    \
      hipLaunchKernelGGL((segment_mm_h2_kernel<NJV, IDX, 8, CW>), grid, block, 0, s, hp);              \
    else if (full == 4)                                                                               \
      hipLaunchKernelGGL((segment_mm_h2_kernel<NJV, IDX, 4, CW>), grid, block, 0, s, hp);              \
    else if (full == 2)                                                                               \
      hipLaunchKernelGGL((segment_mm_h2_kernel<NJV, IDX, 2, CW>), grid, block, 0, s, hp);              \
    else if constexpr (NJV <= 2 && !CW)                                                               \
      hipLaunchKernelGGL((segment_mm_h2_kernel<NJV, IDX, 0, false>), grid, block, 0, s, hp);           \
  } while (0)
#define DGLA_H2I(NJV, IDX)                                                                            \
  do {                                                                                                \
    if (full != 0 && N % (32 * NJV) == 0)                                                             \
      DGLA_H2C(NJV, IDX, true);                                                                       \
    else                                                                                              \
      DGLA_H2C(NJV, IDX, false);                                                                      \
  } while (0)
#define DGLA_H2(NJV)                                                                                  \
  do {                                                                                                \
    if (p.row_index)                                                                                  \
      DGLA_H2I(NJV, true);                                                                            \
    else                                                                                              \
      DGLA_H2I(NJV, false);                                                                           \
  } while (0)
      switch (nj) {
        case 1: DGLA_H2(1); break;
        case 2: DGLA_H2(2); break;
        default: DGLA_H2(4); break;
      }
#undef DGLA_H2
#undef DGLA_H2I
#undef DGLA_H2C
      hipLaunchKernelGGL(h2_fix_weights_kernel, dim3(static_cast<unsigned>(2 * cus)), dim3(256), 0, s, p, flags, fix);  // idle unless a weight element was too small for its column
      hipLaunchKernelGGL(h2_exact_rows_kernel, dim3(static_cast<unsigned>(4 * cus)), dim3(256), 0, s, p, flags, exact_rows);  // idle unless a row was listed
      DGLA_CHECK_HIP(hipGetLastError());
      return 0;
    }
    // fp32: three bf16 planes of the weights, once per call; 64 columns per workgroup
    wp.kp = (p.K + 7) / 8 * 8;
    bf16_t* planes = reinterpret_cast<bf16_t*>(ws + sc.off_planes);
    const int64_t rows = p.num_rel * static_cast<int64_t>(N);
    hipLaunchKernelGGL(split_weights_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((rows * wp.kp + 255) / 256, 4096))),
                       dim3(256), 0, s, static_cast<const float*>(p.bt), planes, rows, p.K, wp.kp);
    wp.planes = planes;
    // 64 columns per workgroup (NJ = 2, 101 KB of LDS).  96 (NJ = 3, 152 KB, three column groups at N = 256 instead
    // of four) was built and measured: 9.41 ms against 7.85 ms at 10 M x 256 x 256 — the time follows the MFMA work
    // per workgroup (x 1.2 with 240 instead of 256 workgroups busy), not the A bytes read.
    wp.ncg = (N + 63) / 64;
    const int groups = std::max(1, cus / 8 / wp.ncg);  // row chunks per XCD
    const dim3 grid(static_cast<unsigned>(groups * wp.ncg * 8));
    // (the PIPE kernels read their pacing words whether they pace or not: always there, always zeroed)
    wp.prog = reinterpret_cast<int*>(ws + sc.off_prog);
    wp.dummy = ws + sc.off_prog + sizeof(int) * (kWsProgWords + kWsQueueWords);
    wp.rot = (variant & 4) ? -1 : (variant & 8) ? 0 : 1;
    static_assert(kWsProgWords >= 64 * 8 * kWsWaves * 4, "pacing words for up to 512 workgroups");
    DGLA_CHECK_HIP(hipMemsetAsync(wp.prog, 0, sizeof(int) * groups * 8 * kWsWaves * 4, s));
    if (int rc = arm_queue(wp.ncg)) return rc;
    // PIPE needs every k-step whole and rounds of exactly 4 lines: K a multiple of 128 floats' worth of lines
    const bool pipe = p.K % 32 == 0 && (p.K / 32) % 4 == 0 && !(variant & 1);
#define DGLA_WS3(NJV)                                                                                       \
  do {                                                                                                      \
    if (p.row_index) {                                                                                      \
      if (pipe)                                                                                             \
        hipLaunchKernelGGL((segment_mm_ws_kernel<DT, NJV, true, true, 4, true>), grid, block, 0, s, wp);     \
      else                                                                                                  \
        hipLaunchKernelGGL((segment_mm_ws_kernel<DT, NJV, true, true, 4, false>), grid, block, 0, s, wp);    \
    } else if (pipe) {                                                                                      \
      hipLaunchKernelGGL((segment_mm_ws_kernel<DT, NJV, true, false, 4, true>), grid, block, 0, s, wp);      \
    } else {                                                                                                \
      hipLaunchKernelGGL((segment_mm_ws_kernel<DT, NJV, true, false, 4, false>), grid, block, 0, s, wp);     \
    }                                                                                                       \
  } while (0)
    DGLA_WS3(2);
#undef DGLA_WS3
    hipLaunchKernelGGL(x3_repair_fwd_kernel, dim3(1024), dim3(256), 0, s, p);  // no-op unless a NaN came out
  } else {
    const int nj = N <= 32 ? 1 : (N <= 64 ? 2 : (N <= 128 ? 4 : 8));
    wp.ncg = (N + 32 * nj - 1) / (32 * nj);
    const int groups = std::max(1, cus / 8 / wp.ncg);
    const dim3 grid(static_cast<unsigned>(groups * wp.ncg * 8));
    if (int rc = arm_queue(wp.ncg)) return rc;
#define DGLA_WS(NJV)                                                                                   \
  do {                                                                                                 \
    if (p.row_index)                                                                                   \
      hipLaunchKernelGGL((segment_mm_ws_kernel<DT, NJV, false, true, 2, false>), grid, block, 0, s, wp);  \
    else                                                                                               \
      hipLaunchKernelGGL((segment_mm_ws_kernel<DT, NJV, false, false, 2, false>), grid, block, 0, s, wp); \
  } while (0)
    switch (nj) {
      case 1: DGLA_WS(1); break;
      case 2: DGLA_WS(2); break;
      case 4: DGLA_WS(4); break;
      default: DGLA_WS(8); break;
    }
#undef DGLA_WS
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename DT>
int run_segment_mm(const void* a, const void* b, void* c, int64_t M, int64_t K, int64_t N,
                   int64_t num_rel, bool b_trans, const int64_t* row_index, char* ws,
                   const MmScratch& sc, hipStream_t s) {
  // Bt = [R, N, K]: the weights with the contraction axis contiguous
  const void* bt = b;
  if (!b_trans) {
    // b: [R, K, N]
    DT* dst = reinterpret_cast<DT*>(ws + sc.off_bt);
    hipLaunchKernelGGL((transpose_weights_kernel<DT>),
                       dim3((N + 31) / 32, (K + 31) / 32, static_cast<unsigned>(num_rel)), dim3(256),
                       0, s, static_cast<const DT*>(b), dst, static_cast<int>(K), static_cast<int>(N));
    bt = dst;
  }
  MmParams p;
  p.a = a;
  p.bt = bt;
  p.c = c;
  p.plan = reinterpret_cast<const int64_t*>(ws + sc.off_plan);
  p.num_rel = num_rel;
  p.K = static_cast<int>(K);
  p.N = static_cast<int>(N);
  constexpr int E = 16 / sizeof(DT);
  p.vec_a = (K % E == 0 && aligned16(a)) ? 1 : 0;
  p.vec_b = (K % E == 0 && aligned16(bt)) ? 1 : 0;
  p.vec_c = (N % E == 0 && aligned16(c)) ? 1 : 0;
  p.row_index = row_index;
  p.n_tiles = 1;
  p.tune = tuning_flags();
  p.nan_flag = reinterpret_cast<uint32_t*>(const_cast<int64_t*>(p.plan) + 3 * num_rel + 2);
  if constexpr (sizeof(DT) == 8) {
    const int64_t total = M * N;
    hipLaunchKernelGGL((segment_mm_plain_kernel<DT>), dim3(static_cast<unsigned>((total + 255) / 256)),
                       dim3(256), 0, s, p, M);
  } else {
    const int64_t max_tiles = ((M + BM - 1) / BM + num_rel + 7) / 8 * 8;  // whole groups of 8 XCDs
    const bool wide = N > 128;  // (fp32 too: 128-wide tiles measured 14.6 vs 11.6 ms at 10 M x 256 x 256)
    p.n_tiles = wide ? (N + 255) / 256 : 1;
    const int64_t blocks = max_tiles * p.n_tiles;
    if (blocks > 0x7fffffffLL) return mfail("segment_mm: too many tiles");
    const dim3 grid(static_cast<unsigned>(blocks)), block(256);
    const bool vec = p.vec_a && p.vec_b && p.vec_c;
    {
      if (ws_eligible(sizeof(DT), M, K, N, num_rel, p.vec_a && p.vec_b, p.vec_c != 0))
        return launch_segment_mm_ws<DT>(p, ws, sc, s);
      if (glds_eligible(sizeof(DT), K, N, p.vec_a && p.vec_b, p.vec_c != 0)) {
        if constexpr (sizeof(DT) == 4) {
          if (!(p.tune & kTuneMmF32)) {  // fp32 operands as three bf16 terms (default)
            if (wide)
              hipLaunchKernelGGL((segment_mm_glds_kernel<DT, 256, 128, 6, 2, true>), grid, block, 0, s, p);
            else
              hipLaunchKernelGGL((segment_mm_glds_kernel<DT, 128, 128, 6, 2, true>), grid, block, 0, s, p);
            hipLaunchKernelGGL(x3_repair_fwd_kernel, dim3(1024), dim3(256), 0, s, p);  // no-op unless a NaN came out
            DGLA_CHECK_HIP(hipGetLastError());
            return 0;
          }
        }
        if (wide)
          hipLaunchKernelGGL((segment_mm_glds_kernel<DT, 256, 128, 6, 2>), grid, block, 0, s, p);
        else
          hipLaunchKernelGGL((segment_mm_glds_kernel<DT, 128, 128, 6, 2>), grid, block, 0, s, p);
        DGLA_CHECK_HIP(hipGetLastError());
        return 0;
      }
    }
    if (wide && vec)
      hipLaunchKernelGGL((segment_mm_kernel<DT, 256, true>), grid, block, 0, s, p);
    else if (wide)
      hipLaunchKernelGGL((segment_mm_kernel<DT, 256, false>), grid, block, 0, s, p);
    else if (vec)
      hipLaunchKernelGGL((segment_mm_kernel<DT, 128, true>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((segment_mm_kernel<DT, 128, false>), grid, block, 0, s, p);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- weight gradient, LDS-direct variant (16-bit storage, contiguous rows) ------------------
// Same split-K slabs, 128 x 128 output tile and fp32 atomics as segment_mm_bwd_b_kernel, but the
// operands are NOT transposed on their way into LDS: global_load_lds_dwordx4 copies 32 rows of
// A and dC (256 bytes = 128 features each) per slab into a ring of NS 16 KB slots, and the MFMA
// fragments — lane <-> feature, registers <-> consecutive contraction rows — are read with
// ds_read_b64_tr_b16, gfx950's transposing LDS load.  Measured semantics (one 16-lane group):
// lane l receives halfword (l & 3) at the addresses supplied by lanes (l >> 2) + 4 j, j = 0..3.
// So lane s supplies row (s >> 2), feature quad (s & 3) of a 4-row x 16-feature block and
// every lane gets ITS feature at 4 consecutive rows.  Contraction index of register j of the
// j2-th read in MFMA k-half kh: row 16 kk + 8 j2 + 4 kh + j — the same on both operands.
//  * LDS rows are unpadded (lane-linear DMA); the 32-byte unit holding features 16 u .. 16 u + 15
//    of row r sits at physical unit u ^ (r & 7), applied to the per-lane SOURCE address and to
//    the fragment reads, so the 8 rows one read touches fall into distinct bank groups.
//  * Rows past the end of the slab read a zero page; feature chunks past the width are clamped
//    (their outputs are never stored).
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int kBwdGldsRows = 32;  // contraction rows per ring slot

template <typename DT>
__device__ __forceinline__ void mfma16(const s16x4 alo, const s16x4 ahi, const s16x4 blo, const s16x4 bhi,
                                       f32x16& acc) {
  const s16x8 a = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
  const s16x8 b = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
  if constexpr (std::is_same<DT, bf16_t>::value)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b16x8, a), __builtin_bit_cast(b16x8, b), acc, 0, 0, 0);
  else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
}

template <typename DT, int NS>
__global__ __launch_bounds__(256, NS >= 4 ? 2 : (NS == 3 ? 3 : 4)) void segment_mm_bwd_b_glds_kernel(const MmBwdParams p) {
  static_assert(sizeof(DT) == 2, "16-bit storage");
  constexpr int kPart = kBwdGldsRows * 256;  // one operand of a slot: 32 rows x 128 features
  constexpr int kSlot = 2 * kPart;
  constexpr int kLoads = 4;                  // DMA instructions per wave and slot (2 per operand)
  __shared__ __attribute__((aligned(1024))) char smem[NS * kSlot];

  const int64_t L = blockIdx.x;
  const int64_t jd = L >> 3;
  const int tiles = p.tiles_i * p.tiles_j;
  const int64_t slab = (jd / tiles) * 8 + (L & 7);
  const int tile = static_cast<int>(jd % tiles);
  const int64_t* slab_off = p.plan;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  if (slab >= slab_off[p.num_rel]) return;
  const int64_t rel = find_segment(slab_off, p.num_rel, slab);
  const int64_t m0 = row_off[rel] + (slab - slab_off[rel]) * p.slab_rows;
  int64_t m1 = m0 + p.slab_rows;
  if (m1 > row_off[rel + 1]) m1 = row_off[rel + 1];
  const int i0 = (tile / p.tiles_j) * BM, j0 = (tile % p.tiles_j) * BN;
  const int D1 = p.D1, D2 = p.D2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA sources.  Instruction n (0, 1) of a wave covers slot rows 8 wave + 4 n .. + 3 of one
  // operand: lane -> row + (lane >> 4), physical 16-byte chunk lane & 15.
  const int l16 = lane & 15;
  int drow[2];
  int64_t offA[2], offC[2];  // byte offset of the lane's piece from the operand's row m0
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r = 8 * wave + 4 * n + (lane >> 4);
    drow[n] = r;
    const int unit = (l16 >> 1) ^ (r & 7);
    const int chunk = (unit << 1) | (l16 & 1);
    int fa = i0 + chunk * 8, fc = j0 + chunk * 8;
    if (fa >= D1) fa = D1 - 8;
    if (fc >= D2) fc = D2 - 8;
    offA[n] = (static_cast<int64_t>(r) * D1 + fa) * 2;
    offC[n] = (static_cast<int64_t>(r) * D2 + fc) * 2;
  }
  const char* __restrict__ baseA = static_cast<const char*>(p.a) + m0 * D1 * 2;
  const char* __restrict__ baseC = static_cast<const char*>(p.dc) + m0 * D2 * 2;
  const int64_t rows = m1 - m0;
  const int nsl = static_cast<int>((rows + kBwdGldsRows - 1) / kBwdGldsRows);
  auto issue = [&](int t) {
    if (t >= nsl) return;
    char* dst = smem + (t % NS) * kSlot + wave * 2048;
    const int64_t mrow = static_cast<int64_t>(t) * kBwdGldsRows;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const bool in = mrow + drow[n] < rows;
      const char* sa = in ? baseA + mrow * D1 * 2 + offA[n] : g_mm_zero_page;
      const char* sc = in ? baseC + mrow * D2 * 2 + offC[n] : g_mm_zero_page;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)sa, (lds_ptr_t)(dst + n * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)sc, (lds_ptr_t)(dst + kPart + n * 1024), 16, 0, 0);
    }
  };
  auto wait_for_slot = [&](int t) {  // slots issued after slot t so far: t + 1 .. min(t + NS - 2, nsl - 1)
    int later = nsl - 1 - t;
    if (later > NS - 2) later = NS - 2;
    switch (later) {
      case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
      case 1: __builtin_amdgcn_s_waitcnt(0x0F70 | (1 * kLoads)); break;
      case 2: __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * kLoads)); break;
      case 3: __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * kLoads)); break;
      default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
  };
  static_assert(NS >= 2 && NS <= 5, "vmcnt switch covers up to 3 slots in flight behind the needed one");

  // fragment addresses: lane = (kh, ih, s): feature 16 ih + (its own), rows 4 kh + (s >> 2) + ...
  const int s4 = lane & 15, ih = (lane >> 4) & 1, kh = lane >> 5;
  const int r7 = 4 * kh + (s4 >> 2);
  int fa_off[2], fb_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ua = ((wm * 2 + i) * 2 + ih) ^ r7;
    const int ub = ((wn * 2 + i) * 2 + ih) ^ r7;
    fa_off[i] = r7 * 256 + ua * 32 + 8 * (s4 & 3);
    fb_off[i] = kPart + r7 * 256 + ub * 32 + 8 * (s4 & 3);
  }
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr_t)smem));

#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue(i);
  for (int t = 0; t < nsl; ++t) {
    wait_for_slot(t);
    __builtin_amdgcn_s_barrier();
    issue(t + NS - 1);
    // The 8 transposing reads of one k-step and their lgkmcnt wait are ONE asm statement: with the
    // builtin, hipcc (ROCm 7.2) cannot tell the reads from the in-flight LDS-DMA writes and puts
    // s_waitcnt vmcnt(0) in front of them, which would drain the ring every iteration.
    const uint32_t slot = lds_base + (t % NS) * kSlot;
    const uint32_t pa0 = slot + fa_off[0], pa1 = slot + fa_off[1], pb0 = slot + fb_off[0], pb1 = slot + fb_off[1];
    s16x4 a00, a01, a10, a11, b00, b01, b10, b11;
#define DGLA_TR_READS(O0, O1)                                                                       \
  asm volatile("ds_read_b64_tr_b16 %0, %8 offset:" #O0 "\n\t"                                        \
               "ds_read_b64_tr_b16 %1, %8 offset:" #O1 "\n\t"                                        \
               "ds_read_b64_tr_b16 %2, %9 offset:" #O0 "\n\t"                                        \
               "ds_read_b64_tr_b16 %3, %9 offset:" #O1 "\n\t"                                        \
               "ds_read_b64_tr_b16 %4, %10 offset:" #O0 "\n\t"                                       \
               "ds_read_b64_tr_b16 %5, %10 offset:" #O1 "\n\t"                                       \
               "ds_read_b64_tr_b16 %6, %11 offset:" #O0 "\n\t"                                       \
               "ds_read_b64_tr_b16 %7, %11 offset:" #O1 "\n\t"                                       \
               "s_waitcnt lgkmcnt(0)"                                                                \
               : "=&v"(a00), "=&v"(a01), "=&v"(a10), "=&v"(a11), "=&v"(b00), "=&v"(b01), "=&v"(b10),  \
                 "=&v"(b11)                                                                          \
               : "v"(pa0), "v"(pa1), "v"(pb0), "v"(pb1)                                              \
               : "memory")
#define DGLA_TR_MFMAS()                          \
  mfma16<DT>(a00, a01, b00, b01, acc[0][0]);     \
  mfma16<DT>(a00, a01, b10, b11, acc[0][1]);     \
  mfma16<DT>(a10, a11, b00, b01, acc[1][0]);     \
  mfma16<DT>(a10, a11, b10, b11, acc[1][1])
    DGLA_TR_READS(0, 2048);     // contraction rows 0-15 of the slot (j2 = 0, 1: rows + 0, + 8)
    DGLA_TR_MFMAS();
    DGLA_TR_READS(4096, 6144);  // rows 16-31
    DGLA_TR_MFMAS();
#undef DGLA_TR_READS
#undef DGLA_TR_MFMAS
  }

  float* out = p.acc + rel * static_cast<int64_t>(D1) * D2;
  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wn * 64 + j * 32 + col_l;
      if (col >= D2) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (row < D1) atomicAdd(out + static_cast<int64_t>(row) * D2 + col, acc[i][j][r]);
      }
    }
}

// fp32 weight gradient, LDS-direct: no transposition is needed at all — the 32x32x2 MFMA takes
// ONE contraction row per k-half, so lane (feature f, k-half kh) reads LDS[row 2 s + kh][f] with a
// plain ds_read_b32 (32 consecutive features = 32 consecutive banks; unpadded, unswizzled rows).
// 16 rows x 128 features of A and of dC per slot, NS slots, counted vmcnt as above.
constexpr int kBwdGldsRowsF32 = 16;

template <int NS, bool X3 = false>  // X3: operands as three bf16 terms (see split3)
__global__ __launch_bounds__(256, 2) void segment_mm_bwd_b_glds_f32_kernel(const MmBwdParams p) {
  constexpr int kPart = kBwdGldsRowsF32 * 512;  // one operand of a slot: 16 rows x 128 features x 4 B
  constexpr int kSlot = 2 * kPart;
  constexpr int kLoads = 4;  // DMA instructions per wave and slot (2 per operand, 2 rows each)
  __shared__ __attribute__((aligned(1024))) char smem[NS * kSlot];

  if constexpr (X3) {  // (uniform) the fallback launch behind segment_mm_bwd_b_h2_kernel
    if (p.only_if && *p.only_if == 0) return;
  }
  const int64_t L = blockIdx.x;
  const int64_t jd = L >> 3;
  const int tiles = p.tiles_i * p.tiles_j;
  const int64_t slab = (jd / tiles) * 8 + (L & 7);
  const int tile = static_cast<int>(jd % tiles);
  const int64_t* slab_off = p.plan;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  if (slab >= slab_off[p.num_rel]) return;
  const int64_t rel = find_segment(slab_off, p.num_rel, slab);
  const int64_t m0 = row_off[rel] + (slab - slab_off[rel]) * p.slab_rows;
  int64_t m1 = m0 + p.slab_rows;
  if (m1 > row_off[rel + 1]) m1 = row_off[rel + 1];
  const int i0 = (tile / p.tiles_j) * BM, j0 = (tile % p.tiles_j) * BN;
  const int D1 = p.D1, D2 = p.D2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // DMA: instruction n (0, 1) of a wave covers slot rows 4 wave + 2 n, + 1 of one operand;
  // lane -> row + (lane >> 5), features 4 (lane & 31) .. + 3
  int drow[2];
  int64_t offA[2], offC[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int r = 4 * wave + 2 * n + (lane >> 5);
    drow[n] = r;
    int fa = i0 + (lane & 31) * 4, fc = j0 + (lane & 31) * 4;
    if (fa >= D1) fa = D1 - 4;
    if (fc >= D2) fc = D2 - 4;
    offA[n] = (static_cast<int64_t>(r) * D1 + fa) * 4;
    offC[n] = (static_cast<int64_t>(r) * D2 + fc) * 4;
  }
  const char* __restrict__ baseA = static_cast<const char*>(p.a) + m0 * D1 * 4;
  const char* __restrict__ baseC = static_cast<const char*>(p.dc) + m0 * D2 * 4;
  const int64_t rows = m1 - m0;
  const int nsl = static_cast<int>((rows + kBwdGldsRowsF32 - 1) / kBwdGldsRowsF32);
  auto issue = [&](int t) {
    if (t >= nsl) return;
    char* dst = smem + (t % NS) * kSlot + wave * 2048;
    const int64_t mrow = static_cast<int64_t>(t) * kBwdGldsRowsF32;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const bool in = mrow + drow[n] < rows;
      const char* sa = in ? baseA + mrow * D1 * 4 + offA[n] : g_mm_zero_page;
      const char* sc = in ? baseC + mrow * D2 * 4 + offC[n] : g_mm_zero_page;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)sa, (lds_ptr_t)(dst + n * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)sc, (lds_ptr_t)(dst + kPart + n * 1024), 16, 0, 0);
    }
  };
  auto wait_for_slot = [&](int t) {
    int later = nsl - 1 - t;
    if (later > NS - 2) later = NS - 2;
    switch (later) {
      case 0: __builtin_amdgcn_s_waitcnt(0x0F70); break;
      case 1: __builtin_amdgcn_s_waitcnt(0x0F70 | (1 * kLoads)); break;
      case 2: __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * kLoads)); break;
      case 3: __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * kLoads)); break;
      default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
    }
  };
  static_assert(NS >= 2 && NS <= 5, "vmcnt switch covers up to 3 slots in flight behind the needed one");

  const int f = lane & 31, kh = lane >> 5;
  const int a_off = kh * 512 + (wm * 64 + f) * 4;
  const int b_off = kPart + kh * 512 + (wn * 64 + f) * 4;

#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue(i);
  for (int t = 0; t < nsl; ++t) {
    wait_for_slot(t);
    __builtin_amdgcn_s_barrier();
    issue(t + NS - 1);
    const char* slot = smem + (t % NS) * kSlot;
    if constexpr (X3) {
      // the slot's 16 contraction rows in ONE 32x32x16 step: lane (feature f, k-half kh) reads its
      // feature at rows 8 kh .. 8 kh + 7 (ds_read_b32, 32 consecutive banks per row as before),
      // splits the eight values into (h, m, l) and runs the six bf16 products per tile
      b16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* pa8 = slot + (wm * 64 + i * 32 + f) * 4 + kh * 8 * 512;
        const char* pb8 = slot + kPart + (wn * 64 + i * 32 + f) * 4 + kh * 8 * 512;
        f32x4 a0, a1, b0, b1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0[r] = *reinterpret_cast<const float*>(pa8 + r * 512);
          a1[r] = *reinterpret_cast<const float*>(pa8 + (4 + r) * 512);
          b0[r] = *reinterpret_cast<const float*>(pb8 + r * 512);
          b1[r] = *reinterpret_cast<const float*>(pb8 + (4 + r) * 512);
        }
        split3(a0, a1, ah[i], am[i], al[i]);
        split3(b0, b1, bh[i], bm[i], bl[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
      continue;
    }
#pragma unroll
    for (int s2 = 0; s2 < kBwdGldsRowsF32 / 2; ++s2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float*>(slot + a_off + s2 * 1024 + i * 128);
        b[i] = *reinterpret_cast<const float*>(slot + b_off + s2 * 1024 + i * 128);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  if constexpr (X3) {  // non-finite operands (see split3): tell x3_repair_bwd_kernel to look
    if (__builtin_expect(any_nan(acc), 0) && lane == 0) atomicOr(p.nan_flag, 1u);
  }
  float* out = p.acc + rel * static_cast<int64_t>(D1) * D2;
  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wn * 64 + j * 32 + col_l;
      if (col >= D2) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (row < D1) atomicAdd(out + static_cast<int64_t>(row) * D2 + col, acc[i][j][r]);
      }
    }
}

// ---- fp32 weight gradient as TWO scaled fp16 terms (round 6; VERDICT r5 Next #4) -------------------------------------
// The X3 kernel above spends six bf16 MFMAs per tile and 16 contraction rows, and every operand half is split by the two
// waves that multiply it: 9.1 ms at 10 M x 256 x 256.  Two fp16 terms (see the forward's H2 kernels: x s = h + 2^-11 l',
// products hh, hl', l'h; 22 bits of x) need three.  The contraction runs over the ROWS here, so the power-of-two scale
// that brings a value into fp16's range must be constant along a COLUMN of A / of dC inside a relation — and the column's
// largest magnitude is only known after the whole segment has been read.  Reading it twice would cost what the split
// saves, so the scale is an ESTIMATE that the main kernel VERIFIES while it converts:
//   * h2_bwd_sample_kernel: largest |x| per (relation, column) over <= kH2bSample rows spread evenly over the segment
//     (33 MB at 8 x 512 columns); h2_bwd_scales_kernel puts it into [2^11, 2^12): five binades of head room for the rows
//     the sample did not see.  A column whose sample is all zero gets the scale 2^115, under which ANY non-zero fp32
//     value leaves a non-zero or infinite pattern in the planes.
//   * an fp16 overflow (a value above 32 x its column's sample maximum, Inf, NaN) turns every output of its row / column
//     into Inf or NaN: the epilogue looks at its accumulators and raises flags[0]; so does anything non-zero in a
//     zero-sample column (found by the test of the next item), a sample maximum that is Inf / NaN / outside 2^+-60, and
//     an overflow of the list below.  A raised flag makes the launches behind the kernel zero the result and run the X3 kernel on
//     the whole call (both exit on their first load otherwise): correct for any input, slow only for columns whose
//     maximum lies more than 16 x above what 2 048 evenly spread rows show.
//   * an element whose scaled magnitude falls below 2^-14 (25 binades under the sample maximum: 1 in 10^7 for continuous
//     data) has a subnormal high term and would keep fewer than 21 bits.  The converting thread finds such elements with
//     one packed 16-bit minimum per pair over |h| | |l|, removes them from both planes and writes (row, column, operand)
//     into a short list; h2_bwd_fix_kernel, launched behind the main kernel and idle unless the list is non-empty, adds
//     their rank-one contributions in plain fp32: A-side element (m, i): dB[i][:] += A[m][i] dC[m][:]; dC-side element
//     (m, j): dB[:][j] += A0[m][:] dC[m][j] with the listed elements of A[m] left out (the same test, recomputed).
//     What the test cannot see: elements whose BOTH terms round to zero (|x s| < 2^-36: 47 binades under a value present
//     in the column's sample) are dropped.
// Data path: the X3 kernel's ring of 16-row fp32 slots filled by global_load_lds; per slot the 256 threads convert it ONCE
// (thread = (operand, k-half, feature): 8 rows of one feature -> one 16-byte piece per plane; fragment reads and piece
// writes are conflict-free), barrier, and every wave reads its eight fragments with ds_read_b128 and issues twelve
// v_mfma_f32_32x32x16_f16 into two accumulator sets (hh; hl' + l'h).
// One barrier per slot: while a wave multiplies slot t it converts slot t + 1 IN PLACE (a wave's 2 KB of fp32 become its 2 KB
// of fp16 pieces: no other wave's data is touched), the conversion's vector instructions placed behind the MFMAs in four
// groups.  Epilogue: (acc0 + 2^-11 acc1) / (scale_A[i] scale_C[j]) — exact power-of-two factors — added with the fp32
// atomics of the split-K scheme.  80 KB of LDS = a ring of five slots (one multiplied, one converted, three in flight), two
// workgroups of eight waves per CU.  What bounds it (profiles/r6/segment_mm_dB_two_term.jsonl): on this chip the vector and
// the matrix instructions of a SIMD do not overlap — compute alone takes VALU (139 instructions x 4 cycles) + MFMA (12 x 32
// cycles) per wave and slot, 4.7 ms — and the memory side (3.9 ms alone) overlaps with that only in part.
constexpr int64_t kH2bMinRows = 16384;  // smaller calls take the three-term kernel
constexpr int kH2bSample = 2048;       // rows per relation the scale estimate looks at
constexpr int kH2bChunks = 16;         // workgroups per relation of the sampling kernel
constexpr int kH2bFixCap = 16384;      // list of elements too small for their column's scale
constexpr uint32_t kH2bTiny = 0x0400;  // fp16 bits of 2^-14: a non-zero (high | low) pattern below it sends the element to the list

struct H2bFix {
  int64_t row;   // physical row of A and dC
  int32_t col;   // column of A (op 0) or dC (op 1)
  int32_t op_rel;  // (relation << 1) | op
};

struct H2bParams {
  MmBwdParams m;
  const float* scale;     // [R][D1 + D2] column scales: A's columns, then dC's
  const float* inv;       // [R][D1 + D2] 1 / scale
  const uint32_t* limit;  // [R][D1 + D2] smallest fp16 magnitude (bits) that must not appear: 0x7c00, or 1 (zero-sample column)
  uint32_t* flags;        // [0]: the estimate failed -> X3 on the whole call; [1]: entries in `fix`
  H2bFix* fix;
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void h2_bwd_sample_kernel(const MmBwdParams p, uint32_t* __restrict__ umax) {
  const int64_t* row_off = p.plan + p.num_rel + 1;
  const int rel = blockIdx.x / kH2bChunks, chunk = blockIdx.x % kH2bChunks;
  const int64_t r0 = row_off[rel], len = row_off[rel + 1] - r0;
  if (len <= 0) return;
  const int64_t ns = len < kH2bSample ? len : kH2bSample;                 // sampled rows: floor(k len / ns), k < ns
  const int64_t k0 = ns * chunk / kH2bChunks, k1 = ns * (chunk + 1) / kH2bChunks;
  const int cols = p.D1 + p.D2;
  const uint32_t* A = static_cast<const uint32_t*>(p.a);
  const uint32_t* C = static_cast<const uint32_t*>(p.dc);
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const uint32_t* base = c < p.D1 ? A + c : C + (c - p.D1);
    const int64_t pitch = c < p.D1 ? p.D1 : p.D2;
    uint32_t mx = 0;
    for (int64_t k = k0; k < k1; ++k) {
      const int64_t m = r0 + (len == ns ? k : k * len / ns);
      const uint32_t a = base[m * pitch] & 0x7fffffffu;
      mx = a > mx ? a : mx;
    }
    if (mx) atomicMax(umax + static_cast<int64_t>(rel) * cols + c, mx);
  }
}

__global__ __launch_bounds__(256) void h2_bwd_scales_kernel(const uint32_t* __restrict__ umax, int64_t n, float* __restrict__ scale,
                                                            float* __restrict__ inv, uint32_t* __restrict__ limit,
                                                            uint32_t* __restrict__ flags) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t u = umax[i];
  const int e = static_cast<int>(u >> 23);
  uint32_t sb = 127u << 23, ib = 127u << 23, lim = 0x7c00u;
  if (u == 0u) {
    sb = 242u << 23;   // 2^115 (2^126 with the low term's 2^11): every non-zero fp32 value, denormals included, leaves a
                       // non-zero (or infinite) pattern in the low plane; a zero stays zero (2^126 itself would make the
                       // low term's factor 2^137 = inf and 0 x inf = NaN)
    lim = 1u;
  } else if (e < kH2MinExp || e > kH2MaxExp) {
    atomicOr(flags, 1u);   // Inf / NaN / a magnitude whose unscaling could leave fp32's range: not this kernel's case
  } else {
    sb = static_cast<uint32_t>(127 + 11 + 127 - e) << 23;   // 2^(11 - (e - 127))
    ib = static_cast<uint32_t>(e - 11) << 23;               // 2^((e - 127) - 11)
  }
  scale[i] = __builtin_bit_cast(float, sb);
  inv[i] = __builtin_bit_cast(float, ib);
  limit[i] = lim;
}

// is the element one the main kernel moved to the list?  (the same arithmetic as its conversion)
__device__ __forceinline__ bool h2b_listed(float x, float s) {
  const float v = x * s;
  const _Float16 h = static_cast<_Float16>(v);
  const _Float16 l = static_cast<_Float16>((v - static_cast<float>(h)) * 2048.f);
  const uint32_t key = (__builtin_bit_cast(unsigned short, h) | __builtin_bit_cast(unsigned short, l)) & 0x7fffu;
  return key != 0u && key < kH2bTiny;
}

constexpr int kH2bSlots = 5;
// EIGHT waves per workgroup: a wave multiplies a 32 x 64 piece of the 128 x 128 tile (64 accumulator registers in two sets)
// and converts 8 elements per slot (waves 0-3: A, waves 4-7: dC), 123 VGPRs, so two workgroups = four waves per SIMD fit.
// (A first form with four waves of 64 x 64 — 215 VGPRs, two waves per SIMD — ran 6.90 ms against 6.63.)
__global__ __launch_bounds__(512, 2) void segment_mm_bwd_b_h2_kernel(const H2bParams hp) {
  const MmBwdParams& p = hp.m;
  constexpr int NS = kH2bSlots;
  constexpr int kPart = kBwdGldsRowsF32 * 512;  // one operand of a slot: 16 rows x 128 features x 4 B
  constexpr int kSlot = 2 * kPart;
  constexpr int kLoads = 2;                     // DMA instructions per wave and slot
  __shared__ __attribute__((aligned(1024))) char smem[NS * kSlot];

  const int64_t L = blockIdx.x;
  const int64_t jd = L >> 3;
  const int tiles = p.tiles_i * p.tiles_j;
  const int64_t slab = (jd / tiles) * 8 + (L & 7);
  const int tile = static_cast<int>(jd % tiles);
  const int64_t* slab_off = p.plan;
  const int64_t* row_off = p.plan + p.num_rel + 1;
  if (slab >= slab_off[p.num_rel]) return;
  const int64_t rel = find_segment(slab_off, p.num_rel, slab);
  const int64_t m0 = row_off[rel] + (slab - slab_off[rel]) * p.slab_rows;
  int64_t m1 = m0 + p.slab_rows;
  if (m1 > row_off[rel + 1]) m1 = row_off[rel + 1];
  const int i0 = (tile / p.tiles_j) * BM, j0 = (tile % p.tiles_j) * BN;
  const int D1 = p.D1, D2 = p.D2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // A tile wm (features 32 wm ..), dC tiles 2 wn, 2 wn + 1
  f32x16 acc0[2], acc1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[j][r] = acc1[j][r] = 0.f;

  // DMA: one instruction per operand and wave: slot rows 2 wave (lanes 0-31), 2 wave + 1 (lanes 32-63)
  const int drow = 2 * wave + (lane >> 5);
  int64_t offA, offC;
  {
    int fa = i0 + (lane & 31) * 4, fc = j0 + (lane & 31) * 4;
    if (fa >= D1) fa = D1 - 4;
    if (fc >= D2) fc = D2 - 4;
    offA = (static_cast<int64_t>(drow) * D1 + fa) * 4;
    offC = (static_cast<int64_t>(drow) * D2 + fc) * 4;
  }
  const char* __restrict__ baseA = static_cast<const char*>(p.a) + m0 * D1 * 4;
  const char* __restrict__ baseC = static_cast<const char*>(p.dc) + m0 * D2 * 4;
  const int64_t rows = m1 - m0;
  const int nsl = static_cast<int>((rows + kBwdGldsRowsF32 - 1) / kBwdGldsRowsF32);
  auto issue = [&](int t, int ring) {   // `ring` = t % NS, a compile-time constant at every call site
    if (t >= nsl) return;
    char* dst = smem + ring * kSlot + wave * 1024;
    const int64_t mrow = static_cast<int64_t>(t) * kBwdGldsRowsF32;
    const bool in = mrow + drow < rows;
    const char* sa = in ? baseA + mrow * D1 * 4 + offA : g_mm_zero_page;
    const char* sc = in ? baseC + mrow * D2 * 4 + offC : g_mm_zero_page;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)sa, (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)sc, (lds_ptr_t)(dst + kPart), 16, 0, 0);
  };
  // this wave's DMA of a slot has landed while `later` slots issued behind it may stay in flight; lgkmcnt(0) with it:
  // the wave's own LDS writes (the converted pieces) are out before the barrier that follows
  auto wait_for_slot = [&](int later) {
    switch (later) {
      case 0: __builtin_amdgcn_s_waitcnt(0x0070); break;
      case 1: __builtin_amdgcn_s_waitcnt(0x0070 | (1 * kLoads)); break;
      case 2: __builtin_amdgcn_s_waitcnt(0x0070 | (2 * kLoads)); break;
      default: __builtin_amdgcn_s_waitcnt(0x0070 | (3 * kLoads)); break;
    }
  };
  static_assert(NS == 5, "the waits below assume a ring of five slots");

  // Conversion task of this thread, the same for A and for dC: feature position fpos, k-half ckh (rows 8 ckh .. + 7).
  // IN PLACE: a wave's 64 threads read the 8 rows x 256 bytes that hold their features' fp32 values and write their
  // 64 + 64 sixteen-byte pieces (high plane, low plane) into the same 2 KB — rows 8 ckh + (j >> 4) (+ 4 for the low plane),
  // byte (fpos >> 6) 256 + (j & 15) 16 for feature j = fpos & 63 — so no other wave's data is touched and the slot needs
  // no second buffer.
  const int fpos = tid & 127, ckh = (tid >> 7) & 1, cop = tid >> 8;   // waves 0-3 convert A, waves 4-7 dC
  const int64_t tab = rel * static_cast<int64_t>(D1 + D2);
  // the column a feature position holds (DMA lanes past the width were clamped to the last four columns; such positions
  // get the scale 0: their products land in accumulator rows / columns that are never stored)
  const bool liveX = cop ? j0 + fpos < D2 : i0 + fpos < D1;
  const int colX = liveX ? (cop ? j0 : i0) + fpos : 0;
  const int64_t tabX = tab + (cop ? D1 : 0) + colX;
  const float sX = liveX ? hp.scale[tabX] : 0.f;
  // an element is LISTED (removed from both planes, added in fp32 by h2_bwd_fix_kernel) when its high term is subnormal
  // (|x s| < 2^-14) and the pair is not zero: key = |h| | |l| as 15-bit patterns is then in [1, 0x400) — a normal high term
  // makes the key >= 0x400 whatever the low term holds, and a zero high term leaves a low term below 2^-14 as well.  In a
  // column whose sample was all zero (limit 1) ANY non-zero pattern takes that path and raises the flag instead.
  // Only one of the workgroups that convert the same rows lists them (tile column 0 for A, tile row 0 for dC).
  const bool listX = cop ? (tile / p.tiles_j) == 0 : (tile % p.tiles_j) == 0;
  const uint32_t thrX = (liveX && hp.limit[tabX] == 1u) ? 0x8000u : kH2bTiny;
  typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  const f32x2v spX = {sX, sX * 2048.f};

  struct Conv {
    uint32_t hw[4], lw[4];
    u16x2 mn;
  };
  auto load8 = [&](const char* src, float (&x)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = *reinterpret_cast<const float*>(src + r * 512);
  };
  auto split_pair = [&](float x0, float x1, f32x2v sp, Conv& c, int q) {   // sp = (scale, 2^11 scale)
    const f32x2v xp = f32x2v{x0, x1};
    f32x2v v, w;   // x s and 2^11 x s (exact): packed multiplies, the scale pair's low / high word for both lanes
    // (the scale pairs are broadcast in registers: with op_sel picking one word of a (scale, 2^11 scale) pair for both lanes the
    // products came out wrong on this chip)
    {
      const f32x2v s_lo = {sp[0], sp[0]}, s_hi = {sp[1], sp[1]};
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(xp), "v"(s_lo));
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(w) : "v"(xp), "v"(s_hi));
    }
    const h16x2v hb = __builtin_convertvector(v, h16x2v);                      // round to nearest
    c.hw[q] = __builtin_bit_cast(uint32_t, hb);
    float d0, d1;                                          // 2^11 (x s - h) = w - 2^11 h: one mixed-precision fma each, exact
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(c.hw[q]), "s"(-2048.f), "v"(w[0]));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(c.hw[q]), "s"(-2048.f), "v"(w[1]));
    const h16x2v lb = __builtin_convertvector(f32x2v{d0, d1}, h16x2v);
    c.lw[q] = __builtin_bit_cast(uint32_t, lb);
    // key = |h| | |l| with the sign bits forced on (one v_or3), minus 0x8001: zero -> 0xffff, a listed pattern -> below 0x3ff
    const u16x2 key = __builtin_bit_cast(u16x2, c.hw[q] | c.lw[q] | 0x80008000u);
    c.mn = __builtin_elementwise_min(c.mn, key - u16x2{0x8001, 0x8001});
  };
  auto put = [&](char* dsth, const Conv& c) {
    *reinterpret_cast<u32x4*>(dsth) = u32x4{c.hw[0], c.hw[1], c.hw[2], c.hw[3]};
    *reinterpret_cast<u32x4*>(dsth + 2048) = u32x4{c.lw[0], c.lw[1], c.lw[2], c.lw[3]};
  };
  // rare: the thread's piece holds an element for the list — take it out (the thread rewrites its own piece)
  auto slow = [&](char* dsth, Conv& c, uint32_t thr, int op, int col, int64_t row0, bool lists) {
    if (thr == 0x8000u) {   // a zero-sample column that is not zero
      atomicOr(hp.flags, 1u);
      return;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t sh = (e & 1) * 16;
      const uint32_t key = ((c.hw[e >> 1] | c.lw[e >> 1]) >> sh) & 0x7fffu;
      if (key != 0u && key < kH2bTiny) {
        c.hw[e >> 1] &= ~(0xffffu << sh);
        c.lw[e >> 1] &= ~(0xffffu << sh);
        if (lists) {
          const uint32_t slot = atomicAdd(hp.flags + 1, 1u);
          if (slot < static_cast<uint32_t>(kH2bFixCap)) {
            H2bFix f;
            f.row = row0 + e;
            f.col = col;
            f.op_rel = static_cast<int32_t>((rel << 1) | op);
            hp.fix[slot] = f;
          } else {
            atomicOr(hp.flags, 1u);
          }
        }
      }
    }
    put(dsth, c);
  };
  auto flagged = [&](const Conv& c, uint32_t thr) { return c.mn[0] < thr - 1u || c.mn[1] < thr - 1u; };

  const int f = lane & 31, kh = lane >> 5;
  const int src_off = cop * kPart + ckh * 8 * 512 + fpos * 4;
  const int dst_off = cop * kPart + (ckh * 8 + ((fpos & 63) >> 4)) * 512 + (fpos >> 6) * 256 + (fpos & 15) * 16;
  // fragments: A tile wm = features 32 wm + f -> piece row 8 kh + 2 (wm & 1) + (f >> 4), byte (wm >> 1) 256;
  // dC tile j = features 64 wn + 32 j + f -> piece row 8 kh + 2 j + (f >> 4), byte wn 256
  const int fragA = (kh * 8 + 2 * (wm & 1) + (f >> 4)) * 512 + (wm >> 1) * 256 + (f & 15) * 16;
  const int fragC = kPart + (kh * 8 + (f >> 4)) * 512 + wn * 256 + (f & 15) * 16;
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue(i, i);
  // slot 0, converted where it lies
  {
    wait_for_slot(nsl - 1 < 3 ? nsl - 1 : 3);
    __builtin_amdgcn_s_barrier();
    float xx[8];
    load8(smem + src_off, xx);
    Conv cx;
    cx.mn = u16x2{0xffff, 0xffff};
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(xx[2 * q], xx[2 * q + 1], spX, cx, q);
    put(smem + dst_off, cx);
    if (__builtin_expect(flagged(cx, thrX), 0)) slow(smem + dst_off, cx, thrX, cop, colX, m0 + ckh * 8, listX);
  }
  // the ring is walked in rounds of NS slots with the round unrolled: every LDS address is a per-thread base plus an immediate
  // (the t % NS arithmetic per slot cost 14 of the loop's 72 vector instructions)
  for (int t0 = 0; t0 < nsl; t0 += NS) {
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int t = t0 + u;
    if (t >= nsl) break;
    // slot t + 1 has landed (slots up to t + 3 are under way), this wave's converted pieces of slot t are out
    const int behind = nsl - 2 - t;
    if (behind >= 0)
      wait_for_slot(behind < 2 ? behind : 2);
    else
      __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();      // every wave: slot t converted, the fragments of slot t - 1 read
    issue(t + NS - 1, (u + NS - 1) % NS);   // into the slot that held t - 1
    const char* stg = smem + u * kSlot;
    char* slot = smem + ((u + 1) % NS) * kSlot;
    const bool more = t + 1 < nsl;     // (uniform; the last pass converts a stale slot nobody reads)
    const char* pa0 = stg + fragA;
    const char* pb0 = stg + fragC;
#define DGLA_H2B_MFMA(j, AH, AL, BH, BL)                                               \
  acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, acc1[j], 0, 0, 0);          \
  acc0[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, acc0[j], 0, 0, 0);          \
  acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, acc1[j], 0, 0, 0);
    Conv cx;
    cx.mn = u16x2{0xffff, 0xffff};
    float x[8];
    const h16x8 ah = *reinterpret_cast<const h16x8*>(pa0), al = *reinterpret_cast<const h16x8*>(pa0 + 2048);
    h16x8 bh = *reinterpret_cast<const h16x8*>(pb0), bl = *reinterpret_cast<const h16x8*>(pb0 + 2048);
    load8(slot + src_off, x);
    DGLA_H2B_MFMA(0, ah, al, bh, bl)
    split_pair(x[0], x[1], spX, cx, 0);
    split_pair(x[2], x[3], spX, cx, 1);
    __builtin_amdgcn_sched_barrier(0);
    bh = *reinterpret_cast<const h16x8*>(pb0 + 1024);
    bl = *reinterpret_cast<const h16x8*>(pb0 + 2048 + 1024);
    DGLA_H2B_MFMA(1, ah, al, bh, bl)
    split_pair(x[4], x[5], spX, cx, 2);
    split_pair(x[6], x[7], spX, cx, 3);
    put(slot + dst_off, cx);
#undef DGLA_H2B_MFMA
    if (__builtin_expect(more && flagged(cx, thrX), 0))
      slow(slot + dst_off, cx, thrX, cop, colX, m0 + static_cast<int64_t>(t + 1) * kBwdGldsRowsF32 + ckh * 8, listX);
  }
  }

  // overflow of a scaled value (a column whose maximum lies above what its sample showed; Inf; NaN) arrives as a
  // non-finite accumulator: inf x = inf or NaN in every output of the element's row / column
  {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) bad |= !(__builtin_fabsf(acc0[j][r]) < __builtin_inff());
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0) && lane == 0) atomicOr(hp.flags, 1u);
  }
  float* out = p.acc + rel * static_cast<int64_t>(D1) * D2;
  const int col_l = lane & 31, rbase = 4 * (lane >> 5);
  const float* invA = hp.inv + tab;
  const float* invC = hp.inv + tab + D1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = j0 + wn * 64 + j * 32 + col_l;
    if (col >= D2) continue;
    const float ic = invC[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + rbase;
      if (row < D1) {
        const float v = __builtin_fmaf(acc1[j][r], 0x1p-11f, acc0[j][r]);
        atomicAdd(out + static_cast<int64_t>(row) * D2 + col, v * invA[row] * ic);
      }
    }
  }
}

// The listed elements' rank-one contributions, in plain fp32 (one workgroup per entry).
__global__ __launch_bounds__(256) void h2_bwd_fix_kernel(const H2bParams hp) {
  const MmBwdParams& p = hp.m;
  if (hp.flags[0] != 0u) return;   // the whole call is redone
  uint32_t n = hp.flags[1];
  if (n == 0u) return;
  if (n > static_cast<uint32_t>(kH2bFixCap)) n = kH2bFixCap;
  const float* A = static_cast<const float*>(p.a);
  const float* dC = static_cast<const float*>(p.dc);
  const int D1 = p.D1, D2 = p.D2;
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    const H2bFix fx = hp.fix[e];
    const int64_t rel = fx.op_rel >> 1;
    float* out = p.acc + rel * static_cast<int64_t>(D1) * D2;
    const float* sA = hp.scale + rel * static_cast<int64_t>(D1 + D2);
    if ((fx.op_rel & 1) == 0) {
      const float x = A[fx.row * D1 + fx.col];
      for (int j = threadIdx.x; j < D2; j += blockDim.x)
        atomicAdd(out + static_cast<int64_t>(fx.col) * D2 + j, x * dC[fx.row * D2 + j]);
    } else {
      const float y = dC[fx.row * D2 + fx.col];
      for (int i = threadIdx.x; i < D1; i += blockDim.x) {
        const float a = A[fx.row * D1 + i];
        if (h2b_listed(a, sA[i])) continue;   // that product was added by the A-side entry
        atomicAdd(out + static_cast<int64_t>(i) * D2 + fx.col, a * y);
      }
    }
  }
}

// flags[0] raised: the result is zeroed again for the X3 kernel launched behind this one.  (Also leaves a copy of the two
// words where dgla_segment_mm_backward_b_last_route can read them after the call's scratch is gone.)
__device__ uint32_t g_h2b_last_flags[2];
__global__ __launch_bounds__(256) void h2_bwd_reset_kernel(const uint32_t* __restrict__ flags, float* __restrict__ acc, int64_t n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_h2b_last_flags[0] = flags[0];
    g_h2b_last_flags[1] = flags[1];
  }
  if (flags[0] == 0u) return;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    acc[i] = 0.f;
}

// Rows per split-K slab: every slab ends in one fp32 atomic add per output element, and the
// atomics (not the MFMAs) bound the kernel when slabs are short (2048-row slabs: 320 M atomics
// = 4.0 ms of a 4.0 ms launch at 10 M rows).  Long slabs cut them; enough slabs must remain to
// fill the chip: aim at ~4 workgroups per slot of (CUs x 3 resident workgroups).
int64_t bwd_slab_rows(int64_t M, int64_t D1, int64_t D2) {
  const int64_t tiles = ((D1 + BM - 1) / BM) * ((D2 + 127) / 128);
  const int64_t want_blocks = int64_t(mm_num_cus()) * 3 * 4;
  int64_t rows = (M * tiles + want_blocks - 1) / want_blocks;
  rows = (rows + 31) / 32 * 32;
  return rows < kMinSlabRows ? kMinSlabRows : rows;
}

template <typename DT>
int run_segment_mm_bwd_b(const void* a, const void* dc, void* db, int64_t M, int64_t D1, int64_t D2,
                         int64_t num_rel, const int64_t* row_index, char* ws, const MmScratch& sc,
                         hipStream_t s) {
  const int64_t* plan = reinterpret_cast<const int64_t*>(ws + sc.off_plan);
  const int64_t out_elems = num_rel * D1 * D2;
  if constexpr (sizeof(DT) == 8) {
    hipLaunchKernelGGL(segment_mm_bwd_b_f64_kernel, dim3(static_cast<unsigned>((out_elems + 255) / 256)),
                       dim3(256), 0, s, static_cast<const double*>(a), static_cast<const double*>(dc),
                       static_cast<double*>(db), plan, num_rel, static_cast<int>(D1),
                       static_cast<int>(D2), row_index);
  } else {
    float* acc = std::is_same<DT, float>::value ? static_cast<float*>(db)
                                                : reinterpret_cast<float*>(ws + sc.off_acc);
    DGLA_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(float) * out_elems, s));
    MmBwdParams p;
    p.a = a;
    p.dc = dc;
    p.acc = acc;
    p.plan = plan;
    p.num_rel = num_rel;
    p.D1 = static_cast<int>(D1);
    p.D2 = static_cast<int>(D2);
    constexpr int E = 16 / sizeof(DT);
    p.vec_a = (D1 % E == 0 && aligned16(a)) ? 1 : 0;
    p.vec_dc = (D2 % E == 0 && aligned16(dc)) ? 1 : 0;
    p.tiles_i = static_cast<int>((D1 + BM - 1) / BM);
    p.tiles_j = static_cast<int>((D2 + BN - 1) / BN);
    p.slab_rows = bwd_slab_rows(M, D1, D2);
    p.row_index = row_index;
    p.nan_flag = reinterpret_cast<uint32_t*>(const_cast<int64_t*>(plan) + 3 * num_rel + 2);
    const int64_t max_slabs = ((M + p.slab_rows - 1) / p.slab_rows + num_rel + 7) / 8 * 8;
    const int64_t blocks = max_slabs * p.tiles_i * p.tiles_j;
    if (blocks >= (int64_t(1) << 24)) return mfail("segment_mm backward: too many tiles (" +
                                                   std::to_string(blocks) + "); split the call");
    bool direct = false;
    if constexpr (sizeof(DT) == 2)
      direct = p.vec_a && p.vec_dc && !row_index && D1 >= 8 && D2 >= 8 && (tuning_flags() & kTuneGlds);
    if constexpr (sizeof(DT) == 2) {
      // Ring depth against occupancy (10 M rows, bf16, profiles/r1/glds_ab.jsonl): with few output
      // tiles per row slab, 5 slots x 2 workgroups per CU (128 KB in flight) wins — 256 x 256:
      // 2.08 ms against 2.57 (3 x 3) / 2.78 (2 x 4) / 2.62 (register-staged); with many tiles the
      // slab rows are re-read from L2 by the other tiles and co-residency wins — 512 x 512:
      // 2.58 ms (2 x 4) against 2.86 (5 x 2) / 3.79 (register-staged).
      if (direct && p.tiles_i * p.tiles_j >= 8)
        hipLaunchKernelGGL((segment_mm_bwd_b_glds_kernel<DT, 2>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
      else if (direct)
        hipLaunchKernelGGL((segment_mm_bwd_b_glds_kernel<DT, 5>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
    }
    if constexpr (sizeof(DT) == 4) {
      direct = p.vec_a && p.vec_dc && !row_index && D1 >= 4 && D2 >= 4 && (tuning_flags() & kTuneGlds);
      const bool x3 = !(tuning_flags() & kTuneMmF32);  // fp32 as an exact 3 x bf16 split (default)
      // Default: two scaled fp16 terms (segment_mm_bwd_b_h2_kernel) with the X3 kernel behind it as the route for inputs
      // whose columns leave the estimated range (it exits on its first load otherwise).  DGLA_TUNE_MM_X3 / _F32: the wider routes.
      // Calls of fewer than kH2bMinRows rows stay on the three-term kernel: nothing to gain there, and its 2^-26 per product
      // keeps segments of one or two rows inside the fp32-level bound of tests/test_mm.py (two fp16 terms: 2^-21.5).
      // DGLA_MM_BWD_H2 = 1 / 0 forces the choice (read per call: tests switch it inside one process).
      const char* e = getenv("DGLA_MM_BWD_H2");
      const int force = e && *e ? atoi(e) : -1;
      const bool h2 = direct && x3 && !(tuning_flags() & kTuneMmX3) && M > 0 && force != 0 && (force == 1 || M >= kH2bMinRows);
      if (h2) {
        H2bParams hp;
        hp.m = p;
        const size_t tab = (static_cast<size_t>(num_rel) * (D1 + D2) * 4 + 255) / 256 * 256;
        char* base = ws + sc.off_h2b;
        uint32_t* umax = reinterpret_cast<uint32_t*>(base);
        hp.scale = reinterpret_cast<const float*>(base + tab);
        hp.inv = reinterpret_cast<const float*>(base + 2 * tab);
        hp.limit = reinterpret_cast<const uint32_t*>(base + 3 * tab);
        hp.flags = reinterpret_cast<uint32_t*>(base + 4 * tab);
        hp.fix = reinterpret_cast<H2bFix*>(base + 4 * tab + 256);
        DGLA_CHECK_HIP(hipMemsetAsync(umax, 0, tab, s));
        DGLA_CHECK_HIP(hipMemsetAsync(hp.flags, 0, 256, s));
        const int64_t ncol = num_rel * (D1 + D2);
        hipLaunchKernelGGL(h2_bwd_sample_kernel, dim3(static_cast<unsigned>(num_rel * kH2bChunks)), dim3(256), 0, s, p, umax);
        hipLaunchKernelGGL(h2_bwd_scales_kernel, dim3(static_cast<unsigned>((ncol + 255) / 256)), dim3(256), 0, s, umax, ncol,
                           const_cast<float*>(hp.scale), const_cast<float*>(hp.inv), const_cast<uint32_t*>(hp.limit), hp.flags);
        hipLaunchKernelGGL(segment_mm_bwd_b_h2_kernel, dim3(static_cast<unsigned>(blocks)), dim3(512), 0, s, hp);
        hipLaunchKernelGGL(h2_bwd_fix_kernel, dim3(256), dim3(256), 0, s, hp);
        hipLaunchKernelGGL(h2_bwd_reset_kernel, dim3(256), dim3(256), 0, s, hp.flags, acc, out_elems);
        p.only_if = hp.flags;
      }
      if (direct && p.tiles_i * p.tiles_j >= 8) {
        if (x3)
          hipLaunchKernelGGL((segment_mm_bwd_b_glds_f32_kernel<2, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
        else
          hipLaunchKernelGGL((segment_mm_bwd_b_glds_f32_kernel<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
      } else if (direct) {
        if (x3)
          hipLaunchKernelGGL((segment_mm_bwd_b_glds_f32_kernel<5, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
        else
          hipLaunchKernelGGL((segment_mm_bwd_b_glds_f32_kernel<5>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
      }
    }
    if constexpr (sizeof(DT) == 4) {
      if (direct && !(tuning_flags() & kTuneMmF32))  // no-op unless an X3 accumulator came out NaN
        hipLaunchKernelGGL(x3_repair_bwd_kernel, dim3(512), dim3(256), 0, s, p);
    }
    if (!direct)
      hipLaunchKernelGGL((segment_mm_bwd_b_kernel<DT>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, p);
    if (!std::is_same<DT, float>::value)
      hipLaunchKernelGGL((convert_from_f32_kernel<DT>), dim3(static_cast<unsigned>(std::min<int64_t>((out_elems + 255) / 256, 4096))),
                         dim3(256), 0, s, acc, static_cast<DT*>(db), out_elems);
  }
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

template <typename Idx, typename DT>
int run_gather_mm(const void* a, const void* b, void* c, const void* ia, const void* ib,
                  const void* ic, int64_t rows, int64_t K, int64_t N, hipStream_t s) {
  const int64_t blocks = (rows * 64 + 255) / 256;
  if (blocks > 0x7fffffffLL) return mfail("gather_mm: too many rows");
  hipLaunchKernelGGL((gather_mm_kernel<Idx, DT>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                     static_cast<const DT*>(a), static_cast<const DT*>(b), static_cast<DT*>(c),
                     static_cast<const Idx*>(ia), static_cast<const Idx*>(ib),
                     static_cast<const Idx*>(ic), rows, static_cast<int>(K), static_cast<int>(N), 0);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int check_mm_common(int idbits, dgla_dtype dtype, int64_t M, int64_t K, int64_t N, int64_t R) {
  if (idbits != 32 && idbits != 64) return mfail("idtype must be int32 or int64");
  if (dtype < DGLA_F32 || dtype > DGLA_BF16) return mfail("unsupported feature dtype");
  if (M < 0 || K < 0 || N < 0 || R < 0) return mfail("negative dimension");
  if (K > 0x3fffffff || N > 0x3fffffff) return mfail("feature dimension too large");
  return 0;
}

}  // namespace
}  // namespace dgla

using namespace dgla;

extern "C" {

size_t dgla_segment_mm_workspace_bytes(dgla_dtype dtype, int64_t num_rel, int64_t d1, int64_t d2) {
  const size_t elem = dtype == DGLA_F64 ? 8 : (dtype == DGLA_F32 ? 4 : 2);
  return std::max(mm_scratch(num_rel, d1, d2, elem, true, true, true).total, mm_scratch(num_rel, d1, d2, elem, false, elem == 2).total);
}

int dgla_segment_mm_indexed(int idtype_bits, dgla_dtype dtype, const void* a, const void* b, void* c,
                            const void* seglen, int seglen_on_host, const int64_t* row_index,
                            int64_t num_rows, int64_t num_rel, int64_t k, int64_t n, int b_trans,
                            void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (check_mm_common(idtype_bits, dtype, num_rows, k, n, num_rel)) return -1;
  if (num_rows == 0 || n == 0 || num_rel == 0) return 0;
  if (!a || !b || !c || !seglen) return mfail("segment_mm: null operand");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, c);
  const size_t elem = dtype == DGLA_F64 ? 8 : (dtype == DGLA_F32 ? 4 : 2);
  const MmScratch sc = mm_scratch(num_rel, k, n, elem, !b_trans, false, true);
  void* owned = nullptr;
  if (!workspace || workspace_bytes < sc.total) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, sc.total, s));
    workspace = owned;
  }
  char* ws = static_cast<char*>(workspace);
  int rc = stage_plan(idtype_bits, seglen, seglen_on_host, num_rel, BM, num_rows, ws, sc, s, true);
  if (rc == 0) {
    hipLaunchKernelGGL(segment_zero_tail_kernel, dim3(256), dim3(256), 0, s,
                       reinterpret_cast<const int64_t*>(ws + sc.off_plan), num_rel, row_index,
                       static_cast<unsigned char*>(c), num_rows, n * static_cast<int64_t>(elem));
    switch (dtype) {
      case DGLA_F32: rc = run_segment_mm<float>(a, b, c, num_rows, k, n, num_rel, b_trans != 0, row_index, ws, sc, s); break;
      case DGLA_F64: rc = run_segment_mm<double>(a, b, c, num_rows, k, n, num_rel, b_trans != 0, row_index, ws, sc, s); break;
      case DGLA_F16: rc = run_segment_mm<f16_t>(a, b, c, num_rows, k, n, num_rel, b_trans != 0, row_index, ws, sc, s); break;
      case DGLA_BF16: rc = run_segment_mm<bf16_t>(a, b, c, num_rows, k, n, num_rel, b_trans != 0, row_index, ws, sc, s); break;
    }
  }
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

int dgla_segment_mm(int idtype_bits, dgla_dtype dtype, const void* a, const void* b, void* c,
                    const void* seglen, int seglen_on_host, int64_t num_rows, int64_t num_rel,
                    int64_t k, int64_t n, int b_trans, void* workspace, size_t workspace_bytes,
                    void* hip_stream) {
  return dgla_segment_mm_indexed(idtype_bits, dtype, a, b, c, seglen, seglen_on_host, nullptr, num_rows,
                                 num_rel, k, n, b_trans, workspace, workspace_bytes, hip_stream);
}

int dgla_segment_mm_backward_b_indexed(int idtype_bits, dgla_dtype dtype, const void* a, const void* dc,
                                       void* db, const void* seglen, int seglen_on_host,
                                       const int64_t* row_index, int64_t num_rows, int64_t num_rel,
                                       int64_t d1, int64_t d2, void* workspace, size_t workspace_bytes,
                                       void* hip_stream) {
  if (check_mm_common(idtype_bits, dtype, num_rows, d1, d2, num_rel)) return -1;
  if (num_rel == 0 || d1 == 0 || d2 == 0) return 0;
  if (!db || !seglen || (num_rows > 0 && (!a || !dc))) return mfail("segment_mm backward: null operand");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, db);
  const size_t elem = dtype == DGLA_F64 ? 8 : (dtype == DGLA_F32 ? 4 : 2);
  const MmScratch sc = mm_scratch(num_rel, d1, d2, elem, false, elem == 2);
  void* owned = nullptr;
  if (!workspace || workspace_bytes < sc.total) {
    DGLA_CHECK_HIP(hipMallocAsync(&owned, sc.total, s));
    workspace = owned;
  }
  char* ws = static_cast<char*>(workspace);
  int rc = stage_plan(idtype_bits, seglen, seglen_on_host, num_rel, static_cast<int>(bwd_slab_rows(num_rows, d1, d2)), num_rows, ws, sc, s);
  if (rc == 0) {
    switch (dtype) {
      case DGLA_F32: rc = run_segment_mm_bwd_b<float>(a, dc, db, num_rows, d1, d2, num_rel, row_index, ws, sc, s); break;
      case DGLA_F64: rc = run_segment_mm_bwd_b<double>(a, dc, db, num_rows, d1, d2, num_rel, row_index, ws, sc, s); break;
      case DGLA_F16: rc = run_segment_mm_bwd_b<f16_t>(a, dc, db, num_rows, d1, d2, num_rel, row_index, ws, sc, s); break;
      case DGLA_BF16: rc = run_segment_mm_bwd_b<bf16_t>(a, dc, db, num_rows, d1, d2, num_rel, row_index, ws, sc, s); break;
    }
  }
  if (owned) (void)hipFreeAsync(owned, s);
  return rc;
}

int dgla_segment_mm_backward_b(int idtype_bits, dgla_dtype dtype, const void* a, const void* dc,
                               void* db, const void* seglen, int seglen_on_host,
                               int64_t num_rows, int64_t num_rel, int64_t d1, int64_t d2,
                               void* workspace, size_t workspace_bytes, void* hip_stream) {
  return dgla_segment_mm_backward_b_indexed(idtype_bits, dtype, a, dc, db, seglen, seglen_on_host, nullptr,
                                            num_rows, num_rel, d1, d2, workspace, workspace_bytes,
                                            hip_stream);
}

int dgla_segment_mm_backward_b_last_route(uint32_t* fell_back, uint32_t* listed_elements) {
  uint32_t w[2] = {0, 0};
  DGLA_CHECK_HIP(hipMemcpyFromSymbol(w, HIP_SYMBOL(g_h2b_last_flags), sizeof(w)));
  if (fell_back) *fell_back = w[0];
  if (listed_elements) *listed_elements = w[1];
  return 0;
}

int dgla_gather_mm(int idtype_bits, dgla_dtype dtype, const void* a, const void* b, void* c,
                   const void* idx_a, const void* idx_b, const void* idx_c, int64_t num_rows,
                   int64_t k, int64_t n, void* hip_stream) {
  if (check_mm_common(idtype_bits, dtype, num_rows, k, n, 0)) return -1;
  if (num_rows == 0 || n == 0) return 0;
  if (!a || !b || !c) return mfail("gather_mm: null operand");
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const DeviceGuard dev(s, c);
#define DGLA_GMM(IDX)                                                                             \
  switch (dtype) {                                                                                \
    case DGLA_F32: return run_gather_mm<IDX, float>(a, b, c, idx_a, idx_b, idx_c, num_rows, k, n, s);   \
    case DGLA_F64: return run_gather_mm<IDX, double>(a, b, c, idx_a, idx_b, idx_c, num_rows, k, n, s);  \
    case DGLA_F16: return run_gather_mm<IDX, f16_t>(a, b, c, idx_a, idx_b, idx_c, num_rows, k, n, s);   \
    case DGLA_BF16: return run_gather_mm<IDX, bf16_t>(a, b, c, idx_a, idx_b, idx_c, num_rows, k, n, s); \
  }
  if (idtype_bits == 32) {
    DGLA_GMM(int32_t)
  } else {
    DGLA_GMM(int64_t)
  }
#undef DGLA_GMM
  return mfail("unsupported feature dtype");
}

}  // extern "C"
