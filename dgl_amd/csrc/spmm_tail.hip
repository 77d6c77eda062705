// Column-sliced tail pass of the CSR g-SpMM (copy_u + sum, fp32 rows of 128 k + 16 bytes).
//
// Why: the merge kernel runs at the L2 <-> fabric REQUEST rate (55 G
// whole-line requests per second, equal to the read-only streaming peak of the box).  A 400-byte
// row costs four requests per gathered edge: three whole lines and one request for the 16-byte
// row tail, which misses the 4 MiB L2 of the XCD because the dense tail array (16 bytes x columns
// = 39 MB on the headline graph) is touched uniformly at random.  That fourth request is 19 % of
// the kernel's time (F = 96, three lines per edge: 3.54 ms against 4.36 ms).
//
// Here the tails are summed by a pass of their own in which they DO hit in L2: the edges are
// regrouped once per graph by COLUMN SLICE (slice = a range of columns whose tails fill ~2.5 MB)
// — a "virtual" CSR with S x num_rows rows, virtual row (s, r) = the edges of row r whose column
// lies in slice s, in CSR order — and the merge path of that CSR is walked slice-major with the
// same XCD-contiguous unit order as the main kernel, so one XCD works on one slice at a time and
// its L2 holds that slice.  Per edge the pass reads a 4-byte column id (streamed) and a 16-byte
// tail (L2 hit); per virtual row it writes one 16-byte partial; a last elementwise kernel adds the
// S partials of every row in slice order and stores the row's last four outputs.  The main kernel
// then gathers three lines per edge and leaves the last 16 bytes of every output row alone.
//
// Reference being replaced: the same SpMMCsrKernel loop as spmm_csr.cuh (src/array/cuda/spmm.cuh:
// 496-543); the summation order of the last four columns becomes (slice, CSR position) instead
// of CSR position — deterministic, inside the 1e-5 bound the sums are held to, and switched off
// by clearing DGLA_TUNE_TAIL_PASS.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

namespace dgla {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool tail_wanted(const unsigned* __restrict__ meta) {
  if (meta == nullptr) return true;
  return 16u * meta[0] < 15u * meta[1];  // the locality probe's verdict, as split_wanted(meta, true) in spmm_csr.cuh
}

struct SliceOf {
  unsigned magic, last;
  __host__ __device__ uint8_t operator()(int32_t c) const {
    const unsigned s = static_cast<unsigned>((static_cast<uint64_t>(static_cast<uint32_t>(c)) * magic) >> 32);
    return static_cast<uint8_t>(s < last ? s : last);
  }
};

// After the stable sort by slice: position i of the virtual CSR takes edge perm[i] (CSR position);
// its row is found in indptr, its virtual row is slice * num_rows + row; vptr[q] = first position
// whose virtual row is >= q, written by the thread that sees the run boundary.
__global__ __launch_bounds__(256) void tail_compress_kernel(
    const uint8_t* __restrict__ skey, const int32_t* __restrict__ perm,
    const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices, int64_t num_rows,
    int64_t nnz, int64_t vrows, int32_t* __restrict__ vptr, int32_t* __restrict__ tcol) {
  auto vrow_of = [&](int64_t i) {
    const int32_t e = perm[i];
    int64_t lo = 0, hi = num_rows;  // largest r with indptr[r] <= e
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (indptr[mid] <= e)
        lo = mid;
      else
        hi = mid - 1;
    }
    return static_cast<int64_t>(skey[i]) * num_rows + lo;
  };
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < nnz; i += stride) {
    tcol[i] = indices[perm[i]];
    const int64_t v = vrow_of(i);
    const int64_t vp = i > 0 ? vrow_of(i - 1) : -1;
    for (int64_t q = vp + 1; q <= v; ++q) vptr[q] = static_cast<int32_t>(i);
    if (i == nnz - 1)
      for (int64_t q = v + 1; q <= vrows; ++q) vptr[q] = static_cast<int32_t>(nnz);
  }
}

__global__ void tail_plan_kernel(const int32_t* __restrict__ vptr, int64_t vrows, int64_t nnz,
                                 int64_t num_waves, int64_t* __restrict__ plan) {
  const int64_t w = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (w > num_waves) return;
  int64_t d = w * kTailWaveItems;
  const int64_t total = vrows + nnz;
  if (d > total) d = total;
  int64_t lo = 0, hi = vrows;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (static_cast<int64_t>(vptr[mid]) + mid <= d)
      lo = mid;
    else
      hi = mid - 1;
  }
  plan[w] = lo;
}

struct TailParams {
  const int32_t* vptr;
  const int32_t* tcol;
  const int64_t* plan;
  int64_t vrows, nnz, num_waves;
  const f4* s2;  // [num_cols] the 16-byte row tails (spmm_split_edges_kernel's dense side array)
  f4* part;      // [vrows] partial tail sums
  int64_t* carry_row;
  f4* carry_val;
  f4* tail_val;
  const unsigned* meta;
  uint32_t tune;
};

// One wave per unit of kTailWaveItems merge items of the virtual CSR; a lane owns T consecutive items.
// Finished partial sums are collected in LDS and leave the wave as whole coalesced lines: written
// straight from the lanes that close the rows they were 16-byte pieces to scattered places, one
// fabric write request each (39 M of them on the headline graph: 0.7 ms at the request rate).
__global__ __launch_bounds__(64 * kWavesPerBlock) void spmm_tail_kernel(const TailParams p) {
  if (!tail_wanted(p.meta)) return;
  constexpr int TI = kTailWaveItems;
  constexpr int T = TI / 64;  // items per lane
  __shared__ int s_cols[kWavesPerBlock][TI];
  __shared__ int s_rend[kWavesPerBlock][TI + 2];
  __shared__ f4 s_out[kWavesPerBlock][TI];
  const int wib = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  unsigned blk = blockIdx.x;
  if (p.tune & kTuneXcd) {  // XCD x walks one contiguous eighth of the slice-major merge path
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u;
    const unsigned x = blk & 7u, i = blk >> 3;
    blk = x * q + (x < r ? x : r) + i;
  }
  const int64_t w = static_cast<int64_t>(blk) * kWavesPerBlock + wib;
  int64_t i0 = 0, j0 = 0;
  int R = 0, nE = 0;
  if (w < p.num_waves) {
    const int64_t total = p.vrows + p.nnz;
    const int64_t d0 = w * TI;
    int64_t d1 = d0 + TI;
    if (d1 > total) d1 = total;
    i0 = p.plan[w];
    const int64_t i1 = p.plan[w + 1];
    j0 = d0 - i0;
    R = static_cast<int>(i1 - i0);
    nE = static_cast<int>((d1 - i1) - j0);
    const int items = nE + R;
    int itemv[T];
#pragma unroll
    for (int k = 0; k < T; ++k) {
      int it = lane + 64 * k;
      if (it >= items) it = items - 1;
      const int32_t* src = it < nE ? p.tcol + (j0 + it) : p.vptr + (i0 + 1 + (it - nE));
      itemv[k] = __builtin_nontemporal_load(src);  // read-once streams: leave the L2 to the tails
    }
    const int64_t first = static_cast<int64_t>(p.vptr[i0]) - j0;
#pragma unroll
    for (int k = 0; k < T; ++k) {
      const int it = lane + 64 * k;
      if (it < nE)
        s_cols[wib][it] = itemv[k];
      else if (it < items)
        s_rend[wib][it - nE + 1] = static_cast<int>(static_cast<int64_t>(itemv[k]) - j0);
    }
    if (lane == 0) s_rend[wib][0] = first < 0 ? -1 : static_cast<int>(first);
  }
  __syncthreads();
  if (w >= p.num_waves) return;

  const int* cols = s_cols[wib];
  const int* rend = s_rend[wib];
  f4* outl = s_out[wib];
  const int items = R + nE;
  int dlo = lane * T;
  if (dlo > items) dlo = items;
  // number of row ends at merge position < dlo (row end t sits at rend[t + 1] + t); the lane's
  // range ends where the next lane's begins
  int t_s;
  {
    int lo = 0, hi = R;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rend[mid + 1] + mid < dlo)
        lo = mid + 1;
      else
        hi = mid;
    }
    t_s = lo;
  }
  int t_e = __shfl_down(t_s, 1, 64);
  if (lane == 63) t_e = R;
  int dhi = dlo + T;
  if (dhi > items) dhi = items;
  const int e_s = dlo - t_s, e_e = dhi - t_e;

  f4 v[T];
#pragma unroll
  for (int u = 0; u < T; ++u) {
    const int ee = e_s + u;
    const int c = ee < e_e ? cols[ee] : 0;
    v[u] = p.s2[c];  // plain load: the slice stays in this XCD's L2
  }

  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 head = {0.f, 0.f, 0.f, 0.f};
  bool has_end = false;
  int first_t = 0;
  int t = t_s;
  int next_end = (t < R) ? rend[t + 1] : 0x7fffffff;
  auto close = [&]() {
    if (!has_end) {
      head = acc;
      has_end = true;
      first_t = t;
    } else {
      outl[t] = acc;
    }
    acc = f4{0.f, 0.f, 0.f, 0.f};
    ++t;
    next_end = (t < R) ? rend[t + 1] : 0x7fffffff;
  };
#pragma unroll
  for (int u = 0; u < T; ++u) {
    const int ee = e_s + u;
    if (ee < e_e) {
      while (ee >= next_end) close();
      acc += v[u];
    }
  }
  while (t < t_e) close();

  // carry-in of every lane: segmented inclusive scan of the trailing partials (a lane that closed
  // a row starts a new segment), then shifted by one lane
  f4 sv = acc;
  int sf = has_end ? 1 : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    f4 pv;
    pv.x = __shfl_up(sv.x, d, 64);
    pv.y = __shfl_up(sv.y, d, 64);
    pv.z = __shfl_up(sv.z, d, 64);
    pv.w = __shfl_up(sv.w, d, 64);
    const int pf = __shfl_up(sf, d, 64);
    if (lane >= d && !sf) {
      sv = pv + sv;
      sf = pf;
    }
  }
  f4 cin;
  cin.x = __shfl_up(sv.x, 1, 64);
  cin.y = __shfl_up(sv.y, 1, 64);
  cin.z = __shfl_up(sv.z, 1, 64);
  cin.w = __shfl_up(sv.w, 1, 64);
  if (lane == 0) cin = f4{0.f, 0.f, 0.f, 0.f};
  const bool row0_began_earlier = rend[0] < 0;
  if (has_end) {
    const f4 tot = cin + head;
    if (first_t == 0 && row0_began_earlier)
      p.tail_val[w] = tot;  // the row began in an earlier unit: the fix-up kernel adds the carries
    else
      outl[first_t] = tot;
  }
  if (lane == 63) {
    const int trailing = nE - (R > 0 ? rend[R] : 0);  // edges of the row still open at the unit's end
    const bool carry = R > 0 ? trailing > 0 : nE > 0;
    p.carry_row[w] = carry ? i0 + R : int64_t(-1);
    if (carry) p.carry_val[w] = sv;
  }
  // the unit's finished rows are consecutive: whole lines leave the wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int i = lane + (row0_began_earlier ? 1 : 0); i < R; i += 64)
    __builtin_nontemporal_store(outl[i], p.part + (i0 + i));
}

// Rows of the virtual CSR that straddle units: carries in unit order, then the closing unit's part.
__global__ __launch_bounds__(256) void spmm_tail_fixup_kernel(const TailParams p) {
  if (!tail_wanted(p.meta)) return;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t s = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; s < p.num_waves; s += stride) {
    const int64_t row = p.carry_row[s];
    if (row < 0) continue;
    if (s > 0 && p.carry_row[s - 1] == row) continue;
    int64_t s2 = s + 1;
    while (s2 < p.num_waves && p.carry_row[s2] == row) ++s2;
    f4 acc = p.carry_val[s];
    for (int64_t q = s + 1; q < s2; ++q) acc += p.carry_val[q];
    acc += p.tail_val[s2];
    p.part[row] = acc;
  }
}

// out[r][F - 4 .. F) = sum over slices (ascending) of part[s][r]; `mean` and `accumulate` as the
// main kernel applies them to the other columns.
__global__ __launch_bounds__(256) void spmm_tail_combine_kernel(
    const f4* __restrict__ part, int64_t num_rows, int slices, float* __restrict__ out, int64_t out_len,
    const int32_t* __restrict__ indptr, int mean, int accumulate, const unsigned* __restrict__ meta) {
  if (!tail_wanted(meta)) return;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; r < num_rows; r += stride) {
    f4 acc = __builtin_nontemporal_load(part + r);
    for (int s = 1; s < slices; ++s) acc += __builtin_nontemporal_load(part + (static_cast<int64_t>(s) * num_rows + r));
    f4* o = reinterpret_cast<f4*>(out + r * out_len + (out_len - 4));
    if (accumulate) {
      acc = *o + acc;
    } else if (mean) {
      const int deg = indptr[r + 1] - indptr[r];
      const float den = static_cast<float>(deg > 1 ? deg : 1);
      acc = acc / den;
    }
    *o = acc;
  }
}

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

using KeyIt = rocprim::transform_iterator<const int32_t*, SliceOf, uint8_t>;
using PosIt = rocprim::counting_iterator<int32_t>;

size_t sort_temp_bytes(int64_t nnz, int bits) {
  size_t bytes = 0;
  const hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, KeyIt(nullptr, SliceOf{1u, 0u}),
                                                 static_cast<uint8_t*>(nullptr), PosIt(0),
                                                 static_cast<int32_t*>(nullptr), static_cast<size_t>(nnz), 0, bits,
                                                 nullptr);
  if (e != hipSuccess || bytes == 0) {
    (void)hipGetLastError();
    bytes = static_cast<size_t>(nnz) * 6 + (size_t(4) << 20);  // no device to ask: an upper bound
  }
  return bytes;
}

int bits_for(int n) {
  int b = 1;
  while ((1 << b) < n) ++b;
  return b;
}

}  // namespace

size_t spmm_tail_build_scratch_bytes(int64_t nnz, int slices) {
  return align256(static_cast<size_t>(nnz)) + align256(static_cast<size_t>(nnz) * 4) +
         align256(sort_temp_bytes(nnz, bits_for(slices)));
}

int spmm_tail_build(const CsrView& csr, int slices, int32_t* vptr, int32_t* tcol, int64_t* plan,
                    int64_t num_waves, char* scratch, hipStream_t s) {
  const int64_t nnz = csr.nnz, vrows = csr.num_rows * slices;
  uint8_t* skey = reinterpret_cast<uint8_t*>(scratch);
  int32_t* perm = reinterpret_cast<int32_t*>(scratch + align256(static_cast<size_t>(nnz)));
  void* temp = scratch + align256(static_cast<size_t>(nnz)) + align256(static_cast<size_t>(nnz) * 4);
  const int bits = bits_for(slices);
  size_t temp_bytes = sort_temp_bytes(nnz, bits);
  // slice of a column: columns * S / num_cols by a multiply-high (any monotone map into [0, S) does)
  const int64_t per = (csr.num_cols + slices - 1) / slices;
  SliceOf fn;
  fn.magic = static_cast<unsigned>(((uint64_t(1) << 32) + per - 1) / per);
  fn.last = static_cast<unsigned>(slices - 1);
  DGLA_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, KeyIt(static_cast<const int32_t*>(csr.indices), fn),
                                           skey, PosIt(0), perm, static_cast<size_t>(nnz), 0, bits, s));
  int64_t blocks = (nnz + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(tail_compress_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, skey, perm,
                     static_cast<const int32_t*>(csr.indptr), static_cast<const int32_t*>(csr.indices),
                     csr.num_rows, nnz, vrows, vptr, tcol);
  DGLA_CHECK_HIP(hipGetLastError());
  const int64_t n = num_waves + 1;
  hipLaunchKernelGGL(tail_plan_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, vptr,
                     vrows, nnz, num_waves, plan);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int spmm_tail_launch(const SpmmTailLaunch& t) {
  TailParams p;
  p.vptr = t.vptr;
  p.tcol = t.tcol;
  p.plan = t.plan;
  p.vrows = t.num_rows * t.slices;
  p.nnz = t.nnz;
  p.num_waves = t.num_waves;
  p.s2 = static_cast<const f4*>(t.s2);
  p.part = static_cast<f4*>(t.part);
  p.carry_row = t.carry_row;
  p.carry_val = static_cast<f4*>(t.carry_val);
  p.tail_val = static_cast<f4*>(t.tail_val);
  p.meta = t.meta;
  p.tune = t.tune;
  const unsigned blocks = static_cast<unsigned>((t.num_waves + kWavesPerBlock - 1) / kWavesPerBlock);
  hipLaunchKernelGGL(spmm_tail_kernel, dim3(blocks), dim3(64 * kWavesPerBlock), 0, t.stream, p);
  DGLA_CHECK_HIP(hipGetLastError());
  const unsigned fblocks = static_cast<unsigned>(std::min<int64_t>((t.num_waves + 255) / 256, 4096));
  hipLaunchKernelGGL(spmm_tail_fixup_kernel, dim3(fblocks), dim3(256), 0, t.stream, p);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

int spmm_tail_combine(const SpmmTailLaunch& t, void* out, int64_t out_len, const void* indptr, bool mean,
                      bool accumulate) {
  const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((t.num_rows + 255) / 256, 1 << 20));
  hipLaunchKernelGGL(spmm_tail_combine_kernel, dim3(blocks), dim3(256), 0, t.stream,
                     static_cast<const f4*>(t.part), t.num_rows, t.slices, static_cast<float*>(out), out_len,
                     static_cast<const int32_t*>(indptr), mean ? 1 : 0, accumulate ? 1 : 0, t.meta);
  DGLA_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace dgla
