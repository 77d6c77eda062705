// Probe of ds_read_b64_tr_b16 (gfx950 transposing LDS load): fills LDS with halfword indices, lets lane L
// point at row L & 15, byte (L >> 4) * 8, and prints which (row, halfword) each lane receives.
// Build: hipcc -O2 --offload-arch=gfx950 benchmarks/probe_ds_read_tr.hip -o build/tr_probe ; output of an
// MI355X run: profiles/r1/ds_read_tr16_b64_probe.txt (used by segment_mm_bwd_b_glds_kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // each lane points at (row = lane & 15) * stride + (lane >> 4) * 8 bytes
  const char* p = reinterpret_cast<const char*>(lds) + (lane & 15) * stride_bytes + (lane >> 4) * 8;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {64, 128}) {
    k<<<1, 64>>>(d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d bytes (row r at halfword %d*r); lane: 4 values as (row,col) in halfwords\n", stride, stride / 2);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (%2d,%2d)", h[l * 4 + j] / (stride / 2), h[l * 4 + j] % (stride / 2));
      printf("\n");
    }
  }
  return 0;
}
