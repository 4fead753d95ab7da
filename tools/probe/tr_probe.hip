// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which element lands in which lane / register position).
// LDS image: u16 value = row*256 + col for a [64 rows][16 cols] tile (32 B per row).  Lane L supplies the address of 4
// contiguous u16: row = 4*(L/16) + (L%16)/4, col0 = 4*(L%4).  Output: the 4 u16 each lane received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ unsigned short tile[64 * 16];
  for (int i = threadIdx.x; i < 64 * 16; i += 64) tile[i] = (unsigned short)((i / 16) * 256 + (i % 16));
  __syncthreads();
  const int L = threadIdx.x;
  const int row = 4 * (L / 16) + (L % 16) / 4, col0 = 4 * (L % 4);
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(&tile[row * 16 + col0]);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[L * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int L = 0; L < 64; ++L) {
    printf("lane %2d (addr row %2d col %2d):", L, 4 * (L / 16) + (L % 16) / 4, 4 * (L % 4));
    for (int e = 0; e < 4; ++e) printf("  (r%2d,c%2d)", h[L * 4 + e] >> 8, h[L * 4 + e] & 255);
    printf("\n");
  }
  return 0;
}
