// Probe: can bf16 MFMA (32x32x16) overlap with VALU work on gfx950, and at what rate?  (fp32 MFMA shares the FMA lanes.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NVALU>   // MODE 0: bf16 32x32x16, 1: fp32 32x32x2
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 2 + i); }
  float fa = seed + threadIdx.x, fb = seed;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (MODE == 0) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
      else acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NVALU; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NVALU>
void run(const char* name, int wg_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 4000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NVALU><<<grid, 256>>>(out, 100, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, NVALU><<<grid, 256>>>(out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)grid * 4 * iters * 4;
  const double flops = mfmas * (MODE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
  printf("%-28s wg/cu=%d nvalu/mfma=%2d : %8.3f ms  %8.1f TFLOP/s  (%.1f ns per MFMA per SIMD-wave)\n", name, wg_per_cu, NVALU, ms,
         flops / ms / 1e9, ms * 1e6 / (4.0 * iters));
  hipFree(out);
}

int main() {
  run<0, 0>("bf16 32x32x16", 1); run<0, 0>("bf16 32x32x16", 2);
  run<0, 4>("bf16 32x32x16", 1); run<0, 4>("bf16 32x32x16", 2);
  run<0, 8>("bf16 32x32x16", 1); run<0, 8>("bf16 32x32x16", 2);
  run<0, 16>("bf16 32x32x16", 1); run<0, 16>("bf16 32x32x16", 2);
  run<1, 0>("fp32 32x32x2", 1); run<1, 0>("fp32 32x32x2", 2);
  run<1, 4>("fp32 32x32x2", 1); run<1, 4>("fp32 32x32x2", 2);
  run<1, 8>("fp32 32x32x2", 2);
  return 0;
}
