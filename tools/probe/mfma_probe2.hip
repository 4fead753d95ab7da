// Probe 2: dependent accumulate chains of bf16 MFMA; cost of the fp32 -> 3 x bf16 split.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 2 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// split 8 floats into 3 planes of 8 bf16 (RNE), feed MFMA so nothing is dead
__device__ __forceinline__ void split8(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    f32x2 v = {x[i], x[i + 1]};
    bf16x2 hh = __builtin_convertvector(v, bf16x2);
    f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
    bf16x2 mm = __builtin_convertvector(r1, bf16x2);
    f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
    bf16x2 ll = __builtin_convertvector(r2, bf16x2);
    h[i] = hh[0]; h[i + 1] = hh[1]; m[i] = mm[0]; m[i + 1] = mm[1]; l[i] = ll[0]; l[i + 1] = ll[1];
  }
}

template <int NSPLIT>   // NSPLIT split8 calls per 24 MFMAs
__global__ __launch_bounds__(256) void splitk(float* out, const float* in, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x * 8 + i];
  bf16x8 h, m, l;
  split8(x, h, m, l);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < NSPLIT; ++s) {
      for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f;     // 8 extra VALU (stand-in for BN+ReLU prologue)
      split8(x, h, m, l);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, m, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m, l, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l, h, acc[2], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 24 - 3 * NSPLIT; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h, l, acc[k & 3], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(50); hipDeviceSynchronize();
  hipEventRecord(e0); f(2000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float *out, *in; hipMalloc(&out, 512 * 256 * 4); hipMalloc(&in, 256 * 8 * 4); hipMemset(in, 0, 256 * 8 * 4);
  for (int wg = 1; wg <= 2; ++wg) {
    float t1 = timeit([&](int n) { chain<1><<<256 * wg, 256>>>(out, n, 1.f); });
    float t2 = timeit([&](int n) { chain<2><<<256 * wg, 256>>>(out, n, 1.f); });
    float t4 = timeit([&](int n) { chain<4><<<256 * wg, 256>>>(out, n, 1.f); });
    float t8 = timeit([&](int n) { chain<8><<<256 * wg, 256>>>(out, n, 1.f); });
    printf("wg/cu=%d ns per MFMA (per SIMD): chain1 %.1f  chain2 %.1f  chain4 %.1f  chain8 %.1f\n", wg,
           t1 * 1e6 / (2000 * 8 * wg), t2 * 1e6 / (2000 * 8 * wg), t4 * 1e6 / (2000 * 8 * wg), t8 * 1e6 / (2000 * 8 * wg));
    float s0 = timeit([&](int n) { splitk<0><<<256 * wg, 256>>>(out, in, n); });
    float s1 = timeit([&](int n) { splitk<1><<<256 * wg, 256>>>(out, in, n); });
    float s2 = timeit([&](int n) { splitk<2><<<256 * wg, 256>>>(out, in, n); });
    float s4 = timeit([&](int n) { splitk<4><<<256 * wg, 256>>>(out, in, n); });
    printf("wg/cu=%d ns per 24 MFMAs with k split8 (8 elts each): k=0 %.1f  k=1 %.1f  k=2 %.1f  k=4 %.1f\n", wg,
           s0 * 1e6 / (2000 * wg), s1 * 1e6 / (2000 * wg), s2 * 1e6 / (2000 * wg), s4 * 1e6 / (2000 * wg));
  }
  return 0;
}
