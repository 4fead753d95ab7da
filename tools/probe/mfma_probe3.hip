// Probe 3: bf16 / fp32 MFMA rate with RANDOM operands cycling through registers (switching activity -> power -> clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: constant operands, 1: random bf16 operands (8 register sets), 2: fp32 mfma random
__global__ __launch_bounds__(256) void probe(float* out, const unsigned* rnd, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const u32x4*>(rnd + ((threadIdx.x * 8 + i) * 4) % 65536);
    b[i] = *reinterpret_cast<const u32x4*>(rnd + ((threadIdx.x * 8 + i) * 4 + 32768) % 65536);
    if (MODE == 0) { a[i] = a[0]; b[i] = b[0]; }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MODE < 2) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[(m + it) & 7]), acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[m].x), __uint_as_float(b[m].y), acc[m & 3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, unsigned* rnd) {
  const int iters = 40000, grid = 256;       // long enough (~10 ms) for the power controller to react
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<MODE><<<grid, 256>>>(out, rnd, 1000);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    probe<MODE><<<grid, 256>>>(out, rnd, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 8;
    const double flops = mfmas * (MODE < 2 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
    printf("%-34s rep %d: %8.3f ms  %8.1f TFLOP/s  %.2f ns/MFMA\n", name, rep, ms, flops / ms / 1e9, ms * 1e6 / (8.0 * iters));
  }
}

int main() {
  float* out; unsigned* rnd;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&rnd, 65536 * 4 + 64);
  unsigned* h = (unsigned*)malloc(65536 * 4);
  for (int i = 0; i < 65536; ++i) {
    // two random bf16 values of magnitude ~1 per dword
    unsigned lo = 0x3F00u | (rand() & 0x80FF), hi = 0x3F00u | (rand() & 0x80FF);
    h[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(rnd, h, 65536 * 4, hipMemcpyHostToDevice);
  run<0>("bf16 32x32x16 constant operands", out, rnd);
  run<1>("bf16 32x32x16 random operands", out, rnd);
  run<2>("fp32 32x32x2 random operands", out, rnd);
  return 0;
}
