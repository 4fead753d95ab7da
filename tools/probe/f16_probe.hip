// Probe 4: (a) sustained rate of v_mfma_f32_32x32x16_f16 with random operands next to the bf16 one (same loop as probe 3);
// (b) are fp16 SUBNORMAL MFMA inputs honoured or flushed?  (c) which instruction a float2 -> half2 round-to-nearest conversion
// becomes is read off the ISA (llvm-objdump), not here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: bf16 random, 1: f16 random
__global__ __launch_bounds__(256) void probe(float* out, const unsigned* rnd, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const u32x4*>(rnd + ((threadIdx.x * 8 + i) * 4) % 65536);
    b[i] = *reinterpret_cast<const u32x4*>(rnd + ((threadIdx.x * 8 + i) * 4 + 32768) % 65536);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MODE == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m]), __builtin_bit_cast(bf16x8, b[(m + it) & 7]), acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m]), __builtin_bit_cast(f16x8, b[(m + it) & 7]), acc[m & 3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one wave: A[i][k] = a_val for k == 0 else 0, B[k][j] = b_val for k == 0 -> every C[i][j] = a_val * b_val
__global__ void subnormal_k(float* out, float a_val, float b_val) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void cvt_k(const float* in, unsigned* out, float s) {
  const int i = threadIdx.x;
  const float x0 = in[2 * i] * s, x1 = in[2 * i + 1] * s;
  const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  const f16x2 l = __builtin_convertvector(f32x2{r0, r1}, f16x2);
  out[2 * i] = __builtin_bit_cast(unsigned, h);
  out[2 * i + 1] = __builtin_bit_cast(unsigned, l);
}

template <int MODE>
void run(const char* name, float* out, unsigned* rnd) {
  const int iters = 40000, grid = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<MODE><<<grid, 256>>>(out, rnd, 1000);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    probe<MODE><<<grid, 256>>>(out, rnd, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 8;
    printf("%-34s rep %d: %8.3f ms  %8.1f TFLOP/s  %.2f ns/MFMA\n", name, rep, ms, mfmas * 2.0 * 32 * 32 * 16 / ms / 1e9, ms * 1e6 / (8.0 * iters));
  }
}

int main() {
  float* out; unsigned* rnd;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&rnd, 65536 * 4 + 64);
  unsigned* h = (unsigned*)malloc(65536 * 4);
  for (int i = 0; i < 65536; ++i) {       // two random 16-bit values of magnitude ~1 in either format (exponent field mid-range)
    unsigned lo = 0x3C00u | (rand() & 0x83FF), hi = 0x3C00u | (rand() & 0x83FF);
    h[i] = lo | (hi << 16);
  }
  (void)hipMemcpy(rnd, h, 65536 * 4, hipMemcpyHostToDevice);
  run<0>("bf16 32x32x16 random operands", out, rnd);
  run<1>("f16  32x32x16 random operands", out, rnd);
  run<0>("bf16 again", out, rnd);
  const float cases[][2] = {{1.f, 1.f}, {ldexpf(1.f, -15), 1024.f}, {ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 4096.f}, {ldexpf(3.f, -24), ldexpf(1.f, -14)}};
  for (auto& c : cases) {
    subnormal_k<<<1, 64>>>(out, c[0], c[1]);
    float r[3]; (void)hipMemcpy(r, out, 12, hipMemcpyDeviceToHost);
    printf("subnormal test: a=%g (as f16 %g) b=%g -> mfma %g   expected %g\n", c[0], r[1], c[1], r[0], (double)r[1] * r[2]);
  }
  return 0;
}
