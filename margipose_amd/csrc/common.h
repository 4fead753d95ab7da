// Shared device helpers for the gfx950 kernels.  Wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/margipose_hip.h"

namespace mpose {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Cell-centre coordinate i*(2/L) - (L-1)/L  (reference dsntnn.py:35-36).
__device__ __forceinline__ float cell_coord(int i, float two_over_l, float first) {
  return fmaf((float)i, two_over_l, first);
}

// MPOSE_CONV_F16X3 (include/margipose_hip.h): exponent k of the power-of-two scale that puts a tensor whose largest magnitude is
// `amax` into fp16's upper range, amax * 2^k in [2^14, 2^15).  The packer, both convolution kernels and the weight-gradient
// kernel all derive k from the same float with this function.
__host__ __device__ __forceinline__ int f16_scale_exp(float amax) {
  unsigned bits;
  __builtin_memcpy(&bits, &amax, 4);
  int e = (int)((bits >> 23) & 0xffu);
  e = e < 27 ? 27 : (e > 254 ? 254 : e);        // amax < 2^-100 (all-zero tensors): 2^114; inf / NaN: 2^-113
  return 141 - e;
}
__host__ __device__ __forceinline__ float pow2f(int k) {       // 2^k, -126 <= k <= 127
  const unsigned bits = (unsigned)(127 + k) << 23;
  float f;
  __builtin_memcpy(&f, &bits, 4);
  return f;
}

struct Ptr3 {
  const float* p[MPOSE_MAX_GROUP];
};
struct MutPtr3 {
  float* p[MPOSE_MAX_GROUP];
};

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace mpose
