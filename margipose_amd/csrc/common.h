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

struct Ptr3 {
  const float* p[MPOSE_MAX_GROUP];
};
struct MutPtr3 {
  float* p[MPOSE_MAX_GROUP];
};

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace mpose
