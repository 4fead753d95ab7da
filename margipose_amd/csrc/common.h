// Shared device helpers for the gfx950 kernels.  Wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/margipose_hip.h"

namespace mpose {

constexpr int kWave = 64;

// Wave-wide reductions with DPP row operations (no LDS crossbar: __shfl_xor compiles to ds_bpermute_b32, ~60 cycles of latency per
// step in a dependent chain of six; the soft-argmax kernels are four such chains long).  quad_perm + row_half_mirror + row_mirror
// leave every 16-lane row holding its total; row_bcast:15 / row_bcast:31 carry the totals up to lane 63; v_readlane broadcasts.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(old), (int)__float_as_uint(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1, 0xf>(0.f, v);        // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E, 0xf>(0.f, v);        // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xf>(0.f, v);       // row_half_mirror
  v += dpp_f32<0x140, 0xf>(0.f, v);       // row_mirror
  v += dpp_f32<0x142, 0xa>(0.f, v);       // row_bcast:15 into rows 1 and 3
  v += dpp_f32<0x143, 0xc>(0.f, v);       // row_bcast:31 into rows 2 and 3
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = __uint_as_float(0xff800000u);
  v = fmaxf(v, dpp_f32<0xB1, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x4E, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x141, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x140, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f32<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_f32<0x143, 0xc>(ninf, v));
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

// Cell-centre coordinate i*(2/L) - (L-1)/L  (reference dsntnn.py:35-36).
__device__ __forceinline__ float cell_coord(int i, float two_over_l, float first) {
  return fmaf((float)i, two_over_l, first);
}

__device__ __forceinline__ void block_amax_commit_one(float m, float* dst);

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

// An activation tensor's "amax slot" is MPOSE_AMAX_SUBSLOTS floats, MPOSE_AMAX_STRIDE floats apart (one cache line each); the
// tensor's largest magnitude is the maximum over them.  Why: every workgroup of a producer pass ends with one atomic max, and
// same-address atomics serialise in L2 -- 2048 workgroups on ONE float tripled the duration of a 20 us elementwise pass, and
// looking before the atomic does not help while the first resident workgroups all still see zero.  Workgroup b uses sub-slot b % 16.
__device__ __forceinline__ float amax_gather(const float* slot) {          // wave-uniform result; call from whole waves
  const int l = threadIdx.x & 63;
  const float v = l < MPOSE_AMAX_SUBSLOTS ? slot[l * MPOSE_AMAX_STRIDE] : 0.f;
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(wave_max(v))));     // (the builtin is integer-typed)
}

// Largest magnitude seen by a 256-thread workgroup -> atomic max on the uint view of its sub-slot (non-negative floats order
// like their bit patterns).  Every thread of the workgroup must call it (it contains a barrier).
__device__ __forceinline__ void block_amax_commit(float m, float* slot) {
  float* dst = slot + (blockIdx.x % MPOSE_AMAX_SUBSLOTS) * MPOSE_AMAX_STRIDE;
  block_amax_commit_one(m, dst);
}
// (single address: the weight tensors' slots, 16 workgroups each)
__device__ __forceinline__ void block_amax_commit_one(float m, float* dst) {
  __shared__ float amax_sm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) amax_sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(amax_sm[0], amax_sm[1]), fmaxf(amax_sm[2], amax_sm[3]));
    if (!(m == m)) m = __uint_as_float(0x7f800000u);          // NaN anywhere -> +inf (ordered above everything)
    // look first: a workgroup that would not raise the value skips the atomic (a stale look costs one extra atomic)
    const unsigned seen = __hip_atomic_load(reinterpret_cast<unsigned*>(dst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(m) > seen) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(m));
  }
}

// Sortable key of a float: keys order like the floats (negative numbers included), and 0 is below the key of every float, so a
// zero-filled word is the identity of an atomic max over keys.  NaN -> the key of +inf (ordered above everything: the scale
// derived from it makes the convolution's result non-finite, like the NaN would).
__device__ __forceinline__ unsigned float_key(float x) {
  if (!(x == x)) x = __uint_as_float(0x7f800000u);
  const unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// One mpose_bn_finalize job: BatchNorm statistics -> (scale, shift, mean, invstd), the running-statistics update, and the exact
// largest relu(scale*x + shift) from the channel extremes.  Run by bn_finalize_k (bn.hip) and -- FRESH -- by the last workgroup
// of the convolution launch whose epilogues accumulated the statistics (conv.hip): those sums and extremes were written by other
// workgroups' atomics moments ago, so they are read at device scope.  All 256 threads of the workgroup call it.
template <bool FRESH>
__device__ __forceinline__ void bn_finalize_job(const mpose_bn_job& j, int train, float eps, float momentum) {
  if (j.eps > 0.f) eps = j.eps;
  const bool want_amax = train && j.minmax != nullptr && j.amax_out != nullptr;
  float amax = 0.f;
  auto ld_f64 = [](const double* p) -> double {
    if (!FRESH) return *p;
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  auto ld_u32 = [](const unsigned* p) -> unsigned {
    if (!FRESH) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int c = threadIdx.x; c < j.C; c += 256) {
    // The conv kernels are bias-free; a producing conv's bias b only shifts the BN input: batch/running mean
    // of (y + b) = mean(y) + b, and  scale*(y + b) + beta - (mean + b)*scale  ==  scale*y + beta - mean*scale.
    const double cb = (j.conv_bias != nullptr) ? (double)j.conv_bias[c] : 0.0;
    double mean, var;
    if (train) {
      const double n = (double)j.count;
      mean = ld_f64(j.stats + 2 * c) / n;
      var = ld_f64(j.stats + 2 * c + 1) / n - mean * mean;
      if (var < 0.0) var = 0.0;
      if (j.running_mean != nullptr) {
        const double unbiased = (j.count > 1) ? var * n / (n - 1.0) : var;
        j.running_mean[c] = (float)((1.0 - momentum) * (double)j.running_mean[c] + momentum * (mean + cb));
        j.running_var[c] = (float)((1.0 - momentum) * (double)j.running_var[c] + momentum * unbiased);
      }
    } else {
      mean = (double)j.running_mean[c] - cb;
      var = (double)j.running_var[c];
    }
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)j.gamma[c] * invstd;
    const float scf = (float)sc, shf = (float)((double)j.beta[c] - mean * sc);
    j.scale[c] = scf;
    j.shift[c] = shf;
    if (j.mean != nullptr) { j.mean[c] = (float)mean; j.invstd[c] = (float)invstd; }
    if (want_amax) {       // relu(scale * x + shift) is monotone in x: its largest value sits at one of the channel's two extremes
      const float vmax = key_float(ld_u32(j.minmax + 2 * c)), vmin = -key_float(ld_u32(j.minmax + 2 * c + 1));
      amax = fmaxf(amax, fmaxf(fmaf(vmax, scf, shf), fmaf(vmin, scf, shf)));       // (fmaxf drops the NaN of an untouched key)
    }
  }
  if (want_amax) block_amax_commit_one(amax, j.amax_out);       // (uniform: every thread of the workgroup gets here)
}

struct Ptr3 {
  const float* p[MPOSE_MAX_GROUP];
};
struct MutPtr3 {
  float* p[MPOSE_MAX_GROUP];
};

inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace mpose
