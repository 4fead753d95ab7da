// Shared device helpers for the gfx950 kernels.  Wavefront = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <tuple>
#include <type_traits>
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
// like their bit patterns).  Every thread of the workgroup (<= 1024 threads) must call it (it contains a barrier).
__device__ __forceinline__ void block_amax_commit(float m, float* slot) {
  float* dst = slot + (blockIdx.x % MPOSE_AMAX_SUBSLOTS) * MPOSE_AMAX_STRIDE;
  block_amax_commit_one(m, dst);
}
// (single address: the weight tensors' slots, 16 workgroups each)
__device__ __forceinline__ void block_amax_commit_one(float m, float* dst) {
  __shared__ float amax_sm[16];
  m = wave_max(m);
  __syncthreads();            // (two calls back to back: thread 0 may still be reading amax_sm of the first one)
  if ((threadIdx.x & 63) == 0) amax_sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = amax_sm[0];
    for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) m = fmaxf(m, amax_sm[w]);
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

// mpose_conv_stat_rows: while non-NULL the convolution launchers add their grid's x extent to *mpose_dry_rows and launch nothing.
extern thread_local int* mpose_dry_rows;
// MPOSE_CONV_STATS_PART buffers are self-describing: float words 0..3 are a header whose first word holds (as an int) the number
// of rows the last launch wrote; the rows start at word 4.  A launch that is one of several writing one buffer (the residue
// launches of an x-dilated kernel) takes its first row and the total from here; {0, 0} = rows 0 .. gridDim.x - 1.
struct PartPhase { int row0, total; };
extern thread_local PartPhase mpose_part_phase;
constexpr int kPartHdr = 4;

// Sum over the rows of an MPOSE_CONV_STATS_PART buffer, float [n_part][ld][NV]: channels c0 .. c0+nc-1 (nc <= blockDim.x), fixed
// order, fp64.  All threads of the workgroup call it; thread t < nc returns with channel c0+t's NV sums in out[].  sh: >= blockDim.x * NV doubles.
template <int NV, bool MAXIMUM = false>
__device__ __forceinline__ void reduce_part_rows(const float* __restrict__ part, int max_rows, int ld, int c0, int nc, double* sh, double (&out)[NV]) {
  const int nt = blockDim.x;
  int n_part = *reinterpret_cast<const int*>(part);      // rows written by the producing launch (header)
  n_part = n_part < 0 ? 0 : (n_part > max_rows ? max_rows : n_part);
  part += kPartHdr;
  const int cw = ((nc + 63) / 64) * 64;            // channel lanes per slice
  const int S = nt / cw > 0 ? nt / cw : 1;         // row slices
  const int t = threadIdx.x, cl = t % cw, sl = t / cw;
  double a[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) a[v] = MAXIMUM ? -__builtin_huge_val() : 0.0;
  if (sl < S && cl < nc) {
    const float* p = part + (size_t)(c0 + cl) * NV;
    int r = sl;
    for (; r + 7 * S < n_part; r += 8 * S) {         // eight rows in flight per thread
      float v[8][NV];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[u][k] = p[(size_t)(r + u * S) * ld * NV + k];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < NV; ++k) a[k] = MAXIMUM ? fmax(a[k], (double)v[u][k]) : a[k] + (double)v[u][k];
    }
    for (; r < n_part; r += S)
#pragma unroll
      for (int k = 0; k < NV; ++k) { const double x = (double)p[(size_t)r * ld * NV + k]; a[k] = MAXIMUM ? fmax(a[k], x) : a[k] + x; }
  }
  __syncthreads();                                   // (sh may still be read from an earlier call)
  if (sl < S && cl < nc) {
#pragma unroll
    for (int k = 0; k < NV; ++k) sh[(size_t)(sl * cw + cl) * NV + k] = a[k];
  }
  __syncthreads();
  if (t < nc) {
#pragma unroll
    for (int k = 0; k < NV; ++k) out[k] = MAXIMUM ? -__builtin_huge_val() : 0.0;
    for (int s_ = 0; s_ < S; ++s_)
#pragma unroll
      for (int k = 0; k < NV; ++k) { const double x = sh[(size_t)(s_ * cw + t) * NV + k]; out[k] = MAXIMUM ? fmax(out[k], x) : out[k] + x; }
  }
}

// reduce_part_rows<2> of `part` and reduce_part_rows<2, true> of `mm_part` (same launch: same row count) in ONE pass: the two
// buffers' loads are in flight together and the slices meet in LDS once -- same rows in the same order per channel, same results.
// sh: >= blockDim.x * 4 doubles.
__device__ __forceinline__ void reduce_part_rows_pair(const float* __restrict__ part, const float* __restrict__ mm_part, int max_rows, int ld,
                                                      int c0, int nc, double* sh, double (&sum)[2], double (&mx)[2]) {
  const int nt = blockDim.x;
  int n_part = *reinterpret_cast<const int*>(part);
  n_part = n_part < 0 ? 0 : (n_part > max_rows ? max_rows : n_part);
  int n_mm = *reinterpret_cast<const int*>(mm_part);
  n_mm = n_mm < 0 ? 0 : (n_mm > max_rows ? max_rows : n_mm);
  part += kPartHdr;
  mm_part += kPartHdr;
  const int cw = ((nc + 63) / 64) * 64;
  const int S = nt / cw > 0 ? nt / cw : 1;
  const int t = threadIdx.x, cl = t % cw, sl = t / cw;
  double a[2] = {0.0, 0.0}, m[2] = {-__builtin_huge_val(), -__builtin_huge_val()};
  if (sl < S && cl < nc) {
    const float2* p = reinterpret_cast<const float2*>(part + (size_t)(c0 + cl) * 2);
    const float2* q = reinterpret_cast<const float2*>(mm_part + (size_t)(c0 + cl) * 2);
    const int n_both = n_part < n_mm ? n_part : n_mm;
    int r = sl;
    for (; r + 7 * S < n_both; r += 8 * S) {         // eight rows of each buffer in flight per thread
      float2 v[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { v[u] = p[(size_t)(r + u * S) * ld]; w[u] = q[(size_t)(r + u * S) * ld]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[0] += (double)v[u].x; a[1] += (double)v[u].y;
        m[0] = fmax(m[0], (double)w[u].x); m[1] = fmax(m[1], (double)w[u].y);
      }
    }
    for (int r2 = r; r2 < n_part; r2 += S) { const float2 v = p[(size_t)r2 * ld]; a[0] += (double)v.x; a[1] += (double)v.y; }
    for (int r2 = r; r2 < n_mm; r2 += S) { const float2 w = q[(size_t)r2 * ld]; m[0] = fmax(m[0], (double)w.x); m[1] = fmax(m[1], (double)w.y); }
  }
  __syncthreads();                                   // (sh may still be read from an earlier call)
  if (sl < S && cl < nc) {
    double* d = sh + (size_t)(sl * cw + cl) * 4;
    d[0] = a[0]; d[1] = a[1]; d[2] = m[0]; d[3] = m[1];
  }
  __syncthreads();
  if (t < nc) {
    sum[0] = 0.0; sum[1] = 0.0; mx[0] = -__builtin_huge_val(); mx[1] = -__builtin_huge_val();
    for (int s_ = 0; s_ < S; ++s_) {
      const double* d = sh + (size_t)(s_ * cw + t) * 4;
      sum[0] += d[0]; sum[1] += d[1]; mx[0] = fmax(mx[0], d[2]); mx[1] = fmax(mx[1], d[3]);
    }
  }
}

// One mpose_bn_finalize job: BatchNorm statistics -> (scale, shift, mean, invstd), the running-statistics update, and the exact
// largest relu(scale*x + shift) from the channel extremes.  Run by bn_finalize_k (bn.hip) and -- FRESH -- by the last workgroup
// of the convolution launch whose epilogues accumulated the statistics (conv.hip): those sums and extremes were written by other
// workgroups' atomics moments ago, so they are read at device scope.  All 256 threads of the workgroup call it.
template <bool FRESH>
__device__ __forceinline__ void bn_finalize_job(const mpose_bn_job& j, int train, float eps, float momentum, double* sh = nullptr,
                                                int part_i = 0, int n_parts = 1, bool bounds = false) {
  // (n_parts > 1: this workgroup takes the part_i-th share of the job's channels, 32-channel granules)
  const int gran = ((j.C + 31) / 32 + n_parts - 1) / n_parts * 32;
  const int c_lo = part_i * gran, c_hi = (c_lo + gran < j.C) ? c_lo + gran : j.C;
  if (j.eps > 0.f) eps = j.eps;
  const bool from_part = !FRESH && train && sh != nullptr && j.part != nullptr;        // MPOSE_CONV_STATS_PART (sh: blockDim.x * 2 doubles)
  const bool want_amax = train && (j.minmax != nullptr || (from_part && j.mm_part != nullptr)) && j.amax_out != nullptr;
  const bool want_bound = train && bounds && j.bound_out != nullptr;
  float amax = 0.f, bound = 0.f;
  const float root_n = sqrtf((float)j.count) * 1.0001f;
  auto ld_f64 = [](const double* p) -> double {
    if (!FRESH) return *p;
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  auto ld_u32 = [](const unsigned* p) -> unsigned {
    if (!FRESH) return *p;
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const int nth = (int)blockDim.x;
  for (int cblk = c_lo; cblk < c_hi; cblk += nth) {
    const int c = cblk + (int)threadIdx.x;
    double psum[2] = {0.0, 0.0}, pmm[2] = {0.0, 0.0};
    // what the channel's result needs besides the statistics, requested BEFORE the row sums are (the launch is a chain of dependent
    // memory round trips on an otherwise idle chip, ten of them per block of the forward pass: every one that overlaps counts)
    const bool cv = c < c_hi;
    const float gamma_c = cv ? j.gamma[c] : 0.f, beta_c = cv ? j.beta[c] : 0.f;
    const float rmean_c = (cv && j.running_mean != nullptr) ? j.running_mean[c] : 0.f, rvar_c = (cv && j.running_mean != nullptr) ? j.running_var[c] : 0.f;
    const float cbias_c = (cv && j.conv_bias != nullptr) ? j.conv_bias[c] : 0.f;
    if (from_part) {       // (uniform per workgroup: the helper contains barriers)
      const int nc = c_hi - cblk < nth ? c_hi - cblk : nth;
      if (want_amax && j.mm_part != nullptr) reduce_part_rows_pair(j.part, j.mm_part, j.n_part, j.part_ld, cblk, nc, sh, psum, pmm);
      else reduce_part_rows<2>(j.part, j.n_part, j.part_ld, cblk, nc, sh, psum);
    }
    if (c >= c_hi) continue;
    // The conv kernels are bias-free; a producing conv's bias b only shifts the BN input: batch/running mean
    // of (y + b) = mean(y) + b, and  scale*(y + b) + beta - (mean + b)*scale  ==  scale*y + beta - mean*scale.
    const double cb = (double)cbias_c;
    double mean, var;
    if (train) {
      const double n = (double)j.count;
      mean = (from_part ? psum[0] : ld_f64(j.stats + 2 * c)) / n;
      var = (from_part ? psum[1] : ld_f64(j.stats + 2 * c + 1)) / n - mean * mean;
      if (var < 0.0) var = 0.0;
      if (j.running_mean != nullptr) {
        const double unbiased = (j.count > 1) ? var * n / (n - 1.0) : var;
        j.running_mean[c] = (float)((1.0 - momentum) * (double)rmean_c + momentum * (mean + cb));
        j.running_var[c] = (float)((1.0 - momentum) * (double)rvar_c + momentum * unbiased);
      }
    } else {
      mean = (double)rmean_c - cb;
      var = (double)rvar_c;
    }
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma_c * invstd;
    const float scf = (float)sc, shf = (float)((double)beta_c - mean * sc);
    j.scale[c] = scf;
    j.shift[c] = shf;
    if (j.mean != nullptr) { j.mean[c] = (float)mean; j.invstd[c] = (float)invstd; }
    if (want_bound) {      // |gamma * xhat + beta| <= |gamma| sqrt(n) + |beta|  (+ the partner BatchNorm's: the residual sum)
      float b_ = fabsf(gamma_c) * root_n + fabsf(beta_c);
      if (j.bound_gamma2 != nullptr) b_ += fabsf(j.bound_gamma2[c]) * root_n + fabsf(j.bound_beta2[c]);
      bound = fmaxf(bound, b_);
    }
    if (want_amax) {       // relu(scale * x + shift) is monotone in x: its largest value sits at one of the channel's two extremes
      const bool pm = from_part && j.mm_part != nullptr;
      const float vmax = pm ? (float)pmm[0] : key_float(ld_u32(j.minmax + 2 * c));
      const float vmin = pm ? -(float)pmm[1] : -key_float(ld_u32(j.minmax + 2 * c + 1));
      amax = fmaxf(amax, fmaxf(fmaf(vmax, scf, shf), fmaf(vmin, scf, shf)));       // (fmaxf drops the NaN of an untouched key / -inf * 0)
    }
  }
  if (want_bound) block_amax_commit_one(bound * 1.0001f, j.bound_out);       // (uniform; contains barriers)
  if (want_amax) {       // (uniform: every thread of the workgroup gets here)
    block_amax_commit_one(amax, j.amax_out);      // (accumulated: the job's channel range may be split over several workgroups)
  }
}

struct Ptr3 {
  const float* p[MPOSE_MAX_GROUP];
};
struct MutPtr3 {
  float* p[MPOSE_MAX_GROUP];
};

inline int launch_status() { return (int)hipGetLastError(); }

// ---- every kernel launch of the library goes through launch(): it is the point where a LAUNCH PLAN records (csrc/plan.hip:
// mpose_plan_begin ... mpose_plan_end keeps, per launch, the kernel, the geometry, a copy of the argument values and which of the
// plan's streams it went to; mpose_plan_replay re-issues the list from one C loop -- a training iteration's ~700 launches for
// 2-3 ms of host time instead of 15-28 ms of Python, on the same two streams with the same dependencies as the eager schedule).
// Recording is process-wide, not per thread: autograd runs the backward pass's launches on its own device thread.
struct Plan;
extern std::atomic<Plan*> g_plan_rec;
void plan_record_launch(Plan* plan, const void* fn, dim3 grid, dim3 block, unsigned lds, hipStream_t stream, void* const* argv,
                        const unsigned* sizes, int n_args);

template <class... P, class... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream, A&&... args) {
  static_assert(sizeof...(P) == sizeof...(A), "launch(): argument count differs from the kernel's parameter list");
  std::tuple<std::remove_cv_t<std::remove_reference_t<P>>...> params{static_cast<P>(args)...};      // the values exactly as the kernel takes them
  std::apply([&](auto&... e) {
    void* argv[] = {static_cast<void*>(&e)...};
    Plan* rec = g_plan_rec.load(std::memory_order_acquire);
    if (rec != nullptr) {
      const unsigned sizes[] = {(unsigned)sizeof(e)...};
      plan_record_launch(rec, reinterpret_cast<const void*>(kernel), grid, block, (unsigned)lds, stream, argv, sizes, (int)sizeof...(P));
    }
    (void)hipLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, argv, lds, stream);      // (errors: launch_status())
  }, params);
}

}  // namespace mpose
