// Producers of PRE-SPLIT activations for conv_h.hip / wgrad.hip (gfx950): every elementwise pass that feeds a convolution of an H2
// block writes its result once as two fp16 planes of x * 2^k (h = rn16(x * 2^k), l = rn16(x * 2^k - h): 22 significant bits), k
// from a BOUND on the tensor's magnitude that exists before the pass runs, in the blocked layout
//       H8[C/8][plane 2][pixel][8]          16 bytes per (pixel, channel octet, plane): the bytes of the fp32 tensor
// -- the convolution and weight-gradient kernels then need no VALU work on their operands and fetch them in 1 KiB contiguous runs.
//
// Replaces, fused: the BN+ReLU between the two convolutions of a ResidualBlock (reference
// models/margipose_model.py:31-35), the block's output sum (:39) and the BatchNorm backward applications (autograd).
#include "common.h"

namespace mpose {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void load8(const float* __restrict__ src, long p, int C, int c0, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(src + p * C + c0);
  const float4 b = *reinterpret_cast<const float4*>(src + p * C + c0 + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* __restrict__ dst, long p, int C, int c0, const float (&v)[8]) {
  *reinterpret_cast<float4*>(dst + p * C + c0) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(dst + p * C + c0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load_vec8(const float* __restrict__ vec, int c0, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(vec + c0), b = *reinterpret_cast<const float4*>(vec + c0 + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// ---- producer-split fp16 planes (conv_h.hip): x * 2^k = h + l, layout H8[C/8][2][npix][8] ----
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h(const float x0, const float x1, unsigned& h, unsigned& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{x0, x1}, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = x0 - (float)hh[0], r1 = x1 - (float)hh[1];
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
// 8 channels of one pixel (already multiplied by the tensor's power-of-two scale) -> the two planes of channel octet c8
__device__ __forceinline__ void store_h2(void* planes, long npix, int c8, long p, const float (&v)[8]) {
  uint4 h, l;
  split2h(v[0], v[1], h.x, l.x);
  split2h(v[2], v[3], h.y, l.y);
  split2h(v[4], v[5], h.z, l.z);
  split2h(v[6], v[7], h.w, l.w);
  uint4* d = reinterpret_cast<uint4*>(planes) + (long)c8 * 2 * npix + p;
  d[0] = h; d[npix] = l;
}

struct SplitH2Args {
  mpose_split_h2_operands op[MPOSE_MAX_GROUP];
  long npix;
  int C, relu;
};

// planes = h2([relu](scale*x + shift) * 2^k)   (scale NULL: identity; k from the tensor's amax slot)
// (C % 32 == 0: whole-line reads + LDS turn, see tile_to_h2 below; otherwise a pixel per lane)
__global__ __launch_bounds__(256) void split_h2_k(SplitH2Args a) {
  const mpose_split_h2_operands& op = a.op[blockIdx.z];
  const int c8 = blockIdx.y * 4 + (threadIdx.x >> 6);
  const long p = (long)blockIdx.x * 64 + (threadIdx.x & 63);
  const float mul = pow2f(f16_scale_exp(amax_gather(op.amax)));
  if (c8 * 8 >= a.C || p >= a.npix) return;
  float v[8];
  load8(op.src, p, a.C, c8 * 8, v);
  if (op.scale != nullptr) {
    float sc[8], sh[8];
    load_vec8(op.scale, c8 * 8, sc);
    load_vec8(op.shift, c8 * 8, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
  }
  if (a.relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= mul;
  store_h2(op.planes, a.npix, c8, p, v);
}

constexpr int kTilePitch = 36;          // floats per pixel row of a staged 64-pixel x 32-channel tile
// The fused producers read and write fp32 NHWC in whole 128-byte lines -- thread (pixel t/8 + 32k, channels 4(t%8)..+3) of a 64-pixel x
// 32-channel workgroup tile: eight lanes per line -- and turn the tile through LDS into the planes' (lane = pixel, wave = channel
// octet) order, 1 KiB contiguous per plane and wave.  (Reading 32 bytes per lane, pixel per lane, costs a one-input pass 19 us
// and a three-input pass 83: every wave instruction touches 64 lines and uses a quarter of each.)
template <int NPX = 64>
__device__ __forceinline__ void tile_to_h2(float (*tile)[kTilePitch], void* planes, long npix, long p0, int c8_0, float mul) {
  __syncthreads();
#pragma unroll
  for (int h = 0; h < NPX / 64; ++h) {
    const int px = (threadIdx.x & 63) + 64 * h, oc = threadIdx.x >> 6;
    const long p = p0 + px;
    if (p < npix) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[px][oc * 8 + e] * mul;
      store_h2(planes, npix, c8_0 + oc, p, v);
    }
  }
}

__global__ __launch_bounds__(256) void split_h2_lines_k(SplitH2Args a) {
  __shared__ float tile[64][kTilePitch];
  const mpose_split_h2_operands& op = a.op[blockIdx.z];
  const float mul = pow2f(f16_scale_exp(amax_gather(op.amax)));
  const long p0 = (long)blockIdx.x * 64;
  const int cg = blockIdx.y * 32, c = cg + (threadIdx.x & 7) * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool affine = op.scale != nullptr;
  if (affine) { sc = *reinterpret_cast<const float4*>(op.scale + c); sh = *reinterpret_cast<const float4*>(op.shift + c); }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int px = (threadIdx.x >> 3) + 32 * k;
    const long p = p0 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < a.npix) {
      v = *reinterpret_cast<const float4*>(op.src + p * a.C + c);
      if (affine) { v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w); }
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    }
    *reinterpret_cast<float4*>(&tile[px][(threadIdx.x & 7) * 4]) = v;
  }
  tile_to_h2(tile, op.planes, a.npix, p0, cg >> 3, mul);
}

struct BnAddH2Args {
  mpose_bn_add_operands op[MPOSE_MAX_GROUP];
  void* h2[MPOSE_MAX_GROUP];
  long npix;
  int C;
};

// out = relu(a_scale*a + a_shift) + (b_scale*b + b_shift)  ->  fp32 NHWC (optional) + the two fp16 planes of out * 2^k, k from the
// BOUND in the tensor's amax slot (written by mpose_bn_finalize before this pass: mpose_bn_job.bound_out)
__global__ __launch_bounds__(256) void bn_add_h2_k(BnAddH2Args a) {
  __shared__ float tile[64][kTilePitch];
  const mpose_bn_add_operands& op = a.op[blockIdx.z];
  const float mul = pow2f(f16_scale_exp(amax_gather(op.out_amax)));
  const long p0 = (long)blockIdx.x * 64;
  const int cg = blockIdx.y * 32, c = cg + (threadIdx.x & 7) * 4;        // (C % 32 == 0: checked by the launcher)
  const float4 sa = *reinterpret_cast<const float4*>(op.a_scale + c), ta = *reinterpret_cast<const float4*>(op.a_shift + c);
  const float4 sb = *reinterpret_cast<const float4*>(op.b_scale + c), tb = *reinterpret_cast<const float4*>(op.b_shift + c);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int px = (threadIdx.x >> 3) + 32 * k;
    const long p = p0 + px;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < a.npix) {
      const float4 x = *reinterpret_cast<const float4*>(op.a + p * a.C + c), y = *reinterpret_cast<const float4*>(op.b + p * a.C + c);
      o.x = fmaxf(fmaf(x.x, sa.x, ta.x), 0.f) + fmaf(y.x, sb.x, tb.x); o.y = fmaxf(fmaf(x.y, sa.y, ta.y), 0.f) + fmaf(y.y, sb.y, tb.y);
      o.z = fmaxf(fmaf(x.z, sa.z, ta.z), 0.f) + fmaf(y.z, sb.z, tb.z); o.w = fmaxf(fmaf(x.w, sa.w, ta.w), 0.f) + fmaf(y.w, sb.w, tb.w);
      if (op.out != nullptr) *reinterpret_cast<float4*>(op.out + p * a.C + c) = o;
    }
    *reinterpret_cast<float4*>(&tile[px][(threadIdx.x & 7) * 4]) = o;
  }
  tile_to_h2(tile, a.h2[blockIdx.z], a.npix, p0, cg >> 3, mul);
}

struct BnApplyH2Args {
  mpose_bn_bwd_apply_operands op[MPOSE_MAX_GROUP];
  void* da_h2[MPOSE_MAX_GROUP];
  void* db_h2[MPOSE_MAX_GROUP];      // (NULL: db as fp32 only, its largest magnitude measured into db_amax)
  long npix;
  int C;
};

// da = k0a*[mask]g + k1a*(a - mean_a) + k2a  ->  fp32 (optional) + fp16 planes (scale from the bound in da_amax: mpose_bn_bwd_coef_job.bound_out);
// db = k0b*g + k1b*(b - mean_b) + k2b  ->  fp32 (optional), its largest magnitude accumulated into db_amax -- or, with db_h2 (round 6:
// the two-input data gradient and the weight gradients read planes too), fp16 planes scaled by the BOUND that db_amax already holds
__global__ __launch_bounds__(256) void bn_bwd_apply_h2_k(BnApplyH2Args a) {
  constexpr int NPX = 128, NK = NPX / 32;       // pixels per workgroup; pixel rows per thread (all their loads in flight together)
  __shared__ float tile[NPX][kTilePitch];
  const mpose_bn_bwd_apply_operands& op = a.op[blockIdx.z];
  const float mul = pow2f(f16_scale_exp(amax_gather(op.da_amax)));
  const long p0 = (long)blockIdx.x * NPX;
  const int cg = blockIdx.y * 32, c = cg + (threadIdx.x & 7) * 4;
  const bool has_b = op.b != nullptr, masked = op.a_scale != nullptr;
  void* const db_planes = a.db_h2[blockIdx.z];
  const float mul_b = (has_b && db_planes != nullptr) ? pow2f(f16_scale_exp(amax_gather(op.db_amax))) : 0.f;
  const float4 k0 = *reinterpret_cast<const float4*>(op.coef_a + c), k1 = *reinterpret_cast<const float4*>(op.coef_a + a.C + c);
  const float4 k2 = *reinterpret_cast<const float4*>(op.coef_a + 2 * a.C + c), mu = *reinterpret_cast<const float4*>(op.coef_a + 3 * a.C + c);
  float4 ms = make_float4(0.f, 0.f, 0.f, 0.f), mt = ms, q0 = ms, q1 = ms, q2 = ms, qm = ms;
  if (masked) { ms = *reinterpret_cast<const float4*>(op.a_scale + c); mt = *reinterpret_cast<const float4*>(op.a_shift + c); }
  if (has_b) {
    q0 = *reinterpret_cast<const float4*>(op.coef_b + c); q1 = *reinterpret_cast<const float4*>(op.coef_b + a.C + c);
    q2 = *reinterpret_cast<const float4*>(op.coef_b + 2 * a.C + c); qm = *reinterpret_cast<const float4*>(op.coef_b + 3 * a.C + c);
  }
  float4 g[NK], x[NK], y[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {               // (rows past the end re-read the last pixel: their results are not stored)
    long p = p0 + (threadIdx.x >> 3) + 32 * k;
    p = p < a.npix ? p : a.npix - 1;
    g[k] = *reinterpret_cast<const float4*>(op.g + p * a.C + c);
    x[k] = *reinterpret_cast<const float4*>(op.a + p * a.C + c);
    y[k] = has_b ? *reinterpret_cast<const float4*>(op.b + p * a.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float amax_b = 0.f;
  float4 dkeep[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int px = (threadIdx.x >> 3) + 32 * k;
    const long p = p0 + px;
    const bool live = p < a.npix;
    dkeep[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ga = g[k];
    if (masked) {
      if (!(fmaf(x[k].x, ms.x, mt.x) > 0.f)) ga.x = 0.f;
      if (!(fmaf(x[k].y, ms.y, mt.y) > 0.f)) ga.y = 0.f;
      if (!(fmaf(x[k].z, ms.z, mt.z) > 0.f)) ga.z = 0.f;
      if (!(fmaf(x[k].w, ms.w, mt.w) > 0.f)) ga.w = 0.f;
    }
    float4 o;
    o.x = fmaf(k1.x, x[k].x - mu.x, fmaf(k0.x, ga.x, k2.x)); o.y = fmaf(k1.y, x[k].y - mu.y, fmaf(k0.y, ga.y, k2.y));
    o.z = fmaf(k1.z, x[k].z - mu.z, fmaf(k0.z, ga.z, k2.z)); o.w = fmaf(k1.w, x[k].w - mu.w, fmaf(k0.w, ga.w, k2.w));
    if (live && op.da != nullptr) *reinterpret_cast<float4*>(op.da + p * a.C + c) = o;
    if (has_b) {
      float4 d;
      d.x = fmaf(q1.x, y[k].x - qm.x, fmaf(q0.x, g[k].x, q2.x)); d.y = fmaf(q1.y, y[k].y - qm.y, fmaf(q0.y, g[k].y, q2.y));
      d.z = fmaf(q1.z, y[k].z - qm.z, fmaf(q0.z, g[k].z, q2.z)); d.w = fmaf(q1.w, y[k].w - qm.w, fmaf(q0.w, g[k].w, q2.w));
      if (live) {
        if (op.db != nullptr) *reinterpret_cast<float4*>(op.db + p * a.C + c) = d;
        amax_b = fmaxf(fmaxf(amax_b, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
      }
      dkeep[k] = d;
    }
    *reinterpret_cast<float4*>(&tile[px][(threadIdx.x & 7) * 4]) = o;
  }
  tile_to_h2<NPX>(tile, a.da_h2[blockIdx.z], a.npix, p0, cg >> 3, mul);
  if (has_b && db_planes != nullptr) {      // (uniform per workgroup) the second gradient through the same tile
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NK; ++k) *reinterpret_cast<float4*>(&tile[(threadIdx.x >> 3) + 32 * k][(threadIdx.x & 7) * 4]) = dkeep[k];
    tile_to_h2<NPX>(tile, db_planes, a.npix, p0, cg >> 3, mul_b);
  } else if (has_b && op.db_amax != nullptr) {
    block_amax_commit(amax_b, op.db_amax);      // (uniform per workgroup)
  }
}

inline dim3 plane_grid(long npix, int C, int n_groups) { return dim3((unsigned)((npix + 63) / 64), (unsigned)((C + 31) / 32), (unsigned)n_groups); }

}  // namespace
}  // namespace mpose

using namespace mpose;



extern "C" int64_t mpose_h2_bytes(int64_t npix, int C) { return npix * (int64_t)((C + 7) / 8) * 2 * 16; }

extern "C" int mpose_split_h2(const mpose_split_h2_operands* ops, int n_groups, int64_t npix, int C, int relu, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 7) || npix < 0 || npix >= (1l << 31)) return MPOSE_EINVAL;
  if (npix == 0) return 0;
  SplitH2Args a{};
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    if (!ops[i].src || !ops[i].planes || !ops[i].amax || (ops[i].scale && !ops[i].shift)) return MPOSE_EINVAL;
  }
  a.npix = npix; a.C = C; a.relu = relu;
  if ((C & 31) == 0) launch(split_h2_lines_k, dim3(plane_grid(npix, C, n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  else launch(split_h2_k, dim3(plane_grid(npix, C, n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}


extern "C" int mpose_bn_add_h2(const mpose_bn_add_operands* ops, void* const* h2, int n_groups, int64_t npix, int C, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 31) || npix < 0 || npix >= (1l << 31)) return MPOSE_EINVAL;
  if (npix == 0) return 0;
  BnAddH2Args a{};
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    a.h2[i] = h2[i];
    if (!h2[i] || !ops[i].a || !ops[i].b || !ops[i].out_amax) return MPOSE_EINVAL;
  }
  a.npix = npix; a.C = C;
  launch(bn_add_h2_k, dim3(plane_grid(npix, C, n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_bn_bwd_apply_h2(const mpose_bn_bwd_apply_operands* ops, void* const* da_h2, void* const* db_h2, int n_groups,
                                     int64_t npix, int C, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 31) || npix < 0 || npix >= (1l << 31)) return MPOSE_EINVAL;
  if (npix == 0) return 0;
  BnApplyH2Args a{};
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    a.da_h2[i] = da_h2[i];
    a.db_h2[i] = db_h2 ? db_h2[i] : nullptr;
    if (!da_h2[i] || !ops[i].g || !ops[i].a || !ops[i].coef_a || !ops[i].da_amax) return MPOSE_EINVAL;
    if (ops[i].b && (!ops[i].coef_b || (!ops[i].db && !a.db_h2[i]))) return MPOSE_EINVAL;
    if (a.db_h2[i] && (!ops[i].b || !ops[i].db_amax)) return MPOSE_EINVAL;       // (planes of db: its bound is read from db_amax)
    if ((ops[i].b != nullptr) != (ops[0].b != nullptr) || (ops[i].db_amax != nullptr) != (ops[0].db_amax != nullptr) ||
        (a.db_h2[i] != nullptr) != (a.db_h2[0] != nullptr))
      return MPOSE_EINVAL;
  }
  a.npix = npix; a.C = C;
  launch(bn_bwd_apply_h2_k, dim3(dim3((unsigned)((npix + 127) / 128), (unsigned)(C / 32), (unsigned)n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

