// Soft-argmax tail for gfx950: flat_softmax + DSNT + JS/Euclidean stage loss, forward and backward.
//
// Replaces (reference src/margipose): dsntnn.py:12-62,84-232 and
// models/margipose_model.py:215-261.  HBM-bound: every kernel reads each heatmap element exactly
// once and writes each result element exactly once; all reductions are wave64 shuffles.
//
// Work decomposition: one workgroup per heatmap row (= one (batch, joint) pair), one 64-lane
// wavefront per plane (xy / zy / xz).  A row of n = H*W <= 4096 fp32 values lives in registers as
// NV float4 per lane (NV = 4 for 32x32, 9 for 48x48, 16 for 64x64), loaded with 16-byte coalesced
// accesses (1 KiB per wave instruction).
#include <stdlib.h>
#include "common.h"

namespace mpose {
namespace {

constexpr float kEps = 1e-24f;   // dsntnn.py:194,199

struct RowGeom {
  int H, W, n4;            // n4 = H*W/4
  float two_over_w, first_w, two_over_h, first_h;
};

__device__ __forceinline__ RowGeom make_geom(int H, int W) {
  RowGeom g;
  g.H = H; g.W = W; g.n4 = (H * W) >> 2;
  g.two_over_w = 2.0f / (float)W; g.first_w = -((float)W - 1.0f) / (float)W;
  g.two_over_h = 2.0f / (float)H; g.first_h = -((float)H - 1.0f) / (float)H;
  return g;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// NT: the access carries the non-temporal hint (`nt`: streamed once, do not keep the line) -- the soft-argmax reads every logit and
// writes every heatmap element exactly once, and at sizes beyond the Infinity Cache the lines it would otherwise leave behind only
// evict what is still to be read.
template <int NV, bool NT = false>
__device__ __forceinline__ void load_row(const float* __restrict__ src, int lane, int n4, float4 (&v)[NV], float fill) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < n4) {
      if (NT) {
        const f32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src) + idx);
        v[i] = make_float4(r.x, r.y, r.z, r.w);
      } else {
        v[i] = reinterpret_cast<const float4*>(src)[idx];
      }
    } else {
      v[i] = make_float4(fill, fill, fill, fill);
    }
  }
}

template <int NV, bool NT = false>
__device__ __forceinline__ void store_row(float* __restrict__ dst, int lane, int n4, const float4 (&v)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < n4) {
      if (NT) {
        const f32x4_t r = {v[i].x, v[i].y, v[i].z, v[i].w};
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4_t*>(dst) + idx);
      } else {
        reinterpret_cast<float4*>(dst)[idx] = v[i];
      }
    }
  }
}

__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {   // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <int NV, bool NT = false>
__device__ __forceinline__ void load_row_bf16(const unsigned short* __restrict__ src, int lane, int n4, float4 (&v)[NV], float fill) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < n4) {
      uint2 r;
      if (NT) {
        const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(src) + idx);
        r.x = t.x; r.y = t.y;
      } else {
        r = reinterpret_cast<const uint2*>(src)[idx];
      }
      v[i] = make_float4(bf16_to_f32(r.x & 0xffff), bf16_to_f32(r.x >> 16), bf16_to_f32(r.y & 0xffff), bf16_to_f32(r.y >> 16));
    } else {
      v[i] = make_float4(fill, fill, fill, fill);
    }
  }
}

template <int NV, bool NT = false>
__device__ __forceinline__ void store_row_bf16(unsigned short* __restrict__ dst, int lane, int n4, const float4 (&v)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = i * 64 + lane;
    if (idx < n4) {
      u32x2_t r;
      r.x = (unsigned)f32_to_bf16(v[i].x) | ((unsigned)f32_to_bf16(v[i].y) << 16);
      r.y = (unsigned)f32_to_bf16(v[i].z) | ((unsigned)f32_to_bf16(v[i].w) << 16);
      if (NT) __builtin_nontemporal_store(r, reinterpret_cast<u32x2_t*>(dst) + idx);
      else reinterpret_cast<u32x2_t*>(dst)[idx] = r;
    }
  }
}

// Coordinates of the 4 elements held in v[i] by `lane`: same row h, columns w0..w0+3 (W % 4 == 0).
__device__ __forceinline__ void elem_hw(const RowGeom& g, int i, int lane, int& h, int& w0) {
  const int e0 = (i * 64 + lane) * 4;
  h = e0 / g.W;
  w0 = e0 - h * g.W;
}

// The row arithmetic of flat_softmax (dsntnn.py:124-130) and dsnt (dsntnn.py:84-96), shared by softmax_dsnt_fwd_k and
// bn_add_softmax_k.  This FILE is compiled with -ffp-contract=on (margipose_amd/build.py): under hipcc's default, `fast`, the
// backend fuses any multiply with any add it meets, whatever the source says -- __fmul_rn / __fadd_rn are plain operators to it -- and
// it turned `x * rs + y * rs` into fma(x, rs, y * rs) in one of the two kernels and not in the other (round 5: the last bit of 15 % of
// the coordinates differed; a `#pragma clang fp contract(off)` does not reach the backend's fusion).  With `on` only a source
// expression `a * b + c` is fused (and the explicit fmaf calls): the two kernels run the same roundings, heatmaps and coordinates are
// bit-identical (tests/test_tail_gpu.py::test_fused_residual_sum_softmax_is_bit_identical).
template <int NV>
__device__ __forceinline__ void row_softmax(float4 (&v)[NV]) {
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
  m = wave_max(m);
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x = expf(__fsub_rn(v[i].x, m)); v[i].y = expf(__fsub_rn(v[i].y, m)); v[i].z = expf(__fsub_rn(v[i].z, m)); v[i].w = expf(__fsub_rn(v[i].w, m));
    s = __fadd_rn(s, __fadd_rn(__fadd_rn(v[i].x, v[i].y), __fadd_rn(v[i].z, v[i].w)));
  }
  s = wave_sum(s);
  const float rs = __fdiv_rn(1.0f, s);
#pragma unroll
  for (int i = 0; i < NV; ++i) { v[i].x = __fmul_rn(v[i].x, rs); v[i].y = __fmul_rn(v[i].y, rs); v[i].z = __fmul_rn(v[i].z, rs); v[i].w = __fmul_rn(v[i].w, rs); }
}

template <int NV>
__device__ __forceinline__ void row_expectation(const float4 (&v)[NV], const RowGeom& g, int lane, float& sx_out, float& sy_out) {
  float sx = 0.0f, sy = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int h, w0;
    elem_hw(g, i, lane, h, w0);
    const float y = cell_coord(h, g.two_over_h, g.first_h);
    const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
    const float x1 = __fadd_rn(x0, g.two_over_w), x2 = __fadd_rn(x0, __fmul_rn(2.0f, g.two_over_w)), x3 = __fadd_rn(x0, __fmul_rn(3.0f, g.two_over_w));
    const float rs = __fadd_rn(__fadd_rn(v[i].x, v[i].y), __fadd_rn(v[i].z, v[i].w));
    sy = fmaf(rs, y, sy);
    sx = __fadd_rn(sx, fmaf(v[i].w, x3, fmaf(v[i].z, x2, fmaf(v[i].y, x1, __fmul_rn(v[i].x, x0)))));
  }
  sx_out = wave_sum(sx);
  sy_out = wave_sum(sy);
}

// ---------------------------------------------------------------------------------------------
// flat_softmax + dsnt (+ heatmaps_to_coords)
// ---------------------------------------------------------------------------------------------
struct SoftmaxArgs {
  const void* logits[MPOSE_MAX_GROUP];
  void* heatmaps[MPOSE_MAX_GROUP];
  float* plane_coords;
  float* xyz;
  int n_planes, rows, H, W;
};

// RPW rows per workgroup (the loads of ALL of them are issued before the first row's arithmetic starts: 4 KB per wave and row in
// flight, half the workgroup dispatches per byte); NT bit 0: non-temporal heatmap stores, bit 1: non-temporal logit loads.  The row
// arithmetic is the same instruction sequence for every (RPW, NT): heatmaps and coordinates are bit-identical across variants.
template <int NV, bool BF_IN, bool BF_OUT, bool EXP, int RPW = 1, int NT = 0>
__global__ __launch_bounds__(64 * MPOSE_MAX_GROUP) void softmax_dsnt_fwd_k(SoftmaxArgs a) {
  __shared__ float s_mu[RPW][MPOSE_MAX_GROUP][2];
  const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const RowGeom g = make_geom(a.H, a.W);
  constexpr bool NT_ST = (NT & 1) != 0, NT_LD = (NT & 2) != 0;

  float4 v[RPW][NV];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int row = blockIdx.x * RPW + k;
    const size_t off = (size_t)(row < a.rows ? row : 0) * (size_t)(a.H * a.W);
    if (BF_IN) load_row_bf16<NV, NT_LD>(reinterpret_cast<const unsigned short*>(a.logits[plane]) + off, lane, g.n4, v[k], EXP ? -INFINITY : 0.0f);
    else load_row<NV, NT_LD>(reinterpret_cast<const float*>(a.logits[plane]) + off, lane, g.n4, v[k], EXP ? -INFINITY : 0.0f);
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int row = blockIdx.x * RPW + k;
    const bool live = row < a.rows;                 // (wave-uniform; a dead row recomputes row 0 and stores nothing)
    const size_t off = (size_t)(live ? row : 0) * (size_t)(a.H * a.W);
    if (EXP) row_softmax<NV>(v[k]);
    if (EXP && a.heatmaps[plane] != nullptr && live) {      // (stored before the expectation sums: the stores drain under them)
      if (BF_OUT) store_row_bf16<NV, NT_ST>(reinterpret_cast<unsigned short*>(a.heatmaps[plane]) + off, lane, g.n4, v[k]);
      else store_row<NV, NT_ST>(reinterpret_cast<float*>(a.heatmaps[plane]) + off, lane, g.n4, v[k]);
    }

    float sx, sy;
    row_expectation<NV>(v[k], g, lane, sx, sy);
    if (lane == 0) {
      s_mu[k][plane][0] = sx; s_mu[k][plane][1] = sy;
      if (a.plane_coords != nullptr && live) {
        float* pc = a.plane_coords + ((size_t)plane * a.rows + row) * 2;
        pc[0] = sx; pc[1] = sy;
      }
    }
  }
  if (a.n_planes == 3 && a.xyz != nullptr) {
    __syncthreads();
    if (threadIdx.x < RPW && (int)(blockIdx.x * RPW + threadIdx.x) < a.rows) {
      const int k = threadIdx.x;
      float* o = a.xyz + (size_t)(blockIdx.x * RPW + k) * 3;
      o[0] = s_mu[k][0][0];
      o[1] = s_mu[k][0][1];
      o[2] = 0.5f * (s_mu[k][1][0] + s_mu[k][2][1]);     // models/margipose_model.py:259
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The last ResidualBlock's residual sum fused with flat_softmax + dsnt (round 4): the column's logits never reach memory.
//   logits[b][j][p] = relu(a_scale*a + a_shift)[b,p,j] + (b_scale*b + b_shift)[b,p,j]     (models/margipose_model.py:34-40, NHWC in)
//   heatmaps = flat_softmax(logits) (dsntnn.py:124-130), plane coordinates = dsnt(heatmaps) (dsntnn.py:84-96)
// One workgroup per (image, column, float4 of joint channels): the four joints' H*W logits live in LDS (16 KB at 32 x 32); phase 1
// reads that float4 of every pixel of the two NHWC tensors (every load of a thread in flight at once) and turns it into 4 rows,
// phase 2 is softmax_dsnt_fwd_k's row arithmetic, instruction for instruction (bit-identical heatmaps), a wave per joint.
// Replaces bn_add_nchw_k + softmax_dsnt_fwd_k: 13.4 MB of logits written and re-read per stage at B = 32, one launch instead of
// two.  (A first form with ONE workgroup per image -- all 17 rows in 68 KB of LDS -- measured 18.4 us against the two launches'
// 14.8: 96 workgroups leave 160 CUs idle.)
// ---------------------------------------------------------------------------------------------
struct BnAddSoftmaxArgs {
  mpose_bn_add_operands op[MPOSE_MAX_GROUP];
  void* heat[MPOSE_MAX_GROUP];
  float* plane_coords;      // (n_groups, B*J, 2)
  int B, P, C, J, H, W;
};

constexpr int kBasThreads = 256, kBasPad = 4, kBasU = 4;     // row pitch P + 4 floats: float4-aligned rows, the turn's stores spread over banks

// ALLJ (round 5): one workgroup per (image, column) with ALL J rows in LDS (70 KB at 32 x 32, 17 joints) -- it reads the first
// ceil(J/4) float4s of every pixel's channel line of both tensors, i.e. every line ONCE, where the four-joint form reads each line
// from five workgroups (fine while the inputs stay in the Infinity Cache; 8x over-fetch from HBM beyond it: 0.10 of the HBM rate at
// B = 2048).  The launcher picks it when the inputs exceed the cache; below, its 96 workgroups would leave CUs idle (18.4 us
// against the four-joint form's 9.7 at B = 32).  Same expressions, same row arithmetic: bit-identical heatmaps.
template <int NV, bool BF_OUT, bool ALLJ>
__global__ __launch_bounds__(kBasThreads) void bn_add_softmax_k(BnAddSoftmaxArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tile[];          // [4][P + kBasPad]: the workgroup's four joints ([J][..]: ALLJ)
  const mpose_bn_add_operands& op = a.op[blockIdx.y];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pitch = a.P + kBasPad;
  if constexpr (ALLJ) {
    // Lanes walk (pixel, float4 of the channel line): eight consecutive lanes share a 128-byte line, the ones whose float4 lies past
    // the J-th channel load nothing -- a wave instruction touches 8 lines.  (One float4 per lane at a 128-byte stride touches 64
    // lines per instruction: that form reached 2 TB/s of line traffic.)  a.C == 32 (the launcher checks); a thread's float4 column
    // is the same in every pass (256 % 8 == 0): its coefficients are loaded once.
    const size_t img0 = (size_t)b * a.P * a.C;
    const int chunk = tid & 7, c0 = chunk * 4;
    const bool live = c0 < a.J;
    float sa[4], ta[4], sb[4], tb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cc = min(c0 + e, a.J - 1);
      sa[e] = op.a_scale[cc]; ta[e] = op.a_shift[cc]; sb[e] = op.b_scale[cc]; tb[e] = op.b_shift[cc];
    }
    constexpr int U = 8;
    const int n_items = a.P * 8;
    for (int base = 0; base < n_items; base += kBasThreads * U) {
      float4 x[U], y[U];              // every load of the pass in flight before the first use
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * kBasThreads + tid;
        if (live && i < n_items) {
          x[u] = *reinterpret_cast<const float4*>(op.a + img0 + (size_t)i * 4);
          y[u] = *reinterpret_cast<const float4*>(op.b + img0 + (size_t)i * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * kBasThreads + tid;
        if (live && i < n_items) {
          const int px = i >> 3;
          const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ys[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < a.J) tile[(c0 + e) * pitch + px] = fmaxf(fmaf(xs[e], sa[e], ta[e]), 0.f) + fmaf(ys[e], sb[e], tb[e]);
        }
      }
    }
    __syncthreads();
    const RowGeom g = make_geom(a.H, a.W);
    for (int j = wave; j < a.J; j += kBasThreads / 64) {      // a wave per joint, four joints at a time
      const float* row = tile + j * pitch;
      float4 v[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = i * 64 + lane;
        v[i] = (idx < g.n4) ? reinterpret_cast<const float4*>(row)[idx] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
      row_softmax<NV>(v);
      float sx, sy;
      row_expectation<NV>(v, g, lane, sx, sy);
      const size_t r = (size_t)b * a.J + j;
      if (BF_OUT) store_row_bf16<NV>(reinterpret_cast<unsigned short*>(a.heat[blockIdx.y]) + r * a.P, lane, g.n4, v);
      else store_row<NV>(reinterpret_cast<float*>(a.heat[blockIdx.y]) + r * a.P, lane, g.n4, v);
      if (lane == 0 && a.plane_coords != nullptr) {
        float* pc = a.plane_coords + ((size_t)blockIdx.y * a.B * a.J + r) * 2;
        pc[0] = sx; pc[1] = sy;
      }
    }
    return;
  }
  const int c0 = blockIdx.z * 4;                                         // the float4 column of channels c0..c0+3
  const size_t img = (size_t)b * a.P * a.C + c0;
  float sa[4], ta[4], sb[4], tb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cc = min(c0 + e, a.J - 1);
    sa[e] = op.a_scale[cc]; ta[e] = op.a_shift[cc]; sb[e] = op.b_scale[cc]; tb[e] = op.b_shift[cc];
  }
  // all of a thread's loads first (kBasU float4 pairs in flight per thread: 4 cover a 32 x 32 image in one pass), then the turn
  for (int base = 0; base < a.P; base += kBasThreads * kBasU) {
    float4 x[kBasU], y[kBasU];
#pragma unroll
    for (int u = 0; u < kBasU; ++u) {
      const int px = base + u * kBasThreads + tid;
      if (px < a.P) {
        x[u] = *reinterpret_cast<const float4*>(op.a + img + (size_t)px * a.C);
        y[u] = *reinterpret_cast<const float4*>(op.b + img + (size_t)px * a.C);
      }
    }
#pragma unroll
    for (int u = 0; u < kBasU; ++u) {
      const int px = base + u * kBasThreads + tid;
      if (px < a.P) {
        const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ys[4] = {y[u].x, y[u].y, y[u].z, y[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)      // (bn_add_nchw_k's expression; rows past J are computed from joint J-1's coefficients and never read)
          tile[e * pitch + px] = fmaxf(fmaf(xs[e], sa[e], ta[e]), 0.f) + fmaf(ys[e], sb[e], tb[e]);
      }
    }
  }
  __syncthreads();
  const RowGeom g = make_geom(a.H, a.W);
  {
    const int j = c0 + wave;                 // a wave per joint
    if (j >= a.J) return;
    const float* row = tile + wave * pitch;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = i * 64 + lane;
      v[i] = (idx < g.n4) ? reinterpret_cast<const float4*>(row)[idx] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    row_softmax<NV>(v);
    float sx, sy;
    row_expectation<NV>(v, g, lane, sx, sy);
    const size_t r = (size_t)b * a.J + j;
    if (BF_OUT) store_row_bf16<NV>(reinterpret_cast<unsigned short*>(a.heat[blockIdx.y]) + r * a.P, lane, g.n4, v);
    else store_row<NV>(reinterpret_cast<float*>(a.heat[blockIdx.y]) + r * a.P, lane, g.n4, v);
    if (lane == 0 && a.plane_coords != nullptr) {
      float* pc = a.plane_coords + ((size_t)blockIdx.y * a.B * a.J + r) * 2;
      pc[0] = sx; pc[1] = sy;
    }
  }
}

constexpr int kBasAllJLds = 144 * 1024;      // LDS the all-joints form may take (a workgroup per CU then)
template <int NV, bool BF_OUT>
static int launch_bn_add_softmax(const BnAddSoftmaxArgs& a, int n_groups, int lds, bool allj, hipStream_t s) {
  static bool raised = false, raised_all = false;            // (per instantiation; the attribute is per function and sticky)
  if (allj) {
    const int lds_all = a.J * (a.P + kBasPad) * 4;
    if (!raised_all) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_add_softmax_k<NV, BF_OUT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kBasAllJLds) != hipSuccess)
        return MPOSE_EINVAL;
      raised_all = true;
    }
    launch(bn_add_softmax_k<NV, BF_OUT, true>, dim3(dim3(a.B, n_groups, 1)), dim3(kBasThreads), lds_all, s, a);
    return 0;
  }
  if (lds > 48 * 1024 && !raised) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_add_softmax_k<NV, BF_OUT, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            72 * 1024) != hipSuccess)
      return MPOSE_EINVAL;
    raised = true;
  }
  launch(bn_add_softmax_k<NV, BF_OUT, false>, dim3(dim3(a.B, n_groups, (a.J + 3) / 4)), dim3(kBasThreads), lds, s, a);
  return 0;
}

// MargiPoseModel.heatmaps_to_coords' merge (models/margipose_model.py:254-261) from the three planes' coordinates
__global__ __launch_bounds__(256) void coords_merge_k(const float* __restrict__ pc, float* __restrict__ xyz, int rows) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  xyz[r * 3] = pc[r * 2];
  xyz[r * 3 + 1] = pc[r * 2 + 1];
  xyz[r * 3 + 2] = 0.5f * (pc[((size_t)rows + r) * 2] + pc[((size_t)2 * rows + r) * 2 + 1]);
}

// ---------------------------------------------------------------------------------------------
// Gaussian target of one plane, regenerated per lane (dsntnn.py:154-195)
// ---------------------------------------------------------------------------------------------
struct Gauss {
  float tx, ty, kx, ky, inv_norm;
};

__device__ __forceinline__ Gauss make_gauss(const RowGeom& g, int lane, float tx, float ty, float sigma) {
  Gauss q;
  q.tx = tx; q.ty = ty;
  const float sdx = 2.0f * sigma / (float)g.W, sdy = 2.0f * sigma / (float)g.H;   // dsntnn.py:179
  q.kx = -0.5f * (1.0f / sdx) * (1.0f / sdx);
  q.ky = -0.5f * (1.0f / sdy) * (1.0f / sdy);
  float ex = 0.0f, ey = 0.0f;
  for (int w = lane; w < g.W; w += 64) { const float d = cell_coord(w, g.two_over_w, g.first_w) - tx; ex += expf(d * d * q.kx); }
  for (int h = lane; h < g.H; h += 64) { const float d = cell_coord(h, g.two_over_h, g.first_h) - ty; ey += expf(d * d * q.ky); }
  ex = wave_sum(ex);
  ey = wave_sum(ey);
  q.inv_norm = 1.0f / (ex * ey + kEps);
  return q;
}

// JS integrand for one element (dsntnn.py:198-207) and its derivative w.r.t. p (SURVEY §8 a-T).
// IEEE division and logf (not v_rcp / v_log): the log RATIOS are O(0.1..1) whenever p is close to the target, so the
// ~1e-7 ABSOLUTE error of the fast forms was a ~4e-6 relative error on the loss gradient that every layer below then
// inherited (round-2 gradient-parity bisect, tests/test_grad_parity_gpu.py); these kernels are latency-bound anyway.
__device__ __forceinline__ float js_term(float p, float gq) {
  const float m = 0.5f * (p + gq);
  const float lp = logf((p + kEps) / (m + kEps));
  const float lg = logf((gq + kEps) / (m + kEps));
  return 0.5f * (p * lp + gq * lg);
}
// The forward VALUE of the same integrand with the hardware's reciprocal and log2 (v_rcp_f32 / v_log_f32, ~1 ulp each): the ratio is
// formed first, so the log's argument is right to 2^-23 and the term to ~1e-7 absolute -- the loss value feeds no gradient (the
// backward recomputes its own terms from the heatmaps with js_dp's exact forms), and at ~15 instead of ~90 instructions per element
// the forward loss kernel is bound by its bytes, not by its transcendentals (round 5: 270 -> ~100 us at B = 2048).
__device__ __forceinline__ float js_term_fast(float p, float gq) {
  const float m = 0.5f * (p + gq);
  const float rm = __builtin_amdgcn_rcpf(m + kEps);
  const float lp = __builtin_amdgcn_logf((p + kEps) * rm) * 0.69314718055994530942f;
  const float lg = __builtin_amdgcn_logf((gq + kEps) * rm) * 0.69314718055994530942f;
  return 0.5f * (p * lp + gq * lg);
}
__device__ __forceinline__ float js_dp(float p, float gq) {
  const float m = 0.5f * (p + gq);
  const float lp = logf((p + kEps) / (m + kEps));
  return 0.5f * (lp + p / (p + kEps) - m / (m + kEps));
}

#ifndef MPOSE_FAST_LOSS_VALUE
#define MPOSE_FAST_LOSS_VALUE 1
#endif
constexpr bool kFastLossValue = MPOSE_FAST_LOSS_VALUE != 0;      // (stage_loss_fwd_k's JS terms; 0: the exact forms, as the gradient uses)

struct LossArgs {
  const float* hm[MPOSE_MAX_GROUP];
  float* g[MPOSE_MAX_GROUP];
  const float* target;   // (rows, 3)
  const float* xyz_in;   // (rows, 3)  (backward)
  const float* dloss;    // (rows)     (backward)
  float* losses;         // (rows)     (forward)
  float* xyz_out;        // (rows, 3)  (forward)
  int rows, H, W;
  float sigma;
  int pixelwise, three_d, accumulate;
};

// target (x,y) of plane: xy->(t0,t1), zy->(t2,t1), xz->(t0,t2)   (models/margipose_model.py:240-242)
__device__ __forceinline__ void plane_target(int plane, const float* t, float& tx, float& ty) {
  tx = (plane == 1) ? t[2] : t[0];
  ty = (plane == 2) ? t[2] : t[1];
}

template <int NV>
__global__ __launch_bounds__(64 * MPOSE_MAX_GROUP) void stage_loss_fwd_k(LossArgs a) {
  __shared__ float s_part[MPOSE_MAX_GROUP][3];
  const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  const float* t = a.target + (size_t)row * 3;
  const bool active = a.three_d || plane == 0;

  float sx = 0.0f, sy = 0.0f, js = 0.0f;
  if (active) {
    float4 v[NV];
    load_row<NV>(a.hm[plane] + off, lane, g.n4, v, 0.0f);
    float tx, ty;
    plane_target(plane, t, tx, ty);
    Gauss q;
    if (a.pixelwise) q = make_gauss(g, lane, tx, ty, a.sigma);
    // when 256 % W == 0 a lane's four columns are the same for every i: their Gaussian factors are hoisted
    const bool fixed_cols = (256 % g.W) == 0;
    float gx_fixed[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.pixelwise && fixed_cols) {
      const int w0 = (lane * 4) % g.W;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float d = cell_coord(w0 + c, g.two_over_w, g.first_w) - q.tx; gx_fixed[c] = expf(d * d * q.kx); }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int h, w0;
      elem_hw(g, i, lane, h, w0);
      const float y = cell_coord(h, g.two_over_h, g.first_h);
      const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
      const float pv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      const float rs = (pv[0] + pv[1]) + (pv[2] + pv[3]);
      sy = fmaf(rs, y, sy);
      float gy = 0.0f;
      if (a.pixelwise) { const float d = y - q.ty; gy = expf(d * d * q.ky) * q.inv_norm; }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x = x0 + (float)c * g.two_over_w;
        sx = fmaf(pv[c], x, sx);
        if (a.pixelwise && (i * 64 + lane) < g.n4) {
          float gxv = gx_fixed[c];
          if (!fixed_cols) { const float d = x - q.tx; gxv = expf(d * d * q.kx); }
          js += kFastLossValue ? js_term_fast(pv[c], gy * gxv) : js_term(pv[c], gy * gxv);
        }
      }
    }
    sx = wave_sum(sx); sy = wave_sum(sy); js = wave_sum(js);
  }
  if (lane == 0) { s_part[plane][0] = sx; s_part[plane][1] = sy; s_part[plane][2] = js; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float x = s_part[0][0], y = s_part[0][1];
    float z = 0.0f, loss;
    if (a.three_d) {
      z = 0.5f * (s_part[1][0] + s_part[2][1]);
      const float dx = x - t[0], dy = y - t[1], dz = z - t[2];
      loss = sqrtf(dx * dx + dy * dy + dz * dz);                         // dsntnn.py:147-150
      loss += (s_part[0][2] + s_part[1][2]) + s_part[2][2];
    } else {
      const float dx = x - t[0], dy = y - t[1];
      loss = sqrtf(dx * dx + dy * dy) + s_part[0][2];
    }
    float* lo = a.losses + row;
    *lo = a.accumulate ? (*lo + loss) : loss;
    if (a.xyz_out != nullptr) { float* o = a.xyz_out + (size_t)row * 3; o[0] = x; o[1] = y; o[2] = z; }
  }
}

template <int NV>
__global__ __launch_bounds__(64 * MPOSE_MAX_GROUP) void stage_loss_bwd_k(LossArgs a) {
  const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  const float* t = a.target + (size_t)row * 3;
  const float* mu = a.xyz_in + (size_t)row * 3;
  const float wgt = a.dloss[row];
  const bool active = a.three_d || plane == 0;
  float* dst = a.g[plane] + off;

  float4 o[NV];
  if (!active) {
    if (a.accumulate) return;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    store_row<NV>(dst, lane, g.n4, o);
    return;
  }
  // e = (mu - t)/|mu - t|  (NaN at zero distance, exactly like the reference's sqrt backward)
  float cx, cy;
  {
    const float dx = mu[0] - t[0], dy = mu[1] - t[1], dz = a.three_d ? (mu[2] - t[2]) : 0.0f;
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    const float ex = dx / dist, ey = dy / dist, ez = dz / dist;
    cx = (plane == 0) ? ex : (plane == 1 ? 0.5f * ez : 0.0f);    // zy: width axis is z
    cy = (plane == 0) ? ey : (plane == 2 ? 0.5f * ez : 0.0f);    // xz: height axis is z
  }
  float4 v[NV];
  load_row<NV>(a.hm[plane] + off, lane, g.n4, v, 0.0f);
  if (a.accumulate) load_row<NV>(dst, lane, g.n4, o, 0.0f);
  float tx, ty;
  plane_target(plane, t, tx, ty);
  Gauss q;
  if (a.pixelwise) q = make_gauss(g, lane, tx, ty, a.sigma);
  const bool fixed_cols = (256 % g.W) == 0;
  float gx_fixed[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.pixelwise && fixed_cols) {
    const int w0f = (lane * 4) % g.W;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float d = cell_coord(w0f + c, g.two_over_w, g.first_w) - q.tx; gx_fixed[c] = expf(d * d * q.kx); }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int h, w0;
    elem_hw(g, i, lane, h, w0);
    const float y = cell_coord(h, g.two_over_h, g.first_h);
    const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
    float gy = 0.0f;
    if (a.pixelwise) { const float d = y - q.ty; gy = expf(d * d * q.ky) * q.inv_norm; }
    const float pv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x = x0 + (float)c * g.two_over_w;
      float d = cx * x + cy * y;
      if (a.pixelwise) {
        float gxv = gx_fixed[c];
        if (!fixed_cols) { const float dd = x - q.tx; gxv = expf(dd * dd * q.kx); }
        d += js_dp(pv[c], gy * gxv);
      }
      r[c] = wgt * d;
    }
    if (a.accumulate) { o[i].x += r[0]; o[i].y += r[1]; o[i].z += r[2]; o[i].w += r[3]; }
    else o[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
  store_row<NV>(dst, lane, g.n4, o);
}

// ---------------------------------------------------------------------------------------------
// softmax backward, dsnt backward
// ---------------------------------------------------------------------------------------------
struct SoftmaxBwdArgs {
  const float* hm[MPOSE_MAX_GROUP];
  const float* g1[MPOSE_MAX_GROUP];
  const float* g2[MPOSE_MAX_GROUP];
  float* dlogits[MPOSE_MAX_GROUP];
  int n;
};

template <int NV>
__global__ __launch_bounds__(64 * MPOSE_MAX_GROUP) void softmax_bwd_k(SoftmaxBwdArgs a) {
  const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n4 = a.n >> 2;
  const size_t off = (size_t)blockIdx.x * (size_t)a.n;
  float4 p[NV], gg[NV];
  load_row<NV>(a.hm[plane] + off, lane, n4, p, 0.0f);
  load_row<NV>(a.g1[plane] + off, lane, n4, gg, 0.0f);
  if (a.g2[plane] != nullptr) {
    float4 h[NV];
    load_row<NV>(a.g2[plane] + off, lane, n4, h, 0.0f);
#pragma unroll
    for (int i = 0; i < NV; ++i) { gg[i].x += h[i].x; gg[i].y += h[i].y; gg[i].z += h[i].z; gg[i].w += h[i].w; }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (p[i].x * gg[i].x + p[i].y * gg[i].y) + (p[i].z * gg[i].z + p[i].w * gg[i].w);
  s = wave_sum(s);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    gg[i].x = p[i].x * (gg[i].x - s); gg[i].y = p[i].y * (gg[i].y - s);
    gg[i].z = p[i].z * (gg[i].z - s); gg[i].w = p[i].w * (gg[i].w - s);
  }
  store_row<NV>(a.dlogits[plane] + off, lane, n4, gg);
}

struct DsntBwdArgs {
  const float* d_plane_coords;   // (n_planes, rows, 2)
  float* d_hm[MPOSE_MAX_GROUP];
  int rows, H, W, accumulate;
};

template <int NV>
__global__ __launch_bounds__(64 * MPOSE_MAX_GROUP) void dsnt_bwd_k(DsntBwdArgs a) {
  const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  const float* d = a.d_plane_coords + ((size_t)plane * a.rows + row) * 2;
  const float dmx = d[0], dmy = d[1];
  float4 o[NV];
  if (a.accumulate) load_row<NV>(a.d_hm[plane] + off, lane, g.n4, o, 0.0f);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int h, w0;
    elem_hw(g, i, lane, h, w0);
    const float yv = dmy * cell_coord(h, g.two_over_h, g.first_h);
    const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
    float4 r = make_float4(fmaf(dmx, x0, yv), fmaf(dmx, x0 + g.two_over_w, yv),
                           fmaf(dmx, x0 + 2.0f * g.two_over_w, yv), fmaf(dmx, x0 + 3.0f * g.two_over_w, yv));
    if (a.accumulate) { r.x += o[i].x; r.y += o[i].y; r.z += o[i].z; r.w += o[i].w; }
    o[i] = r;
  }
  store_row<NV>(a.d_hm[plane] + off, lane, g.n4, o);
}

// ---------------------------------------------------------------------------------------------
// js_reg_losses alone (one plane, explicit means)
// ---------------------------------------------------------------------------------------------
struct JsArgs {
  const float* hm; const float* mu; const float* djs; float* js; float* g;
  int H, W; float sigma;
};

template <int NV, bool BWD>
__global__ __launch_bounds__(64) void js_k(JsArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  float4 v[NV], o[NV];
  load_row<NV>(a.hm + off, lane, g.n4, v, 0.0f);
  const Gauss q = make_gauss(g, lane, a.mu[(size_t)row * 2], a.mu[(size_t)row * 2 + 1], a.sigma);
  const float wgt = BWD ? a.djs[row] : 0.0f;
  float js = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int h, w0;
    elem_hw(g, i, lane, h, w0);
    const float dy = cell_coord(h, g.two_over_h, g.first_h) - q.ty;
    const float gy = expf(dy * dy * q.ky) * q.inv_norm;
    const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
    const float pv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float dx = x0 + (float)c * g.two_over_w - q.tx;
      const float gq = gy * expf(dx * dx * q.kx);
      if (BWD) r[c] = wgt * js_dp(pv[c], gq);
      else if ((i * 64 + lane) < g.n4) js += js_term(pv[c], gq);
    }
    if (BWD) o[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (BWD) {
    store_row<NV>(a.g + off, lane, g.n4, o);
  } else {
    js = wave_sum(js);
    if (lane == 0) a.js[row] = js;
  }
}

// average_loss (dsntnn.py:99-121): single workgroup, n is B*17.
__global__ __launch_bounds__(256) void average_loss_k(const float* __restrict__ losses, const float* __restrict__ mask, float* out2, int n) {
  __shared__ float s_a[4], s_b[4];
  float num = 0.0f, den = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float m = (mask != nullptr) ? mask[i] : 1.0f;
    num = fmaf(losses[i], m, num);
    den += m;
  }
  num = wave_sum(num); den = wave_sum(den);
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = num; s_b[threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
    const float b = fmaxf((s_b[0] + s_b[1]) + (s_b[2] + s_b[3]), 1.0f);
    out2[0] = a / b;
    out2[1] = b;
  }
}

// ---------------------------------------------------------------------------------------------
// Rows beyond 4096 elements (round 6: heatmaps larger than 64 x 64 -- the reference's only constraint on the input size is
// 192 % (H/16) == 0, models/margipose_model.py:87-97, so a 768 x 768 input has 96 x 96 heatmaps).  A row no longer fits a wave's
// registers: one workgroup of 256 threads per (batch, joint) row walks the row of each plane in float4 strides, once per
// reduction (the row stays in L2 between the passes).  Same formulas as the register kernels above, other summation orders.
// fp32 tensors only.
// ---------------------------------------------------------------------------------------------
constexpr int kBigThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {      // all threads call it; every thread gets the sum
  v = wave_sum(v);
  __syncthreads();                                                     // (sh may still be read from an earlier call)
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
// normalisation of the Gaussian target (dsntnn.py:154-195) with the whole workgroup
__device__ __forceinline__ Gauss make_gauss_block(const RowGeom& g, float tx, float ty, float sigma, float* sh) {
  Gauss q;
  q.tx = tx; q.ty = ty;
  const float sdx = 2.0f * sigma / (float)g.W, sdy = 2.0f * sigma / (float)g.H;
  q.kx = -0.5f * (1.0f / sdx) * (1.0f / sdx);
  q.ky = -0.5f * (1.0f / sdy) * (1.0f / sdy);
  float ex = 0.0f, ey = 0.0f;
  for (int w = threadIdx.x; w < g.W; w += kBigThreads) { const float d = cell_coord(w, g.two_over_w, g.first_w) - tx; ex += expf(d * d * q.kx); }
  for (int h = threadIdx.x; h < g.H; h += kBigThreads) { const float d = cell_coord(h, g.two_over_h, g.first_h) - ty; ey += expf(d * d * q.ky); }
  ex = block_sum(ex, sh);
  ey = block_sum(ey, sh);
  q.inv_norm = 1.0f / (ex * ey + kEps);
  return q;
}
__device__ __forceinline__ void big_hw(const RowGeom& g, int idx, int& h, int& w0) {
  const int e0 = idx * 4;
  h = e0 / g.W;
  w0 = e0 - h * g.W;
}

template <bool EXP>
__global__ __launch_bounds__(kBigThreads) void big_softmax_dsnt_fwd_k(SoftmaxArgs a) {
  __shared__ float sh[4];
  __shared__ float s_mu[MPOSE_MAX_GROUP][2];
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  for (int plane = 0; plane < a.n_planes; ++plane) {
    const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.logits[plane]) + off);
    float4* dst = a.heatmaps[plane] != nullptr ? reinterpret_cast<float4*>(reinterpret_cast<float*>(a.heatmaps[plane]) + off) : nullptr;
    float m = 0.0f, rs = 1.0f;
    if (EXP) {
      m = -INFINITY;
      for (int i = threadIdx.x; i < g.n4; i += kBigThreads) { const float4 v = src[i]; m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
      m = block_max(m, sh);
      float s = 0.0f;
      for (int i = threadIdx.x; i < g.n4; i += kBigThreads) { const float4 v = src[i]; s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m)); }
      s = block_sum(s, sh);
      rs = 1.0f / s;
    }
    float sx = 0.0f, sy = 0.0f;
    for (int i = threadIdx.x; i < g.n4; i += kBigThreads) {
      float4 v = src[i];
      if (EXP) {
        v.x = expf(v.x - m) * rs; v.y = expf(v.y - m) * rs; v.z = expf(v.z - m) * rs; v.w = expf(v.w - m) * rs;
        if (dst != nullptr) dst[i] = v;
      }
      int h, w0;
      big_hw(g, i, h, w0);
      const float y = cell_coord(h, g.two_over_h, g.first_h), x0 = cell_coord(w0, g.two_over_w, g.first_w);
      sy = fmaf((v.x + v.y) + (v.z + v.w), y, sy);
      sx += fmaf(v.w, x0 + 3.0f * g.two_over_w, fmaf(v.z, x0 + 2.0f * g.two_over_w, fmaf(v.y, x0 + g.two_over_w, v.x * x0)));
    }
    sx = block_sum(sx, sh);
    sy = block_sum(sy, sh);
    if (threadIdx.x == 0) {
      s_mu[plane][0] = sx; s_mu[plane][1] = sy;
      if (a.plane_coords != nullptr) { float* pc = a.plane_coords + ((size_t)plane * a.rows + row) * 2; pc[0] = sx; pc[1] = sy; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && a.n_planes == 3 && a.xyz != nullptr) {
    float* o = a.xyz + (size_t)row * 3;
    o[0] = s_mu[0][0]; o[1] = s_mu[0][1]; o[2] = 0.5f * (s_mu[1][0] + s_mu[2][1]);     // models/margipose_model.py:259
  }
}

struct BigSoftmaxBwdArgs {
  SoftmaxBwdArgs a;
  int n_planes;
};
__global__ __launch_bounds__(kBigThreads) void big_softmax_bwd_k(BigSoftmaxBwdArgs b) {
  __shared__ float sh[4];
  const SoftmaxBwdArgs& a = b.a;
  const int n4 = a.n >> 2;
  const size_t off = (size_t)blockIdx.x * (size_t)a.n;
  for (int plane = 0; plane < b.n_planes; ++plane) {
    const float4* p = reinterpret_cast<const float4*>(a.hm[plane] + off);
    const float4* g1 = reinterpret_cast<const float4*>(a.g1[plane] + off);
    const float4* g2 = a.g2[plane] != nullptr ? reinterpret_cast<const float4*>(a.g2[plane] + off) : nullptr;
    float4* d = reinterpret_cast<float4*>(a.dlogits[plane] + off);
    float s = 0.0f;
    for (int i = threadIdx.x; i < n4; i += kBigThreads) {
      const float4 pv = p[i];
      float4 gv = g1[i];
      if (g2 != nullptr) { const float4 h = g2[i]; gv.x += h.x; gv.y += h.y; gv.z += h.z; gv.w += h.w; }
      s += (pv.x * gv.x + pv.y * gv.y) + (pv.z * gv.z + pv.w * gv.w);
    }
    s = block_sum(s, sh);
    for (int i = threadIdx.x; i < n4; i += kBigThreads) {
      const float4 pv = p[i];
      float4 gv = g1[i];
      if (g2 != nullptr) { const float4 h = g2[i]; gv.x += h.x; gv.y += h.y; gv.z += h.z; gv.w += h.w; }
      d[i] = make_float4(pv.x * (gv.x - s), pv.y * (gv.y - s), pv.z * (gv.z - s), pv.w * (gv.w - s));
    }
  }
}

struct BigDsntBwdArgs {
  DsntBwdArgs a;
  int n_planes;
};
__global__ __launch_bounds__(kBigThreads) void big_dsnt_bwd_k(BigDsntBwdArgs b) {
  const DsntBwdArgs& a = b.a;
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  for (int plane = 0; plane < b.n_planes; ++plane) {
    const float* d = a.d_plane_coords + ((size_t)plane * a.rows + row) * 2;
    const float dmx = d[0], dmy = d[1];
    float4* o = reinterpret_cast<float4*>(a.d_hm[plane] + off);
    for (int i = threadIdx.x; i < g.n4; i += kBigThreads) {
      int h, w0;
      big_hw(g, i, h, w0);
      const float yv = dmy * cell_coord(h, g.two_over_h, g.first_h);
      const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
      float4 r = make_float4(fmaf(dmx, x0, yv), fmaf(dmx, x0 + g.two_over_w, yv), fmaf(dmx, x0 + 2.0f * g.two_over_w, yv),
                             fmaf(dmx, x0 + 3.0f * g.two_over_w, yv));
      if (a.accumulate) { const float4 t = o[i]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
      o[i] = r;
    }
  }
}

__global__ __launch_bounds__(kBigThreads) void big_stage_loss_fwd_k(LossArgs a) {
  __shared__ float sh[4];
  __shared__ float s_part[MPOSE_MAX_GROUP][3];
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  const float* t = a.target + (size_t)row * 3;
  for (int plane = 0; plane < 3; ++plane) {
    float sx = 0.0f, sy = 0.0f, js = 0.0f;
    if (a.three_d || plane == 0) {        // (uniform)
      const float4* src = reinterpret_cast<const float4*>(a.hm[plane] + off);
      float tx, ty;
      plane_target(plane, t, tx, ty);
      Gauss q{};
      if (a.pixelwise) q = make_gauss_block(g, tx, ty, a.sigma, sh);
      for (int i = threadIdx.x; i < g.n4; i += kBigThreads) {
        const float4 v = src[i];
        int h, w0;
        big_hw(g, i, h, w0);
        const float y = cell_coord(h, g.two_over_h, g.first_h), x0 = cell_coord(w0, g.two_over_w, g.first_w);
        const float pv[4] = {v.x, v.y, v.z, v.w};
        sy = fmaf((pv[0] + pv[1]) + (pv[2] + pv[3]), y, sy);
        float gy = 0.0f;
        if (a.pixelwise) { const float d = y - q.ty; gy = expf(d * d * q.ky) * q.inv_norm; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x = x0 + (float)c * g.two_over_w;
          sx = fmaf(pv[c], x, sx);
          if (a.pixelwise) { const float d = x - q.tx; js += js_term(pv[c], gy * expf(d * d * q.kx)); }
        }
      }
      sx = block_sum(sx, sh); sy = block_sum(sy, sh); js = block_sum(js, sh);
    }
    if (threadIdx.x == 0) { s_part[plane][0] = sx; s_part[plane][1] = sy; s_part[plane][2] = js; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float x = s_part[0][0], y = s_part[0][1];
    float z = 0.0f, loss;
    if (a.three_d) {
      z = 0.5f * (s_part[1][0] + s_part[2][1]);
      const float dx = x - t[0], dy = y - t[1], dz = z - t[2];
      loss = sqrtf(dx * dx + dy * dy + dz * dz) + ((s_part[0][2] + s_part[1][2]) + s_part[2][2]);
    } else {
      const float dx = x - t[0], dy = y - t[1];
      loss = sqrtf(dx * dx + dy * dy) + s_part[0][2];
    }
    float* lo = a.losses + row;
    *lo = a.accumulate ? (*lo + loss) : loss;
    if (a.xyz_out != nullptr) { float* o = a.xyz_out + (size_t)row * 3; o[0] = x; o[1] = y; o[2] = z; }
  }
}

__global__ __launch_bounds__(kBigThreads) void big_stage_loss_bwd_k(LossArgs a) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  const float* t = a.target + (size_t)row * 3;
  const float* mu = a.xyz_in + (size_t)row * 3;
  const float wgt = a.dloss[row];
  const float dx = mu[0] - t[0], dy = mu[1] - t[1], dz = a.three_d ? (mu[2] - t[2]) : 0.0f;
  const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
  const float ex = dx / dist, ey = dy / dist, ez = dz / dist;
  for (int plane = 0; plane < 3; ++plane) {
    float4* dst = reinterpret_cast<float4*>(a.g[plane] + off);
    if (!(a.three_d || plane == 0)) {
      if (!a.accumulate)
        for (int i = threadIdx.x; i < g.n4; i += kBigThreads) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const float cx = (plane == 0) ? ex : (plane == 1 ? 0.5f * ez : 0.0f);
    const float cy = (plane == 0) ? ey : (plane == 2 ? 0.5f * ez : 0.0f);
    const float4* src = reinterpret_cast<const float4*>(a.hm[plane] + off);
    float tx, ty;
    plane_target(plane, t, tx, ty);
    Gauss q{};
    if (a.pixelwise) q = make_gauss_block(g, tx, ty, a.sigma, sh);
    for (int i = threadIdx.x; i < g.n4; i += kBigThreads) {
      const float4 v = src[i];
      int h, w0;
      big_hw(g, i, h, w0);
      const float y = cell_coord(h, g.two_over_h, g.first_h), x0 = cell_coord(w0, g.two_over_w, g.first_w);
      float gy = 0.0f;
      if (a.pixelwise) { const float d = y - q.ty; gy = expf(d * d * q.ky) * q.inv_norm; }
      const float pv[4] = {v.x, v.y, v.z, v.w};
      float r[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float x = x0 + (float)c * g.two_over_w;
        float d = cx * x + cy * y;
        if (a.pixelwise) { const float dd = x - q.tx; d += js_dp(pv[c], gy * expf(dd * dd * q.kx)); }
        r[c] = wgt * d;
      }
      if (a.accumulate) { const float4 o = dst[i]; r[0] += o.x; r[1] += o.y; r[2] += o.z; r[3] += o.w; }
      dst[i] = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(kBigThreads) void big_js_k(JsArgs a) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  const RowGeom g = make_geom(a.H, a.W);
  const size_t off = (size_t)row * (size_t)(a.H * a.W);
  const float4* src = reinterpret_cast<const float4*>(a.hm + off);
  const Gauss q = make_gauss_block(g, a.mu[(size_t)row * 2], a.mu[(size_t)row * 2 + 1], a.sigma, sh);
  const float wgt = BWD ? a.djs[row] : 0.0f;
  float js = 0.0f;
  for (int i = threadIdx.x; i < g.n4; i += kBigThreads) {
    const float4 v = src[i];
    int h, w0;
    big_hw(g, i, h, w0);
    const float dy = cell_coord(h, g.two_over_h, g.first_h) - q.ty;
    const float gy = expf(dy * dy * q.ky) * q.inv_norm;
    const float x0 = cell_coord(w0, g.two_over_w, g.first_w);
    const float pv[4] = {v.x, v.y, v.z, v.w};
    float r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float dxx = x0 + (float)c * g.two_over_w - q.tx;
      const float gq = gy * expf(dxx * dxx * q.kx);
      if (BWD) r[c] = wgt * js_dp(pv[c], gq);
      else js += js_term(pv[c], gq);
    }
    if (BWD) reinterpret_cast<float4*>(a.g + off)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (!BWD) {
    js = block_sum(js, sh);
    if (threadIdx.x == 0) a.js[row] = js;
  }
}

inline bool big_row(long n) { return n > 4096 && (n & 3) == 0 && n <= (1l << 24); }

inline int pick_nv(int n) {
  if (n <= 0 || (n & 3) || n > 4096) return 0;
  const int nv = (n / 4 + 63) / 64;
  return nv <= 4 ? 4 : (nv <= 9 ? 9 : 16);
}

}  // namespace
}  // namespace mpose

using namespace mpose;

#define MPOSE_DISPATCH_NV(nv, CALL)            \
  switch (nv) {                                \
    case 4: { constexpr int NV = 4; CALL; } break;   \
    case 9: { constexpr int NV = 9; CALL; } break;   \
    case 16: { constexpr int NV = 16; CALL; } break; \
    default: return MPOSE_EINVAL;              \
  }

extern "C" int mpose_abi_version(void) { return 16; }   // 2: bf16-plane packed weights, mpose_conv_operands.in1, wgrad tiles, frames/im2col entry points; 3: im2col/col2im_s2, bn_add layout 2; 4: mpose_bn_bwd_coef(eval_mode); 5: plane engine (conv_p.hip, split.hip), pack job layout; 6: mpose_sgd_step, coef job dconv_bias; 7: mpose_bn_bwd_reduce_ws; 8: fused output stage of the plane engine (mpose_conv_operands.epi_*, add_*, out0_planes); 9: MPOSE_CONV_F16X3 (amax fields, pack layout 2, mpose_absmax, mpose_weights_absmax); 10: channel extremes (mpose_conv_operands.mm0, mpose_bn_job.minmax / amax_out), row-of-taps weight gradient; 11: per-axis slot strides (mpose_conv_geom.in_mul_x / out_mul_x), MPOSE_MAX_CLASSES 8; 12: BatchNorm finalisation by the convolution launch's last workgroup (mpose_conv_operands.fin*); 13: MPOSE_CONV_H2_IN (conv_h.hip: producer-split fp16 planes, pack layout 3, mpose_split_h2, out0_amax without the fused stage); 14: launch plans (plan.hip: mpose_plan_*, mpose_stream_wait, recordable fills / copy / loss arithmetic); 15: mpose_conv_wgrad_waves; 16: planes end to end for the H2 blocks (mpose_wgrad_operands.planes_in, mpose_bn_bwd_apply_h2(db_h2), the two-input data gradient on conv_h2r_k)

extern "C" int mpose_softmax_dsnt_fwd(const void* const* logits, void* const* heatmaps, float* plane_coords, float* xyz,
                                      int n_planes, int rows, int H, int W, int io_dtype, void* stream) {
  if (n_planes < 1 || n_planes > MPOSE_MAX_GROUP || rows < 0 || (W & 3)) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  const int nv = pick_nv(H * W);
  SoftmaxArgs a{};
  for (int p = 0; p < n_planes; ++p) { a.logits[p] = logits[p]; a.heatmaps[p] = heatmaps ? heatmaps[p] : nullptr; }
  a.plane_coords = plane_coords; a.xyz = xyz; a.n_planes = n_planes; a.rows = rows; a.H = H; a.W = W;
  hipStream_t s = (hipStream_t)stream;
  if (big_row((long)H * W)) {           // heatmaps beyond 64 x 64: the multi-pass form (fp32 only)
    if (io_dtype != 0) return MPOSE_EINVAL;
    launch(big_softmax_dsnt_fwd_k<true>, dim3(rows), dim3(kBigThreads), 0, s, a);
    return launch_status();
  }
  // variant = rows per workgroup (1, 2) + 16 * NT bits (every variant is bit-identical; profiles/r5_tail_variants.txt).
  // Sizes whose logits + heatmaps exceed the 256 MB Infinity Cache stream (non-temporal both ways, two rows per workgroup);
  // smaller ones keep plain accesses -- their heatmaps are re-read from the cache by the next stage's combiner and the loss kernels.
  constexpr int forced = -1;
  const size_t bytes = (size_t)rows * n_planes * H * W * (io_dtype == 0 ? 8 : (io_dtype == 1 ? 4 : 6));
  const int variant = forced >= 0 ? forced : (bytes > (size_t)256 << 20 ? 2 + 16 * 3 : 1);
  const int rpw = (variant & 15) == 2 && nv <= 9 ? 2 : 1, nt = (variant >> 4) & 3;
  const int grid = (rows + rpw - 1) / rpw;
#define MPOSE_TAIL_LAUNCH(BI, BO)                                                                                              \
  do {                                                                                                                         \
    if (rpw == 2) {                                                                                                            \
      if (nv == 4) { if (nt == 3) launch(softmax_dsnt_fwd_k<4, BI, BO, true, 2, 3>, dim3(grid), dim3(64 * n_planes), 0, s, a);                \
                     else if (nt == 1) launch(softmax_dsnt_fwd_k<4, BI, BO, true, 2, 1>, dim3(grid), dim3(64 * n_planes), 0, s, a);           \
                     else launch(softmax_dsnt_fwd_k<4, BI, BO, true, 2, 0>, dim3(grid), dim3(64 * n_planes), 0, s, a); }                     \
      else { if (nt == 3) launch(softmax_dsnt_fwd_k<9, BI, BO, true, 2, 3>, dim3(grid), dim3(64 * n_planes), 0, s, a);                        \
             else if (nt == 1) launch(softmax_dsnt_fwd_k<9, BI, BO, true, 2, 1>, dim3(grid), dim3(64 * n_planes), 0, s, a);                   \
             else launch(softmax_dsnt_fwd_k<9, BI, BO, true, 2, 0>, dim3(grid), dim3(64 * n_planes), 0, s, a); }                             \
    } else if (nt == 3) {                                                                                                      \
      MPOSE_DISPATCH_NV(nv, (launch(softmax_dsnt_fwd_k<NV, BI, BO, true, 1, 3>, dim3(grid), dim3(64 * n_planes), 0, s, a)));                  \
    } else if (nt == 1) {                                                                                                      \
      MPOSE_DISPATCH_NV(nv, (launch(softmax_dsnt_fwd_k<NV, BI, BO, true, 1, 1>, dim3(grid), dim3(64 * n_planes), 0, s, a)));                  \
    } else {                                                                                                                   \
      MPOSE_DISPATCH_NV(nv, (launch(softmax_dsnt_fwd_k<NV, BI, BO, true, 1, 0>, dim3(grid), dim3(64 * n_planes), 0, s, a)));                  \
    }                                                                                                                          \
  } while (0)
  if (io_dtype == 0) {
    MPOSE_TAIL_LAUNCH(false, false);
  } else if (io_dtype == 1) {
    MPOSE_TAIL_LAUNCH(true, true);
  } else if (io_dtype == 2) {
    MPOSE_TAIL_LAUNCH(false, true);
  } else {
    return MPOSE_EINVAL;
  }
#undef MPOSE_TAIL_LAUNCH
  return launch_status();
}

extern "C" int mpose_bn_add_softmax_fwd(const mpose_bn_add_operands* ops, void* const* heatmaps, float* plane_coords, int n_groups,
                                        int B, int H, int W, int C, int J, int io_dtype, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || B < 0 || (W & 3) || (C & 3) || J < 1 || J > C || (io_dtype != 0 && io_dtype != 2))
    return MPOSE_EINVAL;
  if (B == 0) return 0;
  const int P = H * W;
  const int lds = 4 * (P + kBasPad) * 4;                           // <= 64 KB (pick_nv: P <= 4096)
  BnAddSoftmaxArgs a{};
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    a.heat[i] = heatmaps[i];
    if (!ops[i].a || !ops[i].b || !ops[i].a_scale || !ops[i].a_shift || !ops[i].b_scale || !ops[i].b_shift || !heatmaps[i]) return MPOSE_EINVAL;
  }
  a.plane_coords = plane_coords; a.B = B; a.P = P; a.C = C; a.J = J; a.H = H; a.W = W;
  const int nv = pick_nv(P);
  // the all-joints form (every channel line read once, coalesced) when the two inputs of the launch's columns are large enough to
  // keep its B * n_groups workgroups busy and all J rows fit in LDS
  constexpr int allj_env = -1;
  const bool fits = (long)J * (P + kBasPad) * 4 <= kBasAllJLds && C == 32;
  // (measured crossover at 32 x 32, three columns: B = 64 -- 50 MB of inputs -- 16.6 us four-joint / 21.9 us all-joints,
  //  B = 128 -- 100 MB -- 55.7 / 28.3 us; B = 2048: 1659 / 423 us, the two launches it replaces: 1317 us)
  const bool allj = fits && (allj_env >= 0 ? allj_env != 0 : (long)n_groups * 2 * B * P * C * 4 > (64l << 20));
  int rc = 0;
  if (io_dtype == 0) { MPOSE_DISPATCH_NV(nv, (rc = launch_bn_add_softmax<NV, false>(a, n_groups, lds, allj, (hipStream_t)stream))); }
  else { MPOSE_DISPATCH_NV(nv, (rc = launch_bn_add_softmax<NV, true>(a, n_groups, lds, allj, (hipStream_t)stream))); }
  if (rc) return rc;
  return launch_status();
}

extern "C" int mpose_coords_merge(const float* plane_coords, float* xyz, int rows, void* stream) {
  if (rows < 0 || !plane_coords || !xyz) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  launch(coords_merge_k, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, plane_coords, xyz, rows);
  return launch_status();
}

extern "C" int mpose_dsnt_fwd(const float* const* heatmaps, float* plane_coords, float* xyz, int n_planes, int rows, int H,
                              int W, void* stream) {
  if (n_planes < 1 || n_planes > MPOSE_MAX_GROUP || rows < 0 || (W & 3)) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  const int nv = pick_nv(H * W);
  SoftmaxArgs a{};
  for (int p = 0; p < n_planes; ++p) { a.logits[p] = heatmaps[p]; a.heatmaps[p] = nullptr; }
  a.plane_coords = plane_coords; a.xyz = xyz; a.n_planes = n_planes; a.rows = rows; a.H = H; a.W = W;
  if (big_row((long)H * W)) {
    launch(big_softmax_dsnt_fwd_k<false>, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(softmax_dsnt_fwd_k<NV, false, false, false>, dim3(rows), dim3(64 * n_planes), 0, (hipStream_t)stream, a)));
  return launch_status();
}

extern "C" int mpose_dsnt_bwd(const float* d_plane_coords, float* const* d_heatmaps, int n_planes, int rows, int H, int W,
                              int accumulate, void* stream) {
  if (n_planes < 1 || n_planes > MPOSE_MAX_GROUP || rows < 0 || (W & 3)) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  const int nv = pick_nv(H * W);
  DsntBwdArgs a{};
  a.d_plane_coords = d_plane_coords;
  for (int p = 0; p < n_planes; ++p) a.d_hm[p] = d_heatmaps[p];
  a.rows = rows; a.H = H; a.W = W; a.accumulate = accumulate;
  if (big_row((long)H * W)) {
    launch(big_dsnt_bwd_k, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, BigDsntBwdArgs{a, n_planes});
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(dsnt_bwd_k<NV>, dim3(rows), dim3(64 * n_planes), 0, (hipStream_t)stream, a)));
  return launch_status();
}

extern "C" int mpose_softmax_bwd(const float* const* heatmaps, const float* const* g1, const float* const* g2,
                                 float* const* dlogits, int n_planes, int rows, int n, void* stream) {
  if (n_planes < 1 || n_planes > MPOSE_MAX_GROUP || rows < 0) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  const int nv = pick_nv(n);
  SoftmaxBwdArgs a{};
  for (int p = 0; p < n_planes; ++p) {
    a.hm[p] = heatmaps[p]; a.g1[p] = g1[p]; a.g2[p] = g2 ? g2[p] : nullptr; a.dlogits[p] = dlogits[p];
  }
  a.n = n;
  if (big_row(n)) {
    launch(big_softmax_bwd_k, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, BigSoftmaxBwdArgs{a, n_planes});
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(softmax_bwd_k<NV>, dim3(rows), dim3(64 * n_planes), 0, (hipStream_t)stream, a)));
  return launch_status();
}

static int fill_loss_args(LossArgs& a, const float* const* heatmaps, const float* target, int rows, int H, int W, float sigma,
                          int pixelwise, int three_d, int accumulate) {
  if (rows < 0 || (W & 3) || sigma <= 0.0f) return MPOSE_EINVAL;
  for (int p = 0; p < 3; ++p) a.hm[p] = heatmaps[p];
  a.target = target; a.rows = rows; a.H = H; a.W = W; a.sigma = sigma;
  a.pixelwise = pixelwise; a.three_d = three_d; a.accumulate = accumulate;
  return 0;
}

extern "C" int mpose_stage_loss_fwd(const float* const* heatmaps, const float* target, float* losses, float* xyz_out, int rows,
                                    int H, int W, float sigma, int pixelwise, int three_d, int accumulate, void* stream) {
  LossArgs a{};
  const int rc = fill_loss_args(a, heatmaps, target, rows, H, W, sigma, pixelwise, three_d, accumulate);
  if (rc) return rc;
  if (rows == 0) return 0;
  a.losses = losses; a.xyz_out = xyz_out;
  const int nv = pick_nv(H * W);
  if (big_row((long)H * W)) {
    launch(big_stage_loss_fwd_k, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(stage_loss_fwd_k<NV>, dim3(rows), dim3(64 * 3), 0, (hipStream_t)stream, a)));
  return launch_status();
}

extern "C" int mpose_stage_loss_bwd(const float* const* heatmaps, const float* target, const float* xyz, const float* dloss,
                                    float* const* g, int rows, int H, int W, float sigma, int pixelwise, int three_d,
                                    int accumulate, void* stream) {
  LossArgs a{};
  const int rc = fill_loss_args(a, heatmaps, target, rows, H, W, sigma, pixelwise, three_d, accumulate);
  if (rc) return rc;
  if (rows == 0) return 0;
  a.xyz_in = xyz; a.dloss = dloss;
  for (int p = 0; p < 3; ++p) a.g[p] = g[p];
  const int nv = pick_nv(H * W);
  if (big_row((long)H * W)) {
    launch(big_stage_loss_bwd_k, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(stage_loss_bwd_k<NV>, dim3(rows), dim3(64 * 3), 0, (hipStream_t)stream, a)));
  return launch_status();
}

extern "C" int mpose_js_fwd(const float* heatmaps, const float* mu, float* js, int rows, int H, int W, float sigma, void* stream) {
  if (rows < 0 || (W & 3) || sigma <= 0.0f) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  JsArgs a{heatmaps, mu, nullptr, js, nullptr, H, W, sigma};
  const int nv = pick_nv(H * W);
  if (big_row((long)H * W)) {
    launch(big_js_k<false>, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(js_k<NV, false>, dim3(rows), dim3(64), 0, (hipStream_t)stream, a)));
  return launch_status();
}

extern "C" int mpose_js_bwd(const float* heatmaps, const float* mu, const float* djs, float* g, int rows, int H, int W,
                            float sigma, void* stream) {
  if (rows < 0 || (W & 3) || sigma <= 0.0f) return MPOSE_EINVAL;
  if (rows == 0) return 0;
  JsArgs a{heatmaps, mu, djs, nullptr, g, H, W, sigma};
  const int nv = pick_nv(H * W);
  if (big_row((long)H * W)) {
    launch(big_js_k<true>, dim3(rows), dim3(kBigThreads), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  MPOSE_DISPATCH_NV(nv, (launch(js_k<NV, true>, dim3(rows), dim3(64), 0, (hipStream_t)stream, a)));
  return launch_status();
}

extern "C" int mpose_average_loss_fwd(const float* losses, const float* mask, float* out2, int n, void* stream) {
  if (n < 0) return MPOSE_EINVAL;
  launch(average_loss_k, dim3(1), dim3(256), 0, (hipStream_t)stream, losses, mask, out2, n);
  return launch_status();
}
