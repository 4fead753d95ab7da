// Layout / glue kernels of the MargiPose hot path on gfx950 (all HBM-bound, small):
//   * patch8 stem input transform (NCHW image -> NHWC space-to-depth) and its inverse,
//   * the HeatmapColumn axis permutation (reference models/margipose_model.py:91-97) on NHWC,
//   * HeatmapCombiner + cumulative add (models/margipose_model.py:142-150, :195), fwd and bwd,
//   * gradient fan-in add, NCHW->NHWC padding of the logits gradient, partial-sum reduction.
#include "common.h"

namespace mpose {
namespace {

inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

// out[b][y][x][c*64 + ky*8 + kx] = in[b][c][y*8+ky][x*8+kx]; one thread per 4 consecutive kx.
template <bool INVERSE>
__global__ __launch_bounds__(256) void space_to_depth8_k(const float* __restrict__ src, float* __restrict__ dst, int B, int S) {
  const int G = S / 8;
  const long total4 = (long)B * G * G * 48;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int k4 = (int)(i % 48);
    long r = i / 48;
    const int x = (int)(r % G); r /= G;
    const int y = (int)(r % G);
    const long b = r / G;
    const int k = k4 * 4, c = k >> 6, ky = (k >> 3) & 7, kx = k & 7;
    const long nchw = ((b * 3 + c) * S + (y * 8 + ky)) * S + x * 8 + kx;
    if (!INVERSE) reinterpret_cast<float4*>(dst)[i] = *reinterpret_cast<const float4*>(src + nchw);
    else *reinterpret_cast<float4*>(dst + nchw) = reinterpret_cast<const float4*>(src)[i];
  }
}

struct PermArgs {
  const float* in[MPOSE_MAX_GROUP];
  float* out[MPOSE_MAX_GROUP];
  int space[MPOSE_MAX_GROUP];
  int B, S, C;
};

// NHWC (B,S,S,C), C = n_chunks*S.  zy: out[b][h][w=j][kS+i] = in[b][h][w=i][kS+j]
//                                  xz: out[b][h=j][w][kS+i] = in[b][h=i][w][kS+j]
__global__ __launch_bounds__(256) void axis_permute_k(PermArgs a) {
  const int grp = blockIdx.y;
  const float* __restrict__ in = a.in[grp];
  float* __restrict__ out = a.out[grp];
  const int space = a.space[grp];
  const int S = a.S, C = a.C, c4n = C >> 2;
  const long total4 = (long)a.B * S * S * c4n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % c4n) * 4;
    long r = idx / c4n;
    const int w = (int)(r % S); r /= S;
    const int h = (int)(r % S);
    const long b = r / S;
    float4 v;
    if (space == 0) {
      v = reinterpret_cast<const float4*>(in)[idx];
    } else {
      const int k = c / S, i = c - k * S;      // i..i+3 stay inside the chunk since S % 4 == 0
      float e[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long src = (space == 1) ? (((b * S + h) * S + (i + t)) * C + k * S + w)
                                      : (((b * S + (i + t)) * S + w) * C + k * S + h);
        e[t] = in[src];
      }
      v = make_float4(e[0], e[1], e[2], e[3]);
    }
    reinterpret_cast<float4*>(out)[idx] = v;
  }
}

// ---- HeatmapCombiner ----------------------------------------------------------------------
constexpr int CT = 64;        // pixels per tile
constexpr int CC = 128;       // feature channels

struct CombArgs {
  const float* hm[MPOSE_MAX_GROUP];
  float* d_hm[MPOSE_MAX_GROUP];
  const float* w;             // (C, 3J)
  const float* inp;
  float* out;
  const float* g;
  float* dw_partial;
  int B, J, HW, Q;            // Q = 3J
  int hm_bf16;                // forward only: heatmaps are bf16 (inference storage mode)
};

__global__ __launch_bounds__(256) void combiner_fwd_k(CombArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sW = sm;                       // [Q][CC]
  float* sH = sm + a.Q * CC;            // [Q][CT]
  const int tid = threadIdx.x;
  for (int i = tid; i < a.Q * CC; i += 256) { const int q = i / CC, c = i - q * CC; sW[i] = a.w[c * a.Q + q]; }
  const int tiles_per_img = a.HW / CT;
  const int b = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - b * tiles_per_img) * CT;
  for (int i = tid; i < a.Q * CT; i += 256) {
    const int q = i / CT, px = i - q * CT;
    const int pl = q / a.J, j = q - pl * a.J;
    const long ho = ((long)b * a.J + j) * a.HW + p0 + px;
    sH[i] = a.hm_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.hm[pl])[ho] << 16) : a.hm[pl][ho];
  }
  __syncthreads();
  const int c4 = tid & 31, pr = tid >> 5;
  float4 acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int q = 0; q < a.Q; ++q) {
    const float4 wv = *reinterpret_cast<const float4*>(sW + q * CC + c4 * 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float h = sH[q * CT + pr * 8 + e];
      acc[e].x = fmaf(h, wv.x, acc[e].x); acc[e].y = fmaf(h, wv.y, acc[e].y);
      acc[e].z = fmaf(h, wv.z, acc[e].z); acc[e].w = fmaf(h, wv.w, acc[e].w);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const long o = (((long)b * a.HW + p0 + pr * 8 + e) * CC + c4 * 4);
    const float4 x = *reinterpret_cast<const float4*>(a.inp + o);
    *reinterpret_cast<float4*>(a.out + o) = make_float4(x.x + acc[e].x, x.y + acc[e].y, x.z + acc[e].z, x.w + acc[e].w);
  }
}

constexpr int G_STRIDE = CC + 4;
constexpr int QH_MAX = 32;    // per-thread dW accumulators (Q <= 64)
#ifndef COMB_EXP
#define COMB_EXP 0             // timing experiments: 1 no d_hm loop, 2 no dW loop, 4 no tile loads
#endif

__global__ __launch_bounds__(256) void combiner_bwd_k(CombArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sW = sm;                                // [Q][CC]
  float* sH = sm + a.Q * CC;                     // [2 * QH_MAX][CT]: rows >= Q stay zero (the dW loop below reads them unconditionally)
  float* sG = sH + 2 * QH_MAX * CT;              // [CT][G_STRIDE]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < a.Q * CC; i += 256) { const int q = i / CC, c = i - q * CC; sW[i] = a.w[c * a.Q + q]; }
  const int tiles_per_img = a.HW / CT;
  const int n_tiles = a.B * tiles_per_img;
  for (int i = a.Q * CT + tid; i < 2 * QH_MAX * CT; i += 256) sH[i] = 0.f;
  const int qh = (a.Q + 1) / 2;
  const int dwc = tid & 127, dwq0 = (tid >> 7) * qh;
  float dwacc[QH_MAX];
#pragma unroll
  for (int e = 0; e < QH_MAX; ++e) dwacc[e] = 0.f;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int b = tile / tiles_per_img, p0 = (tile - b * tiles_per_img) * CT;
    __syncthreads();
    if (!(COMB_EXP & 4))
    for (int i = tid; i < a.Q * CT; i += 256) {
      const int q = i / CT, px = i - q * CT;
      const int pl = q / a.J, j = q - pl * a.J;
      const long ho = ((long)b * a.J + j) * a.HW + p0 + px;
    sH[i] = a.hm_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.hm[pl])[ho] << 16) : a.hm[pl][ho];
    }
    if (!(COMB_EXP & 4))
    for (int i = tid; i < CT * (CC / 4); i += 256) {
      const int px = i / (CC / 4), c4 = i - px * (CC / 4);
      *reinterpret_cast<float4*>(sG + px * G_STRIDE + c4 * 4) =
          *reinterpret_cast<const float4*>(a.g + ((long)b * a.HW + p0 + px) * CC + c4 * 4);
    }
    __syncthreads();
    // d_hm[q][px] = sum_c W[c][q] * g[px][c]; wave w handles q = w, w+4, ...; lane = pixel.  All of the wave's (<= 16) q at once:
    // one read of the lane's g per channel quad instead of one per (q, quad), and 16 independent accumulation chains instead
    // of one (the q-outer form took 195 us for 0.86 GFLOP: each output was a chain of 128 dependent fused multiply-adds).
    if (!(COMB_EXP & 1)) {
      float sq[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) sq[e] = 0.f;
#pragma unroll 2
      for (int c4 = 0; c4 < CC / 4; ++c4) {
        const float4 gv = *reinterpret_cast<const float4*>(sG + lane * G_STRIDE + c4 * 4);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int q = wave + 4 * e;
          if (q < a.Q) {                        // (wave-uniform)
            const float4 wv = *reinterpret_cast<const float4*>(sW + q * CC + c4 * 4);
            sq[e] = fmaf(gv.x, wv.x, sq[e]); sq[e] = fmaf(gv.y, wv.y, sq[e]); sq[e] = fmaf(gv.z, wv.z, sq[e]); sq[e] = fmaf(gv.w, wv.w, sq[e]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int q = wave + 4 * e;
        if (q < a.Q) {
          const int pl = q / a.J, j = q - pl * a.J;
          a.d_hm[pl][((long)b * a.J + j) * a.HW + p0 + lane] = sq[e];
        }
      }
    }
    // dW[c][q] += sum_px g[px][c] * hm[q][px]: four pixels per step, every one of the thread's QH_MAX rows of hm (those past Q hold
    // zeros: no conditions in the loop -- with them it took 115 us of the kernel's 177, ~80 cycles per multiply-add)
    for (int px = 0; px < ((COMB_EXP & 2) ? 0 : CT); px += 4) {
      const float g0 = sG[px * G_STRIDE + dwc], g1 = sG[(px + 1) * G_STRIDE + dwc];
      const float g2 = sG[(px + 2) * G_STRIDE + dwc], g3 = sG[(px + 3) * G_STRIDE + dwc];
#pragma unroll
      for (int e = 0; e < QH_MAX; ++e) {
        const float4 h = *reinterpret_cast<const float4*>(sH + (dwq0 + e) * CT + px);
        dwacc[e] = fmaf(g3, h.w, fmaf(g2, h.z, fmaf(g1, h.y, fmaf(g0, h.x, dwacc[e]))));
      }
    }
  }
#pragma unroll
  for (int e = 0; e < QH_MAX; ++e)
    if (e < qh && dwq0 + e < a.Q) a.dw_partial[((long)blockIdx.x * CC + dwc) * a.Q + dwq0 + e] = dwacc[e];
}

__global__ __launch_bounds__(256) void add_k(const float4* __restrict__ x, const float4* __restrict__ y, float4* __restrict__ o, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 p = x[i], q = y[i];
    o[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
  }
}

// (B, J, P) NCHW -> (B, P, Cpad) NHWC, channels >= J zero-filled.
struct PadArgs {
  const float* in[MPOSE_MAX_GROUP];
  float* out[MPOSE_MAX_GROUP];
  int B, J, P, Cpad;
};
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_k(PadArgs a) {
  const float* __restrict__ in = a.in[blockIdx.y];
  float* __restrict__ out = a.out[blockIdx.y];
  const long npix = (long)a.B * a.P;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / a.P;
    const int px = (int)(p - b * a.P);
    for (int c = 0; c < a.Cpad; c += 4) {
      float e[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) e[t] = (c + t < a.J) ? in[(b * a.J + c + t) * a.P + px] : 0.f;
      *reinterpret_cast<float4*>(out + p * a.Cpad + c) = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}

// dst[i] = sum_p src[p][i]: 32 elements per workgroup, the partials dealt to 8 slices of 32 threads (four loads in flight each),
// slices added in a fixed order.  (One thread per element walking all partials in a dependent chain took 63 us for 6.7 MB.)
__global__ __launch_bounds__(256) void reduce_partials_k(const float* __restrict__ src, float* __restrict__ dst, int n_partial, long n, int accumulate) {
  __shared__ float sl[8][32];
  const int el = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + el;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (i < n) {
    int p = slice;
    for (; p + 24 < n_partial; p += 32) {
      t0 += src[(long)p * n + i]; t1 += src[(long)(p + 8) * n + i]; t2 += src[(long)(p + 16) * n + i]; t3 += src[(long)(p + 24) * n + i];
    }
    for (; p < n_partial; p += 8) t0 += src[(long)p * n + i];
  }
  sl[slice][el] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (slice == 0 && i < n) {
    const float s = ((sl[0][el] + sl[1][el]) + (sl[2][el] + sl[3][el])) + ((sl[4][el] + sl[5][el]) + (sl[6][el] + sl[7][el]));
    dst[i] = accumulate ? dst[i] + s : s;
  }
}


// ---- 3x3 pooling (InceptionV4 stem: MaxPool2d(3, stride 2, pad 1) and AvgPool2d(3, 1, 1, count_include_pad=False)) ----
struct PoolArgs {
  const float* in; const float* scale; const float* shift; const float* g; float* out;
  int B, IH, IW, C, OH, OW, ld, kind;
};

__device__ __forceinline__ float4 pool_act(const PoolArgs& a, long pix, int c, const float4& sc, const float4& sh) {
  float4 v = *reinterpret_cast<const float4*>(a.in + pix * a.C + c);
  if (a.scale != nullptr) {
    v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
  }
  return v;
}

__global__ __launch_bounds__(256) void pool3_fwd_k(PoolArgs a) {
  const int c4n = a.C >> 2;
  const int st = a.kind == 0 ? 2 : 1;
  const long total = (long)a.B * a.OH * a.OW * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    long r = i / c4n;
    const int ox = (int)(r % a.OW); r /= a.OW;
    const int oy = (int)(r % a.OH);
    const long b = r / a.OH;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scale != nullptr) { sc = *reinterpret_cast<const float4*>(a.scale + c); sh = *reinterpret_cast<const float4*>(a.shift + c); }
    float4 acc = a.kind == 0 ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    for (int ky = -1; ky <= 1; ++ky)
      for (int kx = -1; kx <= 1; ++kx) {
        const int iy = oy * st + ky, ix = ox * st + kx;
        if (iy < 0 || iy >= a.IH || ix < 0 || ix >= a.IW) continue;
        const float4 v = pool_act(a, (b * a.IH + iy) * a.IW + ix, c, sc, sh);
        if (a.kind == 0) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        ++cnt;
      }
    if (a.kind == 1) { const float rc = 1.0f / (float)cnt; acc.x *= rc; acc.y *= rc; acc.z *= rc; acc.w *= rc; }
    *reinterpret_cast<float4*>(a.out + ((b * a.OH + oy) * a.OW + ox) * a.ld + c) = acc;
  }
}

// Gather form (deterministic, no atomics): one thread per INPUT element sums what the windows containing it send back.
__global__ __launch_bounds__(256) void pool3_bwd_k(PoolArgs a) {
  const int c4n = a.C >> 2;
  const long total = (long)a.B * a.IH * a.IW * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    long r = i / c4n;
    const int ix = (int)(r % a.IW); r /= a.IW;
    const int iy = (int)(r % a.IH);
    const long b = r / a.IH;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scale != nullptr) { sc = *reinterpret_cast<const float4*>(a.scale + c); sh = *reinterpret_cast<const float4*>(a.shift + c); }
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.kind == 1) {
      for (int oy = iy - 1; oy <= iy + 1; ++oy)
        for (int ox = ix - 1; ox <= ix + 1; ++ox) {
          if (oy < 0 || oy >= a.OH || ox < 0 || ox >= a.OW) continue;
          const int ny = min(oy + 1, a.IH - 1) - max(oy - 1, 0) + 1, nx = min(ox + 1, a.IW - 1) - max(ox - 1, 0) + 1;
          const float rc = 1.0f / (float)(ny * nx);
          const float4 gv = *reinterpret_cast<const float4*>(a.g + ((b * a.OH + oy) * a.OW + ox) * a.ld + c);
          d[0] += gv.x * rc; d[1] += gv.y * rc; d[2] += gv.z * rc; d[3] += gv.w * rc;
        }
    } else {
      const float4 me4 = pool_act(a, (b * a.IH + iy) * a.IW + ix, c, sc, sh);
      const float me[4] = {me4.x, me4.y, me4.z, me4.w};
      // windows (stride 2, pad 1) that contain (iy, ix): oy with 2*oy - 1 <= iy <= 2*oy + 1
      for (int oy = (iy + 1) / 2 - ((iy & 1) ? 0 : 0); oy >= 0 && 2 * oy + 1 >= iy; --oy) {
        if (oy >= a.OH || 2 * oy - 1 > iy) continue;
        for (int ox = (ix + 1) / 2; ox >= 0 && 2 * ox + 1 >= ix; --ox) {
          if (ox >= a.OW || 2 * ox - 1 > ix) continue;
          // first position (row-major scan, as ATen records it) holding the window maximum
          float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          int arg[4] = {-1, -1, -1, -1};
          for (int ky = -1; ky <= 1; ++ky)
            for (int kx = -1; kx <= 1; ++kx) {
              const int yy = oy * 2 + ky, xx = ox * 2 + kx;
              if (yy < 0 || yy >= a.IH || xx < 0 || xx >= a.IW) continue;
              const float4 v4 = pool_act(a, (b * a.IH + yy) * a.IW + xx, c, sc, sh);
              const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) if (v[e] > best[e]) { best[e] = v[e]; arg[e] = (ky + 1) * 3 + (kx + 1); }
            }
          const int mine = (iy - oy * 2 + 1) * 3 + (ix - ox * 2 + 1);
          const float4 gv = *reinterpret_cast<const float4*>(a.g + ((b * a.OH + oy) * a.OW + ox) * a.ld + c);
          const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) if (arg[e] == mine) d[e] += g4[e];
          (void)me;
        }
      }
    }
    float4* dst = reinterpret_cast<float4*>(a.out + ((b * a.IH + iy) * a.IW + ix) * a.C + c);
    float4 o = *dst;
    o.x += d[0]; o.y += d[1]; o.z += d[2]; o.w += d[3];
    *dst = o;
  }
}

// Max-pool backward in two passes (the single-pass gather above re-derives up to four 9-element argmaxes per input
// element): pass 1, one thread per OUTPUT window x 4 channels, records the window position (0..8, first maximum in
// row-major order, as ATen does) in a byte; pass 2, one thread per INPUT element x 4 channels, reads the bytes and the
// gradients of the <= 4 windows that contain it.  Deterministic, no atomics.
__global__ __launch_bounds__(256) void maxpool3_argmax_k(PoolArgs a, unsigned char* __restrict__ arg_out) {
  const int c4n = a.C >> 2;
  const long total = (long)a.B * a.OH * a.OW * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    long r = i / c4n;
    const int ox = (int)(r % a.OW); r /= a.OW;
    const int oy = (int)(r % a.OH);
    const long b = r / a.OH;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scale != nullptr) { sc = *reinterpret_cast<const float4*>(a.scale + c); sh = *reinterpret_cast<const float4*>(a.shift + c); }
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned arg[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
      for (int kx = -1; kx <= 1; ++kx) {
        const int yy = oy * 2 + ky, xx = ox * 2 + kx;
        if (yy < 0 || yy >= a.IH || xx < 0 || xx >= a.IW) continue;
        const float4 v4 = pool_act(a, (b * a.IH + yy) * a.IW + xx, c, sc, sh);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (v[e] > best[e]) { best[e] = v[e]; arg[e] = (unsigned)((ky + 1) * 3 + (kx + 1)); }
      }
    *reinterpret_cast<unsigned*>(arg_out + i * 4) = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
  }
}
// The forward max pool that also records the window positions (round 5): a training forward keeps the bytes (1/4 of the pooled
// output) and its backward skips pass 1 above -- the recomputation re-read the whole pre-pool tensor (26 + 42 us per step).  Same
// traversal and the same strict `>` as maxpool3_argmax_k, the stored maximum equals pool3_fwd_k's (the taps' order does not matter to it).
__global__ __launch_bounds__(256) void maxpool3_fwd_arg_k(PoolArgs a, unsigned char* __restrict__ arg_out) {
  const int c4n = a.C >> 2;
  const long total = (long)a.B * a.OH * a.OW * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    long r = i / c4n;
    const int ox = (int)(r % a.OW); r /= a.OW;
    const int oy = (int)(r % a.OH);
    const long b = r / a.OH;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.scale != nullptr) { sc = *reinterpret_cast<const float4*>(a.scale + c); sh = *reinterpret_cast<const float4*>(a.shift + c); }
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned arg[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
      for (int kx = -1; kx <= 1; ++kx) {
        const int yy = oy * 2 + ky, xx = ox * 2 + kx;
        if (yy < 0 || yy >= a.IH || xx < 0 || xx >= a.IW) continue;
        const float4 v4 = pool_act(a, (b * a.IH + yy) * a.IW + xx, c, sc, sh);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (v[e] > best[e]) { best[e] = v[e]; arg[e] = (unsigned)((ky + 1) * 3 + (kx + 1)); }
      }
    *reinterpret_cast<unsigned*>(arg_out + i * 4) = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
    *reinterpret_cast<float4*>(a.out + ((b * a.OH + oy) * a.OW + ox) * a.ld + c) = make_float4(best[0], best[1], best[2], best[3]);
  }
}
__global__ __launch_bounds__(256) void maxpool3_bwd_arg_k(PoolArgs a, const unsigned char* __restrict__ arg_in) {
  const int c4n = a.C >> 2;
  const long total = (long)a.B * a.IH * a.IW * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long r = i / c4n;
    const int ix = (int)(r % a.IW); r /= a.IW;
    const int iy = (int)(r % a.IH);
    const long b = r / a.IH;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    // windows (stride 2, pad 1) containing (iy, ix): 2*oy - 1 <= iy <= 2*oy + 1
    const int oy1 = (iy + 1) >> 1, ox1 = (ix + 1) >> 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int oy = oy1 - u;
      if (oy < 0 || oy >= a.OH || 2 * oy + 1 < iy) continue;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int ox = ox1 - w;
        if (ox < 0 || ox >= a.OW || 2 * ox + 1 < ix) continue;
        const long op = (b * a.OH + oy) * a.OW + ox;
        const unsigned ar = *reinterpret_cast<const unsigned*>(arg_in + (op * c4n + c4) * 4);
        const float4 gv = *reinterpret_cast<const float4*>(a.g + op * a.ld + c4 * 4);
        const unsigned mine = (unsigned)((iy - oy * 2 + 1) * 3 + (ix - ox * 2 + 1));
        d[0] += (ar & 0xff) == mine ? gv.x : 0.f;
        d[1] += ((ar >> 8) & 0xff) == mine ? gv.y : 0.f;
        d[2] += ((ar >> 16) & 0xff) == mine ? gv.z : 0.f;
        d[3] += (ar >> 24) == mine ? gv.w : 0.f;
      }
    }
    float4* dst = reinterpret_cast<float4*>(a.out + i * 4);
    float4 o = *dst;
    o.x += d[0]; o.y += d[1]; o.z += d[2]; o.w += d[3];
    *dst = o;
  }
}

// NCHW (B,C,H,W) <-> NHWC (B,H,W,Cpad), channels >= C zero
__global__ __launch_bounds__(256) void image_to_nhwc_k(const float* __restrict__ x, float* __restrict__ out, int B, int C, long HW, int Cpad) {
  const long npix = (long)B * HW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / HW, px = p - b * HW;
    for (int c = 0; c < Cpad; c += 4) {
      float e[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) e[t] = (c + t < C) ? x[(b * C + c + t) * HW + px] : 0.f;
      *reinterpret_cast<float4*>(out + p * Cpad + c) = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}
// uint8 RGB frames (B, 3, H, W) -> normalised fp32: (x/255 - mean[c]) / std[c]  (reference data_specs.py:6-13,38-39:
// `ImageSpecs.convert` = torchvision `to_tensor` + `normalize_pixels`), written either NHWC zero-padded to Cpad channels
// (the InceptionV4 stem's first load) or NCHW (Cpad == 0; input of the patch8 stem's space-to-depth).
__global__ __launch_bounds__(256) void frames_u8_k(const unsigned char* __restrict__ x, float* __restrict__ out, int B, long HW, int Cpad,
                                                  float s0, float s1, float s2, float t0, float t1, float t2) {
  const long npix = (long)B * HW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / HW, px = p - b * HW;
    const float r = fmaf((float)x[(b * 3 + 0) * HW + px], s0, t0);
    const float g = fmaf((float)x[(b * 3 + 1) * HW + px], s1, t1);
    const float bl = fmaf((float)x[(b * 3 + 2) * HW + px], s2, t2);
    if (Cpad == 0) {
      out[(b * 3 + 0) * HW + px] = r; out[(b * 3 + 1) * HW + px] = g; out[(b * 3 + 2) * HW + px] = bl;
    } else {
      *reinterpret_cast<float4*>(out + p * Cpad) = make_float4(r, g, bl, 0.f);
      for (int c = 4; c < Cpad; c += 4) *reinterpret_cast<float4*>(out + p * Cpad + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
// First layer of the InceptionV4 stem (3x3, stride 2, padding 1 on a 3-channel image): instead of padding the image
// to 32 channels and running a 9-tap convolution over K = 288 (27 real), gather the 27 inputs of every output pixel
// once -- patches (B, H/2, W/2, 32), channel q = c*9 + ky*3 + kx as in the flattened (Cout, 3, 3, 3) weight, 27..31
// zero -- so that the layer is a 1x1 convolution with K = 32.  U8: uint8 frames, normalised on the fly.
template <bool U8>
__global__ __launch_bounds__(256) void im2col_k3s2_k(const void* __restrict__ xin, float* __restrict__ out, int B, int H, int W,
                                                    float s0, float s1, float s2, float t0, float t1, float t2) {
  const int OH = H >> 1, OW = W >> 1;
  const long total = (long)B * OH * OW * 8;                    // float4 pieces
  const long HW = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q4 = (int)(i & 7);
    long r = i >> 3;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const long b = r / OH;
    float e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q4 * 4 + u;
      float v = 0.f;
      if (q < 27) {
        const int c = q / 9, kk = q - c * 9, ky = kk / 3, kx = kk - ky * 3;
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const long o = (b * 3 + c) * HW + (long)iy * W + ix;
          if (U8) {
            const float sc = c == 0 ? s0 : (c == 1 ? s1 : s2), sh = c == 0 ? t0 : (c == 1 ? t1 : t2);
            v = fmaf((float)static_cast<const unsigned char*>(xin)[o], sc, sh);
          } else {
            v = static_cast<const float*>(xin)[o];
          }
        }
      }
      e[u] = v;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(e[0], e[1], e[2], e[3]);
  }
}
// gradient of the gather: dx[b,c,y,x] = sum over the (up to 4) patches that read this pixel
__global__ __launch_bounds__(256) void col2im_k3s2_k(const float* __restrict__ dp, float* __restrict__ dx, int B, int H, int W) {
  const int OH = H >> 1, OW = W >> 1;
  const long total = (long)B * 3 * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    long r = i / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % 3);
    const long b = r / 3;
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = y + 1 - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= OH) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = x + 1 - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= OW) continue;
        s += dp[((b * OH + (ty >> 1)) * OW + (tx >> 1)) * 32 + c * 9 + ky * 3 + kx];
      }
    }
    dx[i] = s;
  }
}
// Generic k x k / stride 2 / pad k/2 gather (ResNet's 7x7 first layer: K = 147 -> Cpad = 160) and its gradient.
template <bool U8>
__global__ __launch_bounds__(256) void im2col_s2_k(const void* __restrict__ xin, float* __restrict__ out, int B, int H, int W, int k,
                                                  int Cpad, float s0, float s1, float s2, float t0, float t1, float t2) {
  const int OH = H >> 1, OW = W >> 1, kk2 = k * k, pad = k >> 1, c4n = Cpad >> 2;
  const long total = (long)B * OH * OW * c4n;
  const long HW = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q4 = (int)(i % c4n);
    long r = i / c4n;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const long b = r / OH;
    float e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q4 * 4 + u;
      float v = 0.f;
      if (q < 3 * kk2) {
        const int c = q / kk2, kk = q - c * kk2, ky = kk / k, kx = kk - ky * k;
        const int iy = 2 * oy - pad + ky, ix = 2 * ox - pad + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const long o = (b * 3 + c) * HW + (long)iy * W + ix;
          if (U8) {
            const float sc = c == 0 ? s0 : (c == 1 ? s1 : s2), sh = c == 0 ? t0 : (c == 1 ? t1 : t2);
            v = fmaf((float)static_cast<const unsigned char*>(xin)[o], sc, sh);
          } else {
            v = static_cast<const float*>(xin)[o];
          }
        }
      }
      e[u] = v;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(e[0], e[1], e[2], e[3]);
  }
}
__global__ __launch_bounds__(256) void col2im_s2_k(const float* __restrict__ dp, float* __restrict__ dx, int B, int H, int W, int k, int Cpad) {
  const int OH = H >> 1, OW = W >> 1, pad = k >> 1;
  const long total = (long)B * 3 * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W);
    long r = i / W;
    const int y = (int)(r % H); r /= H;
    const int c = (int)(r % 3);
    const long b = r / 3;
    float s = 0.f;
    for (int ky = 0; ky < k; ++ky) {
      const int ty = y + pad - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= OH) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int tx = x + pad - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= OW) continue;
        s += dp[((b * OH + (ty >> 1)) * OW + (tx >> 1)) * Cpad + (c * k + ky) * k + kx];
      }
    }
    dx[i] = s;
  }
}
__global__ __launch_bounds__(256) void nhwc_to_image_k(const float* __restrict__ g, float* __restrict__ dx, int B, int C, long HW, int Cpad) {
  const long npix = (long)B * HW;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / HW, px = p - b * HW;
    for (int c = 0; c < C; ++c) dx[(b * C + c) * HW + px] = g[p * Cpad + c];
  }
}

}  // namespace
}  // namespace mpose

using namespace mpose;

extern "C" int mpose_space_to_depth8(const float* x, float* out, int B, int S, void* stream) {
  if (B < 0 || S <= 0 || (S % 8)) return MPOSE_EINVAL;
  if (B == 0) return 0;
  const long total4 = (long)B * (S / 8) * (S / 8) * 48;
  launch(space_to_depth8_k<false>, dim3(grid_for(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, out, B, S);
  return launch_status();
}

extern "C" int mpose_depth_to_space8(const float* g, float* dx, int B, int S, void* stream) {
  if (B < 0 || S <= 0 || (S % 8)) return MPOSE_EINVAL;
  if (B == 0) return 0;
  const long total4 = (long)B * (S / 8) * (S / 8) * 48;
  launch(space_to_depth8_k<true>, dim3(grid_for(total4, 256)), dim3(256), 0, (hipStream_t)stream, g, dx, B, S);
  return launch_status();
}

extern "C" int mpose_axis_permute(const float* const* in, float* const* out, const int* spaces, int n_groups, int B, int S, int C,
                                  void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (S & 3) || (C % S)) return MPOSE_EINVAL;
  PermArgs a{};
  for (int i = 0; i < n_groups; ++i) {
    a.in[i] = in[i]; a.out[i] = out[i]; a.space[i] = spaces[i];
    if (spaces[i] < 0 || spaces[i] > 2) return MPOSE_EINVAL;
  }
  a.B = B; a.S = S; a.C = C;
  const long total4 = (long)B * S * S * C / 4;
  if (total4 == 0) return 0;
  launch(axis_permute_k, dim3(dim3(grid_for(total4, 256), n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

static int combiner_fwd_impl(const void* const* hm, int hm_bf16, const float* w, const float* inp, float* out, int B, int J, int HW,
                             int C, void* stream);
extern "C" int mpose_combiner_fwd(const float* const* hm, const float* w, const float* inp, float* out, int B, int J, int HW, int C,
                                  void* stream) {
  return combiner_fwd_impl(reinterpret_cast<const void* const*>(hm), 0, w, inp, out, B, J, HW, C, stream);
}
extern "C" int mpose_combiner_fwd_bf16(const void* const* hm, const float* w, const float* inp, float* out, int B, int J, int HW,
                                       int C, void* stream) {
  return combiner_fwd_impl(hm, 1, w, inp, out, B, J, HW, C, stream);
}
static int combiner_fwd_impl(const void* const* hm, int hm_bf16, const float* w, const float* inp, float* out, int B, int J, int HW,
                             int C, void* stream) {
  if (C != CC || (HW % CT) || J < 1 || 3 * J > 2 * QH_MAX) return MPOSE_EINVAL;
  if (B == 0) return 0;
  CombArgs a{};
  a.hm_bf16 = hm_bf16;
  for (int p = 0; p < 3; ++p) a.hm[p] = static_cast<const float*>(hm[p]);
  a.w = w; a.inp = inp; a.out = out; a.B = B; a.J = J; a.HW = HW; a.Q = 3 * J;
  const int lds = (a.Q * CC + a.Q * CT) * 4;
  launch(combiner_fwd_k, dim3(B * (HW / CT)), dim3(256), lds, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_combiner_bwd(const float* const* hm, const float* w, const float* g, float* const* d_hm, float* dw_partial,
                                  int n_partial, int B, int J, int HW, int C, void* stream) {
  if (C != CC || (HW % CT) || J < 1 || 3 * J > 2 * QH_MAX || n_partial < 1) return MPOSE_EINVAL;
  CombArgs a{};
  for (int p = 0; p < 3; ++p) { a.hm[p] = hm[p]; a.d_hm[p] = d_hm[p]; }
  a.w = w; a.g = g; a.dw_partial = dw_partial; a.B = B; a.J = J; a.HW = HW; a.Q = 3 * J;
  const int lds = (a.Q * CC + 2 * QH_MAX * CT + CT * G_STRIDE) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(combiner_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  launch(combiner_bwd_k, dim3(n_partial), dim3(256), lds, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_add(const float* x, const float* y, float* out, int64_t n, void* stream) {
  if (n < 0 || (n & 3)) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(add_k, dim3(grid_for(n / 4, 256 * 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y),
                                                                  reinterpret_cast<float4*>(out), n / 4);
  return launch_status();
}

extern "C" int mpose_nchw_to_nhwc_pad(const float* const* in, float* const* out, int n_groups, int B, int J, int P, int Cpad,
                                      void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (Cpad & 3) || J > Cpad) return MPOSE_EINVAL;
  PadArgs a{};
  for (int i = 0; i < n_groups; ++i) { a.in[i] = in[i]; a.out[i] = out[i]; }
  a.B = B; a.J = J; a.P = P; a.Cpad = Cpad;
  if ((long)B * P == 0) return 0;
  launch(nchw_to_nhwc_pad_k, dim3(dim3(grid_for((long)B * P, 256), n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_reduce_partials(const float* src, float* dst, int n_partial, int64_t n, int accumulate, void* stream) {
  if (n_partial < 1 || n < 0) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(reduce_partials_k, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, src, dst, n_partial, n, accumulate);
  return launch_status();
}

static int pool_dims(int kind, int IH, int IW, int& OH, int& OW) {
  if (kind == 0) { OH = (IH + 2 - 3) / 2 + 1; OW = (IW + 2 - 3) / 2 + 1; return 0; }
  if (kind == 1) { OH = IH; OW = IW; return 0; }
  return MPOSE_EINVAL;
}

extern "C" int mpose_pool3_fwd(const float* in, const float* scale, const float* shift, float* out, int B, int IH, int IW, int C,
                               int out_ld, int kind, void* stream) {
  PoolArgs a{};
  if (pool_dims(kind, IH, IW, a.OH, a.OW) || (C & 3) || out_ld < C || (out_ld & 3)) return MPOSE_EINVAL;
  a.in = in; a.scale = scale; a.shift = shift; a.out = out; a.B = B; a.IH = IH; a.IW = IW; a.C = C; a.ld = out_ld; a.kind = kind;
  const long total = (long)B * a.OH * a.OW * (C / 4);
  if (total == 0) return 0;
  launch(pool3_fwd_k, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_pool3_bwd(const float* in, const float* scale, const float* shift, const float* g, float* d_in, int B, int IH,
                               int IW, int C, int g_ld, int kind, void* stream) {
  PoolArgs a{};
  if (pool_dims(kind, IH, IW, a.OH, a.OW) || (C & 3) || g_ld < C || (g_ld & 3)) return MPOSE_EINVAL;
  a.in = in; a.scale = scale; a.shift = shift; a.g = g; a.out = d_in; a.B = B; a.IH = IH; a.IW = IW; a.C = C; a.ld = g_ld; a.kind = kind;
  const long total = (long)B * IH * IW * (C / 4);
  if (total == 0) return 0;
  launch(pool3_bwd_k, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_maxpool3_bwd_ws(const float* in, const float* scale, const float* shift, const float* g, float* d_in,
                                    void* workspace, long workspace_bytes, int B, int IH, int IW, int C, int g_ld, void* stream) {
  PoolArgs a{};
  if (pool_dims(0, IH, IW, a.OH, a.OW) || (C & 3) || g_ld < C || (g_ld & 3)) return MPOSE_EINVAL;
  if (!workspace || workspace_bytes < (long)B * a.OH * a.OW * C) return MPOSE_EINVAL;
  a.in = in; a.scale = scale; a.shift = shift; a.g = g; a.out = d_in; a.B = B; a.IH = IH; a.IW = IW; a.C = C; a.ld = g_ld; a.kind = 0;
  const long total_o = (long)B * a.OH * a.OW * (C / 4), total_i = (long)B * IH * IW * (C / 4);
  if (total_i == 0) return 0;
  launch(maxpool3_argmax_k, dim3(grid_for(total_o, 256)), dim3(256), 0, (hipStream_t)stream, a, static_cast<unsigned char*>(workspace));
  launch(maxpool3_bwd_arg_k, dim3(grid_for(total_i, 256)), dim3(256), 0, (hipStream_t)stream, a, static_cast<const unsigned char*>(workspace));
  return launch_status();
}

extern "C" int mpose_maxpool3_fwd_arg(const float* in, const float* scale, const float* shift, float* out, void* arg_out, long arg_bytes,
                                      int B, int IH, int IW, int C, int out_ld, void* stream) {
  PoolArgs a{};
  if (pool_dims(0, IH, IW, a.OH, a.OW) || (C & 3) || out_ld < C || (out_ld & 3)) return MPOSE_EINVAL;
  if (!arg_out || arg_bytes < (long)B * a.OH * a.OW * C) return MPOSE_EINVAL;
  a.in = in; a.scale = scale; a.shift = shift; a.out = out; a.B = B; a.IH = IH; a.IW = IW; a.C = C; a.ld = out_ld; a.kind = 0;
  const long total = (long)B * a.OH * a.OW * (C / 4);
  if (total == 0) return 0;
  launch(maxpool3_fwd_arg_k, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a, static_cast<unsigned char*>(arg_out));
  return launch_status();
}

extern "C" int mpose_maxpool3_bwd_arg(const float* g, const void* arg, long arg_bytes, float* d_in, int B, int IH, int IW, int C, int g_ld,
                                      void* stream) {
  PoolArgs a{};
  if (pool_dims(0, IH, IW, a.OH, a.OW) || (C & 3) || g_ld < C || (g_ld & 3)) return MPOSE_EINVAL;
  if (!arg || arg_bytes < (long)B * a.OH * a.OW * C) return MPOSE_EINVAL;
  a.g = g; a.out = d_in; a.B = B; a.IH = IH; a.IW = IW; a.C = C; a.ld = g_ld; a.kind = 0;
  const long total_i = (long)B * IH * IW * (C / 4);
  if (total_i == 0) return 0;
  launch(maxpool3_bwd_arg_k, dim3(grid_for(total_i, 256)), dim3(256), 0, (hipStream_t)stream, a, static_cast<const unsigned char*>(arg));
  return launch_status();
}

extern "C" int mpose_image_to_nhwc(const float* x, float* out, int B, int C, int H, int W, int Cpad, void* stream) {
  if (C < 1 || C > Cpad || (Cpad & 3)) return MPOSE_EINVAL;
  if ((long)B * H * W == 0) return 0;
  launch(image_to_nhwc_k, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, out, B, C, (long)H * W, Cpad);
  return launch_status();
}

extern "C" int mpose_frames_u8(const unsigned char* frames, const float* mean3, const float* std3, float* out, int B, int H, int W,
                               int Cpad, void* stream) {
  if (!frames || !out || !mean3 || !std3 || (Cpad & 3) || Cpad < 0) return MPOSE_EINVAL;
  if ((long)B * H * W == 0) return 0;
  float sc[3], sh[3];
  for (int c = 0; c < 3; ++c) {
    if (!(std3[c] > 0.f)) return MPOSE_EINVAL;
    sc[c] = 1.0f / (255.0f * std3[c]);
    sh[c] = -mean3[c] / std3[c];
  }
  launch(frames_u8_k, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, frames, out, B, (long)H * W, Cpad, sc[0], sc[1], sc[2],
                                                                                 sh[0], sh[1], sh[2]);
  return launch_status();
}

extern "C" int mpose_im2col_k3s2(const void* img, int is_u8, const float* mean3, const float* std3, float* out, int B, int H, int W,
                                 void* stream) {
  if (!img || !out || (H & 1) || (W & 1) || H <= 0 || W <= 0) return MPOSE_EINVAL;
  if (B == 0) return 0;
  float sc[3] = {1.f, 1.f, 1.f}, sh[3] = {0.f, 0.f, 0.f};
  if (is_u8) {
    if (!mean3 || !std3) return MPOSE_EINVAL;
    for (int c = 0; c < 3; ++c) {
      if (!(std3[c] > 0.f)) return MPOSE_EINVAL;
      sc[c] = 1.0f / (255.0f * std3[c]);
      sh[c] = -mean3[c] / std3[c];
    }
  }
  const long total = (long)B * (H / 2) * (W / 2) * 8;
  if (is_u8) launch(im2col_k3s2_k<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, sc[0], sc[1], sc[2], sh[0], sh[1], sh[2]);
  else launch(im2col_k3s2_k<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f);
  return launch_status();
}

extern "C" int mpose_im2col_s2(const void* img, int is_u8, const float* mean3, const float* std3, float* out, int B, int H, int W,
                               int k, int Cpad, void* stream) {
  if (!img || !out || (H & 1) || (W & 1) || H <= 0 || W <= 0 || k < 1 || !(k & 1) || (Cpad & 3) || Cpad < 3 * k * k) return MPOSE_EINVAL;
  if (B == 0) return 0;
  float sc[3] = {1.f, 1.f, 1.f}, sh[3] = {0.f, 0.f, 0.f};
  if (is_u8) {
    if (!mean3 || !std3) return MPOSE_EINVAL;
    for (int c = 0; c < 3; ++c) {
      if (!(std3[c] > 0.f)) return MPOSE_EINVAL;
      sc[c] = 1.0f / (255.0f * std3[c]);
      sh[c] = -mean3[c] / std3[c];
    }
  }
  const long total = (long)B * (H / 2) * (W / 2) * (Cpad / 4);
  if (is_u8) launch(im2col_s2_k<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, k, Cpad, sc[0], sc[1], sc[2], sh[0], sh[1], sh[2]);
  else launch(im2col_s2_k<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, k, Cpad, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f);
  return launch_status();
}

extern "C" int mpose_col2im_s2(const float* dpatches, float* dx, int B, int H, int W, int k, int Cpad, void* stream) {
  if (!dpatches || !dx || (H & 1) || (W & 1) || H <= 0 || W <= 0 || k < 1 || !(k & 1) || Cpad < 3 * k * k) return MPOSE_EINVAL;
  if (B == 0) return 0;
  launch(col2im_s2_k, dim3(grid_for((long)B * 3 * H * W, 256)), dim3(256), 0, (hipStream_t)stream, dpatches, dx, B, H, W, k, Cpad);
  return launch_status();
}

extern "C" int mpose_col2im_k3s2(const float* dpatches, float* dx, int B, int H, int W, void* stream) {
  if (!dpatches || !dx || (H & 1) || (W & 1) || H <= 0 || W <= 0) return MPOSE_EINVAL;
  if (B == 0) return 0;
  launch(col2im_k3s2_k, dim3(grid_for((long)B * 3 * H * W, 256)), dim3(256), 0, (hipStream_t)stream, dpatches, dx, B, H, W);
  return launch_status();
}

extern "C" int mpose_nhwc_to_image(const float* g, float* dx, int B, int C, int H, int W, int Cpad, void* stream) {
  if (C < 1 || C > Cpad || (Cpad & 3)) return MPOSE_EINVAL;
  if ((long)B * H * W == 0) return 0;
  launch(nhwc_to_image_k, dim3(grid_for((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, g, dx, B, C, (long)H * W, Cpad);
  return launch_status();
}
