// BatchNorm2d pieces for gfx950 (reference models/margipose_model.py:31,34,37 -- nn.BatchNorm2d
// defaults: eps 1e-5, momentum 0.1, biased batch variance for normalisation, unbiased for the running
// estimate).  The expensive parts of BN never run as their own pass over the activations:
//   * batch sums are reduced in the producing conv's epilogue (conv.hip, fp64 atomics),
//   * normalise(+ReLU) is applied while the consuming conv stages its input tile,
//   * here: the per-channel finalisation, the residual "BN(a) + BN(b)" add, and the backward
//     reductions / coefficient algebra.  All HBM-bound, float4 NHWC accesses.
//
// Backward algebra (x = pre-BN activation, g = upstream gradient, N = B*H*W):
//   xhat = (x - mean) * invstd ; dgamma = sum g*xhat ; dbeta = sum g
//   dx = gamma*invstd * (g - mean(g) - xhat * mean(g*xhat)) = c0*g + c1*(x - mean) + c2
//   with c0 = gamma*invstd, c1 = -c0*invstd*mean(g*xhat), c2 = -c0*mean(g); coefficient rows (c0, c1, c2, mean).
//   (x is centred per element: folding -c1*mean into c2 makes c1*x + c2 a cancelling pair whose rounded constant is a
//   per-channel BIAS on every pixel -- it showed up as a 3-5x excess on sums of dx over pixels, e.g. the shortcut BatchNorm's
//   bias gradient one block further down, in the round-2 gradient-parity bisect.)
#include <stdlib.h>
#include "common.h"

namespace mpose {
namespace {

constexpr int kPartSplit = 6;      // workgroups per job of the partial-row finalize / coefficient launches

// 256 threads per job, or 1024 when the statistics arrive as per-workgroup partial rows (MPOSE_CONV_STATS_PART: a 128-channel
// job then sums 256 rows with eight row slices of 128 channel lanes, 32 rows per thread, eight loads in flight)
__global__ __launch_bounds__(1024) void bn_finalize_k(const mpose_bn_job* __restrict__ jobs, int train, float eps, float momentum) {
  __shared__ double sh[1024 * 4];      // (sums + extremes of a slice meet here in one pass: reduce_part_rows_pair)
  // (common.h; bit 1: the statistics are the jobs' partial rows -- gridDim.y workgroups share a job's channels)
  bn_finalize_job<false>(jobs[blockIdx.x], train & 1, eps, momentum, (train & 2) ? sh : nullptr, (int)blockIdx.y, (int)gridDim.y, (train & 4) != 0);
}

struct BnAddArgs {
  mpose_bn_add_operands op[MPOSE_MAX_GROUP];
  long total4;            // B*P*C/4
  int C, P, c_keep;
};

template <bool RELU_A>
__global__ __launch_bounds__(256) void bn_add_nhwc_k(BnAddArgs a) {
  const mpose_bn_add_operands& op = a.op[blockIdx.y];
  const int c4n = a.C >> 2;
  float amax = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total4; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const float4 x = reinterpret_cast<const float4*>(op.a)[i];
    const float4 y = reinterpret_cast<const float4*>(op.b)[i];
    const float4 sa = *reinterpret_cast<const float4*>(op.a_scale + c), ta = *reinterpret_cast<const float4*>(op.a_shift + c);
    const float4 sb = *reinterpret_cast<const float4*>(op.b_scale + c), tb = *reinterpret_cast<const float4*>(op.b_shift + c);
    float4 o;
    o.x = (RELU_A ? fmaxf(fmaf(x.x, sa.x, ta.x), 0.f) : fmaf(x.x, sa.x, ta.x)) + fmaf(y.x, sb.x, tb.x);
    o.y = (RELU_A ? fmaxf(fmaf(x.y, sa.y, ta.y), 0.f) : fmaf(x.y, sa.y, ta.y)) + fmaf(y.y, sb.y, tb.y);
    o.z = (RELU_A ? fmaxf(fmaf(x.z, sa.z, ta.z), 0.f) : fmaf(x.z, sa.z, ta.z)) + fmaf(y.z, sb.z, tb.z);
    o.w = (RELU_A ? fmaxf(fmaf(x.w, sa.w, ta.w), 0.f) : fmaf(x.w, sa.w, ta.w)) + fmaf(y.w, sb.w, tb.w);
    reinterpret_cast<float4*>(op.out)[i] = o;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (op.out_amax != nullptr) block_amax_commit(amax, op.out_amax);      // (uniform per workgroup: one operand set per blockIdx.y)
}

// NHWC (B, P, C) inputs -> NCHW (B, c_keep, P) output: one thread per pixel, writes coalesced over pixels.
__global__ __launch_bounds__(256) void bn_add_nchw_k(BnAddArgs a, int B) {
  const mpose_bn_add_operands& op = a.op[blockIdx.y];
  const long npix = (long)B * a.P;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long b = p / a.P;
    const int px = (int)(p - b * a.P);
    for (int c = 0; c < a.c_keep; c += 4) {
      const float4 x = *reinterpret_cast<const float4*>(op.a + p * a.C + c);
      const float4 y = *reinterpret_cast<const float4*>(op.b + p * a.C + c);
      const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = c + e;
        if (cc < a.c_keep)
          op.out[(b * a.c_keep + cc) * a.P + px] = fmaxf(fmaf(xs[e], op.a_scale[cc], op.a_shift[cc]), 0.f) + fmaf(ys[e], op.b_scale[cc], op.b_shift[cc]);
      }
    }
  }
}

struct BnReduceArgs {
  mpose_bn_bwd_reduce_operands op[MPOSE_MAX_GROUP];
  long npix;
  int C, pix_per_block;
  double* partials;       // when non-NULL: [group][block][C][4] per-workgroup sums (no atomics; bn_bwd_reduce_finish_k adds them up)
  // mpose_bn_bwd_reduce_coef_ws: the finishing pass also runs these coefficient jobs (those whose `sums` lie inside a group's sums)
  const mpose_bn_bwd_coef_job* coef_jobs;
  int n_coef, coef_eval;
};

// One channel of a coefficient job from its two sums (bn_bwd_coef_k's arithmetic; `c` is the job's channel index).  Returns c0.
__device__ __forceinline__ double coef_channel(const mpose_bn_bwd_coef_job& j, int c, double sg, double sgx, int eval_mode, double mean, double invstd,
                                               double gamma, double* sgxhat_out) {
  const double n = (double)j.count;
  const double sgxhat = invstd * (sgx - mean * sg);
  const double c0 = gamma * invstd;
  const double c1 = eval_mode ? 0.0 : -c0 * invstd * (sgxhat / n);
  const double c2 = eval_mode ? 0.0 : -c0 * (sg / n);
  j.coef[c] = (float)c0;
  j.coef[j.c_stride + c] = (float)c1;
  j.coef[2 * j.c_stride + c] = (float)c2;
  j.coef[3 * j.c_stride + c] = (float)mean;
  if (j.dgamma != nullptr) { j.dgamma[c] = (float)sgxhat; j.dbeta[c] = (float)sg; }
  if (j.dconv_bias != nullptr) j.dconv_bias[c] = eval_mode ? (float)(c0 * sg) : 0.f;
  *sgxhat_out = sgxhat;
  return c0;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_k(BnReduceArgs a) {
  extern __shared__ double sred[];     // [rows_per_pass][C][4]
  const mpose_bn_bwd_reduce_operands& op = a.op[blockIdx.y];
  const int c4n = a.C >> 2;
  const int rows_per_pass = 256 / c4n;
  const int col4 = threadIdx.x % c4n, row0 = threadIdx.x / c4n;
  const bool active = row0 < rows_per_pass;
  const bool has_b = op.b != nullptr;
  const bool masked = op.a_scale != nullptr;
  // fp64 accumulators (products exact): dbeta = sum g and the shortcut's sum g*b are sums with heavy cancellation -- g is a
  // transposed convolution of zero-mean maps -- and fp32 partial sums showed up as a 2.4x excess over the fp32 reference on
  // exactly those gradients in the round-2 parity bisect.  The pass is HBM-bound; the fp64 adds are free.
  double sga0[4] = {0., 0., 0., 0.}, sga1[4] = {0., 0., 0., 0.}, sg[4] = {0., 0., 0., 0.}, sgb[4] = {0., 0., 0., 0.};
  const long p_begin = (long)blockIdx.x * a.pix_per_block;
  const long p_end = min(a.npix, p_begin + a.pix_per_block);
  if (active) {
    float4 ms = make_float4(0.f, 0.f, 0.f, 0.f), mt = ms;
    if (masked) { ms = *reinterpret_cast<const float4*>(op.a_scale + col4 * 4); mt = *reinterpret_cast<const float4*>(op.a_shift + col4 * 4); }
    auto accumulate = [&](const float4 g, const float4 x, const float4 y) {
      const float gv[4] = {g.x, g.y, g.z, g.w}, xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
      const float msv[4] = {ms.x, ms.y, ms.z, ms.w}, mtv[4] = {mt.x, mt.y, mt.z, mt.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double gd = (double)gv[e];
        const double ga = (masked && !(fmaf(xv[e], msv[e], mtv[e]) > 0.f)) ? 0.0 : gd;
        sga0[e] += ga;
        sga1[e] = fma(ga, (double)xv[e], sga1[e]);
        if (has_b) {
          sg[e] += gd;
          sgb[e] = fma(gd, (double)yv[e], sgb[e]);
        }
      }
    };
    const float4* pg = reinterpret_cast<const float4*>(op.g);
    const float4* pa = reinterpret_cast<const float4*>(op.a);
    const float4* pb = reinterpret_cast<const float4*>(has_b ? op.b : op.a);
    long p = p_begin + row0;
    // four pixels (12 independent 16-byte loads) in flight per thread: the pass is latency-bound otherwise
    for (; p + 3 * rows_per_pass < p_end; p += 4 * rows_per_pass) {
      float4 g[4], x[4], y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long o = (p + (long)u * rows_per_pass) * c4n + col4;
        g[u] = pg[o]; x[u] = pa[o];
        y[u] = has_b ? pb[o] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(g[u], x[u], y[u]);
    }
    for (; p < p_end; p += rows_per_pass) {
      const long o = p * c4n + col4;
      accumulate(pg[o], pa[o], has_b ? pb[o] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    double* d = sred + ((long)row0 * a.C + col4 * 4) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[4 * e] = sga0[e]; d[4 * e + 1] = sga1[e]; d[4 * e + 2] = sg[e]; d[4 * e + 3] = sgb[e]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.C; c += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int r = 0; r < rows_per_pass; ++r) {
      const double* d = sred + ((long)r * a.C + c) * 4;
      s0 += d[0]; s1 += d[1]; s2 += d[2]; s3 += d[3];
    }
    if (a.partials != nullptr) {
      double* d = a.partials + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.C + c) * 4;
      d[0] = s0; d[1] = s1; d[2] = s2; d[3] = s3;
    } else {
      atomicAdd(op.sums + (size_t)c * 4, s0);
      atomicAdd(op.sums + (size_t)c * 4 + 1, s1);
      if (has_b) { atomicAdd(op.sums + (size_t)c * 4 + 2, s2); atomicAdd(op.sums + (size_t)c * 4 + 3, s3); }
    }
  }
}

// sums[c][k] = sum over the launch's workgroups of their partial sums, in a fixed order (deterministic).  A workgroup owns 64
// of the C*4 sums; its four waves each add every fourth workgroup's partial (eight loads in flight per thread), then the four
// slices are added in order through LDS.  (The first version walked all partials serially per thread: 39 us for a 23 us pass.)
// EL entries (of a group's C*4 sums) per workgroup, 256 / EL slices of the partial blocks each.  EL = 64: the columns' shape (C <=
// 192, three groups).  A single wide BatchNorm (C = 256 .. 1024: ChatterboxModel) has 1024 .. 4096 entries: with EL = 64 that is
// 16 .. 64 workgroups walking 512 partials each (40 us for 8 MB); EL = 16 spreads it over four times as many (and slices).
template <int EL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_finish_k(BnReduceArgs a, int n_blocks) {
  constexpr int SL = 256 / EL;
  __shared__ double sl[SL][EL];
  const int el = threadIdx.x % EL, slice = threadIdx.x / EL;
  const int e = blockIdx.x * EL + el;
  const int n_e = a.C * 4;
  double t[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
  if (e < n_e) {
    const size_t stride = (size_t)n_e;
    const double* p = a.partials + (size_t)blockIdx.y * n_blocks * stride + e;
    int b = slice;
    for (; b + 7 * SL < n_blocks; b += 8 * SL) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += p[(size_t)(b + SL * u) * stride];
    }
    for (; b < n_blocks; b += SL) t[0] += p[(size_t)b * stride];
  }
  sl[slice][el] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  __shared__ double fin[EL];
  if (slice == 0 && e < n_e) {
    double s;
    if constexpr (SL == 4) {
      s = (sl[0][el] + sl[1][el]) + (sl[2][el] + sl[3][el]);
    } else {
      s = 0.0;
#pragma unroll
      for (int k = 0; k < SL; k += 4) s += (sl[k][el] + sl[k + 1][el]) + (sl[k + 2][el] + sl[k + 3][el]);
    }
    a.op[blockIdx.y].sums[e] = s;
    fin[el] = s;
  }
  if (a.coef_jobs == nullptr) return;       // (uniform)
  // The workgroup holds all four sums of EL / 4 channels: it runs the coefficient jobs that read them (bn_bwd_coef_k's launch, 5-6 us
  // of pure latency behind this one, 22 times per step) -- same sums, same arithmetic, same results.
  __syncthreads();
  if (threadIdx.x < EL / 4) {
    const int ch = blockIdx.x * (EL / 4) + (int)threadIdx.x;       // channel of the group's sums
    if (ch < a.C) {
      const double* base = a.op[blockIdx.y].sums;
      for (int i = 0; i < a.n_coef; ++i) {
        const mpose_bn_bwd_coef_job& j = a.coef_jobs[i];
        if (j.sums_stride != 4) continue;
        const long rel = (long)(j.sums - base);                     // in doubles
        if (rel < 0 || (rel & 3)) continue;
        const long c = (long)ch - (rel >> 2);
        if (c < 0 || c >= j.C) continue;
        const double* f = fin + threadIdx.x * 4;
        const double sg = j.sg_col == 0 ? f[0] : (j.sg_col == 1 ? f[1] : (j.sg_col == 2 ? f[2] : f[3]));
        const double sgx = j.which == 0 ? f[0] : (j.which == 1 ? f[1] : (j.which == 2 ? f[2] : f[3]));
        double sgxhat;
        coef_channel(j, (int)c, sg, sgx, a.coef_eval, (double)j.mean[c], (double)j.invstd[c], (double)j.gamma[c], &sgxhat);
      }
    }
  }
}

// eval_mode: the forward normalised with the RUNNING statistics (constants), so dx = gamma*invstd*g (c1 = c2 = 0);
// dgamma / dbeta keep their formulas with mean / invstd = the running ones (bn_finalize_k stores them in eval mode too).
__global__ __launch_bounds__(1024) void bn_bwd_coef_k(const mpose_bn_bwd_coef_job* __restrict__ jobs, int mode) {
  __shared__ double sh[1024 * 4];
  const mpose_bn_bwd_coef_job j = jobs[blockIdx.x];
  const int eval_mode = mode & 1;
  const bool from_part = j.part != nullptr && !(mode & 2);       // MPOSE_CONV_STATS_PART rows instead of `sums`
  const double n = (double)j.count;
  const int nth = (int)blockDim.x;
  const bool want_bound = (mode & 8) && j.bound_out != nullptr;
  const float gmax = want_bound ? amax_gather(j.g_amax) : 0.f;      // (whole waves call it: uniform)
  float bound = 0.f;
  // (gridDim.y workgroups share a job's channels, 32-channel granules)
  const int gran = ((j.C + 31) / 32 + (int)gridDim.y - 1) / (int)gridDim.y * 32;
  const int c_lo = (int)blockIdx.y * gran, c_hi = (c_lo + gran < j.C) ? c_lo + gran : j.C;
  for (int cblk = c_lo; cblk < c_hi; cblk += nth) {
    const int c = cblk + (int)threadIdx.x;
    double ps[4] = {0.0, 0.0, 0.0, 0.0};
    // (the channel's constants are requested before the row sums are: see bn_finalize_job)
    const bool cv = c < c_hi;
    const float mean_c = cv ? j.mean[c] : 0.f, invstd_c = cv ? j.invstd[c] : 0.f, gamma_c = cv ? j.gamma[c] : 0.f;
    if (from_part) {       // (uniform per workgroup: the helper contains barriers)
      const int nc = c_hi - cblk < nth ? c_hi - cblk : nth;
      if (j.sums_stride == 4) {
        reduce_part_rows<4>(j.part, j.n_part, j.part_ld, cblk, nc, sh, ps);
      } else {
        double p2[2] = {0.0, 0.0};
        reduce_part_rows<2>(j.part, j.n_part, j.part_ld, cblk, nc, sh, p2);
        ps[0] = p2[0]; ps[1] = p2[1];
      }
    }
    if (c >= c_hi) continue;
    // (selects, not ps[runtime index]: a dynamically indexed register array goes to scratch)
    auto pick = [&](int k) { return k == 0 ? ps[0] : (k == 1 ? ps[1] : (k == 2 ? ps[2] : ps[3])); };
    const double sg = from_part ? pick(j.sg_col) : j.sums[(size_t)c * j.sums_stride + j.sg_col];
    const double sgx = from_part ? pick(j.which) : j.sums[(size_t)c * j.sums_stride + j.which];
    double sgxhat;
    const double c0 = coef_channel(j, c, sg, sgx, eval_mode, (double)mean_c, (double)invstd_c, (double)gamma_c, &sgxhat);
    if (want_bound)       // |c0 g + c1 (x - mean) + c2| <= |c0| (gmax + sqrt(n) |mean(g xhat)| + |mean g|):  |x - mean| invstd <= sqrt(n)
      bound = fmaxf(bound, (float)(fabs(c0) * ((double)gmax + (eval_mode ? 0.0 : sqrt(n) * fabs(sgxhat / n) + fabs(sg / n)))));
  }
  if (want_bound) block_amax_commit_one(bound * 1.0001f, j.bound_out);       // (uniform; a bound that is not finite -> +inf, like a NaN maximum)
}

struct BnApplyArgs {
  mpose_bn_bwd_apply_operands op[MPOSE_MAX_GROUP];
  long total4;
  int C;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_k(BnApplyArgs a) {
  const mpose_bn_bwd_apply_operands& op = a.op[blockIdx.y];
  const int c4n = a.C >> 2;
  const bool has_b = op.b != nullptr;
  float amax_a = 0.f, amax_b = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total4; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const float4 g = reinterpret_cast<const float4*>(op.g)[i];
    {
      const float4 x = reinterpret_cast<const float4*>(op.a)[i];
      float4 ga = g;
      if (op.a_scale != nullptr) {
        const float4 ms = *reinterpret_cast<const float4*>(op.a_scale + c), mt = *reinterpret_cast<const float4*>(op.a_shift + c);
        if (!(fmaf(x.x, ms.x, mt.x) > 0.f)) ga.x = 0.f;
        if (!(fmaf(x.y, ms.y, mt.y) > 0.f)) ga.y = 0.f;
        if (!(fmaf(x.z, ms.z, mt.z) > 0.f)) ga.z = 0.f;
        if (!(fmaf(x.w, ms.w, mt.w) > 0.f)) ga.w = 0.f;
      }
      const float4 k0 = *reinterpret_cast<const float4*>(op.coef_a + c);
      const float4 k1 = *reinterpret_cast<const float4*>(op.coef_a + a.C + c);
      const float4 k2 = *reinterpret_cast<const float4*>(op.coef_a + 2 * a.C + c);
      const float4 mu = *reinterpret_cast<const float4*>(op.coef_a + 3 * a.C + c);
      float4 o;
      o.x = fmaf(k1.x, x.x - mu.x, fmaf(k0.x, ga.x, k2.x)); o.y = fmaf(k1.y, x.y - mu.y, fmaf(k0.y, ga.y, k2.y));
      o.z = fmaf(k1.z, x.z - mu.z, fmaf(k0.z, ga.z, k2.z)); o.w = fmaf(k1.w, x.w - mu.w, fmaf(k0.w, ga.w, k2.w));
      reinterpret_cast<float4*>(op.da)[i] = o;
      amax_a = fmaxf(fmaxf(amax_a, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (has_b) {
      const float4 x = reinterpret_cast<const float4*>(op.b)[i];
      const float4 k0 = *reinterpret_cast<const float4*>(op.coef_b + c);
      const float4 k1 = *reinterpret_cast<const float4*>(op.coef_b + a.C + c);
      const float4 k2 = *reinterpret_cast<const float4*>(op.coef_b + 2 * a.C + c);
      const float4 mu = *reinterpret_cast<const float4*>(op.coef_b + 3 * a.C + c);
      float4 o;
      o.x = fmaf(k1.x, x.x - mu.x, fmaf(k0.x, g.x, k2.x)); o.y = fmaf(k1.y, x.y - mu.y, fmaf(k0.y, g.y, k2.y));
      o.z = fmaf(k1.z, x.z - mu.z, fmaf(k0.z, g.z, k2.z)); o.w = fmaf(k1.w, x.w - mu.w, fmaf(k0.w, g.w, k2.w));
      reinterpret_cast<float4*>(op.db)[i] = o;
      amax_b = fmaxf(fmaxf(amax_b, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (op.da_amax != nullptr) block_amax_commit(amax_a, op.da_amax);
  if (has_b && op.db_amax != nullptr) { __syncthreads(); block_amax_commit(amax_b, op.db_amax); }
}

// out = relu(a*scale + shift)  (the stem's BN + ReLU, materialised because the stage input is shared)
__global__ __launch_bounds__(256) void bn_relu_k(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                 float* __restrict__ out, long total4, int C) {
  const int c4n = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 s = *reinterpret_cast<const float4*>(scale + c), t = *reinterpret_cast<const float4*>(shift + c);
    reinterpret_cast<float4*>(out)[i] = make_float4(fmaxf(fmaf(v.x, s.x, t.x), 0.f), fmaxf(fmaf(v.y, s.y, t.y), 0.f),
                                                    fmaxf(fmaf(v.z, s.z, t.z), 0.f), fmaxf(fmaf(v.w, s.w, t.w), 0.f));
  }
}
// gm = g where y > 0 else 0
__global__ __launch_bounds__(256) void relu_bwd_k(const float4* __restrict__ g, const float4* __restrict__ y, float4* __restrict__ gm, long total4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const float4 a = g[i], b = y[i];
    gm[i] = make_float4(b.x > 0.f ? a.x : 0.f, b.y > 0.f ? a.y : 0.f, b.z > 0.f ? a.z : 0.f, b.w > 0.f ? a.w : 0.f);
  }
}

// passes that end with one atomic per workgroup (an amax output) run at most 512 workgroups per tensor: 32 per sub-slot
inline int amax_grid(int blocks, bool amax) { return (amax && blocks > 512) ? 512 : blocks; }

inline int grid_for(long work_items, int per_block) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;      // 256 CUs x 8 blocks, grid-stride beyond that
  return (int)b;
}

}  // namespace
}  // namespace mpose

using namespace mpose;

extern "C" int mpose_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(mpose_conv_geom);
    case 1: return (int)sizeof(mpose_conv_operands);
    case 2: return (int)sizeof(mpose_wgrad_operands);
    case 3: return (int)sizeof(mpose_pack_job);
    case 4: return (int)sizeof(mpose_unpack_job);
    case 5: return (int)sizeof(mpose_bn_job);
    case 6: return (int)sizeof(mpose_bn_bwd_coef_job);
    case 7: return (int)sizeof(mpose_bn_add_operands);
    case 8: return (int)sizeof(mpose_bn_bwd_reduce_operands);
    case 9: return (int)sizeof(mpose_bn_bwd_apply_operands);
    case 10: return (int)sizeof(mpose_split_h2_operands);
    case 11: return (int)sizeof(mpose_sgd_job);
    case 12: return (int)sizeof(mpose_absmax_operands);
    default: return -1;
  }
}

// train: bit 0 = batch statistics; bit 1 = the jobs carry MPOSE_CONV_STATS_PART rows (1024 threads per job); bit 2 = write the jobs' bounds
extern "C" int mpose_bn_finalize(const mpose_bn_job* jobs_dev, int n_jobs, int train, float eps, float momentum, void* stream) {
  if (n_jobs <= 0) return 0;
  // (six workgroups share a job's channels in 32-channel granules: one granule each for the 192-channel layers, four busy for 128)
  if (train & 2) launch(bn_finalize_k, dim3(dim3(n_jobs, kPartSplit)), dim3(1024), 0, (hipStream_t)stream, jobs_dev, train & 7, eps, momentum);
  else launch(bn_finalize_k, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, train & 7, eps, momentum);
  return launch_status();
}

extern "C" int mpose_bn_add_fwd(const mpose_bn_add_operands* ops, int n_groups, int pixels_per_image, int B, int C, int layout,
                                int c_keep, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 3)) return MPOSE_EINVAL;
  BnAddArgs a{};
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.C = C; a.P = pixels_per_image; a.c_keep = c_keep;
  a.total4 = (long)B * pixels_per_image * C / 4;
  if (a.total4 == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (layout == 0) {
    launch(bn_add_nhwc_k<true>, dim3(dim3(amax_grid(grid_for(a.total4, 256), ops[0].out_amax != nullptr), n_groups)), dim3(256), 0, s, a);
  } else if (layout == 2) {             // no ReLU on branch a: bn2(x) + bn_d(shortcut) of a ResNet downsample block
    launch(bn_add_nhwc_k<false>, dim3(dim3(amax_grid(grid_for(a.total4, 256), ops[0].out_amax != nullptr), n_groups)), dim3(256), 0, s, a);
  } else {
    if (c_keep < 1 || c_keep > C) return MPOSE_EINVAL;
    launch(bn_add_nchw_k, dim3(dim3(grid_for((long)B * pixels_per_image, 256), n_groups)), dim3(256), 0, s, a, B);
  }
  return launch_status();
}

extern "C" int mpose_bn_bwd_reduce(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C,
                                   int layout, int c_keep, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 3) || C > 1024 || layout != 0) return MPOSE_EINVAL;
  BnReduceArgs a{};
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.npix = (long)B * pixels_per_image;
  if (a.npix == 0) return 0;
  a.C = C;
  const int rows_per_pass = 256 / (C / 4);
  int blocks = grid_for(a.npix, rows_per_pass * 16);     // (halving this to 8 pixel rows per thread was measured SLOWER: 61 vs 45 us, the fp64 atomics of 2x the workgroups)
  if (blocks > 512) blocks = 512;
  a.pix_per_block = (int)((a.npix + blocks - 1) / blocks);
  const int lds = rows_per_pass * C * 4 * 8;
  launch(bn_bwd_reduce_k, dim3(dim3(blocks, n_groups)), dim3(256), lds, (hipStream_t)stream, a);
  return launch_status();
}

// workgroups per group of the partial-sum form: enough to fill the chip ~4x over all groups, few enough that the partials stay
// ~1 MB per group (1024 per group cost more in the finishing pass than it gained in the main one)
static bool reduce_wide(int n_groups, int C) { return n_groups == 1 && C >= 256; }      // one wide BatchNorm (see bn_bwd_reduce_finish_k)

static long reduce_ws_blocks(int n_groups, long npix, int C) {
  // (one group that is not a wide BatchNorm -- the feature extractor's nodes: 256 workgroups, one per CU; with 1024 the finishing
  //  pass walked four times the partials for nothing gained in the main one: -0.1 ... -0.17 ms per step, profiles/r5_ab_sweeps.txt)
  constexpr long cap1 = 256L;
  const long cap = n_groups >= 3 ? 384 : (n_groups == 2 ? 512 : (reduce_wide(n_groups, C) ? 1024 : cap1));
  // (wide: a workgroup's pass covers 256 / (C/4) <= 4 pixel rows at a time; 32 pixels each keeps >= 4 workgroups per CU in flight)
  long blocks = reduce_wide(n_groups, C) ? (npix + 31) / 32 : (npix + 63) / 64;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return blocks;
}

extern "C" int64_t mpose_bn_bwd_reduce_ws_bytes(int n_groups, int pixels_per_image, int B, int C) {
  return (int64_t)n_groups * reduce_ws_blocks(n_groups, (long)B * pixels_per_image, C) * C * 4 * 8;
}

// Same sums as mpose_bn_bwd_reduce, WRITTEN (not accumulated), through per-workgroup partial sums in a caller-provided workspace
// of at least mpose_bn_bwd_reduce_ws_bytes(): no atomics, more and smaller workgroups, deterministic order.
static int reduce_ws_launch(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C, void* workspace,
                            int64_t workspace_bytes, const mpose_bn_bwd_coef_job* coef_jobs, int n_coef, int coef_eval, void* stream);

extern "C" int mpose_bn_bwd_reduce_ws(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  return reduce_ws_launch(ops, n_groups, pixels_per_image, B, C, workspace, workspace_bytes, nullptr, 0, 0, stream);
}

// mpose_bn_bwd_reduce_ws + mpose_bn_bwd_coef (mode: bit 0 = eval_mode; jobs read from `sums`, no bounds) as two launches instead
// of three: the finishing pass of the reduction runs the coefficient jobs whose `sums` it has just completed.
extern "C" int mpose_bn_bwd_reduce_coef_ws(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C,
                                           void* workspace, int64_t workspace_bytes, const mpose_bn_bwd_coef_job* coef_jobs_dev,
                                           int n_coef_jobs, int coef_mode, void* stream) {
  if (!coef_jobs_dev || n_coef_jobs < 1 || (coef_mode & ~1)) return MPOSE_EINVAL;
  if ((long)B * pixels_per_image == 0) return MPOSE_EINVAL;       // (no finishing pass would run the jobs)
  return reduce_ws_launch(ops, n_groups, pixels_per_image, B, C, workspace, workspace_bytes, coef_jobs_dev, n_coef_jobs, coef_mode & 1, stream);
}

static int reduce_ws_launch(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C, void* workspace,
                            int64_t workspace_bytes, const mpose_bn_bwd_coef_job* coef_jobs, int n_coef, int coef_eval, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 3) || C > 1024 || !workspace) return MPOSE_EINVAL;
  if (workspace_bytes < mpose_bn_bwd_reduce_ws_bytes(n_groups, pixels_per_image, B, C)) return MPOSE_EINVAL;
  BnReduceArgs a{};
  a.coef_jobs = coef_jobs; a.n_coef = n_coef; a.coef_eval = coef_eval;
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.npix = (long)B * pixels_per_image;
  if (a.npix == 0) return 0;
  a.C = C;
  const long blocks = reduce_ws_blocks(n_groups, a.npix, C);
  a.pix_per_block = (int)((a.npix + blocks - 1) / blocks);
  a.partials = reinterpret_cast<double*>(workspace);
  const int rows_per_pass = 256 / (C / 4);
  const int lds = rows_per_pass * C * 4 * 8;
  hipStream_t s = (hipStream_t)stream;
  launch(bn_bwd_reduce_k, dim3(dim3((unsigned)blocks, n_groups)), dim3(256), lds, s, a);
  if (reduce_wide(n_groups, C)) launch(bn_bwd_reduce_finish_k<16>, dim3(dim3((C * 4 + 15) / 16, n_groups)), dim3(256), 0, s, a, (int)blocks);
  else launch(bn_bwd_reduce_finish_k<64>, dim3(dim3((C * 4 + 63) / 64, n_groups)), dim3(256), 0, s, a, (int)blocks);
  return launch_status();
}

// mode: bit 0 = eval_mode, bit 1 = ignore the jobs' partial rows (read `sums`), bit 2 = 1024 threads per job (jobs with partial rows),
// bit 3 = write the jobs' bounds (bound_out)
extern "C" int mpose_bn_bwd_coef(const mpose_bn_bwd_coef_job* jobs_dev, int n_jobs, int mode, void* stream) {
  if (n_jobs <= 0) return 0;
  if (mode & 4) launch(bn_bwd_coef_k, dim3(dim3(n_jobs, kPartSplit)), dim3(1024), 0, (hipStream_t)stream, jobs_dev, mode & 11);
  else launch(bn_bwd_coef_k, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, mode & 11);
  return launch_status();
}

extern "C" int mpose_bn_bwd_apply(const mpose_bn_bwd_apply_operands* ops, int n_groups, int pixels_per_image, int B, int C,
                                  int layout, int c_keep, void* stream) {
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || (C & 3) || layout != 0) return MPOSE_EINVAL;
  BnApplyArgs a{};
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.C = C;
  a.total4 = (long)B * pixels_per_image * C / 4;
  if (a.total4 == 0) return 0;
  launch(bn_bwd_apply_k, dim3(dim3(amax_grid(grid_for(a.total4, 256), ops[0].da_amax != nullptr), n_groups)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int mpose_bn_relu_fwd(const float* x, const float* scale, const float* shift, float* out, int64_t n, int C, void* stream) {
  if (n < 0 || (C & 3) || C <= 0 || (n % C)) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(bn_relu_k, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out, n / 4, C);
  return launch_status();
}

extern "C" int mpose_relu_bwd(const float* g, const float* y, float* gm, int64_t n, void* stream) {
  if (n < 0 || (n & 3)) return MPOSE_EINVAL;
  if (n == 0) return 0;
  launch(relu_bwd_k, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(y),
                                                                    reinterpret_cast<float4*>(gm), n / 4);
  return launch_status();
}
