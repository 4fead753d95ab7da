// Implicit-GEMM convolution engine for gfx950 on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD, bitwise an fp32 fma chain).
//
// Replaces every Conv2d / ConvTranspose2d of reference src/margipose/models/margipose_model.py
// (:33, :67-68, :73-74, :79-82), their data-gradients and their weight-gradients.  One kernel
// family serves all of them through a "tap list" geometry (include/margipose_hip.h):
//     out[slot -> (oy,ox)][n] = sum_{tap} sum_{k} in[slot*in_mul + (dy,dx)][k] * Wp[widx][k][n]
//   * 3x3 / 1x1 stride 1            : one class, in_mul = out_mul = 1
//   * 3x3 / 1x1 stride 2 (down)     : one class, in_mul = 2
//   * transposed 3x3 / 1x1 stride 2 : four output-parity classes, out_mul = 2, 1/2/2/4 taps each
//     (no multiplications by the zeros a dilated formulation would insert)
//   * data-gradients are the same kernel with (dy,dx) negated and the roles of Cin/Cout swapped in
//     the packed weights; the gradient of a down conv is the "up" form and vice versa.
// Fusions: the ResidualBlock's 1x1 shortcut runs in the same launch as its 3x3 (second
// accumulator fed from the centre tap's LDS tile); BatchNorm batch statistics (sum, sum of squares,
// fp64 atomics) are reduced in the epilogue; BN+ReLU of the producer is applied while staging the
// input tile; the ReLU mask and the BN-backward reductions are folded into the dgrad epilogue;
// the xy/zy/xz columns of a stage run as one grouped launch (blockIdx.z).
//
// Layout: activations NHWC fp32, weights pre-packed [widx][K/4][Npad][4] so that both MFMA
// operands are fetched from LDS with one ds_read_b128 per four MFMAs:
//   lane (i = l&31, h = l>>5) holds A[pixel i][k = q*8 + h*4 + j] and B[k][n = l&31], j = 0..3.
// LDS A tile rows are padded to 36 floats (144 B): the 16 lanes of a ds_read_b128 group then hit
// 16 distinct 16-byte slots (9*i mod 16 is a bijection), so the reads are conflict-free.
#include "common.h"
#include <cstdlib>

namespace mpose {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FastDiv {
  unsigned mul, shift;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  unsigned l = 0;
  while ((1u << l) < d) ++l;
  f.shift = l;
  f.mul = (unsigned)(((uint64_t)((1ull << l) - d) << 32) / d) + 1u;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) { return (__umulhi(n, f.mul) + n) >> f.shift; }

constexpr int KC = 32;            // channels per K-chunk
constexpr int A_STRIDE = 36;      // padded floats per A-tile row

struct ConvArgs {
  mpose_conv_geom g;
  mpose_conv_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gw, div_ghw;
  int M;                          // slots per class = B*GH*GW
  int n_mtiles;
  int flags;
};

// Row of the 32x32 accumulator held in register r of lane-half h.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int WM, int WN, int RM, int RN, bool ACC1>
__global__ __launch_bounds__(256, 2) void conv_igemm_k(ConvArgs a) {
  constexpr int BM = 32 * WM * RM;
  constexpr int BN = 32 * WN * RN;
  constexpr int A_LOADS = BM / 32;           // float4 per thread per A tile
  constexpr int W_LOADS = BN / 32;           // float4 per thread per W tile
  constexpr int A_TILE = BM * A_STRIDE;      // floats
  constexpr int W_TILE = BN * KC;            // floats
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                          // [2][A_TILE]
  float* sW = smem + 2 * A_TILE;             // [2][W_TILE]
  int* sTaps = reinterpret_cast<int*>(smem + 2 * A_TILE + 2 * W_TILE);   // [MPOSE_MAX_TAPS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const mpose_conv_geom& g = a.g;
  const int cls = blockIdx.x / a.n_mtiles;
  const int m0 = (blockIdx.x - cls * a.n_mtiles) * BM;
  const int n0 = blockIdx.y * BN;
  const mpose_conv_operands& op = a.op[blockIdx.z];
  const int n_taps = g.cls[cls].n_taps;
  if (tid < MPOSE_MAX_TAPS) {
    const mpose_tap t = g.cls[cls].taps[tid];
    sTaps[tid] = (int)(unsigned char)t.dy | ((int)(unsigned char)t.dx << 8) | ((int)(unsigned char)t.widx << 16) | ((int)(unsigned char)t.acc << 24);
  }

  // ---- per-thread staging state, fixed for the whole K loop ----
  // row_off[j]: element offset of the slot's anchor pixel (+ this thread's 4-channel column);
  // row_taps[j]: bit t set when tap t of this class reads an in-bounds pixel for that row.
  const int a_col4 = tid & 7;
  __syncthreads();   // sTaps visible
  unsigned row_off[A_LOADS], row_taps[A_LOADS];
#pragma unroll
  for (int j = 0; j < A_LOADS; ++j) {
    const int row = (tid >> 3) + 32 * j;
    const unsigned m = (unsigned)(m0 + row);
    row_off[j] = 0; row_taps[j] = 0;
    if ((int)m < a.M) {
      const unsigned b = fdiv(m, a.div_ghw);
      const unsigned rem = m - b * (unsigned)(g.GH * g.GW);
      const unsigned gy = fdiv(rem, a.div_gw);
      const unsigned gx = rem - gy * (unsigned)g.GW;
      const int iy0 = (int)gy * g.in_mul, ix0 = (int)gx * g.in_mul;
      row_off[j] = ((b * (unsigned)g.IH + (unsigned)iy0) * (unsigned)g.IW + (unsigned)ix0) * (unsigned)g.Cin + (unsigned)(a_col4 * 4);
      for (int t = 0; t < n_taps; ++t) {
        const int tp = sTaps[t];
        const int iy = iy0 + (int)(signed char)(tp & 0xff), ix = ix0 + (int)(signed char)((tp >> 8) & 0xff);
        if (iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) row_taps[j] |= 1u << t;
      }
    }
  }
  const int n_chunks = g.Cin / KC;
  const int n_iter = n_chunks * n_taps;
  const int k4_total = g.Cin >> 2;
  const unsigned w_lane0 = (unsigned)(((tid / BN) * g.Npad0 + (tid % BN)) * 4);                       // Npad0 == Npad1 when ACC1
  const unsigned w_lane1 = (unsigned)((((tid + 256) / BN) * g.Npad0 + ((tid + 256) % BN)) * 4);

  f32x16 acc0[RM][RN];
  f32x16 acc1[RM][RN];
#pragma unroll
  for (int rm = 0; rm < RM; ++rm)
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[rm][rn][r] = 0.0f; acc1[rm][rn][r] = 0.0f; }
    }

  float4 ra[A_LOADS];
  float4 rw0 = make_float4(0.f, 0.f, 0.f, 0.f), rw1 = rw0;   // named (not an array): keeps the W staging in VGPRs
  static_assert(W_LOADS <= 2, "W tile staging assumes at most two float4 per thread");
  float4 rsc = make_float4(1.f, 1.f, 1.f, 1.f), rsh = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned ra_ok = 0;                      // bit j: row j of the staged tile is a real (in-bounds) pixel
  const bool pro = op.in_scale != nullptr;

  // Branch-free issue of every global load of tile `it`: the tap contributes ONE wave-uniform element offset,
  // per-row validity is a precomputed bit; invalid rows read element 0 and are zeroed at LDS-store time.
  auto load_regs = [&](int it) {
    const int c = it / n_taps;
    const int t = it - c * n_taps;
    const int tp = __builtin_amdgcn_readfirstlane(sTaps[t]);
    const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
    const int widx = (tp >> 16) & 0xff;
    const bool second = ACC1 && ((tp >> 24) & 0xff);
    const unsigned tap_off = (unsigned)((dy * g.IW + dx) * g.Cin + c * KC);
    ra_ok = 0;
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      const bool ok = (row_taps[j] >> t) & 1u;
      const unsigned off = ok ? row_off[j] + tap_off : (unsigned)(a_col4 * 4);
      ra[j] = *reinterpret_cast<const float4*>(op.in + off);
      ra_ok |= (ok ? 1u : 0u) << j;
    }
    const float* wsrc = (second ? op.w1 : op.w0) + ((long)(widx * k4_total + c * (KC / 4)) * g.Npad0 + n0) * 4;
    rw0 = *reinterpret_cast<const float4*>(wsrc + w_lane0);
    if (W_LOADS > 1) rw1 = *reinterpret_cast<const float4*>(wsrc + w_lane1);
    if (pro) {
      rsc = *reinterpret_cast<const float4*>(op.in_scale + c * KC + a_col4 * 4);
      rsh = *reinterpret_cast<const float4*>(op.in_shift + c * KC + a_col4 * 4);
    }
  };
  // BN + ReLU of the producing layer (when requested) and the zero padding are applied here, after the
  // MFMA phase, so the loads above are never waited for early.
  auto store_lds = [&](int buf) {
    float* dA = sA + buf * A_TILE;
    float* dW = sW + buf * W_TILE;
#pragma unroll
    for (int j = 0; j < A_LOADS; ++j) {
      const int row = (tid >> 3) + 32 * j;
      float4 v = ra[j];
      if (pro) {
        v.x = fmaxf(fmaf(v.x, rsc.x, rsh.x), 0.f); v.y = fmaxf(fmaf(v.y, rsc.y, rsh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, rsc.z, rsh.z), 0.f); v.w = fmaxf(fmaf(v.w, rsc.w, rsh.w), 0.f);
      }
      if (!((ra_ok >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(dA + row * A_STRIDE + a_col4 * 4) = v;
    }
    *reinterpret_cast<float4*>(dW + tid * 4) = rw0;
    if (W_LOADS > 1) *reinterpret_cast<float4*>(dW + (tid + 256) * 4) = rw1;
  };
  // One (chunk, tap) tile = a 32-long fp32 FMA chain per output element, accumulated into a FRESH register
  // tile and then added to the running sum.  Error grows like sqrt(32) + sqrt(#tiles) ulps instead of
  // sqrt(K) for one K-long chain (K = 1152 for a 3x3 over 128 channels): ~5x closer to the fp64 result, which
  // matters because every rounding-induced ReLU-mask flip costs ~1e-3 relative error in the gradients.
  // LDS fragments of k-group q+1 are requested before the MFMAs of group q are issued.
  auto compute = [&](int buf, bool second) {
    const float* cA = sA + buf * A_TILE + (wm * RM * 32 + li) * A_STRIDE + lh * 4;
    const float* cW = sW + buf * W_TILE + (lh * BN + wn * RN * 32 + li) * 4;
    f32x16 part[RM][RN];
    float4 fa[2][RM], fb[2][RN];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) fa[0][rm] = *reinterpret_cast<const float4*>(cA + rm * 32 * A_STRIDE);
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) fb[0][rn] = *reinterpret_cast<const float4*>(cW + rn * 32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q < 3) {
#pragma unroll
        for (int rm = 0; rm < RM; ++rm) fa[nxt][rm] = *reinterpret_cast<const float4*>(cA + rm * 32 * A_STRIDE + (q + 1) * 8);
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) fb[nxt][rn] = *reinterpret_cast<const float4*>(cW + ((q + 1) * 2 * BN + rn * 32) * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int rm = 0; rm < RM; ++rm)
#pragma unroll
          for (int rn = 0; rn < RN; ++rn) {
            f32x16 c;
            if (q == 0 && e == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) c[r] = 0.0f;
            } else {
              c = part[rm][rn];
            }
            const float av = e == 0 ? fa[cur][rm].x : (e == 1 ? fa[cur][rm].y : (e == 2 ? fa[cur][rm].z : fa[cur][rm].w));
            const float bv = e == 0 ? fb[cur][rn].x : (e == 1 ? fb[cur][rn].y : (e == 2 ? fb[cur][rn].z : fb[cur][rn].w));
            part[rm][rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        if (ACC1 && second) acc1[rm][rn] += part[rm][rn];
        else acc0[rm][rn] += part[rm][rn];
      }
  };

  // ---- main loop: the loads of tile it+1 are in flight across the MFMAs of tile it; one barrier per tile ----
  if (n_iter > 0) {
    load_regs(0);
    store_lds(0);
  }
  __syncthreads();
#ifdef MPOSE_ABLATE
  const int abl = a.flags >> 8;
  if (abl & 16) {   // experiment: distinct static priorities for the blocks that share a CU
    const unsigned flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned pr = (flat >> 8) % 3u;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
  }
  if (abl & 32) {   // experiment: one-time stagger
    const unsigned flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned pr = (flat >> 8) % 3u;
    for (unsigned k = 0; k < pr * 24; ++k) __builtin_amdgcn_s_sleep(127);
  }
  for (int it = 0; it < n_iter; ++it) {
    const int buf = it & 1;
    const int nx = it + 1 < n_iter ? it + 1 : it;
    if (!(abl & 1)) load_regs(nx);
    const int c = it / n_taps;
    const bool second = ACC1 && ((__builtin_amdgcn_readfirstlane(sTaps[it - c * n_taps]) >> 24) & 0xff);
    if (!(abl & 8)) compute(buf, second);
    if (!(abl & 2)) store_lds(buf ^ 1);
    if (!(abl & 4)) __syncthreads();
  }
#else
  for (int it = 0; it < n_iter; ++it) {
    const int buf = it & 1;
    const int nx = it + 1 < n_iter ? it + 1 : it;        // always prefetch (the last one is a harmless repeat): no branch
    load_regs(nx);
    const int c = it / n_taps;
    const bool second = ACC1 && ((__builtin_amdgcn_readfirstlane(sTaps[it - c * n_taps]) >> 24) & 0xff);
    compute(buf, second);
    store_lds(buf ^ 1);
    __syncthreads();
  }
#endif

  // ---- epilogue ----
  float* sRed = smem;    // [2 sets][4 waves][RN*32][2] floats, pipeline LDS is free after the last barrier
  const int oyc = g.cls[cls].oy, oxc = g.cls[cls].ox;
  // output pixel offsets of this lane's 16 rows per rm
#pragma unroll
  for (int set = 0; set < (ACC1 ? 2 : 1); ++set) {
    float* outp = set ? op.out1 : op.out0;
    const int cout = set ? g.Cout1 : g.Cout0;
    double* stats = set ? op.stats1 : op.stats0;
    const bool masked = (set == 0) && op.mask_src != nullptr;
    const bool accumulate = (set == 0) && (a.flags & 1);
    float csum[RN], csq[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) { csum[rn] = 0.f; csq[rn] = 0.f; }
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * RM + rm) * 32 + acc_row(r, lh);
        const unsigned m = (unsigned)(m0 + row);
        const bool row_ok = (int)m < a.M;
        long pix = 0;
        if (row_ok) {
          const unsigned b = fdiv(m, a.div_ghw);
          const unsigned rem = m - b * (unsigned)(g.GH * g.GW);
          const unsigned gy = fdiv(rem, a.div_gw);
          const unsigned gx = rem - gy * (unsigned)g.GW;
          pix = ((long)b * g.OH + (gy * g.out_mul + oyc)) * g.OW + (gx * g.out_mul + oxc);
        }
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
          const int n = n0 + (wn * RN + rn) * 32 + li;
          float v = set ? acc1[rm][rn][r] : acc0[rm][rn][r];
          if (row_ok && n < cout) {
            const long o = pix * cout + n;
            float second_factor = v;
            if (masked) {
              const float src = op.mask_src[o];
              if (!(fmaf(src, op.mask_scale[n], op.mask_shift[n]) > 0.f)) v = 0.f;
              second_factor = src;
            }
            if (accumulate) v += outp[o];
            outp[o] = v;
            csum[rn] += v;
            csq[rn] = fmaf(v, second_factor, csq[rn]);
          }
        }
      }
    }
    if (stats != nullptr) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        csum[rn] += __shfl_xor(csum[rn], 32, 64);
        csq[rn] += __shfl_xor(csq[rn], 32, 64);
        if (lh == 0) {
          float* d = sRed + ((set * 4 + wave) * (RN * 32) + rn * 32 + li) * 2;
          d[0] = csum[rn]; d[1] = csq[rn];
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int set = 0; set < (ACC1 ? 2 : 1); ++set) {
    double* stats = set ? op.stats1 : op.stats0;
    const int cout = set ? g.Cout1 : g.Cout0;
    if (stats != nullptr && tid < BN) {
      const int cwn = tid / (RN * 32), cc = tid - cwn * (RN * 32);
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        const float* d = sRed + ((set * 4 + (w * WN + cwn)) * (RN * 32) + cc) * 2;
        s += d[0]; q += d[1];
      }
      const int n = n0 + tid;
      if (n < cout) {
        atomicAdd(stats + (size_t)n * 2, (double)s);
        atomicAdd(stats + (size_t)n * 2 + 1, (double)q);
      }
    }
  }
}

template <int WM, int WN, int RM, int RN, bool ACC1>
int launch_conv(const ConvArgs& a, int n_groups, hipStream_t s) {
  constexpr int BM = 32 * WM * RM, BN = 32 * WN * RN;
  int lds = (2 * BM * A_STRIDE + 2 * BN * KC) * 4 + MPOSE_MAX_TAPS * 4;
  static const int extra_lds = getenv("MPOSE_DEBUG_EXTRA_LDS") ? atoi(getenv("MPOSE_DEBUG_EXTRA_LDS")) : 0;   // occupancy experiments
  lds += extra_lds;
  const int cmax = a.g.Cout1 > a.g.Cout0 ? a.g.Cout1 : a.g.Cout0;
  dim3 grid(a.n_mtiles * a.g.n_classes, (cmax + BN - 1) / BN, n_groups);
  conv_igemm_k<WM, WN, RM, RN, ACC1><<<grid, 256, lds, s>>>(a);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dWp[split][widx][k/4][n][4] = sum_{slots in split} X_tap[slot][k] * G[slot][n]
// MFMA roles: i = k (input channel), j = n (output channel), reduction index = slot.
// ---------------------------------------------------------------------------------------------
constexpr int KP = 32;            // slots per K-step

struct WgradArgs {
  mpose_conv_geom g;
  mpose_wgrad_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gw, div_ghw;
  int M;
  int n_split, slots_per_split;
  int n_entries;                  // flat (class, tap) entries
  int entry_cls[MPOSE_MAX_CLASSES * MPOSE_MAX_TAPS];
  int entry_tap[MPOSE_MAX_CLASSES * MPOSE_MAX_TAPS];
  int n_widx0, n_widx1;
};

template <int TI>   // input-channel tile: 128 (4x1 waves, 2 acc tiles), 64 (2x2 waves) or 32 (1x2 waves, 2 idle)
__global__ __launch_bounds__(256, 2) void conv_wgrad_k(WgradArgs a) {
  constexpr int WM = TI == 128 ? 4 : (TI == 64 ? 2 : 1);
  constexpr int WN = TI == 128 ? 1 : 2;
  constexpr int RN = 64 / (32 * WN);
  constexpr int X_LOADS = KP * TI / 4 / 256;   // 4 (TI=128) or 2 (TI=64)
  constexpr int G_LOADS = KP * 64 / 4 / 256;   // 2
  constexpr int X_TILE = KP * TI, G_TILE = KP * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                 // [2][X_TILE]
  float* sG = smem + 2 * X_TILE;    // [2][G_TILE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const bool wactive = wave < WM * WN;
  const int li = lane & 31, lh = lane >> 5;
  const mpose_conv_geom& g = a.g;
  const int entry = blockIdx.x / a.n_split;
  const int split = blockIdx.x - entry * a.n_split;
  const int cls = a.entry_cls[entry];
  const mpose_tap tap = g.cls[cls].taps[a.entry_tap[entry]];
  const bool second = tap.acc != 0;
  const int npad = second ? g.Npad1 : g.Npad0;
  const int n_ctiles = npad / 64;
  const int k_tile = blockIdx.y / n_ctiles, n_tile = blockIdx.y - k_tile * n_ctiles;
  if (k_tile * TI >= g.Cin) return;
  const int k0 = k_tile * TI, n0 = n_tile * 64;
  const mpose_wgrad_operands& op = a.op[blockIdx.z];
  const float* gout = second ? op.gout1 : op.gout0;
  const int cout = second ? g.Cout1 : g.Cout0;
  float* dw = second ? op.dw1 : op.dw0;
  const int n_widx = second ? a.n_widx1 : a.n_widx0;
  const int oyc = g.cls[cls].oy, oxc = g.cls[cls].ox;
  const int dy = tap.dy, dx = tap.dx;

  const int m_begin = split * a.slots_per_split;
  const int m_end = min(a.M, m_begin + a.slots_per_split);
  const int n_steps = (m_end - m_begin + KP - 1) / KP;

  constexpr int XC4 = TI / 4;                  // float4 per X row
  const int x_col4 = tid % XC4, x_row0 = tid / XC4;
  constexpr int X_ROWSTEP = 256 / XC4;
  const int g_col4 = tid & 15, g_row0 = tid >> 4;
  const bool pro = op.in_scale != nullptr;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pro) {
    sc = *reinterpret_cast<const float4*>(op.in_scale + k0 + x_col4 * 4);
    sh = *reinterpret_cast<const float4*>(op.in_shift + k0 + x_col4 * 4);
  }
  const bool g_col_ok = (n0 + g_col4 * 4) < cout;     // cout % 4 == 0

  f32x16 acc[RN];
#pragma unroll
  for (int rn = 0; rn < RN; ++rn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rn][r] = 0.0f;

  float4 rx[X_LOADS], rg[G_LOADS];
  unsigned rx_ok = 0, rg_ok = 0;
  auto decomp = [&](int m, unsigned& b, unsigned& gy, unsigned& gx) {
    b = fdiv((unsigned)m, a.div_ghw);
    const unsigned rem = (unsigned)m - b * (unsigned)(g.GH * g.GW);
    gy = fdiv(rem, a.div_gw);
    gx = rem - gy * (unsigned)g.GW;
  };
  // Branch-free issue (clamped addresses, masks applied at LDS-store time): see conv_igemm_k.
  auto load_regs = [&](int step) {
    const int mb = m_begin + step * KP;
    rx_ok = 0; rg_ok = 0;
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
      const int m = mb + x_row0 + X_ROWSTEP * j;
      const int mc = m < m_end ? m : m_begin;
      unsigned b, gy, gx;
      decomp(mc, b, gy, gx);
      const int iy = (int)gy * g.in_mul + dy, ix = (int)gx * g.in_mul + dx;
      const bool ok = m < m_end && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
      const long off = ok ? ((((long)b * g.IH + iy) * g.IW + ix) * g.Cin) : 0l;
      rx[j] = *reinterpret_cast<const float4*>(op.in + off + k0 + x_col4 * 4);
      rx_ok |= (ok ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < G_LOADS; ++j) {
      const int m = mb + g_row0 + 16 * j;
      const bool ok = m < m_end && g_col_ok;
      const int mc = m < m_end ? m : m_begin;
      unsigned b, gy, gx;
      decomp(mc, b, gy, gx);
      const long pix = ((long)b * g.OH + (gy * g.out_mul + oyc)) * g.OW + (gx * g.out_mul + oxc);
      const long off = ok ? (pix * cout + n0 + g_col4 * 4) : 0l;
      rg[j] = *reinterpret_cast<const float4*>(gout + off);
      rg_ok |= (ok ? 1u : 0u) << j;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
      float4 v = rx[j];
      if (pro) {
        v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
      }
      if (!((rx_ok >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(sX + buf * X_TILE + (x_row0 + X_ROWSTEP * j) * TI + x_col4 * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < G_LOADS; ++j) {
      float4 v = rg[j];
      if (!((rg_ok >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(sG + buf * G_TILE + (g_row0 + 16 * j) * 64 + g_col4 * 4) = v;
    }
  };

  if (n_steps > 0) { load_regs(0); store_lds(0); }
  __syncthreads();
  for (int st = 0; st < n_steps; ++st) {
    const int buf = st & 1;
    if (st + 1 < n_steps) load_regs(st + 1);
    const float* cX = sX + buf * X_TILE;
    const float* cG = sG + buf * G_TILE;
    if (wactive) {
#pragma unroll
    for (int kk = 0; kk < KP / 2; ++kk) {
      const float av = cX[(kk * 2 + lh) * TI + wm * 32 + li];
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float bv = cG[(kk * 2 + lh) * 64 + (wn * RN + rn) * 32 + li];
        acc[rn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rn], 0, 0, 0);
      }
    }
    }
    if (st + 1 < n_steps) store_lds(buf ^ 1);
    __syncthreads();
  }

  // epilogue: rows i = input channel; regs 4*rg..4*rg+3 are 4 consecutive channels -> one float4
  if (!wactive) return;
  const int k4_total = g.Cin >> 2;
  float* base = dw + ((long)(split * n_widx + tap.widx) * k4_total) * npad * 4;
#pragma unroll
  for (int rn = 0; rn < RN; ++rn) {
    const int n = n0 + (wn * RN + rn) * 32 + li;
#pragma unroll
    for (int rgp = 0; rgp < 4; ++rgp) {
      const int k4 = (k0 + wm * 32) / 4 + 2 * rgp + lh;
      const float4 v = make_float4(acc[rn][4 * rgp], acc[rn][4 * rgp + 1], acc[rn][4 * rgp + 2], acc[rn][4 * rgp + 3]);
      *reinterpret_cast<float4*>(base + ((long)k4 * npad + n) * 4) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight packing / gradient unpacking (one launch for all convolutions of the model)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weights_k(const mpose_pack_job* __restrict__ jobs) {
  const mpose_pack_job j = jobs[blockIdx.y];
  const long total = (long)j.T * j.Kpad * j.Npad;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k_lo = (int)(e & 3);
    long r = e >> 2;
    const int n = (int)(r % j.Npad); r /= j.Npad;
    const int k4 = (int)(r % (j.Kpad / 4));
    const int t = (int)(r / (j.Kpad / 4));
    const int k = k4 * 4 + k_lo;
    float v = 0.f;
    if (n < j.N && k < j.K) v = j.src[n * j.sn + k * j.sk + t * j.st];
    j.dst[e] = v;
  }
}

__global__ __launch_bounds__(256) void unpack_wgrads_k(const mpose_unpack_job* __restrict__ jobs) {
  const mpose_unpack_job j = jobs[blockIdx.y];
  const long total = (long)j.T * j.K * j.N;
  const long split_stride = (long)j.T * j.Kpad * j.Npad;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    // enumerate (t, k, n) with n fastest so that reads of the packed layout are contiguous in n
    const int n = (int)(e % j.N);
    long r = e / j.N;
    const int k = (int)(r % j.K);
    const int t = (int)(r / j.K);
    const long src = (((long)t * (j.Kpad / 4) + (k >> 2)) * j.Npad + n) * 4 + (k & 3);
    float s = 0.f;
    for (int sp = 0; sp < j.n_split; ++sp) s += j.src[src + sp * split_stride];
    float* d = j.dst + n * j.sn + k * j.sk + t * j.st;
    *d = j.accumulate ? (*d + s) : s;
  }
}

}  // namespace
}  // namespace mpose

using namespace mpose;

static int check_geom(const mpose_conv_geom* g) {
  if (!g || g->Cin <= 0 || (g->Cin % KC) || g->n_classes < 1 || g->n_classes > MPOSE_MAX_CLASSES) return MPOSE_EINVAL;
  if (g->Npad0 <= 0 || (g->Npad0 % 32)) return MPOSE_EINVAL;
  if ((g->in_mul != 1 && g->in_mul != 2) || (g->out_mul != 1 && g->out_mul != 2)) return MPOSE_EINVAL;
  for (int c = 0; c < g->n_classes; ++c)
    if (g->cls[c].n_taps < 0 || g->cls[c].n_taps > MPOSE_MAX_TAPS) return MPOSE_EINVAL;
  if ((long)g->B * g->GH * g->GW >= (1l << 26)) return MPOSE_EINVAL;
  return 0;
}

extern "C" int mpose_conv_fwd(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups, int flags,
                              void* stream) {
  int rc = check_geom(geom);
  if (rc) return rc;
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP) return MPOSE_EINVAL;
  ConvArgs a{};
  a.g = *geom;
  bool acc1 = false;
  for (int c = 0; c < geom->n_classes; ++c)
    for (int t = 0; t < geom->cls[c].n_taps; ++t) acc1 |= geom->cls[c].taps[t].acc != 0;
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    if (!ops[i].in || !ops[i].w0 || !ops[i].out0) return MPOSE_EINVAL;
    if (acc1 && (!ops[i].w1 || !ops[i].out1)) return MPOSE_EINVAL;
  }
  if (acc1 && geom->Npad1 != geom->Npad0) return MPOSE_EINVAL;
  a.M = geom->B * geom->GH * geom->GW;
  if (a.M == 0) return 0;
  a.div_gw = make_fastdiv((unsigned)geom->GW);
  a.div_ghw = make_fastdiv((unsigned)(geom->GH * geom->GW));
  a.flags = flags;
#ifdef MPOSE_ABLATE
  if (getenv("MPOSE_ABLATE")) a.flags |= atoi(getenv("MPOSE_ABLATE")) << 8;
#endif
  hipStream_t s = (hipStream_t)stream;
  const int npad = geom->Npad0;
  // Tile selection: 128x64 when there is enough work to fill 256 CUs, else 64x64; N = 32 tiles for the
  // 17(->32)-channel layers.
  const int cmax = (acc1 && geom->Cout1 > geom->Cout0) ? geom->Cout1 : geom->Cout0;
  if (cmax > npad) return MPOSE_EINVAL;
  if (cmax <= 32) {
    a.n_mtiles = (a.M + 127) / 128;
    return acc1 ? launch_conv<4, 1, 1, 1, true>(a, n_groups, s) : launch_conv<4, 1, 1, 1, false>(a, n_groups, s);
  }
  if (npad % 64) return MPOSE_EINVAL;
  const long blocks128 = (long)((a.M + 127) / 128) * geom->n_classes * ((cmax + 63) / 64) * n_groups;
  if (blocks128 >= 1024) {
    a.n_mtiles = (a.M + 127) / 128;
    return acc1 ? launch_conv<4, 1, 1, 2, true>(a, n_groups, s) : launch_conv<4, 1, 1, 2, false>(a, n_groups, s);
  }
  a.n_mtiles = (a.M + 63) / 64;
  return acc1 ? launch_conv<2, 2, 1, 1, true>(a, n_groups, s) : launch_conv<2, 2, 1, 1, false>(a, n_groups, s);
}

extern "C" int mpose_conv_wgrad(const mpose_conv_geom* geom, const mpose_wgrad_operands* ops, int n_groups, int n_split,
                                void* stream) {
  int rc = check_geom(geom);
  if (rc) return rc;
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || n_split < 1) return MPOSE_EINVAL;
  if ((geom->Npad0 % 64) || (geom->Cout0 % 4)) return MPOSE_EINVAL;
  WgradArgs a{};
  a.g = *geom;
  bool acc1 = false;
  int max0 = -1, max1 = -1;
  for (int c = 0; c < geom->n_classes; ++c)
    for (int t = 0; t < geom->cls[c].n_taps; ++t) {
      const mpose_tap& tp = geom->cls[c].taps[t];
      a.entry_cls[a.n_entries] = c;
      a.entry_tap[a.n_entries] = t;
      ++a.n_entries;
      if (tp.acc) { acc1 = true; if (tp.widx > max1) max1 = tp.widx; }
      else if (tp.widx > max0) max0 = tp.widx;
    }
  a.n_widx0 = max0 + 1;
  a.n_widx1 = max1 + 1;
  if (acc1 && ((geom->Npad1 % 64) || (geom->Cout1 % 4) || geom->Npad1 != geom->Npad0)) return MPOSE_EINVAL;
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    if (!ops[i].in || !ops[i].gout0 || !ops[i].dw0) return MPOSE_EINVAL;
    if (acc1 && (!ops[i].gout1 || !ops[i].dw1)) return MPOSE_EINVAL;
  }
  a.M = geom->B * geom->GH * geom->GW;
  if (a.M == 0 || a.n_entries == 0) return 0;
  a.div_gw = make_fastdiv((unsigned)geom->GW);
  a.div_ghw = make_fastdiv((unsigned)(geom->GH * geom->GW));
  a.n_split = n_split;
  a.slots_per_split = ((a.M + n_split - 1) / n_split + KP - 1) / KP * KP;
  hipStream_t s = (hipStream_t)stream;
  const int n_ctiles = geom->Npad0 / 64;
  if (geom->Cin % 128 == 0) {
    dim3 grid(a.n_entries * n_split, (geom->Cin / 128) * n_ctiles, n_groups);
    const int lds = (2 * KP * 128 + 2 * KP * 64) * 4;
    conv_wgrad_k<128><<<grid, 256, lds, s>>>(a);
  } else if (geom->Cin % 64 == 0) {
    dim3 grid(a.n_entries * n_split, (geom->Cin / 64) * n_ctiles, n_groups);
    const int lds = (2 * KP * 64 + 2 * KP * 64) * 4;
    conv_wgrad_k<64><<<grid, 256, lds, s>>>(a);
  } else {
    dim3 grid(a.n_entries * n_split, (geom->Cin / 32) * n_ctiles, n_groups);
    const int lds = (2 * KP * 32 + 2 * KP * 64) * 4;
    conv_wgrad_k<32><<<grid, 256, lds, s>>>(a);
  }
  return launch_status();
}

extern "C" int mpose_pack_weights(const mpose_pack_job* jobs_dev, int n_jobs, int max_elems_per_job, void* stream) {
  if (n_jobs <= 0) return 0;
  int bx = (max_elems_per_job + 256 * 8 - 1) / (256 * 8);
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  pack_weights_k<<<dim3(bx, n_jobs), 256, 0, (hipStream_t)stream>>>(jobs_dev);
  return launch_status();
}

extern "C" int mpose_unpack_wgrads(const mpose_unpack_job* jobs_dev, int n_jobs, int max_elems_per_job, void* stream) {
  if (n_jobs <= 0) return 0;
  int bx = (max_elems_per_job + 256 * 8 - 1) / (256 * 8);
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  unpack_wgrads_k<<<dim3(bx, n_jobs), 256, 0, (hipStream_t)stream>>>(jobs_dev);
  return launch_status();
}
