// Implicit-GEMM convolution engine for gfx950: fp32 convolutions computed on the 16-bit matrix cores with split operands --
// by default THREE fp16 products per multiply-add of two-way split, per-tensor-scaled operands (MPOSE_CONV_F16X3, NPL = 2
// below), on request six bf16 products of three-way split ones ("bf16x6", NPL = 3) -- forward, data-gradient and weight-gradient
// alike.  (conv_p.hip: the engine that reads PRE-SPLIT activations; wgrad.hip: the row-of-taps weight gradient, round 3.)
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
// Layout: activations NHWC fp32; weights pre-split into three bf16 planes and packed
// [widx][K/16][plane][Npad][half][8] -- one 16-byte v_mfma_f32_32x32x16_bf16 B fragment per (n, half):
//   lane (i = l&31, h = l>>5) holds A[pixel i][k = 16 s + 8 h + 0..7] and B[k = 16 s + 8 h + 0..7][n = l&31].
#include "common.h"
#include <stdlib.h>

#ifndef CV_PART
#define CV_PART 0          // 0: this unit; 3 / 4: conv_igemm_k's RN = 3 / 4 instantiations alone (conv_rn3.hip, conv_rn4.hip; see the end of the anonymous namespace)
#endif
#ifndef MFMA_SPREAD
#define MFMA_SPREAD 1      // 0: round 2's rm-major MFMA chains (A/B builds)
#endif
#ifndef CV_EXP
#define CV_EXP 0      // timing experiments (tools/igemm_exp.sh; wrong results): 32 no scale gathers, 1 no epilogue, 2 no K loop, 4 no split-K exchange,
                      // 8 every B fragment from one place (L1-resident weights), 16 every A row from one place, 64 no statistics atomics
#endif

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

constexpr int KC = 32;              // channels per K-chunk = two 16-deep MFMA k-groups
// LDS bytes per pixel row of one 16-bit plane: 64 B of data, no padding.  The four 16-byte chunks of a row are stored XOR-swizzled
// by row bits 2-3 (physical chunk = chunk ^ ((row >> 2) & 3)): the staging writes (ds_write_b64: 32 lanes = 4 consecutive rows x
// 8 lanes, the same swizzle for all four) and the fragment reads (ds_read_b128: 16 lanes = 16 consecutive rows of one logical
// chunk -> (row & 3, chunk ^ (row >> 2)) takes all 16 values) both touch every bank once.  Round 1's 80-byte pitch made the reads
// conflict-free but let row 3 of a write alias row 0 (20 r mod 64 = 0, 20, 40, 60: SQ_LDS_BANK_CONFLICT 36-40 % of the LDS cycles).
constexpr int A_ROW_B = 64;
constexpr int A_PLANE_B = 64 * A_ROW_B;
constexpr int A_TILE_B = 3 * A_PLANE_B;   // hi / mid / lo planes of one 64-pixel x 32-channel tile
// Row-group staging (ROWG; stride-1 3x3, first pass): one staged tile serves the three dx taps of a kernel row.  It holds the
// 66 consecutive input pixels  m0-1+dy*IW .. m0+64+dy*IW  (rows 0..65) plus SIXTEEN all-zero rows (80..95) that the fragment
// reads of out-of-image taps are redirected to: a redirected lane takes zero row 80 + (its row & 15), i.e. the banks (and the
// chunk swizzle) of the row it would have read, so the 16 lanes of a ds_read_b128 still touch every bank once (one shared zero
// row cost a two-way conflict wherever a 16-lane group held an image-border pixel: 0.09-0.27 of the LDS cycles).
// Two planes per tile: the row-group loop exists in the three-product form only.
constexpr int RG_ROWS = 96;
constexpr int RG_ZERO_ROW = 80;
constexpr int RG_PLANE_B = RG_ROWS * A_ROW_B;
constexpr int RG_TILE_B = 2 * RG_PLANE_B;

struct ConvArgs {
  mpose_conv_geom g;
  mpose_conv_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gw, div_ghw;
  int M;                          // slots per class = B*GH*GW
  int n_mtiles;
  int flags;
  int part_row0, part_rows;       // MPOSE_CONV_STATS_PART: first row this launch writes, rows in the buffers' headers
  int in_bias;                    // bytes: max negative tap shift, folded into the input buffer base
  int xcd_order;                  // conv_igemm_k: tiles taken in XCD-contiguous order (see the kernel's head)
};

// Row of the 32x32 accumulator held in register r of lane-half h.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
__device__ __forceinline__ u32x4 buf_load4u(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}

constexpr unsigned kOob = 0xFFFFFFF0u;      // voffset that the buffer unit treats as out of range -> returns 0

// fp32 -> three bf16 planes with x == hi + mid + lo up to 2^-27 |x| (round-to-nearest at every level; the two
// subtractions are exact in fp32).  gfx950 has v_cvt_pk_bf16_f32, so a float4 costs ~20 VALU instructions.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Two elements at a time: v_cvt_pk_bf16_f32 rounds a pair; the residuals are computed with SCALAR v_sub_f32 --
// packed-f32 VALU (v_pk_add_f32) beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md), so the file is
// also built with -fno-slp-vectorize.
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }
__device__ __forceinline__ void split2(const float x0, const float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
  const float r0 = x0 - bf_lo(h), r1 = x1 - bf_hi(h);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float q0 = r0 - bf_lo(m), q1 = r1 - bf_hi(m);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2));
}
__device__ __forceinline__ void split4(const float4 v, uint2& h, uint2& m, uint2& l) {
  split2(v.x, v.y, h.x, m.x, l.x);
  split2(v.z, v.w, h.y, m.y, l.y);
}

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// MPOSE_CONV_F16X3: x (already multiplied by the tensor's power-of-two scale, |x| < 2^15) -> two fp16 values with
// x == h + l up to 2^-22 |x| (round to nearest twice; the subtraction is exact).  v_cvt_pk_f16_f32 rounds a pair.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(const float x0, const float x1, unsigned& h, unsigned& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{x0, x1}, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = x0 - (float)hh[0], r1 = x1 - (float)hh[1];
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Forward / data-gradient kernel: fp32 convolution on the 16-bit matrix cores (NPL = 2: three fp16 products, the default;
// NPL = 3: six bf16 products -- the form the next paragraph derives; the fp16 form's own argument is at MPOSE_CONV_F16X3 in the header).
//
// Why: gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs on the SIMD's 32 fp32 FMA lanes -- measured 147 TFLOP/s
// for the whole chip (tools/probe/mfma_probe.hip) -- while v_mfma_f32_32x32x16_bf16 sustains 1.9 PFLOP/s.  An fp32
// value is the exact sum of three bf16 values (8 significant bits each), so  a*b = sum_ij a_i*b_j ; the three
// smallest cross terms (mid*lo, lo*mid, lo*lo <= 2^-26 |ab|) are below fp32's own rounding and are dropped.  Six
// bf16 MFMAs (products exact, fp32 accumulation) therefore replace eight fp32 MFMAs per 16 channels at 2.2x the
// speed, with the same error bound as an fp32 FMA chain (tests/test_conv_gpu.py compares both against fp64).
//
// Cost model (measured, tools/probe/): a bf16 MFMA occupies its SIMD for ~16.5 ns whatever the occupancy, a
// dependent accumulate chain costs nothing extra, and VALU instructions of the same SIMD do not hide under it
// (+1..1.6 ns each).  So: ONE wave per SIMD with a fat 64-pixel x 32*RN-channel register tile (512 registers),
// deep register prefetch instead of occupancy, and as little VALU per MFMA as possible:
//   * all addresses are  buffer descriptor + loop-invariant per-lane VGPR offset + wave-uniform SGPR offset;
//   * zero padding is done by the buffer unit (out-of-bounds rows get voffset 0xFFFFFFF0 and read 0);
//   * a workgroup is 4 INDEPENDENT wavefronts (no workgroup barrier in the K loop);
//   * A operand: the wave stages ITS OWN 64 rows x 32 channels per (chunk, tap): coalesced 128-byte row reads,
//     optional BN+ReLU of the producer, the 3-way bf16 split, then three ds_write_b64 into a wave-private,
//     double-buffered LDS tile whose ds_read_b128 delivers MFMA fragment order (row pitch 80 B: conflict-free);
//   * B operand: weights are split and laid out in fragment order ONCE per step by pack_weights_k, so every lane
//     fetches 16-byte fragments straight from L2 into a register ring a whole tile (96 MFMAs) ahead of use.
// KS (split-K inside the workgroup): KS waves share one 64-row tile and take 1/KS of the (chunk, tap) tiles each;
// chosen per launch so that the number of waves is close to a multiple of the 1024 SIMDs.  The parts are summed
// through the (by then dead) A-tile LDS region in a fixed order.
// MODE 0: one pass.  MODE 1: taps with acc == 1 are a second pass into a second output (fused shortcut, forward).
// MODE 2 (MPOSE_CONV_SUM_INPUTS): they are a second pass over a second INPUT that keeps accumulating into the same
// tile -- dX = conv_in^T(dC1) + shortcut^T(dSC) -- with one epilogue.
// NPL = 3: the six-product bf16 form above.  NPL = 2 (MPOSE_CONV_F16X3): operands split into TWO fp16 planes after a
// per-tensor power-of-two scale, three products per k-group (al*bh, ah*bl, ah*bh), accumulators scaled back (exactly) after
// the K loop -- half the matrix work, two thirds of the LDS and L2 fragment traffic, 4 instead of 5.5 VALU per element.
// ROWG (stride-1 3x3 first passes, three-product form): the operand split -- all of the loop's VALU work -- runs once per
// (channel chunk, kernel ROW) instead of once per tap: see the row-group K loop below.
// SLIM (single-pass row-group launches with 64-channel wave tiles, three-product form): TWO workgroups per CU.  A 64-channel
// wave tile needs <= 256 registers, and a row-group tile of 68 rows (66 are read) with ONE block of zero rows per wave instead of
// one per plane and buffer brings the workgroup to 79 KB of LDS: the second workgroup's waves issue their MFMAs into the gaps
// the first one's leave (prologue, exchange, epilogue, every s_waitcnt) -- see slim_tile().
template <int RN, int MODE, int NPL, bool ROWG>
constexpr bool slim_tile() { return ROWG && MODE == 0 && RN <= 2 && NPL <= 2; }
constexpr int RG_SLIM_ROWS = 68;

// (round 6) the per-tap form of the narrow tiles, unsplit: 68 KB of LDS with two staged planes -- two workgroups per CU once the
// registers fit too (the feature extractors' 32- / 64-channel layers, stride-2 and output-parity launches: short K loops between a
// prologue and an epilogue, which a second resident workgroup fills: 65 -> 50 us and 49 -> 37 us on the launches that fitted as they were)
// (which variants fit 256 registers without scratch was read off the build log: the 96-channel tile only without a prologue / second
//  output in the three-product form)
template <int RN, int MODE, int KS, bool PRO, int NPL, bool ROWG>
constexpr bool pair_tile() {
  return !ROWG && KS == 1 && NPL <= 2 && MODE <= 1 && (RN <= 2 || (RN == 3 && (NPL == 1 || (MODE == 0 && !PRO))));
}

template <int RN, int MODE, int KS, bool PRO, int NPL, bool ROWG>
__global__ __launch_bounds__(256, ((slim_tile<RN, MODE, NPL, ROWG>() || pair_tile<RN, MODE, KS, PRO, NPL, ROWG>()) ? 2 : 1)) void conv_igemm_k(ConvArgs a) {
  constexpr bool ACC1 = MODE == 1, SUM2 = MODE == 2;
  constexpr bool F16 = NPL <= 2;          // NPL == 1: the h planes only (MPOSE_CONV_F16X1: operands rounded to fp16, one product)
  constexpr int NPM = F16 ? 2 : 3;        // planes of the packed weights in memory
  constexpr bool SLIM = slim_tile<RN, MODE, NPL, ROWG>();
  constexpr int RG_PL = (SLIM ? RG_SLIM_ROWS : RG_ROWS) * A_ROW_B;     // bytes of one plane of a row-group tile
  constexpr int RG_TL = 2 * RG_PL;
  constexpr int PT_B = (F16 && KS == 1) ? 2 * A_PLANE_B : A_TILE_B;     // a per-tap tile (K-split launches park accumulators in the same area and keep its size): the planes the form stages (round 6: two for the fp16 forms -- 68 KB
                                                           // per workgroup instead of 100, so that the narrow tiles' launches, whose registers allow it,
                                                           // run two workgroups per CU: their short K loops are all prologue and epilogue otherwise)
  constexpr int TILE_B = ROWG ? RG_TL : PT_B;              // LDS bytes reserved per staging buffer
  constexpr int WAVE_B = 2 * TILE_B + (SLIM ? 16 * A_ROW_B : 0);      // a wave's two buffers (+ its zero rows)
  static_assert(!ROWG || F16, "row-group tiles hold two planes");
  constexpr int NPASS = MODE ? 2 : 1;
  constexpr int BM = 256 / KS;
  constexpr int BN = 32 * RN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* sA_all = smem_raw;                                               // [4 waves][WAVE_B]
  float* sRed = reinterpret_cast<float*>(smem_raw + 4 * WAVE_B);                  // [2 sets][4 waves][BN][2]
  unsigned* sRow = reinterpret_cast<unsigned*>(smem_raw + 4 * WAVE_B + 2 * 4 * BN * 2 * 4);   // [4 waves][64] output row offsets
  float* sMM = reinterpret_cast<float*>(smem_raw + 4 * WAVE_B + 2 * 4 * BN * 2 * 4 + 4 * 64 * 4);   // [4 waves][BN][2] channel extremes of out0

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const mpose_conv_geom& g = a.g;
  // XCD-aware tile order: workgroup L of a column's (x fastest) dispatch order runs on XCD L % 8 (observed placement, speed only),
  // and every XCD has a private 4 MiB L2.  XCD c takes the CONTIGUOUS run c of the pixel tiles, their channel tiles innermost:
  // neighbouring pixel tiles (which share 3x3 halo rows) and the channel tiles of one pixel tile (which read the same input) are
  // dispatched 8 workgroups apart on ONE L2 instead of one on each -- measured 2.1x the algorithmic bytes at the L2s' memory side
  // on the two-input data gradient before (profiles/r4_pmc_traffic.json).
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd_order) {
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, w = L >> 3;
    by = w % gridDim.y;
    bx = (L & 7u) * (gridDim.x >> 3) + w / gridDim.y;
  }
  const int cls = bx / a.n_mtiles;
  const int kh = wave % KS;                                                       // K part of this wave
  const int m0 = (bx - cls * a.n_mtiles) * BM + (wave / KS) * 64;                  // first row of THIS wave
  const int n0 = by * BN;
  const mpose_conv_operands& op = a.op[blockIdx.z];
  const int n_taps = g.cls[cls].n_taps;
  // taps are read from the kernel arguments (scalar loads): {dy, dx, widx, acc} packed in one dword
  // ... once: lane t keeps tap t in a VGPR and v_readlane hands it to the scalar unit whenever a tile needs it (no
  // scalar-memory load, whose lgkmcnt(0) would also drain the LDS reads in flight, inside the K loop)
  const int lane_tap = (int)(threadIdx.x & 63) < MPOSE_MAX_TAPS
      ? *reinterpret_cast<const int*>(&g.cls[cls].taps[(threadIdx.x & 63) < MPOSE_MAX_TAPS ? (threadIdx.x & 63) : 0]) : 0;
  auto tap_word = [&](int t) { return __builtin_amdgcn_readlane(lane_tap, t); };
  unsigned char* sA = sA_all + wave * WAVE_B;
  // swizzled chunk offsets inside a 64-byte row (see A_ROW_B): staging writes 8 bytes of chunk (lane & 7) >> 1, fragment reads
  // take chunk 2 s + (lane >> 5) of row (lane & 31)
  const int st_off_even = ((((lane & 7) >> 1) ^ (lane >> 5)) << 4) + ((lane & 1) << 3);
  const int st_off_odd = ((((lane & 7) >> 1) ^ ((lane >> 5) | 2)) << 4) + ((lane & 1) << 3);
  const int rd_sw = ((lane & 31) >> 2) & 3;
  const int rd_off0 = ((0 + (lane >> 5)) ^ rd_sw) << 4, rd_off1 = ((2 + (lane >> 5)) ^ rd_sw) << 4;

  // ---- per-lane staging state (loop invariant) ----
  // row_voff[j]: BYTE offset of the slot's anchor pixel + this lane's 16-byte channel column;
  // row_taps[j]: bit t = tap t reads an in-bounds pixel for that row (filled in after the first loads are issued).
  const int a_col4 = lane & 7;
  const int in_ld = g.in_ld > 0 ? g.in_ld : g.Cin;
  unsigned row_voff[8], row_taps[8];
  int row_yx[8];                           // iy0 | ix0 << 16 of the row's anchor pixel (rows beyond M: never valid)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned m = (unsigned)(m0 + (lane >> 3) + 8 * j);
    row_voff[j] = 0; row_taps[j] = 0; row_yx[j] = 0x7000;
    if ((int)m < a.M) {
      const unsigned b = fdiv(m, a.div_ghw);
      const unsigned rem = m - b * (unsigned)(g.GH * g.GW);
      const unsigned gy = fdiv(rem, a.div_gw);
      const unsigned gx = rem - gy * (unsigned)g.GW;
      const int iy0 = (int)gy * g.in_mul, ix0 = (int)gx * g.in_mul_x;
      row_voff[j] = (((b * (unsigned)g.IH + (unsigned)iy0) * (unsigned)g.IW + (unsigned)ix0) * (unsigned)in_ld + (unsigned)(a_col4 * 4)) * 4u;
      row_yx[j] = iy0 | (ix0 << 16);
    }
  }
  auto mark_tap = [&](int t) {             // set bit t of row_taps where tap t stays inside the image
    const int tp = tap_word(t);
    const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int iy = (row_yx[j] & 0xffff) + dy, ix = (row_yx[j] >> 16) + dx;
      if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) row_taps[j] |= 1u << t;
    }
  };
  const int n_chunks = g.Cin / KC;
  const int k16_total = g.Cin >> 4;
  const int npad = g.Npad0;                                       // == Npad1 when ACC1
  const unsigned plane_b = (unsigned)npad * 32u;                   // bytes of one (k-group, plane) slab of packed weights
  constexpr bool pro = PRO;                                        // BN + ReLU of the producer applied while staging
  // Buffer descriptors.  The input base is moved back by `a.in_bias` bytes so that the (possibly negative) tap
  // shift becomes a non-negative SGPR offset.
  const __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(op.in)) - a.in_bias, 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(SUM2 ? op.in1 : op.in)) - a.in_bias, 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.w0), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MODE ? op.w1 : op.w0), 0, 0xFFFFFF00, 0x00020000);
  const unsigned w_voff = (unsigned)(((n0 + li) * 2 + lh) * 16);     // this lane's 16-byte fragment inside a slab

  f32x16 acc0[2][RN];
  float4 ra[ROWG ? 9 : 8];               // A rows in flight: global -> registers -> (BN+ReLU, split) -> LDS
  float4 rsc_c = make_float4(1.f, 1.f, 1.f, 1.f), rsh_c = make_float4(0.f, 0.f, 0.f, 0.f), rsc_n = rsc_c, rsh_n = rsh_c;
  unsigned pad_c = 0, pad_n = 0;         // bit j: row j of the tile being staged / being loaded is padding (PRO only)
  u32x4 fb[2][RN][NPL];                  // B-fragment ring: [k-group][column block][plane] of the NEXT use
  u32x4 afA[2][NPL], afB[2][NPL];        // A fragments [row block][plane] of k-group 0 / 1
  // F16: scale exponents of this launch's tensors (SGPRs).  Pass `set` multiplies input ka[set] with weights kw[set].
  int ka0 = 0, ka1 = 0, kw0 = 0, kw1 = 0;
  if (F16 && !(CV_EXP & 32)) {
    ka0 = f16_scale_exp(amax_gather(op.in_amax));
    ka1 = SUM2 ? f16_scale_exp(amax_gather(op.in1_amax)) : ka0;
    kw0 = f16_scale_exp(*op.w0_amax);
    kw1 = MODE ? f16_scale_exp(*op.w1_amax) : kw0;
    if (SUM2) {
      // The first pass's accumulators are re-expressed in the second pass's units (2^((ka1 + kw1) - (ka0 + kw0))) before the second
      // input accumulates on top.  A second input that is all zeros (a shortcut BatchNorm with gamma == 0) or 2^60 times smaller
      // than the first would make that factor overflow fp32: cap it -- the second input is then scaled less than its own maximum
      // allows, which costs it precision it cannot contribute anyway (its whole sum is below the first's rounding error).
      const int excess = (ka1 + kw1) - (ka0 + kw0) - 60;
      if (excess > 0) ka1 -= excess;
    }
  }
  const int oyc = g.cls[cls].oy, oxc = g.cls[cls].ox;

  // Taps with acc == 0 (the convolution proper) come first in a class's tap list, taps with acc == 1 (the fused
  // 1x1 shortcut of a ResidualBlock: same input, second weight set, second output) last.  The two sets run as
  // two passes of the same pipeline with ONE accumulator tile, so the fused launch keeps the full 128-wide tile.
  bool masks_done = false;
  int n_taps0 = 0;
  for (int t = 0; t < n_taps; ++t) n_taps0 += (((tap_word(t) >> 24) & 0xff) == 0) ? 1 : 0;

#pragma unroll 1
  for (int set = 0; set < NPASS; ++set) {
    const int t_lo = set ? n_taps0 : 0;
    const int nt = set ? n_taps - n_taps0 : (MODE ? n_taps0 : n_taps);
    const int n_iter = n_chunks * nt;
    const __amdgpu_buffer_rsrc_t rs_w = set ? rs_w1 : rs_w0;
    const __amdgpu_buffer_rsrc_t rs_in = set ? rs_in1 : rs_in0;
    const float a_mul = F16 ? pow2f(set ? ka1 : ka0) : 1.f;
    if (!SUM2 || set == 0) {
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc0[rm][rn][r] = 0.0f;
    }

    // wave-uniform (SGPR) description of tile `it`
    struct TileInfo { unsigned a_soff; unsigned w_soff; int t; int c; };
    auto tile_ct = [&](int c, int t) {           // tile = (channel chunk c, tap position t inside this pass's tap range)
      TileInfo ti;
      ti.c = c;
      ti.t = t_lo + t;
      const int tp = tap_word(ti.t);
      const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
      const int widx = (tp >> 16) & 0xff;
      ti.a_soff = (CV_EXP & 16) ? (unsigned)a.in_bias : (unsigned)(((dy * g.IW + dx) * in_ld + ti.c * KC) * 4 + a.in_bias);
      ti.w_soff = (CV_EXP & 8) ? 0u : (unsigned)(widx * k16_total + ti.c * (KC / 16)) * (unsigned)NPM * plane_b;
      return ti;
    };
    auto tile_info = [&](int it) { const int c = it / nt; return tile_ct(c, it - c * nt); };
    auto load_b = [&](const TileInfo& ti, int s_, int rn) {
      const unsigned so = ti.w_soff + (unsigned)s_ * (unsigned)NPM * plane_b;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        fb[s_][rn][pl] = buf_load4u(rs_w, w_voff + (unsigned)(rn * 1024), so + (unsigned)pl * plane_b);
    };
    // one row (16 bytes per lane) of the A tile described by `ti`
    auto load_row = [&](const TileInfo& ti, int j, float4& dst, unsigned& padbits) {
      const unsigned inv = ((row_taps[j] >> ti.t) & 1u) - 1u;       // 0 = in bounds, 0xFFFFFFFF = padding
      if (pro) padbits = (padbits & ~(1u << j)) | (inv & (1u << j));
      dst = buf_load4(rs_in, row_voff[j] | (inv & kOob), ti.a_soff);
    };
    auto load_scale = [&](const TileInfo& ti, float4& sc, float4& sh) {
      if (pro) {
        sc = *reinterpret_cast<const float4*>(op.in_scale + ti.c * KC + a_col4 * 4);
        sh = *reinterpret_cast<const float4*>(op.in_shift + ti.c * KC + a_col4 * 4);
        if (F16) {     // relu(s*x + t) * 2^k == relu((s 2^k) x + t 2^k): the tensor scale rides on the prologue (exact)
          sc.x *= a_mul; sc.y *= a_mul; sc.z *= a_mul; sc.w *= a_mul;
          sh.x *= a_mul; sh.y *= a_mul; sh.z *= a_mul; sh.w *= a_mul;
        }
      }
    };
    // BN + ReLU of the producing layer (PRO) and the bf16 split happen at LDS-store time; padding rows are already
    // zero from the buffer unit and are re-zeroed only on the PRO path (relu(shift) != 0).
    auto stage = [&](int buf, int j, float4 v, unsigned padbits, const float4& sc, const float4& sh) {
      if (pro) {
        const bool pad = (padbits >> j) & 1u;
        v.x = pad ? 0.f : fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = pad ? 0.f : fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        v.z = pad ? 0.f : fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = pad ? 0.f : fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
      }
      // row (lane >> 3) + 8 j: its swizzle is ((lane >> 5) + 2 j) & 3 -- one of two per-lane values (j is a compile-time constant)
      unsigned char* dA = sA + buf * PT_B + ((lane >> 3) + 8 * j) * A_ROW_B + ((j & 1) ? st_off_odd : st_off_even);
      if constexpr (NPL == 1) {
        if (!pro) { v.x *= a_mul; v.y *= a_mul; v.z *= a_mul; v.w *= a_mul; }
        uint2 h;
        h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, f16x2));
        h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, f16x2));
        *reinterpret_cast<uint2*>(dA) = h;
      } else if constexpr (F16) {
        if (!pro) { v.x *= a_mul; v.y *= a_mul; v.z *= a_mul; v.w *= a_mul; }
        uint2 h, l;
        split2h(v.x, v.y, h.x, l.x);
        split2h(v.z, v.w, h.y, l.y);
        *reinterpret_cast<uint2*>(dA) = h;
        *reinterpret_cast<uint2*>(dA + A_PLANE_B) = l;
      } else {
        uint2 h, m, l;
        split4(v, h, m, l);
        *reinterpret_cast<uint2*>(dA) = h;
        *reinterpret_cast<uint2*>(dA + A_PLANE_B) = m;
        *reinterpret_cast<uint2*>(dA + 2 * A_PLANE_B) = l;
      }
    };
    auto load_a_row = [&](const TileInfo& ti, int j) { load_row(ti, j, ra[j], pad_n); };
    auto stage_row = [&](int buf, int j) { stage(buf, j, ra[j], pad_c, rsc_c, rsh_c); };
    auto rotate_stage_state = [&]() { pad_c = pad_n; rsc_c = rsc_n; rsh_c = rsh_n; };
    auto read_frags = [&](int buf, int s_, u32x4 (&af)[2][NPL]) {
      const unsigned char* cA = sA + buf * PT_B + li * A_ROW_B + (s_ ? rd_off1 : rd_off0);    // (rows li and li + 32 swizzle alike)
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          af[rm][pl] = *reinterpret_cast<const u32x4*>(cA + rm * 32 * A_ROW_B + pl * A_PLANE_B);
    };
    // The six bf16 products of one 16-channel k-group (smallest terms first) for all 2 x RN accumulator blocks,
    // each column block followed by the refill of its B fragments with the same k-group of tile `nb`.  `side(rn)`
    // is the staging work the caller wants issued between the MFMAs of column block rn.
    auto mfma_group = [&](int s_, const u32x4 (&af)[2][NPL], const TileInfo& nb, auto&& side) {
      if constexpr (NPL == 2 && MFMA_SPREAD) {
        // Three-product form, column blocks in PAIRS, products outermost: al*bh over the pair's four accumulators, then ah*bl,
        // then ah*bh -- the accumulation order of every accumulator is what it was (results bit-identical), but two consecutive
        // MFMAs never share an accumulator.  Why: an instruction issued between two MFMAs on the SAME accumulator costs ~43
        // cycles, between MFMAs on different ones ~6 and up to five of them hide (MI355X_MICROARCH.md, "one wave per SIMD"), so
        // with rm-major chains of three the staging work could only sit behind the chains -- exposed.  The pattern below hands
        // every gap between two MFMAs a share of the block pair's other instructions.
        constexpr int STEP = RN >= 2 ? 2 : 1;
#pragma unroll
        for (int rn0 = 0; rn0 < RN; rn0 += STEP) {
          constexpr int dummy = 0;
          const int nblk = (rn0 + STEP <= RN) ? STEP : RN - rn0;
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < STEP; ++j)
              if (j < nblk) {
#pragma unroll
                for (int rm = 0; rm < 2; ++rm)
                  acc0[rm][rn0 + j] = mfma_f16(af[rm][p == 0 ? 1 : 0], fb[s_][rn0 + j][p == 1 ? 1 : 0], acc0[rm][rn0 + j]);
              }
#pragma unroll
          for (int j = 0; j < STEP; ++j)
            if (j < nblk) { load_b(nb, s_, rn0 + j); side(rn0 + j); }
#pragma unroll
          for (int m = 0; m < 6 * STEP; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);
            if (m % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        return;
      }
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        if constexpr (NPL == 1) {
#pragma unroll
          for (int rm = 0; rm < 2; ++rm) acc0[rm][rn] = mfma_f16(af[rm][0], fb[s_][rn][0], acc0[rm][rn]);
        } else if constexpr (F16) {
          const u32x4 bh = fb[s_][rn][0], bl = fb[s_][rn][1];
#pragma unroll
          for (int rm = 0; rm < 2; ++rm) {
            f32x16 c = acc0[rm][rn];
            c = mfma_f16(af[rm][1], bh, c);
            c = mfma_f16(af[rm][0], bl, c);
            c = mfma_f16(af[rm][0], bh, c);
            acc0[rm][rn] = c;
          }
        } else {
          const bf16x8 bh = as_bf16x8(fb[s_][rn][0]), bm = as_bf16x8(fb[s_][rn][1]), bl = as_bf16x8(fb[s_][rn][NPL - 1]);
#pragma unroll
          for (int rm = 0; rm < 2; ++rm) {
            f32x16 c = acc0[rm][rn];
            c = mfma_bf16(as_bf16x8(af[rm][NPL - 1]), bh, c);
            c = mfma_bf16(as_bf16x8(af[rm][0]), bl, c);
            c = mfma_bf16(as_bf16x8(af[rm][1]), bm, c);
            c = mfma_bf16(as_bf16x8(af[rm][1]), bh, c);
            c = mfma_bf16(as_bf16x8(af[rm][0]), bm, c);
            c = mfma_bf16(as_bf16x8(af[rm][0]), bh, c);
            acc0[rm][rn] = c;
          }
        }
        load_b(nb, s_, rn);
        side(rn);
        // Pin the column block: without the fence hipcc sinks every global load of the body to its end and hoists their
        // consumers to its start (s_waitcnt vmcnt(23) ... vmcnt(0) in front of the first MFMAs: a prefetch distance of a few
        // instructions) -- harmless behind 96 MFMAs per tile step, exposed behind 48.
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // Software pipeline (ONE wave per SIMD: every latency has to be covered by this wave's own MFMAs).  Time is
    // cut in half-bodies of 12*RN MFMAs; half 2k runs k-group 0 of tile k, half 2k+1 its k-group 1:
    //   LDS fragments of (tile k, group s)   read during the half before their use;
    //   staging of tile k (split -> LDS)     spread over halves 2k-3, 2k-2, into the buffer tile k-2 has left;
    //   global loads of tile k               issued row by row as staging frees the registers: one body ahead;
    //   B fragments of (tile k, group s)     loaded right after group s of tile k-1 used the registers.
    // A loop body is { half 2k+1, half 2k+2 } so that all of this is one basic block for the scheduler.
    const bool use_rowg = ROWG && set == 0;
    if constexpr (CV_EXP & 2) {
    } else if (use_rowg) {
      if constexpr (ROWG) {
        // ---- row-group K loop (first pass of a stride-1 3x3: taps 3*ky .. 3*ky+2 share dy and have dx in {-1,0,1}) ----
        // One staged tile per (32-channel chunk, kernel row): 66 consecutive input pixels, scaled and split ONCE, serve the
        // three dx taps -- the fragment read of tap dx starts at staged row 1+dx.  Zero padding moves to the fragment read:
        // a lane whose (pixel, tap) falls outside the image reads the all-zero row instead (no masking at staging time, so
        // the BN+ReLU prologue needs no special case either).  Staging work, A loads and LDS writes per MFMA fall ~2.7x.
        unsigned char* sG = sA;
        const int ld4 = in_ld * 4;
        const int q_lane = m0 + (lane >> 3);                   // pixel staged by this lane in piece 0 is q_lane + qs
        const unsigned rg_voff = (unsigned)q_lane * (unsigned)ld4 + (unsigned)(a_col4 * 16);
        unsigned fr_taps[2] = {0u, 0u};                        // bit t: tap t of output pixel m0 + 32*rm + li is inside the image
        struct GroupInfo { unsigned a_soff; int qs; int c; int t; };
        auto group_ck = [&](int c, int ky) {
          GroupInfo G;
          G.c = c; G.t = 3 * ky;
          const int dy = (int)(signed char)(tap_word(G.t) & 0xff);
          G.qs = dy * g.IW - 1;
          G.a_soff = (CV_EXP & 16) ? (unsigned)(a.in_bias + in_ld * 4 * (g.IW + 1)) : (unsigned)((G.qs * in_ld + c * KC) * 4 + a.in_bias);
          return G;
        };
        auto load_piece = [&](const GroupInfo& G, int j) {       // staged rows 8j .. 8j+7, 16 bytes per lane
          const int q = q_lane + G.qs + 8 * j;
          const unsigned vo = (unsigned)q < (unsigned)a.M ? rg_voff : kOob;     // never touch memory outside the tensor
          ra[j] = buf_load4(rs_in, vo, G.a_soff + (unsigned)(j * 8 * ld4));
        };
        auto load_scale_c = [&](int c, float4& sc, float4& sh) {
          if (pro) {
            sc = *reinterpret_cast<const float4*>(op.in_scale + c * KC + a_col4 * 4);
            sh = *reinterpret_cast<const float4*>(op.in_shift + c * KC + a_col4 * 4);
            if (F16) {
              sc.x *= a_mul; sc.y *= a_mul; sc.z *= a_mul; sc.w *= a_mul;
              sh.x *= a_mul; sh.y *= a_mul; sh.z *= a_mul; sh.w *= a_mul;
            }
          }
        };
        auto stage_piece = [&](int buf, int j, const float4& sc, const float4& sh) {
          float4 v = ra[j];
          if (pro) {
            v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
          }
          unsigned char* dA = sG + buf * RG_TL + ((lane >> 3) + 8 * j) * A_ROW_B + ((j & 1) ? st_off_odd : st_off_even);
          if (j < 8 || (lane >> 3) < 2) {                        // the last piece is rows 64, 65 only
            if constexpr (NPL == 1) {
              if (!pro) { v.x *= a_mul; v.y *= a_mul; v.z *= a_mul; v.w *= a_mul; }
              uint2 h;
              h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, f16x2));
              h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, f16x2));
              *reinterpret_cast<uint2*>(dA) = h;
            } else if constexpr (F16) {
              if (!pro) { v.x *= a_mul; v.y *= a_mul; v.z *= a_mul; v.w *= a_mul; }
              uint2 h, l;
              split2h(v.x, v.y, h.x, l.x);
              split2h(v.z, v.w, h.y, l.y);
              *reinterpret_cast<uint2*>(dA) = h;
              *reinterpret_cast<uint2*>(dA + RG_PL) = l;
            } else {
              uint2 h, m, l;
              split4(v, h, m, l);
              *reinterpret_cast<uint2*>(dA) = h;
              *reinterpret_cast<uint2*>(dA + RG_PL) = m;
              *reinterpret_cast<uint2*>(dA + 2 * RG_PL) = l;
            }
          }
        };
        // LDS offsets of this lane's fragment rows for tap t, and the chunk swizzle of those rows (rows li+1+dx and
        // li+33+dx swizzle alike; the zero row holds zeros in every chunk)
        // (fa[rm]: plane 0; fa[2 + rm]: plane 1 -- SLIM keeps one block of zero rows per wave, behind its two buffers, so the
        //  plane offset of an out-of-image lane is 0 there and cannot be an immediate of the read)
        auto frag_addr = [&](int buf, int t, unsigned (&fa)[4], int& fsw) {
          const int dx = (int)(signed char)((tap_word(t) >> 8) & 0xff);
          fsw = ((li + 1 + dx) >> 2) & 3;
#pragma unroll
          for (int rm = 0; rm < 2; ++rm) {
            const bool ok = (fr_taps[rm] >> t) & 1u;
            const int row = li + rm * 32 + 1 + dx;
            if constexpr (SLIM) {
              fa[rm] = ok ? (unsigned)(row * A_ROW_B + buf * RG_TL) : (unsigned)(2 * RG_TL + (row & 15) * A_ROW_B);
              fa[2 + rm] = ok ? fa[rm] + (unsigned)RG_PL : fa[rm];
            } else {
              fa[rm] = (unsigned)((ok ? row : RG_ZERO_ROW + (row & 15)) * A_ROW_B) + (unsigned)(buf * RG_TL);
              fa[2 + rm] = fa[rm];
            }
          }
        };
        auto read_frags_g = [&](int s_, const unsigned (&fa)[4], int fsw, u32x4 (&af)[2][NPL]) {
          const unsigned chunk = (unsigned)(((s_ * 2 + lh) ^ fsw) << 4);
#pragma unroll
          for (int rm = 0; rm < 2; ++rm)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
              if constexpr (SLIM) af[rm][pl] = *reinterpret_cast<const u32x4*>(sG + fa[pl ? 2 + rm : rm] + chunk);
              else af[rm][pl] = *reinterpret_cast<const u32x4*>(sG + fa[rm] + chunk + pl * RG_PL);
            }
        };
        const int n_grp = n_chunks * 3;
        const int g_begin = (KS == 1) ? 0 : (n_grp * kh) / KS;
        const int g_end = (KS == 1) ? n_grp : (n_grp * (kh + 1)) / KS;
        if (g_begin < g_end) {
          int c2 = g_begin / 3, ky2 = g_begin - 3 * c2, idx2 = g_begin;
          auto next_group = [&]() {              // advance the cursor (clamped at the last group: repeats are harmless)
            if (idx2 + 1 < g_end) { ++idx2; if (++ky2 == 3) { ky2 = 0; ++c2; } }
            return group_ck(c2, ky2);
          };
          GroupInfo gc = group_ck(c2, ky2);
          GroupInfo gn = next_group();
          float4 sc_n = rsc_c, sh_n = rsh_c, sc_2 = rsc_c, sh_2 = rsh_c;
          {
            const TileInfo tb0 = tile_ct(gc.c, gc.t);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
              for (int rn = 0; rn < RN; ++rn) load_b(tb0, s_, rn);
            float4 sc0 = rsc_c, sh0 = rsh_c;
            load_scale_c(gc.c, sc0, sh0);
#pragma unroll
            for (int j = 0; j < 9; ++j) load_piece(gc, j);
            load_scale_c(gn.c, sc_n, sh_n);
            // while those loads fly: the zero rows and the per-lane tap masks
            if constexpr (SLIM) {
              *reinterpret_cast<u32x4*>(sG + 2 * RG_TL + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            } else {
#pragma unroll
              for (int buf = 0; buf < 2; ++buf)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)       // 16 rows x 64 bytes = 64 lanes x 16 bytes
                  *reinterpret_cast<u32x4*>(sG + buf * RG_TL + pl * RG_PL + RG_ZERO_ROW * A_ROW_B + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int rm = 0; rm < 2; ++rm) {
              const unsigned m = (unsigned)(m0 + rm * 32 + li);
              if ((int)m < a.M) {
                const unsigned b = fdiv(m, a.div_ghw);
                const unsigned rem = m - b * (unsigned)(g.GH * g.GW);
                const int gy = (int)fdiv(rem, a.div_gw);
                const int gx = (int)rem - gy * g.GW;
                for (int t = 0; t < 9; ++t) {
                  const int tp = tap_word(t);
                  const int iy = gy + (int)(signed char)(tp & 0xff), ix = gx + (int)(signed char)((tp >> 8) & 0xff);
                  if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) fr_taps[rm] |= 1u << t;
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) { stage_piece(0, j, sc0, sh0); load_piece(gn, j); }
          }
          __builtin_amdgcn_wave_barrier();
          unsigned fa[4];
          int fsw;
          frag_addr(0, gc.t, fa, fsw);
          read_frags_g(0, fa, fsw, afA);
          int bcur = 0;
#pragma unroll 1
          for (int G = g_begin; G < g_end; ++G) {
            const GroupInfo g2 = next_group();
            load_scale_c(g2.c, sc_2, sh_2);
            const TileInfo tc1 = tile_ct(gc.c, gc.t + 1), tc2 = tile_ct(gc.c, gc.t + 2), tn0 = tile_ct(gn.c, gn.t);
            const int bnxt = bcur ^ 1;
            auto side = [&](int h, int rn) {       // staging of the next group, spread over half-steps 0..4
#pragma unroll
              for (int j = 2 * h + (rn * 2) / RN; j < 2 * h + ((rn + 1) * 2) / RN; ++j)
                if (j < 9) { stage_piece(bnxt, j, sc_n, sh_n); load_piece(g2, j); }
            };
            read_frags_g(1, fa, fsw, afB);                          // tap 0, k-group 1
            mfma_group(0, afA, tc1, [&](int rn) { side(0, rn); });
            frag_addr(bcur, gc.t + 1, fa, fsw);
            read_frags_g(0, fa, fsw, afA);                          // tap 1, k-group 0
            mfma_group(1, afB, tc1, [&](int rn) { side(1, rn); });
            read_frags_g(1, fa, fsw, afB);
            mfma_group(0, afA, tc2, [&](int rn) { side(2, rn); });
            frag_addr(bcur, gc.t + 2, fa, fsw);
            read_frags_g(0, fa, fsw, afA);                          // tap 2, k-group 0
            mfma_group(1, afB, tc2, [&](int rn) { side(3, rn); });
            read_frags_g(1, fa, fsw, afB);
            mfma_group(0, afA, tn0, [&](int rn) { side(4, rn); });
            frag_addr(bnxt, gn.t, fa, fsw);
            read_frags_g(0, fa, fsw, afA);                          // next group, tap 0, k-group 0
            mfma_group(1, afB, tn0, [&](int) {});
            gc = gn; gn = g2; bcur = bnxt; sc_n = sc_2; sh_n = sh_2;
          }
        }
      }
    } else {
    const int it_begin = (KS == 1) ? 0 : (n_iter * kh) / KS;
    const int it_end = (KS == 1) ? n_iter : (n_iter * (kh + 1)) / KS;
    if (it_begin < it_end) {
      auto tile_at = [&](int it) { return tile_info(it < it_end ? it : it_end - 1); };   // repeats at the tail are harmless
      const TileInfo t0 = tile_at(it_begin);
      TileInfo t1 = tile_at(it_begin + 1), t2 = tile_at(it_begin + 2);
      // (chunk, tap) of the tile three ahead, advanced incrementally (a division per tile is ~18 scalar instructions)
      int idx3 = it_begin + 3 < it_end ? it_begin + 3 : it_end - 1;
      int c3 = idx3 / nt, tt3 = idx3 - c3 * nt;
      // prologue: tiles t0 and t1 are fetched together (one exposed memory latency instead of two); t1 sits in
      // registers that the fragment arrays take over afterwards
      float4 rb[8], sc1 = rsc_c, sh1 = rsh_c;
      unsigned pad1 = 0;
      {
        float4 sc0 = rsc_c, sh0 = rsh_c;
        unsigned pad0 = 0;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int rn = 0; rn < RN; ++rn) load_b(t0, s_, rn);
        if (!masks_done) { mark_tap(t0.t); mark_tap(t1.t); }     // just enough of the tap masks to get going
        load_scale(t0, sc0, sh0);
#pragma unroll
        for (int j = 0; j < 8; ++j) load_row(t0, j, ra[j], pad0);
        load_scale(t1, sc1, sh1);
#pragma unroll
        for (int j = 0; j < 8; ++j) load_row(t1, j, rb[j], pad1);
        if (!masks_done) {                                       // the rest of the masks while those loads fly
          for (int t = 0; t < n_taps; ++t) mark_tap(t);
          masks_done = true;
        }
        load_scale(t2, rsc_n, rsh_n);
#pragma unroll
        for (int j = 0; j < 8; ++j) { stage(it_begin & 1, j, ra[j], pad0, sc0, sh0); load_row(t2, j, ra[j], pad_n); }
      }
      __builtin_amdgcn_wave_barrier();
      read_frags(it_begin & 1, 0, afA);
      read_frags(it_begin & 1, 1, afB);
      mfma_group(0, afA, t1, [&](int rn) {        // the second tile is staged in the shadow of the first MFMAs
#pragma unroll
        for (int j = rn * 8 / RN; j < (rn + 1) * 8 / RN; ++j) stage((it_begin + 1) & 1, j, rb[j], pad1, sc1, sh1);
      });
      __builtin_amdgcn_wave_barrier();
      // bodies: k = it_begin .. it_end-2
      for (int k = it_begin; k + 1 < it_end; ++k) {
        const int bk = k & 1;                      // buffer of tile k (free: its last fragments are in afB) = tile k+2's
        const TileInfo t3 = tile_ct(c3, tt3);
        if (idx3 + 1 < it_end) { ++idx3; if (++tt3 == nt) { tt3 = 0; ++c3; } }
        rotate_stage_state();
        load_scale(t3, rsc_n, rsh_n);
        read_frags(bk ^ 1, 0, afA);                // tile k+1, k-group 0
        mfma_group(1, afB, t1, [&](int rn) {       // tile k, k-group 1; refill with tile k+1
#pragma unroll
          for (int j = rn * 4 / RN; j < (rn + 1) * 4 / RN; ++j) { stage_row(bk, j); load_a_row(t3, j); }
        });
        read_frags(bk ^ 1, 1, afB);                // tile k+1, k-group 1
        mfma_group(0, afA, t2, [&](int rn) {       // tile k+1, k-group 0; refill with tile k+2
#pragma unroll
          for (int j = 4 + rn * 4 / RN; j < 4 + (rn + 1) * 4 / RN; ++j) { stage_row(bk, j); load_a_row(t3, j); }
        });
        t1 = t2; t2 = t3;
      }
      mfma_group(1, afB, t1, [&](int) {});          // last tile, k-group 1
    }
    }

    if constexpr (F16) {
      // back to the tensors' own units (v_ldexp_f32: exact).  Under SUM2 the first pass is re-expressed in the second pass's
      // units instead, so that both inputs accumulate into one tile.
      const int k_this = (set ? ka1 : ka0) + (set ? kw1 : kw0);
      const int shift = (SUM2 && set == 0) ? (ka1 + kw1) - k_this : -k_this;
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc0[rm][rn][r] = __builtin_ldexpf(acc0[rm][rn][r], shift);
    }
    if (SUM2 && set == 0) continue;          // the second input accumulates on top; one exchange + epilogue after it
    // ---- split-K exchange.  The KS waves of a group hold partial sums of the same 64 x 32*RN tile.  Each of them OWNS
    //      a share of its 2*RN accumulator blocks (KS = 2: the row block rm == kh; KS = 4: row block kh & 1, column
    //      blocks of half kh >> 1), parks the blocks it does not own in LDS, adds the other waves' parts of the blocks
    //      it owns (in increasing kh order: deterministic) and then runs the epilogue for those only -- the exchange
    //      and the epilogue are shared by the group instead of being one wave's job while the others idle. ----
    auto owns = [&](int rm, int rn) {
      if (KS == 1) return true;
      if (KS == 2) return rm == kh;
      return rm == (kh & 1) && ((rn * 2) / RN) == (kh >> 1);
    };
    if constexpr (KS > 1 && !(CV_EXP & 4)) {
      constexpr int BLK = 16 * 64;                       // floats of one accumulator block
      constexpr int STRIDE = (KS == 2 ? RN : 2 * RN - RN / 2) * BLK;    // most blocks a wave can have to park
      static_assert(4 * STRIDE * 4 <= 4 * WAVE_B, "exchange area must fit in the A-tile region");
      __syncthreads();                                   // every wave is done with its A tiles
      float* ex_all = reinterpret_cast<float*>(sA_all);  // [wave][parked blocks, in (rm, rn) order][16][64]
      auto owned_by = [&](int k, int rm, int rn) {       // ownership rule for the wave with K part k
        return KS == 2 ? rm == k : (rm == (k & 1) && ((rn * 2) / RN) == (k >> 1));
      };
      if constexpr (KS == 2) {                           // parked: the other row block, slot = column block
        float* mine = ex_all + (size_t)wave * STRIDE;
        const float* theirs = ex_all + (size_t)(wave ^ 1) * STRIDE;
#pragma unroll
        for (int rm = 0; rm < 2; ++rm)
          if (rm != kh) {
#pragma unroll
            for (int rn = 0; rn < RN; ++rn)
#pragma unroll
              for (int r = 0; r < 16; ++r) mine[(rn * 16 + r) * 64 + lane] = acc0[rm][rn][r];
          }
        __syncthreads();
#pragma unroll
        for (int rm = 0; rm < 2; ++rm)
          if (rm == kh) {
#pragma unroll
            for (int rn = 0; rn < RN; ++rn)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc0[rm][rn][r] += theirs[(rn * 16 + r) * 64 + lane];
          }
      } else {
        {
          float* mine = ex_all + (size_t)wave * STRIDE;
          int slot = 0;
#pragma unroll
          for (int rm = 0; rm < 2; ++rm)
#pragma unroll
            for (int rn = 0; rn < RN; ++rn)
              if (!owned_by(kh, rm, rn)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(slot * 16 + r) * 64 + lane] = acc0[rm][rn][r];
                ++slot;
              }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < KS; ++p) {
          if (p == kh) continue;
          const float* theirs = ex_all + (size_t)(wave - kh + p) * STRIDE;
          int slot = 0;                                  // position among the blocks wave p parked
#pragma unroll
          for (int rm = 0; rm < 2; ++rm)
#pragma unroll
            for (int rn = 0; rn < RN; ++rn)
              if (!owned_by(p, rm, rn)) {
                if (owned_by(kh, rm, rn)) {
#pragma unroll
                  for (int r = 0; r < 16; ++r) acc0[rm][rn][r] += theirs[(slot * 16 + r) * 64 + lane];
                }
                ++slot;
              }
        }
      }
      if (ACC1) __syncthreads();                         // the exchange area is the next pass's A tiles
    }

    // ---- epilogue (branch-free: buffer stores/loads, rows beyond M get an out-of-range offset and are dropped) ----
    const int oset = SUM2 ? 0 : set;             // which output this epilogue writes
    float* outp = oset ? op.out1 : op.out0;
    const int cout = oset ? g.Cout1 : g.Cout0;
    double* stats = oset ? op.stats1 : op.stats0;
    const bool masked = (oset == 0) && op.mask_src != nullptr;
    const bool accumulate = (oset == 0) && (a.flags & 1);
    // fused output stage (inference): y = [relu](es * conv + et) [+ as * add_src + at], and max |y| for the next convolution
    const bool epi = (oset == 0) && op.epi_scale0 != nullptr;
    const bool epi_add = epi && op.add_src != nullptr;
    const bool epi_relu = (a.flags & MPOSE_CONV_EPI_RELU0) != 0;
    float epi_amax = 0.f;
    const bool plain_amax = !epi && (oset == 0) && op.out0_amax != nullptr;      // max |out0| as stored, without the fused output stage
    // BatchNorm-backward sums of the tensor's consumer, taken from the values as they are stored (see mpose_conv_operands.red_*)
    const bool red = (oset == 0) && op.red_sums != nullptr;
    float rs0[RN], rs1[RN], rs2[RN], rs3[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) rs0[rn] = rs1[rn] = rs2[rn] = rs3[rn] = 0.f;
    const int old_ = oset ? g.out_ld1 : g.out_ld0;
    const int out_ld = old_ > 0 ? old_ : cout;
    const unsigned out_bytes = (unsigned)((((long)g.B * g.OH * g.OW - 1) * out_ld + cout) * 4);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(outp, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? op.mask_src : outp), 0, out_bytes, 0x00020000);
    float csum[RN], csq[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) csum[rn] = csq[rn] = 0.f;
    // per-channel extremes of out0 as stored (mpose_conv_operands.mm0): max v and max -v over the rows that exist
    const bool want_mm = (oset == 0) && op.mm0 != nullptr;
    const float kNegInf = __uint_as_float(0xff800000u);
    float vmx[RN], vng[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) vmx[rn] = vng[rn] = kNegInf;
    // Output row table: lane l works out the byte offset of output pixel m0 + l ONCE (two divisions per lane
    // instead of two per accumulator row); rows beyond M, and every row of a non-writing wave, get an offset the
    // buffer unit rejects (stores dropped, loads return 0).  Accumulator register group rg of lane half h holds
    // rows 8*rg + 4*h .. +3 = one ds_read_b128 of the table.
    {
      const unsigned m = (unsigned)(m0 + lane);
      const unsigned mm = (int)m < a.M ? m : 0u;
      const unsigned b = fdiv(mm, a.div_ghw);
      const unsigned rem = mm - b * (unsigned)(g.GH * g.GW);
      const unsigned gy = fdiv(rem, a.div_gw);
      const unsigned gx = rem - gy * (unsigned)g.GW;
      const unsigned pix = (b * (unsigned)g.OH + (gy * g.out_mul + oyc)) * (unsigned)g.OW + (gx * g.out_mul_x + oxc);
      const bool ok = (int)m < a.M;
      sRow[wave * 64 + lane] = ok ? pix * (unsigned)out_ld * 4u : 0xFFFFF000u;
      __builtin_amdgcn_wave_barrier();
    }
    const unsigned col_off = (unsigned)((n0 + li) * 4);
#pragma unroll
    for (int rm = 0; rm < ((CV_EXP & 1) ? 0 : 2); ++rm) {
      unsigned voff[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const u32x4 e = *reinterpret_cast<const u32x4*>(sRow + wave * 64 + rm * 32 + 8 * rg + 4 * lh);
        voff[4 * rg] = e.x + col_off; voff[4 * rg + 1] = e.y + col_off; voff[4 * rg + 2] = e.z + col_off; voff[4 * rg + 3] = e.w + col_off;
      }
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const int nb = n0 + rn * 32;                 // wave-uniform: the whole 32-column group is in or out (cout % 32 == 0)
        if (nb < cout && owns(rm, rn)) {
          const int n = nb + li;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc0[rm][rn][r];
          if (masked) {
            const float msc = op.mask_scale[n], msh = op.mask_shift[n];
            float src[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) src[r] = buf_load1(rs_m, voff[r] + (unsigned)(rn * 128), 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (!(fmaf(src[r], msc, msh) > 0.f)) v[r] = 0.f;
              csq[rn] = fmaf(v[r], src[r], csq[rn]);
            }
          }
          if (accumulate) {
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = buf_load1(rs_o, voff[r] + (unsigned)(rn * 128), 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += old[r];
          }
          if (epi) {
            const float es = op.epi_scale0[n], et = op.epi_shift0[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              v[r] = fmaf(v[r], es, et);
              if (epi_relu) v[r] = fmaxf(v[r], 0.f);
            }
            if (epi_add) {
              const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.add_src), 0, out_bytes, 0x00020000);
              const float as = op.add_scale[n], at = op.add_shift[n];
              float ad[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) ad[r] = buf_load1(rs_a, voff[r] + (unsigned)(rn * 128), 0);
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] += fmaf(ad[r], as, at);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)       // (rows beyond M: their stores are dropped, but relu(et) + at need not be zero)
              epi_amax = fmaxf(epi_amax, voff[r] < 0xFFFFF000u ? fabsf(v[r]) : 0.f);
          }
          if (red) {
            const __amdgpu_buffer_rsrc_t rs_ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.red_a), 0, out_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.red_b), 0, out_bytes, 0x00020000);
            const float ms = op.red_scale[n], mt = op.red_shift[n];
            float xa[16], xb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              xa[r] = buf_load1(rs_ra, voff[r] + (unsigned)(rn * 128), 0);
              xb[r] = buf_load1(rs_rb, voff[r] + (unsigned)(rn * 128), 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {       // (rows beyond M hold v = 0 and read 0: they add nothing)
              const float ga = fmaf(xa[r], ms, mt) > 0.f ? v[r] : 0.f;
              rs0[rn] += ga;
              rs1[rn] = fmaf(ga, xa[r], rs1[rn]);
              rs2[rn] += v[r];
              rs3[rn] = fmaf(v[r], xb[r], rs3[rn]);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_o, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0);
            // rows beyond M accumulated zeros (their inputs were read as 0); blocks of other owners are skipped above
            csum[rn] += v[r];
            if (!masked) csq[rn] = fmaf(v[r], v[r], csq[rn]);
          }
          if (plain_amax) {
#pragma unroll
            for (int r = 0; r < 16; ++r) epi_amax = fmaxf(epi_amax, fabsf(v[r]));      // (rows beyond M hold zeros unless accumulated into: the caller's bound stays a bound)
          }
          if (want_mm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool ok = voff[r] < 0xFFFFF000u;
              vmx[rn] = fmaxf(vmx[rn], ok ? v[r] : kNegInf);
              vng[rn] = fmaxf(vng[rn], ok ? -v[r] : kNegInf);
            }
          }
        }
      }
    }
    if (want_mm) {       // (waves that own no block of a column group contribute the identity)
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float a_ = fmaxf(vmx[rn], __shfl_xor(vmx[rn], 32, 64)), b_ = fmaxf(vng[rn], __shfl_xor(vng[rn], 32, 64));
        if (lh == 0) { sMM[(wave * BN + rn * 32 + li) * 2] = a_; sMM[(wave * BN + rn * 32 + li) * 2 + 1] = b_; }
      }
    }
    if ((epi || plain_amax) && op.out0_amax != nullptr) {      // one look-then-atomic per wave into the workgroup's sub-slot (common.h)
      float m = wave_max(epi_amax);
      if (lane == 0) {
        if (!(m == m)) m = __uint_as_float(0x7f800000u);
        unsigned* dst = reinterpret_cast<unsigned*>(op.out0_amax + (bx % MPOSE_AMAX_SUBSLOTS) * MPOSE_AMAX_STRIDE);
        if (__float_as_uint(m) > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, __float_as_uint(m));
      }
    }
    if (red) {       // (sRed is free: red_* and stats* exclude each other) [4 waves][BN][4]
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float t0 = rs0[rn] + __shfl_xor(rs0[rn], 32, 64), t1 = rs1[rn] + __shfl_xor(rs1[rn], 32, 64);
        const float t2 = rs2[rn] + __shfl_xor(rs2[rn], 32, 64), t3 = rs3[rn] + __shfl_xor(rs3[rn], 32, 64);
        if (lh == 0) *reinterpret_cast<float4*>(sRed + (wave * BN + rn * 32 + li) * 4) = make_float4(t0, t1, t2, t3);
      }
    }
    if (stats != nullptr) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float s_ = csum[rn] + __shfl_xor(csum[rn], 32, 64);
        const float q_ = csq[rn] + __shfl_xor(csq[rn], 32, 64);
        if (lh == 0) {
          float* d = sRed + ((oset * 4 + wave) * BN + rn * 32 + li) * 2;
          d[0] = s_; d[1] = q_;
        }
      }
    }
  }
  __syncthreads();
  // MPOSE_CONV_STATS_PART: this workgroup's sums go to row blockIdx.x of fp32 partial buffers with plain stores (the finalize /
  // coefficient kernels add the rows up); otherwise fp64 atomics into the (C, 2) / (C, 4) accumulators.
  const bool part = (a.flags & MPOSE_CONV_STATS_PART) != 0;
  const int prow = a.part_row0 + (int)bx;
  const bool hdr_writer = prow == 0 && by == 0;
#pragma unroll
  for (int set = 0; set < (ACC1 ? 2 : 1); ++set) {
    double* stats = set ? op.stats1 : op.stats0;
    const int cout = set ? g.Cout1 : g.Cout0;
    if (stats != nullptr && tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float* d = sRed + ((set * 4 + w) * BN + tid) * 2;
        s += d[0]; q += d[1];
      }
      const int n = n0 + tid;
      if (n < cout && !(CV_EXP & 64)) {
        if (part) {
          float* pb = reinterpret_cast<float*>(stats);
          if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
          reinterpret_cast<float2*>(pb + kPartHdr)[(size_t)prow * cout + n] = make_float2(s, q);
        } else {
          atomicAdd(stats + (size_t)n * 2, (double)s);
          atomicAdd(stats + (size_t)n * 2 + 1, (double)q);
        }
      }
    }
  }
  if (op.mm0 != nullptr && tid < BN && n0 + tid < g.Cout0) {
    float a_ = sMM[tid * 2], b_ = sMM[tid * 2 + 1];
#pragma unroll
    for (int w = 1; w < 4; ++w) { a_ = fmaxf(a_, sMM[(w * BN + tid) * 2]); b_ = fmaxf(b_, sMM[(w * BN + tid) * 2 + 1]); }
    if (part) {
      float* pb = reinterpret_cast<float*>(op.mm0);
      if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
      reinterpret_cast<float2*>(pb + kPartHdr)[(size_t)prow * g.Cout0 + n0 + tid] = make_float2(a_, b_);
    } else {
      atomicMax(op.mm0 + (size_t)(n0 + tid) * 2, float_key(a_));
      atomicMax(op.mm0 + (size_t)(n0 + tid) * 2 + 1, float_key(b_));
    }
  }
  if (op.red_sums != nullptr && tid < BN && n0 + tid < g.Cout0) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 d = *reinterpret_cast<const float4*>(sRed + (w * BN + tid) * 4);
      t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
    }
    if (part) {
      float* pb = reinterpret_cast<float*>(op.red_sums);
      if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
      reinterpret_cast<float4*>(pb + kPartHdr)[(size_t)prow * g.Cout0 + n0 + tid] = t;
    } else if (!(CV_EXP & 64)) {
      double* d = op.red_sums + (size_t)(n0 + tid) * 4;
      atomicAdd(d, (double)t.x); atomicAdd(d + 1, (double)t.y); atomicAdd(d + 2, (double)t.z); atomicAdd(d + 3, (double)t.w);
    }
  }
  // The last workgroup of this group's launch to get here turns the statistics into the BatchNorm vectors (mpose_bn_finalize's
  // job, common.h).  Every workgroup waits until its device-scope atomics have been ACKNOWLEDGED (vmcnt: they are performed at
  // the memory side, beyond the XCDs' private L2s) before it draws its ticket there too, so whoever draws the last ticket has
  // all of them behind it and reads them with device-scope loads.  No workgroup waits for another, and no cache is flushed: a
  // __threadfence() here -- an agent-scope release writes the XCD's dirty L2 lines back -- cost 2.8 ms per training step.
  if (op.fin_count != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* ticket = reinterpret_cast<unsigned*>(sRow);
    if (tid == 0) *ticket = __hip_atomic_fetch_add(op.fin_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*ticket == gridDim.x * gridDim.y - 1) {
      bn_finalize_job<true>(*reinterpret_cast<const mpose_bn_job*>(op.fin0), 1, op.fin_eps, op.fin_momentum);
      if (ACC1 && op.fin1 != nullptr) bn_finalize_job<true>(*reinterpret_cast<const mpose_bn_job*>(op.fin1), 1, op.fin_eps, op.fin_momentum);
      if (tid == 0) __hip_atomic_store(op.fin_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int RN, int MODE, int KS, bool PRO, int NPL, bool ROWG>
int launch_conv(const ConvArgs& a0, int n_groups, hipStream_t s) {
  constexpr int BN = 32 * RN;
  constexpr bool SLIM = slim_tile<RN, MODE, NPL, ROWG>();
  constexpr int wave_b = SLIM ? 2 * 2 * RG_SLIM_ROWS * A_ROW_B + 16 * A_ROW_B : 2 * (ROWG ? RG_TILE_B : ((NPL <= 2 && KS == 1) ? 2 * A_PLANE_B : A_TILE_B));
  constexpr int lds = 4 * wave_b + 2 * 4 * BN * 2 * 4 + 4 * 64 * 4 + 4 * BN * 2 * 4;
  static_assert(!SLIM || 2 * lds <= 160 * 1024, "two workgroups per CU");
  if (mpose_dry_rows) { *mpose_dry_rows += ((a0.M + 256 / KS - 1) / (256 / KS)) * a0.g.n_classes; return 0; }     // (mpose_conv_stat_rows)
  static bool attr_set = false;          // > 64 KiB of dynamic LDS has to be requested once per kernel
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_k<RN, MODE, KS, PRO, NPL, ROWG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MPOSE_EINVAL;
    attr_set = true;
  }
  ConvArgs a = a0;
  a.n_mtiles = (a.M + 256 / KS - 1) / (256 / KS);
  const int cmax = a.g.Cout1 > a.g.Cout0 ? a.g.Cout1 : a.g.Cout0;
  dim3 grid(a.n_mtiles * a.g.n_classes, (cmax + BN - 1) / BN, n_groups);
  a.part_row0 = mpose_part_phase.row0;
  a.part_rows = mpose_part_phase.total > 0 ? mpose_part_phase.total : (int)grid.x;
  // (row-group launches only -- the columns' convolutions: 308 -> 226 MB per launch on the two-input data gradient, 131 -> 73 MB on
  //  the 64-channel 3x3, durations within 1 %; the feature extractor's wide launches, several channel tiles per pixel tile with an
  //  unsplit K, measured 17-25 % SLOWER in this order and keep the plain one: tools/pmc_xcd.sh)
  a.xcd_order = (ROWG && (grid.x & 7u) == 0 && grid.x >= 16) ? 1 : 0;
  launch(conv_igemm_k<RN, MODE, KS, PRO, NPL, ROWG>, dim3(grid), dim3(256), lds, s, a);
  return launch_status();
}

// Split-K factor: one workgroup (4 waves, one per SIMD) owns a CU, so a launch runs in ceil(WGs / 256) rounds, and
// every round pays a fixed cost (prologue burst, exchange, epilogue, workgroup turnover) next to its share of the K
// loop.  Cost model in microseconds, constants from the round-1 ablations (tools/experiments/):
//     rounds * (9.5 + exchange(ks) + ceil(n_iter / ks) * (0.65 + 0.49 * RN))
// -- e.g. 128->128 @32^2: KS 1/2/4 = 207/174/213 predicted, 195/175/202 measured; 192->192 @16^2: 124/136/125
// predicted, 98/108/103 measured (same order).  The choice is made for a NOMINAL batch of 32 images, not the actual
// one: the summation order of a sample then does not depend on how many other samples share its launch, so a
// data-parallel shard reproduces the full batch's per-sample results bit for bit in the six-product form (MPOSE_F16X3=0); in the
// default three-product form a tensor's scale follows the largest magnitude in the LOCAL batch, so shards agree to fp32 rounding.
template <int RN, int NPL, bool ROWG, int MODE>
inline int pick_ks(const ConvArgs& a, int cmax, int n_groups) {
  const int n_iter = (a.g.Cin / KC) * a.g.cls[0].n_taps;
  const long m_nominal = 32l * a.g.GH * a.g.GW;
  int best = 1;
  double best_cost = 0.0;
  // (two workgroups per CU: 512 slots per round; and the split that bounds the accumulation chain of a training launch -- an
  //  unsplit 128-channel 3x3 is ONE chain of 216 -- is kept whatever the model says: max error 8.8e-7 vs 5.4e-7 of the fp64 result
  //  where torch's fp32 convolution shows 2.8e-7)
  constexpr bool SLIM = slim_tile<RN, MODE, NPL, ROWG>();
  // (round 4: unsplit is the default in training too -- on a common ReLU piece the step's gradients sit at 1.14x the fp32 CPU path's
  //  median error with every 128-channel launch in this form (tests/test_grad_parity_gpu.py, 'igemm' case: 4.0e-6 / max 1.3e-5),
  //  the contract is 1e-4, and the step gains 0.2 ms: 23.68 -> 23.47.  MPOSE_SLIM=1 restores the split for training launches.)
  const bool chain_bound = false;      // (round 3 kept training's two-way K split on these launches; round 4 measured the unsplit form inside the gates)
#ifdef CV_KS_FORCE          // (launch-plan experiments, debug builds: -DCV_KS_FORCE=<RN * 10 + ks>, e.g. 32 = the 96-wide tiles split two ways)
  if (RN == CV_KS_FORCE / 10 && n_iter >= 2 * (CV_KS_FORCE % 10)) return CV_KS_FORCE % 10;
#endif
  for (int ks = 1; ks <= (RN > 1 ? 4 : 2); ks *= 2) {
    if (ks > 1 && n_iter < 2 * ks) break;
    const long wgs = ((m_nominal + 256 / ks - 1) / (256 / ks)) * a.g.n_classes * ((cmax + 32 * RN - 1) / (32 * RN)) * n_groups;
    const long rounds = SLIM ? (wgs + 511) / 512 : (wgs + 255) / 256;
    const double exchange = ks == 1 ? 0.0 : (ks == 2 ? 1.5 : 2.5);
    // (three instead of six products per k-group.  With row-group staging a tap costs ~0.3 us per 32 output channels -- the
    //  192-channel layers: 2.7 us per row group of three taps at RN = 3 -- and the 128-channel launch measured FASTER with 384
    //  workgroups of twelve row groups than with 768 of six (25.5 vs 26.2-27.1 ms per training step), which puts the fixed
    //  cost of a round at >= 17 us there.  But an unsplit 128-channel 3x3 is one chain of 216 accumulations per block instead
    //  of two of 108, and the gradient-parity gates felt it (same-piece median 4.2e-6 -> 5.5e-6, 1.6x the fp32 oracle's): the
    //  faster constants are used for INFERENCE launches only -- those with the fused output stage -- where the forward error
    //  stays at ~1e-5 of the 1e-4 gate; training keeps the split that bounds the chain.)
    const bool fast = ROWG && a.op[0].epi_scale0 != nullptr;
    const double per_iter = fast ? 0.05 + 0.29 * RN : (NPL <= 2 ? 0.55 + 0.27 * RN : 0.65 + 0.49 * RN);
    const double fixed = fast ? 18.0 : 9.5;
    const double cost = (double)rounds * (fixed + exchange + (double)((n_iter + ks - 1) / ks) * per_iter);
    if (ks == 1 || cost < 0.97 * best_cost || (chain_bound && best == 1)) { best = ks; best_cost = cost; }     // ties go to the smaller split
  }
  return best;
}

// Row-group staging applies to the first pass of a stride-1 3x3 over an input of the output's size: one class,
// nine acc == 0 taps first, each consecutive triple sharing dy with dx in {-1, 0, 1} (any dy: a kernel dilated along y --
// models/chatterbox_model.py:143-150 -- stages the 66 pixels around m0 + dy * IW like any other row).
inline bool rowg_eligible(const mpose_conv_geom& g) {
  if (g.n_classes != 1 || g.in_mul != 1 || g.in_mul_x != 1 || g.IH != g.GH || g.IW != g.GW || (g.Cin % KC)) return false;
  const mpose_tap_class& c = g.cls[0];
  int n0 = 0;
  for (int t = 0; t < c.n_taps; ++t) {
    if (c.taps[t].acc == 0) { if (n0 != t) return false; ++n0; }
  }
  if (n0 != 9) return false;
  for (int t = 0; t < 9; ++t) {
    if (c.taps[t].dy != c.taps[3 * (t / 3)].dy) return false;
    if (c.taps[t].dx < -1 || c.taps[t].dx > 1) return false;
  }
  return true;
}

template <int RN, int MODE, bool PRO, int NPL, bool ROWG>
int launch_conv_kp(const ConvArgs& a, int cmax, int n_groups, hipStream_t s) {
  const int ks = pick_ks<RN, NPL, ROWG, MODE>(a, cmax, n_groups);
  if constexpr (RN > 1) {                  // (32-wide tiles never profit from a 4-way split)
    if (ks == 4) return launch_conv<RN, MODE, 4, PRO, NPL, ROWG>(a, n_groups, s);
  }
  if (ks >= 2) return launch_conv<RN, MODE, 2, PRO, NPL, ROWG>(a, n_groups, s);
  return launch_conv<RN, MODE, 1, PRO, NPL, ROWG>(a, n_groups, s);
}
template <int RN, int NPL, bool ROWG>
int launch_conv_ks_p(const ConvArgs& a, int mode, int cmax, int n_groups, hipStream_t s) {
  if (mode == 1) return launch_conv_kp<RN, 1, false, NPL, ROWG>(a, cmax, n_groups, s);
  if (mode == 2) return launch_conv_kp<RN, 2, false, NPL, ROWG>(a, cmax, n_groups, s);
  if (a.op[0].in_scale != nullptr) return launch_conv_kp<RN, 0, true, NPL, ROWG>(a, cmax, n_groups, s);
  return launch_conv_kp<RN, 0, false, NPL, ROWG>(a, cmax, n_groups, s);
}
static constexpr int rowg_env() { return 1; }      // (the row-group loop: a run-time switch until round 6)
template <int RN>
int launch_conv_ks(const ConvArgs& a, int mode, int cmax, int n_groups, hipStream_t s) {
  if (a.flags & MPOSE_CONV_F16X1) {          // (with MPOSE_CONV_F16X3: same operands and scales, the h x h product only)
    if constexpr (RN > 1) {
      if (rowg_env() && rowg_eligible(a.g)) return launch_conv_ks_p<RN, 1, true>(a, mode, cmax, n_groups, s);
    }
    return launch_conv_ks_p<RN, 1, false>(a, mode, cmax, n_groups, s);
  }
  if (a.flags & MPOSE_CONV_F16X3) {
    // (round 5: the 32-wide tiles too -- the feature extractor's 32-channel 3x3 layers and the columns' last 32-channel block spent
    //  their loop on the split, nine times per input element; MPOSE_CONV_ROWG=2 restores the per-tap loop for them)
    if (rowg_env() && (RN > 1 || rowg_env() != 2) && rowg_eligible(a.g)) return launch_conv_ks_p<RN, 2, true>(a, mode, cmax, n_groups, s);
    return launch_conv_ks_p<RN, 2, false>(a, mode, cmax, n_groups, s);
  }
  return launch_conv_ks_p<RN, 3, false>(a, mode, cmax, n_groups, s);
}

#if CV_PART == 0     // (weight gradients, weight packing and the C entry points: the main translation unit only)
// ---------------------------------------------------------------------------------------------
// Weight gradient: dWp[split][widx][k/4][n][4] = sum_{slots in split} X_tap[slot][k] * G[slot][n]
// MFMA roles: i = k (input channel), j = n (output channel), reduction index = slot (pixel); bf16x6 like the
// forward kernel (both operands are split in registers, products exact, fp32 accumulation).
//
// v_mfma_f32_32x32x16_bf16 wants, per lane (i, h), EIGHT consecutive reduction indices = 8 pixels of one channel.
// In NHWC that is 8 dword loads with a pixel stride, each one a coalesced 128-byte line across the 32 lanes of
// a half -- so the operands go straight from L2 to registers (no LDS), are split there (5.5 VALU per element)
// and feed 6*KB*NB MFMAs per 16-pixel group.  The wave tile is as fat as the accumulator file allows
// (32*KB input x 32*NB output channels, up to 128 x 128 = all 256 AGPRs) because the split cost per MFMA falls
// with (KB+NB)/(KB*NB); one wave per SIMD, the next group's 8*(KB+NB) loads in flight during the current MFMAs.
// Pixels are walked in OCTETS (8 consecutive slots of one slot row; GW % 8 == 0): lane half h of a group takes
// octet 2q+h.  The scalar unit walks rows/octets (padding rows are skipped); columns shifted out of the image by
// the tap are zeroed per pixel, and anything outside the tensor reads 0 through the buffer range check.
// A workgroup = 4 waves that split the slot rows of one (tap, tile, split); their tiles are summed in a fixed
// order through LDS (deterministic) and written as split-K partials for unpack_wgrads_k.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  mpose_conv_geom g;
  mpose_wgrad_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gh;
  int n_split, rows_per_split;    // slot rows (b, gy) per split; a workgroup's 4 waves share one split
  int n_entries;                  // flat (class, tap) entries
  int entry_cls[MPOSE_MAX_CLASSES * MPOSE_MAX_TAPS];
  int entry_tap[MPOSE_MAX_CLASSES * MPOSE_MAX_TAPS];
  int n_widx0, n_widx1;
  // XCD-aware work order: workgroup b runs on XCD b % 8 (observed, speed only), and each XCD has a private 4 MiB
  // L2.  The work list is ordered (group, pixel split) outermost, (k/n tile, tap) innermost, and XCD c takes the
  // contiguous chunk c of it, so all taps x tiles that re-read one pixel split's X and dY slices run on ONE L2 at
  // about the same time (instead of every L2 pulling every slice: measured 5-9x over-fetch before).
  int n_ytiles, n_groups, chunk, total;
};

// F16 (operands carry in_amax / gout*_amax): both operands are scaled by their tensor's power of two and split into TWO fp16
// values, three products per block (MPOSE_CONV_F16X3 in the header), accumulators scaled back before the cross-wave sum.
template <int KB, int NB, bool F16>   // wave tile: 32*KB input channels x 32*NB output channels
__global__ __launch_bounds__(256, 1) void conv_wgrad_k(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // cross-wave reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const mpose_conv_geom& g = a.g;
  const unsigned slot_id = blockIdx.x >> 3, work = (blockIdx.x & 7u) * (unsigned)a.chunk + slot_id;
  if (slot_id >= (unsigned)a.chunk || work >= (unsigned)a.total) return;
  unsigned wrk = work;
  const int entry = wrk % (unsigned)a.n_entries; wrk /= (unsigned)a.n_entries;
  const int ytile = wrk % (unsigned)a.n_ytiles; wrk /= (unsigned)a.n_ytiles;
  const int split = wrk % (unsigned)a.n_split;
  const int group = wrk / (unsigned)a.n_split;
  const int cls = a.entry_cls[entry];
  const mpose_tap tap = g.cls[cls].taps[a.entry_tap[entry]];
  const bool second = tap.acc != 0;
  const int npad = second ? g.Npad1 : g.Npad0;
  const int cout = second ? g.Cout1 : g.Cout0;
  const int n_ctiles = cout / (32 * NB);
  const int k_tile = ytile / n_ctiles, n_tile = ytile - k_tile * n_ctiles;
  const int k0 = k_tile * 32 * KB, n0 = n_tile * 32 * NB;
  const mpose_wgrad_operands& op = a.op[group];
  const float* gout = second ? op.gout1 : op.gout0;
  float* dw = second ? op.dw1 : op.dw0;
  const int n_widx = second ? a.n_widx1 : a.n_widx0;
  const int oyc = g.cls[cls].oy, oxc = g.cls[cls].ox;
  const int dy = tap.dy, dx = tap.dx;

  // slot rows of this wave
  const int n_rows = g.B * g.GH;
  const int r_split0 = split * a.rows_per_split;
  const int r_split1 = min(n_rows, r_split0 + a.rows_per_split);
  const int rpw = (a.rows_per_split + 3) >> 2;
  const int r_begin = min(r_split1, r_split0 + wave * rpw);
  const int r_end = min(r_split1, r_begin + rpw);
  // valid slot columns for this tap: 0 <= gx*in_mul_x + dx < IW
  int gx_lo = 0, gx_hi = g.GW;
  while (gx_lo < g.GW && gx_lo * g.in_mul_x + dx < 0) ++gx_lo;
  while (gx_hi > gx_lo && (gx_hi - 1) * g.in_mul_x + dx >= g.IW) --gx_hi;
  const int n_oct = g.GW >> 3;

  const int in_ld = g.in_ld > 0 ? g.in_ld : g.Cin;
  const int gld_ = second ? g.out_ld1 : g.out_ld0;
  const int g_ld = gld_ > 0 ? gld_ : cout;
  // exact extents: a pixel outside the tensor (a negative offset wraps to a huge unsigned one) reads 0
  const unsigned x_bytes = (unsigned)(((long)g.B * g.IH * g.IW - 1) * in_ld * 4 + (long)g.Cin * 4);
  const unsigned g_bytes = (unsigned)(((long)g.B * g.OH * g.OW - 1) * g_ld * 4 + (long)cout * 4);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.in), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, g_bytes, 0x00020000);
  const int x_pix = g.in_mul_x * in_ld * 4, g_pix = g.out_mul_x * g_ld * 4;        // byte stride between consecutive slots
  const int x_lane = (k0 + li) * 4, g_lane = (n0 + li) * 4;
  const bool pro = op.in_scale != nullptr;
  const int kx = F16 ? f16_scale_exp(amax_gather(op.in_amax)) : 0;
  const int kg = F16 ? f16_scale_exp(amax_gather(second ? op.gout1_amax : op.gout0_amax)) : 0;
  const float x_mul = pow2f(kx), g_mul = pow2f(kg);
  const bool x1 = F16 && a.op[0].single_product != 0;
  float psc[KB], psh[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    psc[kb] = pro ? op.in_scale[k0 + kb * 32 + li] : 1.f;
    psh[kb] = pro ? op.in_shift[k0 + kb * 32 + li] : 0.f;
    if (F16 && pro) { psc[kb] *= x_mul; psh[kb] *= x_mul; }      // relu(s x + t) 2^k == relu((s 2^k) x + t 2^k)
  }

  // ---- scalar cursor over (slot row, octet); only rows whose tap-shifted input row is in bounds ----
  struct Cursor { int r, o; int xs, gs; };       // xs/gs: byte offsets of the row's first slot (tap shift applied to x)
  auto seek = [&](Cursor& c) {                   // move to the first valid row at or after c.r
    while (c.r < r_end) {
      const unsigned b = fdiv((unsigned)c.r, a.div_gh);
      const int gy = c.r - (int)b * g.GH;
      const int iy = gy * g.in_mul + dy;
      c.xs = __builtin_amdgcn_readfirstlane((((int)b * g.IH + iy) * g.IW + dx) * in_ld * 4);
      c.gs = __builtin_amdgcn_readfirstlane((((int)b * g.OH + gy * g.out_mul + oyc) * g.OW + oxc) * g_ld * 4);
      if (iy >= 0 && iy < g.IH) break;
      ++c.r;
    }
  };
  struct Octet { int xs, gs; unsigned mask; };
  auto take = [&](Cursor& c) {                   // current octet (or an all-masked dummy past the end), then advance
    Octet o;
    if (c.r < r_end) {
      const int gx0 = c.o * 8;
      o.xs = c.xs + gx0 * x_pix;
      o.gs = c.gs + gx0 * g_pix;
      const int lo = min(8, max(0, gx_lo - gx0)), hi = min(8, max(0, gx_hi - gx0));
      o.mask = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
      if (++c.o == n_oct) { c.o = 0; ++c.r; seek(c); }
    } else {
      o.xs = 0; o.gs = 0; o.mask = 0;
    }
    return o;
  };
  Cursor cur; cur.r = r_begin; cur.o = 0; cur.xs = 0; cur.gs = 0;
  seek(cur);
  int n_groups16 = 0;                            // 16-pixel groups of this wave
  {
    Cursor cnt = cur;
    int octs = 0;
    while (cnt.r < r_end) { octs += n_oct; ++cnt.r; seek(cnt); }
    n_groups16 = (octs + 1) >> 1;
  }

  f32x16 acc[KB][NB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kb][nb][r] = 0.0f;

  // Software pipeline (one wave per SIMD; VALU work hides only in the shadow of independent MFMAs, ~5 per MFMA).
  // A 16-pixel group q is KB "regions" of 6*NB MFMAs; region kb multiplies A(q,kb) with all B(q,*) while the
  // VALU (a) splits A of the NEXT region (of group q+1 after the last one) and (b) splits its share of group q+1's
  // B blocks.  Every raw value register is re-loaded right after the split that consumed it, with the data it will
  // deliver one whole group later: G of group q+2, X of group q+1 (block 0: q+2).
  float rx[KB][8], rgv[NB][8];
  struct Lane { int xv, gv; unsigned mask; };
  auto next_lane = [&]() {                       // lane half h takes octet h of the next group
    const Octet o0 = take(cur);
    const Octet o1 = take(cur);
    Lane ln;
    ln.xv = (lh ? o1.xs : o0.xs) + x_lane;
    ln.gv = (lh ? o1.gs : o0.gs) + g_lane;
    ln.mask = lh ? o1.mask : o0.mask;
    return ln;
  };
  auto load_x = [&](const Lane& ln, int kb) {
#pragma unroll
    for (int j = 0; j < 8; ++j)                   // per pixel: may fall outside the tensor -> reads 0
      rx[kb][j] = buf_load1(rs_x, (unsigned)(ln.xv + j * x_pix) + (unsigned)(kb * 128), 0);
  };
  auto load_g = [&](const Lane& ln, int nb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) rgv[nb][j] = buf_load1(rs_g, (unsigned)ln.gv + (unsigned)(nb * 128), (unsigned)(j * g_pix));
  };
  struct Frag { u32x4 h, m, l; };          // (F16: h and l only)
  auto split8 = [&](const float (&v)[8], Frag& f, const float mul) {
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (F16) split2h(v[2 * q] * mul, v[2 * q + 1] * mul, hh[q], ll[q]);
      else split2(v[2 * q], v[2 * q + 1], hh[q], mm[q], ll[q]);
    }
    f.h = u32x4{hh[0], hh[1], hh[2], hh[3]}; f.l = u32x4{ll[0], ll[1], ll[2], ll[3]};
    if constexpr (!F16) f.m = u32x4{mm[0], mm[1], mm[2], mm[3]};
  };
  auto split_x = [&](int kb, unsigned mask, Frag& f) {       // BN+ReLU prologue, column mask, split
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = rx[kb][j];
      if (pro) v = fmaxf(fmaf(v, psc[kb], psh[kb]), 0.f);
      xv[j] = ((mask >> j) & 1u) ? v : 0.f;
    }
    split8(xv, f, (F16 && !pro) ? x_mul : 1.f);
  };

  // B fragments of the next group wait in a wave-private LDS area (48 registers less than a second register set)
  u32x4* sB = reinterpret_cast<u32x4*>(smem) + wave * (NB * 3 * 64) + lane;      // [nb][plane][lane]
  if (n_groups16 > 0) {
    Frag bc[NB], aF[2];
    Lane ln1 = next_lane();                       // group 0
    unsigned mask_q = ln1.mask, mask_q1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load_g(ln1, nb);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) load_x(ln1, kb);
    ln1 = next_lane();                            // group 1
    mask_q1 = ln1.mask;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { split8(rgv[nb], bc[nb], g_mul); load_g(ln1, nb); }
    split_x(0, mask_q, aF[0]);
    load_x(ln1, 0);

    for (int q = 0; q < n_groups16; ++q) {
      const Lane ln2 = next_lane();               // group q+2 (past the end: all-masked dummy octets at offset 0)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const Frag& ac = aF[kb & 1];
        Frag& an = aF[(kb + 1) & 1];
        // VALU side of the region: this region's share of group q+1's B blocks (parked in LDS), next region's A
#pragma unroll
        for (int nb = kb * NB / KB; nb < (kb + 1) * NB / KB; ++nb) {
          Frag t;
          split8(rgv[nb], t, g_mul);
          load_g(ln2, nb);
          sB[(nb * 3 + 0) * 64] = t.h; sB[(nb * 3 + 2) * 64] = t.l;
          if constexpr (!F16) sB[(nb * 3 + 1) * 64] = t.m;
        }
        if (kb + 1 < KB) { split_x(kb + 1, mask_q, an); load_x(ln1, kb + 1); }
        else { split_x(0, mask_q1, an); load_x(ln2, 0); }
        // matrix side
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          f32x16 c = acc[kb][nb];
          if constexpr (F16) {
            if (!x1) {                             // (wave-uniform: MPOSE_CONV_F16X1 keeps the h x h product only)
              c = mfma_f16(ac.l, bc[nb].h, c);
              c = mfma_f16(ac.h, bc[nb].l, c);
            }
            c = mfma_f16(ac.h, bc[nb].h, c);
          } else {
            c = mfma_bf16(as_bf16x8(ac.l), as_bf16x8(bc[nb].h), c);
            c = mfma_bf16(as_bf16x8(ac.h), as_bf16x8(bc[nb].l), c);
            c = mfma_bf16(as_bf16x8(ac.m), as_bf16x8(bc[nb].m), c);
            c = mfma_bf16(as_bf16x8(ac.m), as_bf16x8(bc[nb].h), c);
            c = mfma_bf16(as_bf16x8(ac.h), as_bf16x8(bc[nb].m), c);
            c = mfma_bf16(as_bf16x8(ac.h), as_bf16x8(bc[nb].h), c);
          }
          acc[kb][nb] = c;
          if (kb == KB - 1) {                     // last use in this group: fetch the next group's fragments
            bc[nb].h = sB[(nb * 3 + 0) * 64]; bc[nb].l = sB[(nb * 3 + 2) * 64];
            if constexpr (!F16) bc[nb].m = sB[(nb * 3 + 1) * 64];
          }
        }
        // interleave: the region's VALU instructions (~6 per MFMA with six products, ~9 with three) go into the MFMAs' shadows
#pragma unroll
        for (int i = 0; i < (F16 ? 3 : 6) * NB; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, F16 ? 9 : 6, 0);
          if (F16 || i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);        // keep each region's loads and splits inside the region
      }
      if (KB & 1) aF[0] = aF[1];
      mask_q = mask_q1; mask_q1 = ln2.mask; ln1 = ln2;
    }
  }
  if constexpr (F16) {                            // back to the tensors' own units (exact)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kb][nb][r] = __builtin_ldexpf(acc[kb][nb][r], -(kx + kg));
  }
  __syncthreads();                                // the reduction below reuses the LDS of slower waves' B areas

  // ---- deterministic cross-wave sum through LDS.  An "item" is one float4 per lane = 4 consecutive input
  //      channels of one (kb, nb, register group): every wave parks its items in its own LDS region, then wave w
  //      sums items w, w+4, ... over the regions in the fixed order 0,1,2,3 and writes them to the split-K partial
  //      buffer (layout [split][widx][K/4][Npad][4]).  Tiles of more than 32 items go in two halves (LDS size). ----
  constexpr int ITEMS = KB * NB * 4, HALVES = ITEMS > 32 ? 2 : 1, IPH = ITEMS / HALVES;
  float4* sm4 = reinterpret_cast<float4*>(smem);                     // [4 regions][IPH][64]
  const int k4_total = g.Cin >> 2;
  float* base = dw + ((long)(split * n_widx + tap.widx) * k4_total) * npad * 4;
#pragma unroll
  for (int half = 0; half < HALVES; ++half) {
    if (half) __syncthreads();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int item = (kb * NB + nb) * 4 + rg;
          if (item >= half * IPH && item < (half + 1) * IPH)
            sm4[(wave * IPH + (item - half * IPH)) * 64 + lane] =
                make_float4(acc[kb][nb][4 * rg], acc[kb][nb][4 * rg + 1], acc[kb][nb][4 * rg + 2], acc[kb][nb][4 * rg + 3]);
        }
    __syncthreads();
    for (int il = wave; il < IPH; il += 4) {
      const float4 v0 = sm4[(0 * IPH + il) * 64 + lane], v1 = sm4[(1 * IPH + il) * 64 + lane];
      const float4 v2 = sm4[(2 * IPH + il) * 64 + lane], v3 = sm4[(3 * IPH + il) * 64 + lane];
      const float4 v = make_float4(((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y,
                                   ((v0.z + v1.z) + v2.z) + v3.z, ((v0.w + v1.w) + v2.w) + v3.w);
      const int item = half * IPH + il;
      const int rg = item & 3, blk = item >> 2;
      const int kb = blk / NB, nb = blk - kb * NB;
      const int k4 = (k0 + kb * 32) / 4 + 2 * rg + lh;
      const int n = n0 + nb * 32 + li;
      *reinterpret_cast<float4*>(base + ((long)k4 * npad + n) * 4) = v;
    }
  }
}

template <int KB, int NB, bool F16>
int launch_wgrad_p(const WgradArgs& a, hipStream_t s) {
  constexpr int items = KB * NB * 4;
  constexpr int lds_red = 4 * (items > 32 ? items / 2 : items) * 64 * 16, lds_b = 4 * NB * 3 * 64 * 16;
  constexpr int lds = lds_red > lds_b ? lds_red : lds_b;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_k<KB, NB, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MPOSE_EINVAL;
    attr_set = true;
  }
  launch(conv_wgrad_k<KB, NB, F16>, dim3(dim3(8 * a.chunk)), dim3(256), lds, s, a);
  return launch_status();
}
template <int KB, int NB>
int launch_wgrad(const WgradArgs& a, hipStream_t s) {
  return a.op[0].in_amax != nullptr ? launch_wgrad_p<KB, NB, true>(a, s) : launch_wgrad_p<KB, NB, false>(a, s);
}
template <int KB>
int launch_wgrad_n(const WgradArgs& a, int nb, hipStream_t s) {
  switch (nb) {
    case 4: return launch_wgrad<KB, 4>(a, s);
    case 3: return launch_wgrad<KB, 3>(a, s);
    case 2: return launch_wgrad<KB, 2>(a, s);
    default: return launch_wgrad<KB, 1>(a, s);
  }
}
inline int wgrad_blocks(int c) { return c % 128 == 0 ? 4 : (c % 96 == 0 ? 3 : (c % 64 == 0 ? 2 : 1)); }

// ---------------------------------------------------------------------------------------------
// Weight packing / gradient unpacking (one launch for all convolutions of the model)
// ---------------------------------------------------------------------------------------------
// dst layout (bf16): [T][Kpad/16][plane 3][Npad][half 2][8]  -- one 16-byte MFMA B fragment per (n, half);
// layout 1 (conv_p.hip): [T][Kpad/16][plane 3][half 2][Npad][8] -- 64 consecutive columns of a half = one 1 KiB DMA
// Jobs the tiled packer takes (round 5): the fp16 two-plane layouts of weights whose taps are contiguous and whose k or n index is
// the next-faster one -- every Conv2d / ConvTranspose2d weight, forward and data-gradient view, up to 12 taps.
constexpr int kPackTiledMaxT = 12;
constexpr int kPackU = 9;
__device__ __forceinline__ bool pack_tiled_ok(const mpose_pack_job& j) {
  return (j.layout == 2 || j.layout == 3) && j.st == 1 && j.T >= 1 && j.T <= kPackTiledMaxT && (j.sk == j.T || j.sn == j.T) && (j.Kpad & 15) == 0;
}

// One workgroup per (16 k, 64 n) tile and all taps: the tile is read in SOURCE order (runs of 16 T or 64 T contiguous floats:
// every line is used whole -- the element-wise packer below walks a tap at a time and fetches every line once per tap), turned in
// LDS, and written as 16-byte fragments in the packed order.  Same values, bit for bit, as pack_weights_k.
__global__ __launch_bounds__(256) void pack_weights_tiled_k(const mpose_pack_job* __restrict__ jobs) {
  __shared__ float tile[kPackTiledMaxT * 16 * 65];          // [t][k_lo][n] (+1: the turn's reads spread over the banks)
  const mpose_pack_job j = jobs[blockIdx.y];
  if (!pack_tiled_ok(j)) return;
  const int T = j.T, k16n = j.Kpad / 16, nblk = (j.Npad + 63) / 64;
  const float w_mul = j.amax != nullptr ? pow2f(f16_scale_exp(*j.amax)) : 1.f;
  const long plane = (long)j.Npad * 16;
  const bool k_fast = j.sk == j.T;              // (k, t) contiguous for a fixed n; otherwise (n, t) contiguous for a fixed k
  for (int tl = blockIdx.x; tl < k16n * nblk; tl += gridDim.x) {
    const int k16 = tl / nblk, n0 = (tl - k16 * nblk) * 64, k0 = k16 * 16;
    const int run = (k_fast ? 16 : 64) * T, n_run = k_fast ? 64 : 16;
    __syncthreads();                            // (the previous tile's readers are done)
    // 16 threads walk a run (64 bytes per row and instruction), 16 runs per pass; kPackU loads of a thread in flight at a time
    const float inv_t = 1.0f / (float)T;
    const int lane16 = threadIdx.x & 15, row16 = threadIdx.x >> 4;
    for (int o = row16; o < n_run; o += 16) {
      for (int r0 = lane16; r0 < run; r0 += 16 * kPackU) {
        float v[kPackU];
        int rr[kPackU];
#pragma unroll
        for (int u = 0; u < kPackU; ++u) {
          const int r = r0 + 16 * u;
          rr[u] = r;
          const int inner = (int)(((float)r + 0.5f) * inv_t), tt = r - inner * T;     // (exact: r < 64 * 12)
          const int nn = k_fast ? o : inner, kl = k_fast ? inner : o;
          v[u] = 0.f;
          if (r < run && n0 + nn < j.N && k0 + kl < j.K) v[u] = j.src[(long)(n0 + nn) * j.sn + (long)(k0 + kl) * j.sk + tt];
        }
#pragma unroll
        for (int u = 0; u < kPackU; ++u) {
          const int r = rr[u];
          if (r < run) {
            const int inner = (int)(((float)r + 0.5f) * inv_t), tt = r - inner * T;
            const int nn = k_fast ? o : inner, kl = k_fast ? inner : o;
            tile[(tt * 16 + kl) * 65 + nn] = v[u];
          }
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * 128; i += 256) {          // (t, half, n): one 16-byte fragment per plane
      const int t = i >> 7, half = (i >> 6) & 1, nn = i & 63;
      if (n0 + nn >= j.Npad) continue;
      f16x8 h, l;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float vs = tile[(t * 16 + half * 8 + q) * 65 + nn] * w_mul;
        const _Float16 hh = (_Float16)vs;
        h[q] = hh;
        l[q] = (_Float16)(vs - (float)hh);
      }
      _Float16* d = reinterpret_cast<_Float16*>(j.dst) + ((long)(t * k16n + k16) * 2) * plane +
                    (j.layout == 3 ? ((long)half * j.Npad + n0 + nn) * 8 : (long)(n0 + nn) * 16 + half * 8);
      *reinterpret_cast<f16x8*>(d) = h;
      *reinterpret_cast<f16x8*>(d + plane) = l;
    }
  }
}

__global__ __launch_bounds__(256) void pack_weights_k(const mpose_pack_job* __restrict__ jobs, int skip_tiled) {
  const mpose_pack_job j = jobs[blockIdx.y];
  if (skip_tiled && pack_tiled_ok(j)) return;      // (pack_weights_tiled_k's)
  const long total = (long)j.T * j.Kpad * j.Npad;
  __bf16* dst = reinterpret_cast<__bf16*>(j.dst);
  const long plane = (long)j.Npad * 16;
  const float w_mul = ((j.layout == 2 || j.layout == 3) && j.amax != nullptr) ? pow2f(f16_scale_exp(*j.amax)) : 1.f;
  // (32-bit index arithmetic: a job is at most taps x Kpad x Npad < 2^31 elements (max_elems_per_job is an int), and the three
  //  64-bit divisions per element this loop used to do were most of its instructions)
  const unsigned k16n = (unsigned)j.Kpad / 16u, npad = (unsigned)j.Npad;
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < (unsigned)total; e += gridDim.x * 256u) {
    const int k_lo = (int)(e & 15u);             // half * 8 + j
    unsigned r = e >> 4;
    const unsigned r1_ = r / npad;
    const int n = (int)(r - r1_ * npad);
    const int t = (int)(r1_ / k16n);
    const int k16 = (int)(r1_ - (unsigned)t * k16n);
    const int k = k16 * 16 + k_lo;
    float v = 0.f;
    if (n < j.N && k < j.K) v = j.src[n * j.sn + k * j.sk + t * j.st];
    if (j.layout == 2 || j.layout == 3) {           // two fp16 planes of w * 2^k (MPOSE_CONV_F16X3; layout 3: conv_h.hip's B tiles)
      const float vs = v * w_mul;
      const _Float16 h = (_Float16)vs;
      const _Float16 l = (_Float16)(vs - (float)h);
      _Float16* d = reinterpret_cast<_Float16*>(j.dst) + ((long)(t * (j.Kpad / 16) + k16) * 2) * plane +
                    (j.layout == 3 ? ((long)(k_lo >> 3) * j.Npad + n) * 8 + (k_lo & 7) : (long)n * 16 + k_lo);
      d[0] = h; d[plane] = l;
      continue;
    }
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    __bf16* d = dst + ((long)(t * (j.Kpad / 16) + k16) * 3) * plane +
                (j.layout == 1 ? ((long)(k_lo >> 3) * j.Npad + n) * 8 + (k_lo & 7) : (long)n * 16 + k_lo);
    d[0] = h; d[plane] = m; d[2 * plane] = l;
  }
}

__global__ __launch_bounds__(256) void unpack_wgrads_k(const mpose_unpack_job* __restrict__ jobs) {
  const mpose_unpack_job j = jobs[blockIdx.y];
  const long split_stride4 = (long)j.T * j.Kpad * j.Npad / 4;       // float4 items per split
  const int k4n = (j.K + 3) >> 2;
  const long total = (long)j.T * k4n * j.N;
  const float4* __restrict__ src4 = reinterpret_cast<const float4*>(j.src);
  // One thread per (tap, 4 input channels, output channel): the packed float4 of a (k4, n) pair, consecutive n on consecutive
  // lanes -- a wave reads 1 KiB runs of every split, eight splits in flight per thread (the pass is a 2 GB read per step: with
  // one dword per lane and four in flight it ran at 2.6 TB/s).  Each element is still summed in split order: same bits as before.
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e % j.N);
    const long r = e / j.N;
    const int k4 = (int)(r % k4n), t = (int)(r / k4n);
    const float4* p = src4 + ((long)t * (j.Kpad / 4) + k4) * j.Npad + n;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int sp = 0;
    for (; sp + 8 <= j.n_split; sp += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(sp + u) * split_stride4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; sp < j.n_split; ++sp) {
      const float4 v = p[sp * split_stride4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* d = j.dst + (long)n * j.sn + (long)(k4 * 4) * j.sk + (long)t * j.st;
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (k4 * 4 + i < j.K) {
        float* dt = d + (long)i * j.sk;
        *dt = j.accumulate ? (*dt + sv[i]) : sv[i];
      }
    }
  }
}

#endif   // CV_PART == 0
}  // namespace
}  // namespace mpose

// conv_igemm_k's 96- and 128-channel tiles -- two thirds of its instantiations by compile time -- are built in translation units of
// their own: conv_rn3.hip / conv_rn4.hip compile THIS file with CV_PART = 3 / 4 (the kernel, its launchers and one entry point each;
// ConvArgs is the same struct in every unit, passed as an untyped pointer because its type lives in an anonymous namespace).
namespace mpose {
int conv_igemm_rn3(const void* args, int mode, int cmax, int n_groups, hipStream_t s);
int conv_igemm_rn4(const void* args, int mode, int cmax, int n_groups, hipStream_t s);
#if CV_PART == 3
int conv_igemm_rn3(const void* args, int mode, int cmax, int n_groups, hipStream_t s) {
  return launch_conv_ks<3>(*static_cast<const ConvArgs*>(args), mode, cmax, n_groups, s);
}
#elif CV_PART == 4
int conv_igemm_rn4(const void* args, int mode, int cmax, int n_groups, hipStream_t s) {
  return launch_conv_ks<4>(*static_cast<const ConvArgs*>(args), mode, cmax, n_groups, s);
}
#endif
}  // namespace mpose

#if CV_PART == 0

using namespace mpose;

thread_local int* mpose::mpose_dry_rows = nullptr;
thread_local mpose::PartPhase mpose::mpose_part_phase = {0, 0};

int mpose_conv_h2_launch(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups, int flags, int mode,
                         int cmax, void* stream);          // conv_h.hip
int mpose_wgrad_rows_units(const mpose_conv_geom* geom);                                                                       // wgrad.hip
int mpose_wgrad_rows_launch(const mpose_conv_geom* geom, const mpose_wgrad_operands* ops, int n_groups, int n_split, void* stream);
int mpose_wgrad_rows_waves(const mpose_conv_geom* geom);
int mpose_wgrad_rows_occupancy(const mpose_conv_geom* geom);

// in_mul_x / out_mul_x = 0 ("as along y") filled in: what the kernels and the checks below read
static mpose_conv_geom normalised(const mpose_conv_geom* g) {
  mpose_conv_geom n = *g;
  if (!n.in_mul_x) n.in_mul_x = n.in_mul;
  if (!n.out_mul_x) n.out_mul_x = n.out_mul;
  return n;
}

static int check_geom(const mpose_conv_geom* g) {
  if (!g || g->Cin <= 0 || (g->Cin % KC) || g->n_classes < 1 || g->n_classes > MPOSE_MAX_CLASSES) return MPOSE_EINVAL;
  if (g->Npad0 <= 0 || (g->Npad0 % 32)) return MPOSE_EINVAL;
  if (g->in_mul < 1 || g->in_mul > 8 || g->out_mul < 1 || g->out_mul > 8) return MPOSE_EINVAL;
  if (g->in_mul_x < 0 || g->in_mul_x > 8 || g->out_mul_x < 0 || g->out_mul_x > 8) return MPOSE_EINVAL;
  for (int c = 0; c < g->n_classes; ++c)
    if (g->cls[c].n_taps < 0 || g->cls[c].n_taps > MPOSE_MAX_TAPS) return MPOSE_EINVAL;
  if ((long)g->B * g->GH * g->GW >= (1l << 26)) return MPOSE_EINVAL;
  return 0;
}

// A stride-1 convolution whose taps are all a multiple of d pixels apart along x (dilation d) is d independent undilated
// convolutions, one per residue of x mod d: pixel x = d*q + r of an NHWC row IS pixel q of a row whose pixels are d*ld floats
// apart, starting r*ld floats in.  The kernels that want |dx| <= 1 -- the row-of-taps weight gradient (wgrad.hip), row-group
// staging in conv_igemm_k -- then take it one residue per launch.  Returns d and fills the per-residue geometry, or 1.
static int x_phase_geom(const mpose_conv_geom& g, mpose_conv_geom* phase) {
  if (g.n_classes != 1 || g.in_mul != 1 || g.out_mul != 1 || g.in_mul_x != 1 || g.out_mul_x != 1) return 1;
  if (g.IH != g.GH || g.IW != g.GW || g.OH != g.GH || g.OW != g.GW || g.cls[0].oy || g.cls[0].ox) return 1;
  int d = 0;
  for (int t = 0; t < g.cls[0].n_taps; ++t) {
    if (g.cls[0].taps[t].acc) return 1;
    int v = g.cls[0].taps[t].dx < 0 ? -g.cls[0].taps[t].dx : g.cls[0].taps[t].dx;
    while (v) { const int m = d % v; d = v; v = m; }        // gcd
  }
  if (d <= 1 || (g.IW % d) || ((g.IW / d) & 7)) return 1;
  mpose_conv_geom p = g;
  p.IW = p.OW = p.GW = g.IW / d;
  p.in_ld = d * (g.in_ld > 0 ? g.in_ld : g.Cin);
  p.out_ld0 = d * (g.out_ld0 > 0 ? g.out_ld0 : g.Cout0);
  for (int t = 0; t < p.cls[0].n_taps; ++t) p.cls[0].taps[t].dx = (int8_t)(p.cls[0].taps[t].dx / d);
  *phase = p;
  return d;
}

extern "C" int mpose_conv_fwd(const mpose_conv_geom* geom_, const mpose_conv_operands* ops, int n_groups, int flags,
                              void* stream) {
  int rc = check_geom(geom_);
  if (rc) return rc;
  const mpose_conv_geom gn = normalised(geom_);
  const mpose_conv_geom* geom = &gn;
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP) return MPOSE_EINVAL;
  ConvArgs a{};
  a.g = *geom;
  bool acc1 = false;
  for (int c = 0; c < geom->n_classes; ++c)
    for (int t = 0; t < geom->cls[c].n_taps; ++t) acc1 |= geom->cls[c].taps[t].acc != 0;
  const bool sum_inputs = (flags & MPOSE_CONV_SUM_INPUTS) != 0;
  if (sum_inputs) {                        // acc taps feed the SAME output from a second input: a single-pass launch
    if (!acc1) return MPOSE_EINVAL;
    for (int i = 0; i < n_groups; ++i)
      if (!ops[i].in1 || !ops[i].w1 || ops[i].in_scale) return MPOSE_EINVAL;
    acc1 = false;
  }
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    if (!ops[i].in || !ops[i].w0) return MPOSE_EINVAL;
    if (!ops[i].out0 && !((flags & MPOSE_CONV_PLANES_IN) && ops[i].out0_planes)) return MPOSE_EINVAL;
    if (!(flags & MPOSE_CONV_PLANES_IN) && ops[i].out0_planes) return MPOSE_EINVAL;
    if ((flags & MPOSE_CONV_PLANES_IN) && ops[i].out0_amax) return MPOSE_EINVAL;
    if (ops[i].epi_scale0 && (!ops[i].epi_shift0 || ops[i].stats0 || ops[i].mask_src || (flags & MPOSE_CONV_ACCUMULATE))) return MPOSE_EINVAL;
    if (ops[i].add_src && (!ops[i].epi_scale0 || !ops[i].add_scale || !ops[i].add_shift)) return MPOSE_EINVAL;
    if ((ops[i].epi_scale0 != nullptr) != (ops[0].epi_scale0 != nullptr) || (ops[i].add_src != nullptr) != (ops[0].add_src != nullptr)) return MPOSE_EINVAL;
    if (ops[i].out0_amax && !ops[i].epi_scale0 && ((flags & MPOSE_CONV_PLANES_IN) || !(flags & MPOSE_CONV_F16X3) || (flags & MPOSE_CONV_ACCUMULATE))) return MPOSE_EINVAL;
    if (ops[i].red_sums && ((flags & MPOSE_CONV_PLANES_IN) || !ops[i].red_a || !ops[i].red_b || !ops[i].red_scale || !ops[i].red_shift ||
                            ops[i].stats0 || ops[i].stats1 || ops[i].epi_scale0 || (acc1 && !sum_inputs)))
      return MPOSE_EINVAL;
    if ((ops[i].red_sums != nullptr) != (ops[0].red_sums != nullptr)) return MPOSE_EINVAL;
    if (ops[i].mm0 && ((flags & MPOSE_CONV_PLANES_IN) || !ops[i].stats0)) return MPOSE_EINVAL;
    if ((flags & MPOSE_CONV_H2_IN) && (ops[i].epi_scale0 || ops[i].add_src || ops[i].out0_planes || ops[i].fin_count || ops[i].in_scale ||
                                       (flags & (MPOSE_CONV_ACCUMULATE | MPOSE_CONV_PLANES_IN)) || !(flags & MPOSE_CONV_F16X3)))
      return MPOSE_EINVAL;
    if (ops[i].fin_count && ((flags & MPOSE_CONV_PLANES_IN) || !ops[i].fin0 || !ops[i].stats0 || (ops[i].fin1 && (!acc1 || !ops[i].stats1)) ||
                             sum_inputs)) return MPOSE_EINVAL;
    if (!ops[i].fin_count && (ops[i].fin0 || ops[i].fin1)) return MPOSE_EINVAL;
    if ((flags & MPOSE_CONV_STATS_PART) && (ops[i].fin_count || (flags & MPOSE_CONV_PLANES_IN))) return MPOSE_EINVAL;
    if (acc1 && (!ops[i].w1 || !ops[i].out1)) return MPOSE_EINVAL;
    if ((ops[i].in_scale != nullptr) != (ops[0].in_scale != nullptr)) return MPOSE_EINVAL;
    if (ops[i].in_scale && (acc1 || !ops[i].in_shift)) return MPOSE_EINVAL;
  }
  if (acc1 && geom->Npad1 != geom->Npad0) return MPOSE_EINVAL;
  // x-dilated 3x3 in the three-product form: one launch per residue of x, each eligible for row-group staging
  if ((flags & MPOSE_CONV_F16X3) && !(flags & MPOSE_CONV_PLANES_IN) && !acc1 && !sum_inputs && rowg_env()) {
    mpose_conv_geom pg;
    const int d = x_phase_geom(*geom, &pg);
    bool simple = d > 1 && rowg_eligible(pg);
    for (int i = 0; simple && i < n_groups; ++i)
      simple = !ops[i].mask_src && !ops[i].epi_scale0 && !ops[i].add_src && !ops[i].out0_planes && !ops[i].out0_amax &&
               !ops[i].red_sums && !ops[i].fin_count;
    if (simple) {
      const int in_ld = geom->in_ld > 0 ? geom->in_ld : geom->Cin, o_ld = geom->out_ld0 > 0 ? geom->out_ld0 : geom->Cout0;
      int rows_phase = 0;
      if ((flags & MPOSE_CONV_STATS_PART) && !mpose_dry_rows) {     // each residue's launch writes its own block of partial rows
        rows_phase = mpose_conv_stat_rows(&pg, ops, n_groups, flags);
        if (rows_phase < 0) return rows_phase;
      }
      for (int r = 0; r < d; ++r) {
        mpose_conv_operands po[MPOSE_MAX_GROUP];
        for (int i = 0; i < n_groups; ++i) {
          po[i] = ops[i];
          po[i].in = ops[i].in + (long)r * in_ld;
          po[i].out0 = ops[i].out0 + (long)r * o_ld;
        }
        if (rows_phase) mpose_part_phase = PartPhase{r * rows_phase, d * rows_phase};
        rc = mpose_conv_fwd(&pg, po, n_groups, flags, stream);
        mpose_part_phase = PartPhase{0, 0};
        if (rc) return rc;
      }
      return 0;
    }
  }
  a.M = geom->B * geom->GH * geom->GW;
  if (a.M == 0) return 0;
  if (flags & MPOSE_CONV_PLANES_IN) return MPOSE_EINVAL;      // (round 2's plane engine left the library in round 6)
  if (flags & MPOSE_CONV_BF16) return MPOSE_EINVAL;
  if ((flags & MPOSE_CONV_F16X1) && !(flags & MPOSE_CONV_F16X3)) return MPOSE_EINVAL;
  if (flags & MPOSE_CONV_F16X3) {
    for (int i = 0; i < n_groups; ++i) {
      if (!ops[i].in_amax || !ops[i].w0_amax) return MPOSE_EINVAL;
      if (sum_inputs && (!ops[i].in1_amax || !ops[i].w1_amax)) return MPOSE_EINVAL;
      if (acc1 && !ops[i].w1_amax) return MPOSE_EINVAL;
    }
  }
  if (flags & MPOSE_CONV_H2_IN) {          // producer-split fp16 planes: conv_h.hip
    if ((geom->Cout0 % 32) || (acc1 && (geom->Cout1 % 32)) || (geom->Npad0 % 64) || (geom->Cin % 16)) return MPOSE_EINVAL;
    const int cm = (acc1 && geom->Cout1 > geom->Cout0) ? geom->Cout1 : geom->Cout0;
    if (cm > geom->Npad0) return MPOSE_EINVAL;
    const int ldm = geom->out_ld0 > geom->out_ld1 ? geom->out_ld0 : geom->out_ld1;
    if ((long)geom->B * geom->OH * geom->OW * (ldm > cm ? ldm : cm) * 4 >= 0xFFFFF000l) return MPOSE_EINVAL;
    rc = mpose_conv_h2_launch(geom, ops, n_groups, flags, sum_inputs ? 2 : (acc1 ? 1 : 0), cm, stream);
    return rc == MPOSE_ENOSYS ? MPOSE_EINVAL : rc;
  }
  a.div_gw = make_fastdiv((unsigned)geom->GW);
  a.div_ghw = make_fastdiv((unsigned)(geom->GH * geom->GW));
  a.flags = flags;
  {
    long min_shift = 0;
    for (int c = 0; c < geom->n_classes; ++c)
      for (int t = 0; t < geom->cls[c].n_taps; ++t) {
        const long sft = ((long)geom->cls[c].taps[t].dy * geom->IW + geom->cls[c].taps[t].dx) * (geom->in_ld > 0 ? geom->in_ld : geom->Cin) * 4;
        if (sft < min_shift) min_shift = sft;
      }
    a.in_bias = (int)(-min_shift);
    const long in_bytes = (long)geom->B * geom->IH * geom->IW * (geom->in_ld > 0 ? geom->in_ld : geom->Cin) * 4;
    if (in_bytes + a.in_bias >= 0xFFFFFF00l - (1l << 20)) return MPOSE_EINVAL;       // 32-bit buffer offsets
  }
  hipStream_t s = (hipStream_t)stream;
  const int npad = geom->Npad0;
  const int cmax = (acc1 && geom->Cout1 > geom->Cout0) ? geom->Cout1 : geom->Cout0;
  if (cmax > npad) return MPOSE_EINVAL;
  if ((geom->Cout0 % 32) || (acc1 && (geom->Cout1 % 32))) return MPOSE_EINVAL;
  {
    const int ldm = geom->out_ld0 > geom->out_ld1 ? geom->out_ld0 : geom->out_ld1;
    if ((long)geom->B * geom->OH * geom->OW * (ldm > cmax ? ldm : cmax) * 4 >= 0xFFFFF000l) return MPOSE_EINVAL;
  }
  if (geom->Cin % 16) return MPOSE_EINVAL;
  const int mode = sum_inputs ? 2 : (acc1 ? 1 : 0);
  if (cmax <= 32) return launch_conv_ks<1>(a, mode, cmax, n_groups, s);
  if (npad % 64) return MPOSE_EINVAL;
  // widest wave tile that divides the padded N: 128 channels (RN=4), 96 (RN=3) or 64 (RN=2)
  // Single-pass row-group launches in the three-product form with the FUSED OUTPUT STAGE (inference: the second 3x3 of a block)
  // run with 64-channel wave tiles, two workgroups per CU (slim_tile() above), where the widest tile would be 128 channels.
  // Measured on the 128 -> 128 launches at 32 x 32, B = 32: unsplit (one chain of 216 accumulations per output) 112 -> 97 us
  // forward, 114 -> 91 us data-gradient; with the two-way K split that training keeps for its accumulation chains (pick_ks) the
  // gain is gone (114-120 -> 122, 113 -> 111 us), and at 192 channels the 96-channel tiles are faster (58 vs 65 us).  So:
  // inference launches only -- until round 4, which measured the training step with them: MPOSE_SLIM=2 (every eligible launch,
  // without the rule that keeps training's K split) is now the default; 1: inference launches only (round 3's plan); 0: never.
  constexpr int slim = 2;
  // (the single-product mode MPOSE_CONV_F16X1 accumulates a third as often: unsplit everywhere, training included)
  if (slim && (slim == 2 || a.op[0].epi_scale0 != nullptr || (flags & MPOSE_CONV_F16X1)) && mode == 0 && (flags & MPOSE_CONV_F16X3) &&
      (cmax % 128) == 0 && rowg_env() && rowg_eligible(a.g))
    return launch_conv_ks<2>(a, mode, cmax, n_groups, s);
  if (cmax % 128 == 0) return conv_igemm_rn4(&a, mode, cmax, n_groups, s);
  if (cmax % 96 == 0 && npad % 96 == 0) return conv_igemm_rn3(&a, mode, cmax, n_groups, s);
  return launch_conv_ks<2>(a, mode, cmax, n_groups, s);
}

extern "C" int mpose_conv_stat_rows(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups, int flags) {
  int rows = 0;
  int* const outer = mpose_dry_rows;
  mpose_dry_rows = &rows;
  const int rc = mpose_conv_fwd(geom, ops, n_groups, flags, nullptr);
  mpose_dry_rows = outer;
  return rc ? (rc < 0 ? rc : -rc) : rows;
}

static int x_phases(const mpose_conv_geom& g, mpose_conv_geom* phase) {       // ... whose weight gradient the row form takes
  mpose_conv_geom p;
  const int d = x_phase_geom(g, &p);
  if (d <= 1 || mpose_wgrad_rows_units(&p) <= 0) return 1;
  if (phase) *phase = p;
  return d;
}

extern "C" int mpose_conv_wgrad_phases(const mpose_conv_geom* geom_) {
  if (check_geom(geom_)) return -1;
  const mpose_conv_geom gn = normalised(geom_);
  return x_phases(gn, nullptr);
}

extern "C" int mpose_conv_wgrad_tiles(const mpose_conv_geom* geom_) {
  if (check_geom(geom_)) return -1;
  const mpose_conv_geom gn = normalised(geom_);
  const mpose_conv_geom* geom = &gn;
  if (const int rows = mpose_wgrad_rows_units(geom)) return rows;      // the row-of-taps kernel (wgrad.hip) takes this geometry
  {
    mpose_conv_geom pg;
    if (x_phases(gn, &pg) > 1) return mpose_wgrad_rows_units(&pg);     // ... one residue of x at a time (mpose_conv_wgrad_phases)
  }
  int entries = 0;
  for (int c = 0; c < geom->n_classes; ++c) entries += geom->cls[c].n_taps;
  return entries * (geom->Cin / (32 * wgrad_blocks(geom->Cin))) * (geom->Cout0 / (32 * wgrad_blocks(geom->Cout0)));
}

extern "C" int mpose_conv_wgrad_occupancy(const mpose_conv_geom* geom_) {
  if (check_geom(geom_)) return -1;
  const mpose_conv_geom gn = normalised(geom_);
  if (mpose_wgrad_rows_units(&gn)) return mpose_wgrad_rows_occupancy(&gn);
  {
    mpose_conv_geom pg;
    if (x_phases(gn, &pg) > 1 && mpose_wgrad_rows_units(&pg)) return mpose_wgrad_rows_occupancy(&pg);
  }
  return 1;
}

extern "C" int mpose_conv_wgrad_waves(const mpose_conv_geom* geom_) {
  if (check_geom(geom_)) return -1;
  const mpose_conv_geom gn = normalised(geom_);
  if (mpose_wgrad_rows_units(&gn)) return mpose_wgrad_rows_waves(&gn);
  {
    mpose_conv_geom pg;
    if (x_phases(gn, &pg) > 1 && mpose_wgrad_rows_units(&pg)) return mpose_wgrad_rows_waves(&pg);
  }
  return 4;
}

extern "C" int mpose_conv_wgrad(const mpose_conv_geom* geom_, const mpose_wgrad_operands* ops, int n_groups, int n_split,
                                void* stream) {
  int rc = check_geom(geom_);
  if (rc) return rc;
  const mpose_conv_geom gn = normalised(geom_);
  const mpose_conv_geom* geom = &gn;
  if (n_groups < 1 || n_groups > MPOSE_MAX_GROUP || n_split < 1) return MPOSE_EINVAL;
  if ((geom->Npad0 % 64) || (geom->Cout0 % 32) || (geom->GW & 7)) return MPOSE_EINVAL;
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
  if (acc1 && ((geom->Npad1 % 64) || (geom->Cout1 % 32) || geom->Npad1 != geom->Npad0 || geom->Cout1 != geom->Cout0)) return MPOSE_EINVAL;
  for (int i = 0; i < n_groups; ++i) {
    a.op[i] = ops[i];
    if (!ops[i].in || !ops[i].gout0 || !ops[i].dw0) return MPOSE_EINVAL;
    if (acc1 && (!ops[i].gout1 || !ops[i].dw1)) return MPOSE_EINVAL;
    if (ops[i].in_scale && !ops[i].in_shift) return MPOSE_EINVAL;
    if ((ops[i].in_amax != nullptr) != (ops[0].in_amax != nullptr)) return MPOSE_EINVAL;
    if (ops[i].in_amax && (!ops[i].gout0_amax || (acc1 && !ops[i].gout1_amax))) return MPOSE_EINVAL;
    if ((ops[i].single_product != 0) != (ops[0].single_product != 0) || (ops[i].single_product && !ops[i].in_amax)) return MPOSE_EINVAL;
    if ((ops[i].planes_in != 0) != (ops[0].planes_in != 0) || (ops[i].planes_in && !ops[i].in_amax)) return MPOSE_EINVAL;
  }
  const int n_rows = geom->B * geom->GH;
  if (n_rows == 0 || a.n_entries == 0) return 0;
  const long in_bytes = (long)geom->B * geom->IH * geom->IW * (geom->in_ld > 0 ? geom->in_ld : geom->Cin) * 4;
  const long g_bytes = (long)geom->B * geom->OH * geom->OW * (geom->out_ld0 > geom->Cout0 ? geom->out_ld0 : geom->Cout0) * 4;
  if (in_bytes >= 0x7FFFFF00l || g_bytes >= 0x7FFFFF00l) return MPOSE_EINVAL;        // signed 32-bit byte offsets
  // stride-1 geometries in the three-product form: one staged operand pair per kernel ROW of taps (wgrad.hip)
  rc = mpose_wgrad_rows_launch(geom, ops, n_groups, n_split, stream);
  if (rc != MPOSE_ENOSYS) return rc;
  if (ops[0].planes_in) return MPOSE_EINVAL;       // (plane operands: the row form or nothing)
  {   // x-dilated kernels: one launch of the row form per residue of x, n_split / d partials each (x_phases above)
    mpose_conv_geom pg;
    const int d = x_phases(*geom, &pg);
    if (d > 1 && n_split % d == 0 && ops[0].in_amax) {
      const long split_stride = (long)a.n_widx0 * geom->Cin * geom->Npad0;       // floats per partial
      const int in_ld = geom->in_ld > 0 ? geom->in_ld : geom->Cin;
      const int g_ld = geom->out_ld0 > 0 ? geom->out_ld0 : geom->Cout0;
      for (int r = 0; r < d; ++r) {
        mpose_wgrad_operands po[MPOSE_MAX_GROUP];
        for (int i = 0; i < n_groups; ++i) {
          po[i] = ops[i];
          po[i].in = ops[i].in + (long)r * in_ld;
          po[i].gout0 = ops[i].gout0 + (long)r * g_ld;
          po[i].dw0 = ops[i].dw0 + (long)r * (n_split / d) * split_stride;
        }
        rc = mpose_wgrad_rows_launch(&pg, po, n_groups, n_split / d, stream);
        if (rc) return rc == MPOSE_ENOSYS ? MPOSE_EINVAL : rc;
      }
      return 0;
    }
  }
  a.div_gh = make_fastdiv((unsigned)geom->GH);
  a.n_split = n_split;
  a.rows_per_split = (n_rows + n_split - 1) / n_split;
  hipStream_t s = (hipStream_t)stream;
  const int kb = wgrad_blocks(geom->Cin), nb = wgrad_blocks(geom->Cout0);
  a.n_ytiles = (geom->Cin / (32 * kb)) * (geom->Cout0 / (32 * nb));
  a.n_groups = n_groups;
  a.total = a.n_entries * a.n_ytiles * n_split * n_groups;
  a.chunk = (a.total + 7) / 8;
  switch (kb) {
    case 4: return launch_wgrad_n<4>(a, nb, s);
    case 3: return launch_wgrad_n<3>(a, nb, s);
    case 2: return launch_wgrad_n<2>(a, nb, s);
    default: return launch_wgrad_n<1>(a, nb, s);
  }
}

extern "C" int mpose_pack_weights(const mpose_pack_job* jobs_dev, int n_jobs, int max_elems_per_job, void* stream) {
  if (n_jobs <= 0) return 0;
  int bx = (max_elems_per_job + 256 * 8 - 1) / (256 * 8);
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  constexpr int tiled = 1;      // (the tiled packer for the large weights, the element-wise one for the rest)
  if (tiled) launch(pack_weights_tiled_k, dim3(dim3(bx > 48 ? 48 : bx, n_jobs)), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  launch(pack_weights_k, dim3(dim3(tiled && bx > 32 ? 32 : bx, n_jobs)), dim3(256), 0, (hipStream_t)stream, jobs_dev, tiled);
  return launch_status();
}

extern "C" int mpose_unpack_wgrads(const mpose_unpack_job* jobs_dev, int n_jobs, int max_elems_per_job, void* stream) {
  if (n_jobs <= 0) return 0;
  int bx = (max_elems_per_job / 4 + 255) / 256;      // float4 items of the largest job / 256 (smaller jobs: the surplus workgroups leave at once)
  if (bx < 1) bx = 1;
  if (bx > 384) bx = 384;
  launch(unpack_wgrads_k, dim3(dim3(bx, n_jobs)), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  return launch_status();
}
#endif   // CV_PART == 0
