// Weight gradient of the stride-1 convolutions on gfx950, "row of taps" form (round 3).
//
//     dW[tap (dy,dx)][k][n] = sum over pixels (b,y,x) of  X[b, y+dy, x+dx, k] * G[b, y, x, n]
//
// Same arithmetic as conv.hip's conv_wgrad_k<.., F16 = true> (MPOSE_CONV_F16X3: both operands scaled by their tensor's power of
// two, split into two fp16 values, three exact products per multiply-add accumulated in fp32 by v_mfma_f32_32x32x16_f16), a
// different division of labour.  conv_wgrad_k gave every WAVE a 128 x 128 tile of one tap and let it gather and split its own
// operands in registers: 64 pixel-strided dword gathers and ~300 VALU instructions per 48 MFMAs, all issued by the one wave per
// SIMD that also issues the MFMAs (MFMA-busy 0.17-0.27).  Here
//   * a WORKGROUP owns a (32*KB*WK input channels) x (32*NB*WN output channels) tile for one kernel ROW: the three taps dx = -1, 0,
//     +1 of a row multiply the same gradient pixels with the same input pixels shifted by one, so one staged input octet (+ one
//     halo pixel on each side) and one staged gradient octet feed three accumulator sets;
//   * the operands are fetched ONCE per workgroup with coalesced 16-byte loads (NHWC fp32, as every producer writes them),
//     scaled, split and written to LDS as two fp16 planes [pixel][channel] -- 3-4 VALU per element, shared by the four waves:
//     ~60 VALU per 36 MFMAs and wave;
//   * the MFMA wants 8 consecutive PIXELS of one channel per lane -- the transpose of what NHWC offers.  ds_read_b64_tr_b16 does
//     it in the LDS crossbar: within a 16-lane group lane L supplies the address of (pixel L/4, channels 4(L%4)..+3) and receives
//     (pixels 0..3, channel L)  (tools/probe/tr_probe.hip).  A tap shift is an address offset of that read.
//   * pixels are walked in OCTETS (8 consecutive pixels of one image row, GW % 8 == 0), two per MFMA K step (lane half h takes
//     octet h); rows whose tap-shifted input row lies outside the image are skipped, the halo pixel of an octet at the image
//     border is read as zero (buffer range check).
// Units of work: (kernel row | single tap) x (k tile, n tile) x pixel split x column group; a workgroup's four waves own disjoint
// sub-tiles, so there is no cross-wave sum; the split-K partials go to the same [split][widx][K/4][Npad][4] buffers that
// mpose_unpack_wgrads reduces.
//
// Replaces the weight gradients of Conv2d(k = 3 | 1, stride 1) of reference src/margipose/models/margipose_model.py:33,36 (and
// the fused 1x1 shortcut :37) inside autograd's backward.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.h"

#ifndef WG_TRAFFIC_EXP
#define WG_TRAFFIC_EXP 0       // WgRowsArgs::exp_flags of every launch (memory-side timing experiments: wrong results, debug builds only)
#endif

#ifndef WG_EXP
#define WG_EXP 0      // timing experiments (tools/wgrad_exp.sh; wrong results): 1 no fragment reads, 2 no staging, 4 no barrier, 8 no MFMA
#endif

namespace mpose {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

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

constexpr unsigned kBig = 0x40000000u;       // added to a voffset (up to three times): beyond every tensor (< 1 GiB), the buffer unit returns 0
constexpr int MAX_UNITS = 8;

struct RowUnit {
  int8_t dy, ntap, acc, pad;       // ntap == 3: taps dx = -1, 0, +1 (widx < 0: not a tap of the kernel, computed and dropped); ntap == 1: dx = 0
  int8_t widx[4];
  int x_off, g_off;                // byte offsets of the unit's VIEW of the input / the gradient (strided geometries, see build_units)
};

struct WgRowsArgs {
  mpose_wgrad_operands op[MPOSE_MAX_GROUP];
  int H, W, n_rows;                // slot grid == input == output spatial size; n_rows = B * H
  int Cin, Cout;                   // storage channel counts
  int x_pix, g_pix0, g_pix1;       // pixel strides in BYTES (of the views the slot grid walks: 2 pixels of the tensor for a stride-2 side)
  int x_row, g_row0, g_row1;       // row strides in BYTES of the same views
  unsigned x_bytes, g_bytes0, g_bytes1;
  int npad;
  int n_widx0, n_widx1;
  int n_units;
  RowUnit units[MAX_UNITS];
  int n_ktiles, n_ntiles;
  int n_split, rows_per_split;
  int n_groups, chunk, total;
  unsigned x_slab, g_slab;         // PL (operands as H8 planes): bytes of one (channel octet, plane) slab of the input / the gradient = pixels * 16
  int exp_flags;                   // timing experiments (compile-time -DWG_TRAFFIC_EXP=<bits>, 0 in every shipped build): 1 = no operand traffic, 2 = no partial-sum stores, 4 = operands re-read from one place (wrong results)
  FastDiv div_h;
};

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0);
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}
__device__ __forceinline__ void split2h(const float x0, const float x1, unsigned& h, unsigned& l) {
  const f16x2 hh = __builtin_convertvector(f32x2{x0, x1}, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = fmaf((float)hh[0], -1.0f, x0), r1 = fmaf((float)hh[1], -1.0f, x1);      // (exact; the form v_fma_mix_f32 matches)
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
__device__ __forceinline__ f32x16 mfma_f16(const f16x8 a, const f16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// 8 consecutive pixels of one channel: two transposing reads (pixels 0..3 and 4..7 of the octet)
__device__ __forceinline__ f16x8 read_tr8(const unsigned char* p, int pitch) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p + 4 * pitch));
  return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// LDS row pitch (bytes) of a [pixel][C channels] fp16 image such that the 4 pixel rows x 64 bytes a 32-lane group of a
// transposing read touches fall into four different 64-byte bank windows: pitch = 64 or 192 (mod 256).
constexpr int lds_pitch(int C) {
  const int raw = 2 * C;
  const int m = raw % 256;
  return m == 64 || m == 192 ? raw : (m < 64 ? raw + (64 - m) : (m < 192 ? raw + (192 - m) : raw + (320 - m)));
}

template <int WK, int WN, int KB, int NB>
struct Cfg {
  static constexpr int NW = WK * WN, NTH = 64 * NW;              // waves / threads per workgroup
  static constexpr int KT = 32 * KB * WK, NT = 32 * NB * WN;
  static constexpr int QX = KT / 4, QG = NT / 4;                 // float4 items per pixel
  static constexpr int NXI = (16 * QX + NTH - 1) / NTH;          // body items (2 octets x 8 pixels x QX) per thread
  static constexpr int NGI = (16 * QG + NTH - 1) / NTH;
  static constexpr bool X_OSTATIC = (8 * QX) % NTH == 0, G_OSTATIC = (8 * QG) % NTH == 0;     // an item's octet is the same for all threads
  static constexpr int NPX = 10;                                 // staged input pixels per octet: halo, 8, halo
  static constexpr int PX = lds_pitch(KT), PG = lds_pitch(NT);
  static constexpr int HW = KT % 128 == 0 ? 2 : 4;               // floats of a halo pixel per thread
  static constexpr int XPL_DATA = 2 * NPX * PX, GPL_DATA = 2 * 8 * PG;       // bytes of one fp16 plane (two octets) ...
  static constexpr int XPL = XPL_DATA + 8 * NTH, GPL = GPL_DATA + 8 * NTH;   // ... + a dump area: 8 bytes per thread for stores that must go nowhere
  static constexpr int BUF = 2 * XPL + 2 * GPL;                  // h and l planes of both operands
  static constexpr int LDS = 2 * BUF;
  static_assert(4 * KT / HW <= NTH, "one halo item per thread at most");
};

// PL (round 6): both operands arrive as H8 planes -- [C/8][plane][pixel][8] fp16 of x * 2^k, written once by their producer
// (split.hip) with the scale their amax slot prescribes: a staging item is one 16-byte load of 8 channels of one plane and one
// ds_write_b128 into the same [pixel][channel] LDS image the transposing reads want -- no prologue, scale or split VALU at all.
template <int WK, int WN, int KB, int NB, bool PRO, bool PAIR, bool X1, bool PL = false>
__global__ __launch_bounds__(64 * WK * WN, WK * WN >= 4 ? WK * WN / 4 : 2) void conv_wgrad_rows_k(WgRowsArgs a) {
  using C = Cfg<WK, WN, KB, NB>;
  static_assert(!PL || (!PRO && C::KT % 32 == 0 && C::NT % 32 == 0 && C::KT <= C::NTH), "plane operands: no prologue, whole 32-channel groups");
  static_assert(C::NW == 1 || C::NW == 2 || C::NW == 4 || C::NW == 8, "one or two waves per SIMD; one / two waves per workgroup for the 32-channel tiles");
  constexpr int KT = C::KT, NT = C::NT, QX = C::QX, QG = C::QG, NXI = C::NXI, NGI = C::NGI, NPX = C::NPX, PX = C::PX, PG = C::PG, NTH = C::NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wk = wave % WK, wn = wave / WK;

  // ---- which unit of work (XCD-aware order: workgroup b runs on XCD b % 8, and XCD c takes the contiguous chunk c of the work
  //      list, whose innermost index is the unit: the kernel rows that re-read one pixel range share one L2) ----
  const unsigned slot_id = blockIdx.x >> 3, work = (blockIdx.x & 7u) * (unsigned)a.chunk + slot_id;
  if (slot_id >= (unsigned)a.chunk || work >= (unsigned)a.total) return;
  unsigned wrk = work;
  const int ui = wrk % (unsigned)a.n_units; wrk /= (unsigned)a.n_units;
  const int n_yt = a.n_ktiles * a.n_ntiles;
  const int ytile = wrk % (unsigned)n_yt; wrk /= (unsigned)n_yt;
  const int split = wrk % (unsigned)a.n_split;
  const int group = wrk / (unsigned)a.n_split;
  const RowUnit u = a.units[ui];
  const bool second = u.acc != 0;
  const int k_tile = ytile / a.n_ntiles, n_tile = ytile - k_tile * a.n_ntiles;
  const int k0 = k_tile * KT, n0 = n_tile * NT;
  const mpose_wgrad_operands& op = a.op[group];
  const float* gout = second ? op.gout1 : op.gout0;
  float* dw = second ? op.dw1 : op.dw0;
  const int n_widx = second ? a.n_widx1 : a.n_widx0;
  const int g_pix = second ? a.g_pix1 : a.g_pix0;
  const int x_pix = a.x_pix;
  const int dy = u.dy;
  const bool three = u.ntap == 3;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.in), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, second ? a.g_bytes1 : a.g_bytes0, 0x00020000);
  constexpr bool pro = PRO;                  // BatchNorm + ReLU of the producer applied while staging (all groups alike)
  const int kx = f16_scale_exp(amax_gather(op.in_amax));
  const int kg = f16_scale_exp(amax_gather(second ? op.gout1_amax : op.gout0_amax));
  const float x_mul = pow2f(kx), g_mul = pow2f(kg);

  // ---- per-thread staging items (loop invariant).  The body pixels of both octets are one list of 16 * QX float4 items handed out
  //      thread by thread: item -> (octet, pixel j, channel quad q); kept per item: the global byte offset relative to the octet,
  //      which octet, and the LDS byte offset inside a plane (threads without an item: out-of-range offset, dump area). ----
  unsigned xb_voff[NXI], xb_lds[NXI], g_voff[NGI], g_lds[NGI];
  bool xb_o[NXI], g_o[NGI];
  float4 xb_sc[NXI], xb_sh[NXI];
  // (PL: an octet's 2 * KT items of 16 bytes are numbered r = [c8 / 4][plane][pixel / 4][c8 % 4][pixel % 4]: sixteen consecutive
  //  lanes store 4 pixel rows x 4 channel octets = sixteen different 16-byte bank groups of a plane (row pitch = 64 or 192 mod 256),
  //  and four consecutive lanes read 64 contiguous bytes of one (octet, plane) slab)
  auto pl_item = [&](int r, int& j, int& pl, int& c8) { j = (r & 3) | (((r >> 4) & 1) << 2); pl = (r >> 5) & 1; c8 = ((r >> 6) << 2) | ((r >> 2) & 3); };
  const unsigned x_slab = a.x_slab, g_slab = a.g_slab;
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int it = tid + NTH * i;
    if constexpr (PL) {
      const int o = it / (8 * QX), r = it - o * (8 * QX);
      int j, pl, c8;
      pl_item(r, j, pl, c8);
      const bool ok = it < 16 * QX && k0 + 8 * c8 < a.Cin;
      xb_o[i] = o != 0;
      xb_voff[i] = ok && !(a.exp_flags & 1) ? (unsigned)(((k0 >> 3) + c8) * 2 + pl) * x_slab + (unsigned)(j * 16) : kBig;
      xb_lds[i] = it < 16 * QX ? (unsigned)(pl * C::XPL + (o * NPX + j + 1) * PX + c8 * 16) : (unsigned)(C::XPL_DATA + (tid >> 1) * 16);
      continue;
    }
    const int o = it / (8 * QX), r = it - o * (8 * QX), j = r / QX, q = r - j * QX;
    const bool ok = it < 16 * QX && k0 + 4 * q < a.Cin;
    xb_o[i] = o != 0;
    xb_voff[i] = ok && !(a.exp_flags & 1) ? (unsigned)(j * x_pix + (k0 + 4 * q) * 4) : kBig;
    xb_lds[i] = it < 16 * QX ? (unsigned)((o * NPX + j + 1) * PX + q * 8) : (unsigned)(C::XPL_DATA + tid * 8);
    xb_sc[i] = make_float4(x_mul, x_mul, x_mul, x_mul);
    xb_sh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pro && ok) {
      const float4 s = *reinterpret_cast<const float4*>(op.in_scale + k0 + 4 * q), t = *reinterpret_cast<const float4*>(op.in_shift + k0 + 4 * q);
      xb_sc[i] = make_float4(s.x * x_mul, s.y * x_mul, s.z * x_mul, s.w * x_mul);       // relu(s x + t) 2^k == relu((s 2^k) x + t 2^k)
      xb_sh[i] = make_float4(t.x * x_mul, t.y * x_mul, t.z * x_mul, t.w * x_mul);
    }
  }
#pragma unroll
  for (int i = 0; i < NGI; ++i) {
    const int it = tid + NTH * i;
    if constexpr (PL) {
      const int o = it / (8 * QG), r = it - o * (8 * QG);
      int j, pl, c8;
      pl_item(r, j, pl, c8);
      const bool ok = it < 16 * QG && n0 + 8 * c8 < a.Cout;
      g_o[i] = o != 0;
      g_voff[i] = ok && !(a.exp_flags & 1) ? (unsigned)(((n0 >> 3) + c8) * 2 + pl) * g_slab + (unsigned)(j * 16) : kBig;
      g_lds[i] = it < 16 * QG ? (unsigned)(pl * C::GPL + (o * 8 + j) * PG + c8 * 16) : (unsigned)(C::GPL_DATA + (tid >> 1) * 16);
      continue;
    }
    const int o = it / (8 * QG), r = it - o * (8 * QG), j = r / QG, q = r - j * QG;
    const bool ok = it < 16 * QG && n0 + 4 * q < a.Cout;
    g_o[i] = o != 0;
    g_voff[i] = ok && !(a.exp_flags & 1) ? (unsigned)(j * g_pix + (n0 + 4 * q) * 4) : kBig;
    g_lds[i] = it < 16 * QG ? (unsigned)((o * 8 + j) * PG + q * 8) : (unsigned)(C::GPL_DATA + tid * 8);
  }
  // halo pixels of the two octets (three-tap units), HW floats per thread: thread t takes (octet, side) = t / TPS, channels
  // (t % TPS) * HW .. + HW-1.  Threads beyond 4 * TPS read nothing (out-of-range offset) and store into the dump area.
  // (PL: a halo pixel is KT / 4 items of 16 bytes -- (channel octet, plane) -- one per thread: tid = [octet, side][c8][plane])
  constexpr int HW = PL ? 4 : C::HW, TPS = PL ? KT / 4 : KT / HW;
  const int h_id = tid / TPS, h_oct = (h_id >> 1) & 1, h_side = h_id & 1, h_c = PL ? ((tid % TPS) >> 1) * 8 : (tid % TPS) * HW;
  const int h_pl = PL ? (tid % TPS) & 1 : 0;
  const bool h_ok = h_id < 4 && k0 + h_c < a.Cin;
  const unsigned h_col = PL ? (unsigned)(((k0 + h_c) >> 3) * 2 + h_pl) * x_slab : (unsigned)((k0 + h_c) * 4);
  const unsigned h_lds = h_id < 4 ? (unsigned)(h_pl * C::XPL + (h_oct * NPX + (h_side ? NPX - 1 : 0)) * PX + h_c * 2)
                                  : (unsigned)(C::XPL_DATA + (PL ? (tid >> 1) * 16 : tid * 8));
  float4 h_sc = make_float4(x_mul, x_mul, x_mul, x_mul), h_sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pro && h_ok) {
    const float* ps = op.in_scale + k0 + h_c;
    const float* pt = op.in_shift + k0 + h_c;
    h_sc.x = ps[0] * x_mul; h_sc.y = ps[1] * x_mul; h_sh.x = pt[0] * x_mul; h_sh.y = pt[1] * x_mul;
    if (HW == 4) { h_sc.z = ps[2] * x_mul; h_sc.w = ps[3] * x_mul; h_sh.z = pt[2] * x_mul; h_sh.w = pt[3] * x_mul; }
  }

  // ---- fragment read addresses (bytes inside a plane): lane = (pixel half lh -> octet, 16-channel half, pixel e, channel quad) ----
  const int f_e = (lane & 15) >> 2, f_j = lane & 3, f_c16 = (lane >> 4) & 1;
  const unsigned fa_x = (unsigned)((lh * NPX + f_e) * PX + (wk * KB * 32 + f_c16 * 16 + f_j * 4) * 2);
  const unsigned fa_g = (unsigned)((lh * 8 + f_e) * PG + (wn * NB * 32 + f_c16 * 16 + f_j * 4) * 2);

  // ---- scalar cursor over (valid slot row, octet).  A slot row r = b*H + gy is valid for this kernel row when 0 <= gy + dy < H:
  //      per image the Hv = H - |dy| rows gy = vy + max(0, -dy), vy = 0 .. Hv-1, and inside an image consecutive valid rows are
  //      consecutive in memory: the byte offsets of the octets advance by 8 pixels, plus |dy| rows at an image boundary.  The
  //      split's row range is mapped to ordinals of valid rows once. ----
  const int n_oct = a.W >> 3;
  const int ady = dy < 0 ? -dy : dy;
  const int Hv = a.H - ady;                               // (<= 0: no valid row at all)
  const int x_row0 = dy > 0 ? dy : 0, g_row0 = dy < 0 ? -dy : 0;      // input / gradient row of vy = 0
  auto valid_before = [&](int r) {                        // number of valid rows among slot rows 0 .. r-1
    const int b = (int)fdiv((unsigned)r, a.div_h);
    const int gy = r - b * a.H;
    const int part = min(max(gy - g_row0, 0), Hv);
    return b * Hv + part;
  };
  const int r_split0 = min(a.n_rows, split * a.rows_per_split);
  const int r_end = min(a.n_rows, r_split0 + a.rows_per_split);
  const int v_begin = Hv > 0 ? valid_before(r_split0) : 0, v_end = Hv > 0 ? valid_before(r_end) : 0;
  const int n_octets = (v_end - v_begin) * n_oct;
  const int n_steps = (n_octets + 1) >> 1;
  const bool exp_still = (a.exp_flags & 4) != 0;          // (timing experiment: every step re-reads the first octets -> cache-resident operands)
  const int x_rowb = a.x_row, g_rowb = second ? a.g_row1 : a.g_row0;
  const unsigned x_step = exp_still ? 0u : (unsigned)(8 * x_pix), g_step = exp_still ? 0u : (unsigned)(8 * g_pix);
  const unsigned x_jump = exp_still ? 0u : (unsigned)(ady * x_rowb), g_jump = exp_still ? 0u : (unsigned)(ady * g_rowb);
  // (a view's rows need not follow each other in memory -- every second row and pixel of a tensor: the gap is added at a row's end)
  const unsigned x_gap = exp_still ? 0u : (unsigned)(x_rowb - a.W * x_pix), g_gap = exp_still ? 0u : (unsigned)(g_rowb - a.W * g_pix);
  struct Cursor { unsigned xs, gs; int vy, o, left; };    // left = octets still to hand out (<= 0: dummy octets, gradient reads 0)
  Cursor cur;
  {
    const int hv = Hv > 0 ? Hv : 1;
    const int b = v_begin / hv;
    cur.vy = v_begin - b * hv; cur.o = 0; cur.left = n_octets;
    cur.xs = (unsigned)((b * a.H + cur.vy + x_row0) * x_rowb + u.x_off);
    cur.gs = (unsigned)((b * a.H + cur.vy + g_row0) * g_rowb + u.g_off);
  }
  struct Oct { unsigned xs, gs, hl, hr, gpen; };     // body offsets (bytes), halo offsets (or kBig), gradient penalty (0 / kBig)
  auto take = [&](Cursor& c) {
    Oct o;
    o.xs = c.xs; o.gs = c.gs;
    o.hl = c.o > 0 ? c.xs - (unsigned)x_pix : kBig;
    o.hr = c.o + 1 < n_oct ? c.xs + x_step : kBig;
    o.gpen = c.left > 0 ? 0u : kBig;
    const bool row_done = c.o + 1 == n_oct;
    const bool img_done = row_done && c.vy + 1 == Hv;
    c.xs += x_step + (row_done ? x_gap : 0u) + (img_done ? x_jump : 0u);
    c.gs += g_step + (row_done ? g_gap : 0u) + (img_done ? g_jump : 0u);
    c.left -= 1;
    c.o = row_done ? 0 : c.o + 1;
    c.vy = img_done ? 0 : c.vy + (row_done ? 1 : 0);
    return o;
  };

  struct Oct2 { Oct o[2]; };
  // The two octets of a step.  PAIR (an even number of octets per row): they are neighbours in one row, which halves the
  // scalar work -- the inner halo pixels are each other's border pixels, and there is no odd tail.
  auto take2 = [&](Cursor& c, Oct2& oo) {
    if constexpr (PAIR) {
      const unsigned xs1 = c.xs + x_step, pen = c.left > 0 ? 0u : kBig;
      oo.o[0].xs = c.xs; oo.o[0].gs = c.gs; oo.o[1].xs = xs1; oo.o[1].gs = c.gs + g_step;
      oo.o[0].hl = c.o > 0 ? c.xs - (unsigned)x_pix : kBig;
      oo.o[0].hr = xs1;
      oo.o[1].hl = xs1 - (unsigned)x_pix;
      oo.o[1].hr = c.o + 2 < n_oct ? xs1 + x_step : kBig;
      oo.o[0].gpen = pen; oo.o[1].gpen = pen;
      const bool row_done = c.o + 2 == n_oct;
      const bool img_done = row_done && c.vy + 1 == Hv;
      c.xs = xs1 + x_step + (row_done ? x_gap : 0u) + (img_done ? x_jump : 0u);
      c.gs += 2 * g_step + (row_done ? g_gap : 0u) + (img_done ? g_jump : 0u);
      c.left -= 2;
      c.o = row_done ? 0 : c.o + 2;
      c.vy = img_done ? 0 : c.vy + (row_done ? 1 : 0);
    } else {
      oo.o[0] = take(c); oo.o[1] = take(c);
    }
  };

  // ---- staging: global -> registers -> (BN + ReLU prologue, scale, split) -> LDS.  The operands of a step are a list of ITEMS
  //      (one load per thread each): X body of octet 0 / 1 (NXB each), G of octet 0 / 1 (NGB each), the X halos (three-tap units).
  //      Item i of step s+1 is staged, and its registers refilled with item i of step s+2, between two MFMAs of step s.  All of
  //      it is branch-free (threads without an item read out of range and store into the dump area), so that a step is ONE basic
  //      block and the staging instructions can be interleaved with the MFMAs one by one. ----
  constexpr int I_G = NXI, I_HALO = NXI + NGI;
  float4 raw[I_HALO + 1];
  unsigned h_base_prev = kBig;          // offset the halo registers were loaded from (HW == 4; HW == 2 keeps it in raw[].z)
  auto load_item = [&](auto ic, const Oct2& oo) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I < I_G) {
      if constexpr (C::X_OSTATIC) raw[I] = buf_load4(rs_x, xb_voff[I], oo.o[(I * NTH) / (8 * QX)].xs);
      else raw[I] = buf_load4(rs_x, xb_voff[I] + (xb_o[I] ? oo.o[1].xs : oo.o[0].xs), 0);
    } else if constexpr (I < I_HALO) {
      constexpr int i = I - I_G;
      if constexpr (C::G_OSTATIC) {
        constexpr int o = (i * NTH) / (8 * QG);
        raw[I] = buf_load4(rs_g, g_voff[i] + oo.o[o].gpen, oo.o[o].gs);
      } else {
        raw[I] = buf_load4(rs_g, g_voff[i] + (g_o[i] ? oo.o[1].gs + oo.o[1].gpen : oo.o[0].gs + oo.o[0].gpen), 0);
      }
    } else {
      const unsigned s0 = h_side ? oo.o[0].hr : oo.o[0].hl, s1 = h_side ? oo.o[1].hr : oo.o[1].hl;
      const unsigned base = h_ok ? (h_oct ? s1 : s0) : kBig;
      if constexpr (HW == 4) {
        raw[I] = buf_load4(rs_x, h_col + base, 0);
        h_base_prev = base;
      } else {
        const uint2 r = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs_x, (int)(h_col + base), 0, 0));
        raw[I] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(base), 0.f);     // (.z: the offset, for `keep`)
      }
    }
  };
  auto split_store4 = [&](unsigned char* plane_h, int plane_b, unsigned off, const float4& v) {
    if constexpr (X1) {                 // MPOSE_CONV_F16X1: the operands are ROUNDED to fp16 (the h pieces alone)
      uint2 h;
      h.x = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, f16x2));
      h.y = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, f16x2));
      *reinterpret_cast<uint2*>(plane_h + off) = h;
    } else {
      uint2 h, l;
      split2h(v.x, v.y, h.x, l.x);
      split2h(v.z, v.w, h.y, l.y);
      *reinterpret_cast<uint2*>(plane_h + off) = h;
      *reinterpret_cast<uint2*>(plane_h + plane_b + off) = l;
    }
  };
  auto prologue4 = [&](float4 v, const float4& sc, const float4& sh) {
    if (pro) {
      v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
      v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    } else {
      v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
    }
    return v;
  };
  auto stage_item = [&](auto ic, unsigned char* buf) {
    constexpr int I = decltype(ic)::value;
    if constexpr (PL) {                // the producer's bytes as they are (an out-of-range halo or a dummy octet was read as zeros)
      if constexpr (I < I_G) *reinterpret_cast<float4*>(buf + xb_lds[I]) = raw[I];
      else if constexpr (I < I_HALO) *reinterpret_cast<float4*>(buf + 2 * C::XPL + g_lds[I - I_G]) = raw[I];
      else *reinterpret_cast<float4*>(buf + h_lds) = raw[I];
    } else if constexpr (I < I_G) {
      split_store4(buf, C::XPL, xb_lds[I], prologue4(raw[I], xb_sc[I], xb_sh[I]));
    } else if constexpr (I < I_HALO) {
      float4 v = raw[I];
      v.x *= g_mul; v.y *= g_mul; v.z *= g_mul; v.w *= g_mul;
      split_store4(buf + 2 * C::XPL, C::GPL, g_lds[I - I_G], v);
    } else {
      // (a padding pixel under the prologue: relu(shift) need not be zero -> the pieces are cleared)
      if constexpr (HW == 4) {
        float4 v = prologue4(raw[I], h_sc, h_sh);
        if (pro && h_base_prev >= kBig) v = make_float4(0.f, 0.f, 0.f, 0.f);
        split_store4(buf, C::XPL, h_lds, v);
      } else {
        float4 v = prologue4(raw[I], h_sc, h_sh);
        if (pro && __float_as_uint(raw[I].z) >= kBig) { v.x = 0.f; v.y = 0.f; }
        unsigned h, l;
        split2h(v.x, v.y, h, l);
        *reinterpret_cast<unsigned*>(buf + h_lds) = h;
        if constexpr (!X1) *reinterpret_cast<unsigned*>(buf + C::XPL + h_lds) = l;
      }
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  auto body = [&](auto ntap_c) {
    constexpr int NTAP = decltype(ntap_c)::value;
    constexpr int NPROD = X1 ? 1 : 3;                  // products per multiply-add
    constexpr int NPH = NPROD * NTAP;                  // phases of a step: (tap, product), KB*NB MFMAs each
    constexpr int TOFF = NTAP == 3 ? 0 : 1;            // single tap: dx = 0 = staged pixel 1
    constexpr int NI = I_HALO + (NTAP == 3 ? 1 : 0);   // (no halo for a single tap)
    auto for_items = [&](auto&& f) {                   // f(integral_constant<int, I>) for I = 0 .. NI-1
      [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, NI>{});
    };
    f32x16 acc[NTAP][KB][NB];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][kb][nb][r] = 0.0f;

    if (n_steps > 0) {
      // Schedule of step s (on LDS buffer s & 1; fragments of a step are prefetched by the step before it):
      //   phases 0 .. PB-1   MFMAs of the first taps | stage step s+1 into the other buffer, reload its registers with step s+2 |
      //                      fragment reads of the later taps
      //   barrier            every wave's stores of step s+1 have landed; nobody reads buffer s & 1 through LDS any more
      //   phases PB .. NPH-1 MFMAs of the last tap | fragment reads of step s+1's gradient and first tap | cursor of step s+3
      constexpr int PB = NPH - NPH / 3;
      constexpr int NSP = (NTAP == 3 && PB > 1) ? PB - 1 : PB;     // phases that carry staging items
      Oct2 oo;
      f16x8 bh[2][NB], bl[2][NB], ah[2][KB], al[2][KB];
      auto read_b = [&](const unsigned char* cb, f16x8 (&h)[NB], f16x8 (&l)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          h[nb] = read_tr8(cb + 2 * C::XPL + fa_g + nb * 64, PG);
          if constexpr (!X1) l[nb] = read_tr8(cb + 2 * C::XPL + C::GPL + fa_g + nb * 64, PG);
        }
      };
      auto read_a = [&](const unsigned char* cb, int t, f16x8 (&h)[KB], f16x8 (&l)[KB]) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          h[kb] = read_tr8(cb + fa_x + (t + TOFF) * PX + kb * 64, PX);
          if constexpr (!X1) l[kb] = read_tr8(cb + C::XPL + fa_x + (t + TOFF) * PX + kb * 64, PX);
        }
      };
      take2(cur, oo);
      for_items([&](auto ic) { load_item(ic, oo); });
      take2(cur, oo);
      for_items([&](auto ic) { stage_item(ic, smem); load_item(ic, oo); });       // step 0 -> LDS, step 1 -> registers
      take2(cur, oo);                                                              // step 2
      lds_barrier();
      read_b(smem, bh[0], bl[0]);
      read_a(smem, 0, ah[0], al[0]);
      auto step = [&](auto buf_c) {
        constexpr int BUFI = decltype(buf_c)::value;
        const unsigned char* cb = smem + BUFI * C::BUF;
        unsigned char* nb_ = smem + (BUFI ^ 1) * C::BUF;
        // One phase = the KB*NB MFMAs of one (tap, product) + the other work that belongs beside them.  The phases before the
        // barrier form one scheduling region, the phases after it another: inside a region the MFMAs are spread evenly and
        // every gap between two of them gets a few of the other instructions (at most five hide behind an MFMA when a SIMD
        // holds one wave: MI355X_MICROARCH.md).
        auto phase = [&](auto ph_c) {
          constexpr int PH = decltype(ph_c)::value, t = PH / NPROD, p = X1 ? 2 : PH % NPROD;      // (p == 2: the h x h product)
          constexpr int aset = (t + BUFI) & 1;
          constexpr int i_lo = PH < NSP ? (PH * NI + NSP - 1) / NSP : NI, i_hi = PH < NSP ? ((PH + 1) * NI + NSP - 1) / NSP : NI;
          if constexpr (PH % NPROD == 0 && t + 1 < NTAP && !(WG_EXP & 1)) read_a(cb, t + 1, ah[aset ^ 1], al[aset ^ 1]);      // the next tap, a tap's phases ahead
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              if constexpr (!(WG_EXP & 8)) acc[t][kb][nb] = mfma_f16(p == 0 ? al[aset][kb] : ah[aset][kb], p == 1 ? bl[BUFI][nb] : bh[BUFI][nb], acc[t][kb][nb]);
              else asm volatile("" :: "v"(al[aset][kb]), "v"(ah[aset][kb]), "v"(bl[BUFI][nb]), "v"(bh[BUFI][nb]));
          [&]<int... Js>(std::integer_sequence<int, Js...>) {
            (((WG_EXP & 2) ? (void)0 : stage_item(std::integral_constant<int, i_lo + Js>{}, nb_), load_item(std::integral_constant<int, i_lo + Js>{}, oo)), ...);
          }(std::make_integer_sequence<int, i_hi - i_lo>{});
        };
        __builtin_amdgcn_sched_barrier(0);
        [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (phase(std::integral_constant<int, Ps>{}), ...); }(std::make_integer_sequence<int, PB>{});
#pragma unroll
        for (int m = 0; m < PB * KB * NB; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          if (m % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(WG_EXP & 4)) lds_barrier();
        if constexpr (!(WG_EXP & 1)) {                  // the next step's first fragments
          read_b(nb_, bh[BUFI ^ 1], bl[BUFI ^ 1]);
          read_a(nb_, 0, ah[BUFI ^ 1], al[BUFI ^ 1]);
        }
        take2(cur, oo);                                 // octets of the next step's loads
        [&]<int... Ps>(std::integer_sequence<int, Ps...>) { (phase(std::integral_constant<int, PB + Ps>{}), ...); }(std::make_integer_sequence<int, NPH - PB>{});
#pragma unroll
        for (int m = 0; m < (NPH - PB) * KB * NB; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      int s = 0;
#pragma unroll 1
      for (; s + 1 < n_steps; s += 2) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
      }
      if (s < n_steps) step(std::integral_constant<int, 0>{});
    }

    // ---- epilogue: back to the tensors' own units (exact), then the wave's sub-tile of every tap as split-K partials
    //      [split][widx][K/4][Npad][4]: a float4 = 4 consecutive input channels of one output channel ----
    const int k4_total = a.Cin >> 2;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      if (u.widx[t] < 0) continue;           // (a position of the row the kernel has no tap at)
      float* base = dw + ((long)(split * n_widx + u.widx[t]) * k4_total) * a.npad * 4;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int n = n0 + (wn * NB + nb) * 32 + li;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int k4 = (k0 + (wk * KB + kb) * 32) / 4 + 2 * rg + lh;
            if (k4 < k4_total && n < a.npad && n0 + (wn * NB + nb) * 32 < a.Cout && !(a.exp_flags & 2)) {
              const float4 v = make_float4(__builtin_ldexpf(acc[t][kb][nb][4 * rg], -(kx + kg)), __builtin_ldexpf(acc[t][kb][nb][4 * rg + 1], -(kx + kg)),
                                           __builtin_ldexpf(acc[t][kb][nb][4 * rg + 2], -(kx + kg)), __builtin_ldexpf(acc[t][kb][nb][4 * rg + 3], -(kx + kg)));
              *reinterpret_cast<float4*>(base + ((long)k4 * a.npad + n) * 4) = v;
            }
          }
        }
    }
  };
  if (three) body(std::integral_constant<int, 3>{});
  else body(std::integral_constant<int, 1>{});
}

template <int WK, int WN, int KB, int NB, bool PRO, bool PAIR, bool X1, bool PL = false>
int launch_rows_ppx(WgRowsArgs& a, hipStream_t s) {
  using C = Cfg<WK, WN, KB, NB>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rows_k<WK, WN, KB, NB, PRO, PAIR, X1, PL>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess)
      return MPOSE_EINVAL;
    attr_set = true;
  }
  a.n_ktiles = (a.Cin + C::KT - 1) / C::KT;
  a.n_ntiles = (a.Cout + C::NT - 1) / C::NT;
  a.total = a.n_units * a.n_ktiles * a.n_ntiles * a.n_split * a.n_groups;
  a.chunk = (a.total + 7) / 8;
  launch(conv_wgrad_rows_k<WK, WN, KB, NB, PRO, PAIR, X1, PL>, dim3(dim3(8 * a.chunk)), dim3(C::NTH), C::LDS, s, a);
  return launch_status();
}
template <int WK, int WN, int KB, int NB, bool PRO, bool PAIR>
int launch_rows_pp(WgRowsArgs& a, hipStream_t s) {
  return a.op[0].single_product ? launch_rows_ppx<WK, WN, KB, NB, PRO, PAIR, true>(a, s) : launch_rows_ppx<WK, WN, KB, NB, PRO, PAIR, false>(a, s);
}
template <int WK, int WN, int KB, int NB, bool PRO>
int launch_rows_p(WgRowsArgs& a, hipStream_t s) {
  return (a.W & 15) == 0 ? launch_rows_pp<WK, WN, KB, NB, PRO, true>(a, s) : launch_rows_pp<WK, WN, KB, NB, PRO, false>(a, s);
}
template <int WK, int WN, int KB, int NB>
int launch_rows(WgRowsArgs& a, hipStream_t s) {
  return a.op[0].in_scale != nullptr ? launch_rows_p<WK, WN, KB, NB, true>(a, s) : launch_rows_p<WK, WN, KB, NB, false>(a, s);
}

// tile shape per (Cin, Cout): 0 = 128 x 128 (2x2 waves of 64 x 64), 1 = 192 x 64 (2x2 waves of 96 x 32), 2 = 128 x 32 (4x1 waves of
// 32 x 32), 3 = 64 x 64 (2x2 waves of 32 x 32).  MPOSE_WGRAD_192=1 (experiments) runs shapes 0 / 1 with EIGHT waves per workgroup
// (two per SIMD; 2x4 waves of 64 x 32 / 96 x 32, tiles 128 x 128 / 192 x 128): measured slower -- 95 vs 88 us on the 128-channel
// layers -- because eight waves read 1.75x the fragment bytes from LDS for the same MFMAs (profiles/r3_wgrad_loop_parts.txt).
// 4 = 32 x 32 as ONE wave, 5 = 32 x 64 as two (round 5): the feature extractor's 32-channel layers at 128 x 128 ran shape 3 with
// one / two of its four waves working -- a step of the kernel is latency-bound at that width (two octets staged per barrier), so
// the launch's duration follows its number of workgroups, not their size: a quarter / half of the footprint per workgroup lets the
// caller split the pixels 4x / 2x further on the same CUs (mpose_conv_wgrad_waves).
constexpr bool narrow_env() { return true; }
inline int rows_shape(int cin, int cout) {
  if (cin % 192 == 0 && cout % 64 == 0) return 1;
  if (narrow_env() && cin <= 32 && cout <= 32) return 4;
  if (narrow_env() && cin <= 32 && cout <= 64) return 5;
  if (cout <= 32) return cin > 64 ? 2 : 3;
  if (cin <= 64 && cout <= 64) return 3;
  return 0;
}
constexpr bool wide192() { return false; }      // (round 3's eight-wave tiles: measured slower, not instantiated any more)
inline void shape_tiles(int shape, int& kt, int& nt) {
  switch (shape) {
    case 1: kt = 192; nt = wide192() ? 128 : 64; break;
    case 2: kt = 128; nt = 32; break;
    case 3: kt = 64; nt = 64; break;
    case 4: kt = 32; nt = 32; break;
    case 5: kt = 32; nt = 64; break;
    default: kt = 128; nt = 128; break;
  }
}

constexpr bool strided_env() { return true; }      // (the stride-2 geometries as strided views: round 5)
constexpr int rows_env() { return 1; }      // (2: the row form also for launches of single taps -- measured slower)

// Kernel rows / single taps of a geometry, or -1 when the row form does not apply (then conv.hip's conv_wgrad_k runs).
//
// What the kernel walks is a slot grid on which a VIEW of the input and a VIEW of the gradient are correlated at shifts |dx| <= 1:
//   * stride 1: both views are the tensors;
//   * a stride-2 Conv2d (in_mul = 2; round 5): input pixel 2 gy + dy = 2 (gy + ay) + ry -- the taps split by the residues (ry, rx)
//     of their offsets, and residue (ry, rx) sees the view "every second row and pixel of the input, starting at (ry, rx)" at the
//     shifts (ay, ax): 3 x 3 with padding 1 = one single tap, two pairs {-1, 0} and two rows of pairs;
//   * a stride-2 ConvTranspose2d (out_mul = 2, one class per output phase): the same with the GRADIENT as the strided side -- class
//     (oy, ox) pairs the input at (gy + dy, gx + dx) with every second row and pixel of the gradient starting at (oy, ox).
// A unit = (class, accumulator, residue, ay) with its taps ax in {-1, 0, +1}; a pair runs as a row of three whose missing tap is
// computed and dropped (widx -1).
int build_units(const mpose_conv_geom* g, RowUnit* units, int* n_widx0, int* n_widx1) {
  const int im = g->in_mul, om = g->out_mul;
  if (g->in_mul_x != im || g->out_mul_x != om) return -1;
  if (!((im == 1 && om == 1) || (strided_env() && ((im == 2 && om == 1) || (im == 1 && om == 2))))) return -1;
  if (g->IH != g->GH * im || g->IW != g->GW * im || g->OH != g->GH * om || g->OW != g->GW * om) return -1;
  if ((g->GW & 7) || (g->Cin & 3) || g->n_classes != om * om) return -1;
  const int in_ld = g->in_ld > 0 ? g->in_ld : g->Cin;
  const int g_ld0 = g->out_ld0 > 0 ? g->out_ld0 : g->Cout0, g_ld1 = g->out_ld1 > 0 ? g->out_ld1 : g->Cout1;
  auto fmod_ = [](int v, int m) { return ((v % m) + m) % m; };
  int n_units = 0, max0 = -1, max1 = -1, matched = 0, total = 0;
  for (int ci = 0; ci < g->n_classes; ++ci) {
    const mpose_tap_class& c = g->cls[ci];
    if (c.oy < 0 || c.oy >= om || c.ox < 0 || c.ox >= om) return -1;
    total += c.n_taps;
    for (int acc = 0; acc < 2; ++acc)
      for (int ry = 0; ry < im; ++ry)
        for (int rx = 0; rx < im; ++rx)
          for (int ay = -8; ay <= 8; ++ay) {          // (dilated kernels: rows 2, 4 pixels apart)
            int widx[3] = {-1, -1, -1}, cnt = 0;
            for (int t = 0; t < c.n_taps; ++t) {
              const mpose_tap& tp = c.taps[t];
              const int ty = fmod_(tp.dy, im), tx = fmod_(tp.dx, im);
              if ((tp.acc != 0) != (acc != 0) || ty != ry || tx != rx || (tp.dy - ty) / im != ay) continue;
              const int ax = (tp.dx - tx) / im;
              if (ax < -1 || ax > 1 || widx[ax + 1] >= 0 || tp.widx < 0) return -1;
              widx[ax + 1] = tp.widx;
              ++cnt;
            }
            if (!cnt) continue;
            matched += cnt;
            if (n_units == MAX_UNITS) return -1;
            RowUnit u{};
            u.dy = (int8_t)ay; u.acc = (int8_t)acc;
            if (cnt == 1 && widx[1] >= 0) { u.ntap = 1; u.widx[0] = (int8_t)widx[1]; }
            else if (im == 1 && om == 1 && cnt != 3) return -1;      // (stride 1: whole rows or single taps, as before)
            else { u.ntap = 3; u.widx[0] = (int8_t)widx[0]; u.widx[1] = (int8_t)widx[1]; u.widx[2] = (int8_t)widx[2]; }
            u.x_off = (ry * g->IW + rx) * in_ld * 4;
            u.g_off = (c.oy * g->OW + c.ox) * (acc ? g_ld1 : g_ld0) * 4;
            units[n_units++] = u;
          }
    for (int t = 0; t < c.n_taps; ++t) {
      const mpose_tap& tp = c.taps[t];
      if (tp.acc) { if (tp.widx > max1) max1 = tp.widx; } else if (tp.widx > max0) max0 = tp.widx;
    }
  }
  if (matched != total) return -1;         // (a tap further than eight rows away)
  *n_widx0 = max0 + 1;
  *n_widx1 = max1 + 1;
  // a launch of single taps only (1x1 convolutions, k x 1 columns) stages as much per step as a kernel row for a third of the
  // MFMAs: conv_wgrad_k keeps those unless MPOSE_WGRAD_ROWS=2
  bool any3 = false;
  for (int i = 0; i < n_units; ++i) any3 |= units[i].ntap == 3;
  if (!any3 && rows_env() < 2) return -1;
  return n_units;
}

}  // namespace
}  // namespace mpose

using namespace mpose;

// Work units (workgroups per pixel split and column group) of the row form for this geometry, or 0 when it does not apply.
int mpose_wgrad_rows_units(const mpose_conv_geom* geom) {
  if (!rows_env()) return 0;
  RowUnit units[MAX_UNITS];
  int w0, w1;
  const int n = build_units(geom, units, &w0, &w1);
  if (n <= 0) return 0;
  int kt, nt;
  shape_tiles(rows_shape(geom->Cin, geom->Cout0), kt, nt);
  return n * ((geom->Cin + kt - 1) / kt) * ((geom->Cout0 + nt - 1) / nt);
}

// Workgroups of the row form that share a CU for this geometry: the narrow tiles (shapes 2, 3: <= 114 + 48 registers, 44 KB of LDS)
// run three per CU, the wide ones one.
int mpose_wgrad_rows_occupancy(const mpose_conv_geom* geom) {
  const int shape = rows_shape(geom->Cin, geom->Cout0);
  return (shape >= 2 && shape <= 5) ? 3 : 1;
}

// Waves per workgroup of the row form for this geometry (4; 1 / 2 for the 32-channel tiles).
int mpose_wgrad_rows_waves(const mpose_conv_geom* geom) {
  const int shape = rows_shape(geom->Cin, geom->Cout0);
  return shape == 4 ? 1 : (shape == 5 ? 2 : 4);
}

// Called by mpose_conv_wgrad (conv.hip) after it validated geometry and operands.  Returns MPOSE_ENOSYS when the row form does
// not apply (the caller then runs conv_wgrad_k).
int mpose_wgrad_rows_launch(const mpose_conv_geom* geom, const mpose_wgrad_operands* ops, int n_groups, int n_split, void* stream) {
  if (!rows_env() || !ops[0].in_amax) return MPOSE_ENOSYS;
  WgRowsArgs a{};
  a.n_units = build_units(geom, a.units, &a.n_widx0, &a.n_widx1);
  if (a.n_units <= 0) return MPOSE_ENOSYS;
  bool acc1 = false;
  for (int i = 0; i < a.n_units; ++i) acc1 |= a.units[i].acc != 0;
  if (acc1 && geom->Cout1 != geom->Cout0) return MPOSE_ENOSYS;
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.H = geom->GH; a.W = geom->GW; a.n_rows = geom->B * geom->GH;
  a.Cin = geom->Cin; a.Cout = geom->Cout0;
  const int in_ld = geom->in_ld > 0 ? geom->in_ld : geom->Cin;
  const int g_ld0 = geom->out_ld0 > 0 ? geom->out_ld0 : geom->Cout0;
  const int g_ld1 = geom->out_ld1 > 0 ? geom->out_ld1 : geom->Cout1;
  const int im = geom->in_mul, om = geom->out_mul;           // (the views of a strided side: every im-th / om-th row and pixel)
  a.x_pix = in_ld * 4 * im; a.g_pix0 = g_ld0 * 4 * om; a.g_pix1 = g_ld1 * 4 * om;
  a.x_row = geom->IW * in_ld * 4 * im; a.g_row0 = geom->OW * g_ld0 * 4 * om; a.g_row1 = geom->OW * g_ld1 * 4 * om;
  const long npix_i = (long)geom->B * geom->IH * geom->IW, npix_o = (long)geom->B * geom->OH * geom->OW;
  if (npix_i * in_ld * 4 >= (long)kBig || npix_o * (g_ld0 > g_ld1 ? g_ld0 : g_ld1) * 4 >= (long)kBig) return MPOSE_ENOSYS;
  a.x_bytes = (unsigned)((npix_i - 1) * in_ld * 4 + (long)geom->Cin * 4);
  a.g_bytes0 = (unsigned)((npix_o - 1) * g_ld0 * 4 + (long)geom->Cout0 * 4);
  a.g_bytes1 = acc1 ? (unsigned)((npix_o - 1) * g_ld1 * 4 + (long)geom->Cout1 * 4) : 0u;
  a.npad = geom->Npad0;
  a.n_split = n_split;
  a.rows_per_split = (a.n_rows + n_split - 1) / n_split;
  a.n_groups = n_groups;
  a.div_h = make_fastdiv((unsigned)geom->GH);
  a.exp_flags = WG_TRAFFIC_EXP;
  hipStream_t s = (hipStream_t)stream;
  if (ops[0].planes_in) {
    // operands as H8 planes (mpose_wgrad_operands.planes_in): the stride-1 geometries of the 128-channel tile, dense tensors, whole
    // channel octets, an even number of pixel octets per row; a pixel is 16 bytes of one (channel octet, plane) slab
    if (im != 1 || om != 1 || geom->in_ld > 0 || geom->out_ld0 > 0 || geom->out_ld1 > 0 || (geom->GW & 15) || (geom->Cin & 31) ||
        (geom->Cout0 & 31) || rows_shape(geom->Cin, geom->Cout0) != 0)
      return MPOSE_EINVAL;
    for (int i = 0; i < n_groups; ++i)
      if (!ops[i].planes_in || ops[i].in_scale) return MPOSE_EINVAL;
    if (npix_i * 4 * geom->Cin >= (long)kBig || npix_o * 4 * geom->Cout0 >= (long)kBig) return MPOSE_EINVAL;
    a.x_pix = a.g_pix0 = a.g_pix1 = 16;
    a.x_row = geom->IW * 16; a.g_row0 = a.g_row1 = geom->OW * 16;
    a.x_slab = (unsigned)(npix_i * 16); a.g_slab = (unsigned)(npix_o * 16);
    a.x_bytes = (unsigned)(npix_i * 4 * geom->Cin);
    a.g_bytes0 = a.g_bytes1 = (unsigned)(npix_o * 4 * geom->Cout0);
    return ops[0].single_product ? launch_rows_ppx<2, 2, 2, 2, false, true, true, true>(a, s)
                                 : launch_rows_ppx<2, 2, 2, 2, false, true, false, true>(a, s);
  }
  switch (rows_shape(geom->Cin, geom->Cout0)) {
    case 1: return launch_rows<2, 2, 3, 1>(a, s);
    case 2: return launch_rows<4, 1, 1, 1>(a, s);
    case 3: return launch_rows<2, 2, 1, 1>(a, s);
    case 4: return launch_rows<1, 1, 1, 1>(a, s);
    case 5: return launch_rows<1, 2, 1, 1>(a, s);
    default: return launch_rows<2, 2, 2, 2>(a, s);
  }
}
