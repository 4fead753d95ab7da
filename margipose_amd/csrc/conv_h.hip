// Implicit-GEMM convolution on PRODUCER-SPLIT fp16 operands for gfx950 (round 4: the "H2" engine).
//
// Arithmetic: the three-product fp16 form of conv.hip (MPOSE_CONV_F16X3 in include/margipose_hip.h) -- an fp32 convolution
// as  a*b ~= a_l*b_h + a_h*b_l + a_h*b_h  over operands x * 2^k = h + l (two fp16 values, 22 significant bits), fp32
// accumulation on v_mfma_f32_32x32x16_f16 -- but NOTHING of the split happens here any more:
//   * activations arrive as two fp16 planes (h, l) of x * 2^k in the blocked layout
//         H8[C/8][plane 2][pixel][8]                   (16 bytes per pixel per (channel octet, plane): 4 bytes per element,
//     the size of the fp32 tensor they replace), written ONCE by the elementwise pass that produces the tensor (split.hip:
//     BatchNorm+ReLU, the residual sum, the BatchNorm-backward application) with the scale k that the tensor's amax SLOT
//     prescribes -- a bound on the tensor's largest magnitude that exists BEFORE the producer runs (bn.hip: the coefficient
//     kernels derive it; a loose bound costs no precision, see MPOSE_CONV_H2_IN);
//   * weights: the same two planes, packed [widx][K/16][plane][k half][Npad][8] by pack_weights_k (layout 3).
// conv_igemm_k (conv.hip) spent a third of its non-MFMA instructions scaling and splitting fp32 operands inside the K loop,
// on the one wave per SIMD that also had to issue the MFMAs (MFMA busy 0.25-0.33).  Here both operands go global -> LDS by DMA
// (buffer_load_dwordx4 ... lds: 64 consecutive pixels or output channels x 16 B = 1 KiB per wave instruction, zero padding by
// the buffer unit's range check), the workgroup SHARES its tiles through LDS, TWO workgroups are resident per CU (two waves
// per SIMD, <= 256 registers), and the K loop is fragment reads + MFMAs only.
//
// Accumulation: TWO accumulators per output block -- `acc` takes the h x h products (one rounding of the running sum per
// 16 channels: 72 for a 128-channel 3x3), `acx` the two cross products (2^-11 of the sum: its roundings do not matter).  That
// bounds the fp32 accumulation chain better than conv_igemm_k's two-way split-K (108) without any exchange between waves, so
// training launches run unsplit.
//
// Structure (conv_p.hip's, re-cut for 3 products): 256 threads = 4 waves as WM x WN, wave tile 32*RM pixels x 32*RN channels.
// A ring slot holds KST k-groups (16 input channels each) of one tap: per k-group A [plane][k half][BM pixels][16 B] and
// B [plane][k half][BNL columns][16 B] (every ds_read_b128: 32 consecutive lanes on 32 consecutive 16-byte words, conflict
// free).  ONE s_barrier per slot; the DMA of slot s+NBUF is issued when slot s has been drained into registers; fragments
// of k-group q+1 are read while k-group q multiplies (single register set, refilled as blocks retire).
// Geometry (tap lists, classes), fused shortcut (MODE 1: a second pass into a second output), two-input sum (MODE 2: dX =
// conv_in^T(dC1) + shortcut^T(dSC)), BatchNorm statistics / channel extremes / ReLU mask / consumer BatchNorm-backward sums /
// output amax epilogues: the contracts of include/margipose_hip.h (mpose_conv_operands).
//
// Replaces Conv2d / ConvTranspose2d of reference src/margipose/models/margipose_model.py:33,67-82 and their data-gradients
// inside the columns.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace mpose {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_p;

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

constexpr unsigned kOob = 0xFFFFFFF0u;      // voffset beyond num_records: the buffer unit returns (and the DMA stores) zeros

struct ConvHArgs {
  mpose_conv_geom g;
  mpose_conv_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gw, div_ghw;
  int M;                          // slots per class = B*GH*GW
  int n_mtiles;
  int flags;
  int part_row0, part_rows;       // MPOSE_CONV_STATS_PART: first row this launch writes, rows in the buffers' headers
  unsigned in_slab;               // bytes of one (channel octet, plane) slab of the input: B*IH*IW*16
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff, unsigned soff) {
  // 64 lanes x 16 B: lane l's bytes land at lds_wave_base + 16*l (the LDS address is wave-uniform, carried in M0)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_p)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

__device__ __forceinline__ f32x16 mfma_f16(const f16x8 a, const f16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifndef SCHED_HINTS
#define SCHED_HINTS 1   // sched_group_barrier patterns in conv_h2r_k's K loop (0: hipcc's own order; A/B builds)
#endif
#ifndef CH_EXP
#define CH_EXP 0      // timing experiments (tools/h2_exp.sh; wrong results): 1 no A traffic after the first tap of a chunk, 2 no B traffic,
                      // 4 plain tile order, 8 no DMA at all, 16 no MFMAs, 32 no fragment reads, 64 no epilogue stores, 128 no K loop
#endif

template <int WM, int WN, int RM, int RN, int KST, int NBUF, int MODE, bool DUAL>
__global__ __launch_bounds__(256, 2) void conv_h2_k(ConvHArgs a) {
  static_assert(WM * WN == 4, "four waves");
  static_assert(RM == 1 || RM == 2, "row blocks per wave");
  constexpr bool ACC1 = MODE == 1, SUM2 = MODE == 2;
  constexpr int NPASS = MODE ? 2 : 1;
  constexpr int BM = 32 * RM * WM, BN = 32 * RN * WN, BNL = (BN + 63) / 64 * 64;
  constexpr int SGN = BM / 64, NGN = BNL / 64;                 // 64-pixel / 64-column groups = DMA instructions per (plane, half)
  static_assert(SGN == 1 || SGN == 2 || SGN == 4, "slot groups must divide the wave count");
  constexpr int A_B = 4 * BM * 16, B_B = 4 * BNL * 16, SUB_B = A_B + B_B, BUF_B = KST * SUB_B;   // per k-group: [plane][half][row][16 B]
  constexpr int NA = KST * 4 * SGN, NB = KST * 4 * NGN, TOT = NA + NB;     // DMA instructions per slot (workgroup)
  static_assert(TOT % 4 == 0, "every wave issues the same number of DMA instructions per slot");
  constexpr int LW = TOT / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* sRow = reinterpret_cast<unsigned*>(smem + NBUF * BUF_B);                 // [BM] output row byte offsets
  float* sRed = reinterpret_cast<float*>(smem + NBUF * BUF_B + BM * 4);              // [2 sets][4 waves][32*RN][2]  (or [4 waves][32*RN][4])
  float* sMM = sRed + 2 * 4 * 32 * RN * 2;                                           // [4 waves][32*RN][2] channel extremes of out0

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  const mpose_conv_geom& g = a.g;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed placement, speed only); an XCD gets a CONTIGUOUS run of pixel
  // tiles, so neighbouring tiles share their halo rows (and a class's tiles their weights) in one L2.
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0 && !(CH_EXP & 4)) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
  const int cls = bid / a.n_mtiles;
  const int m0 = (bid - cls * a.n_mtiles) * BM;
  const int n0 = blockIdx.y * BN;
  const mpose_conv_operands& op = a.op[blockIdx.z];
  const int n_taps = g.cls[cls].n_taps;
  // lane t keeps tap t ({dy, dx, widx, acc} in one dword); v_readlane hands it to the scalar unit
  const int lane_tap = lane < MPOSE_MAX_TAPS ? *reinterpret_cast<const int*>(&g.cls[cls].taps[lane < MPOSE_MAX_TAPS ? lane : 0]) : 0;
  auto tap_word = [&](int t) { return __builtin_amdgcn_readlane(lane_tap, t); };

  // ---- this lane's input pixel for the A-tile DMA (slot group sg of the wave, pixel `lane` of the group) ----
  const int sg = wave % SGN;
  unsigned pix_off = 0, row_taps = 0;          // byte offset of the anchor pixel inside a slab; bit t: tap t in bounds
  {
    const unsigned m = (unsigned)(m0 + sg * 64 + lane);
    const bool in_m = (int)m < a.M;
    const unsigned mm = in_m ? m : 0u;
    const unsigned b = fdiv(mm, a.div_ghw);
    const unsigned rem = mm - b * (unsigned)(g.GH * g.GW);
    const unsigned gy = fdiv(rem, a.div_gw);
    const unsigned gx = rem - gy * (unsigned)g.GW;
    const int iy0 = (int)gy * g.in_mul, ix0 = (int)gx * g.in_mul_x;
    pix_off = ((b * (unsigned)g.IH + (unsigned)iy0) * (unsigned)g.IW + (unsigned)ix0) * 16u;
    for (int t = 0; t < n_taps; ++t) {
      const int tp = tap_word(t);
      const int iy = iy0 + (int)(signed char)(tp & 0xff), ix = ix0 + (int)(signed char)((tp >> 8) & 0xff);
      if (in_m && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) row_taps |= 1u << t;
    }
  }
  // ---- output row table: byte offset of output pixel m0 + i, or an offset the buffer unit rejects ----
  const int oyc = g.cls[cls].oy, oxc = g.cls[cls].ox;
  auto fill_rows = [&](int out_ld) {
    if (tid < BM) {
      const unsigned m = (unsigned)(m0 + tid);
      const unsigned mm = (int)m < a.M ? m : 0u;
      const unsigned b = fdiv(mm, a.div_ghw);
      const unsigned rem = mm - b * (unsigned)(g.GH * g.GW);
      const unsigned gy = fdiv(rem, a.div_gw);
      const unsigned gx = rem - gy * (unsigned)g.GW;
      const unsigned pix = (b * (unsigned)g.OH + (gy * g.out_mul + oyc)) * (unsigned)g.OW + (gx * g.out_mul_x + oxc);
      sRow[tid] = (int)m < a.M ? pix * (unsigned)out_ld * 4u : 0xFFFFF000u;
    }
  };

  const int k16_total = g.Cin >> 4;
  const int n_chunks = k16_total / KST;                            // (Cin % (16 * KST) == 0: checked by the launcher)
  const int npad = g.Npad0;                                        // == Npad1 when a second weight set is used
  const unsigned w_plane_b = (unsigned)npad * 16u;                 // bytes of one (plane, half) slab of packed weights
  const __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.in), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SUM2 ? op.in1 : op.in), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.w0), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MODE ? op.w1 : op.w0), 0, 0xFFFFFF00, 0x00020000);

  // scale exponents of this launch's tensors: pass `set` multiplies input ka[set] with weights kw[set]
  const int ka0 = f16_scale_exp(amax_gather(op.in_amax));
  const int ka1 = SUM2 ? f16_scale_exp(amax_gather(op.in1_amax)) : ka0;
  const int kw0 = f16_scale_exp(*op.w0_amax);
  const int kw1 = MODE ? f16_scale_exp(*op.w1_amax) : kw0;
  // MODE 2: the first pass's accumulators are re-expressed in the second pass's units (a power of two: exact) before the second
  // input accumulates on top.  A second input 2^60 times smaller than the first (a shortcut BatchNorm with gamma == 0 gives an
  // all-zero one) would make that factor overflow fp32; its planes are already written with the scale its slot prescribes, so it
  // cannot be scaled less as conv.hip does -- it is DROPPED instead: its whole sum is below 2^-60 of the first input's scale.
  const bool skip1 = SUM2 && (ka1 + kw1) - (ka0 + kw0) > 60;
  f32x16 acc[RM][RN], acx[RM][DUAL ? RN : 1];
  // taps with acc == 0 first, taps with acc == 1 (second weight set) last
  int n_taps0 = 0;
  for (int t = 0; t < n_taps; ++t) n_taps0 += (((tap_word(t) >> 24) & 0xff) == 0) ? 1 : 0;

#pragma unroll 1
  for (int set = 0; set < NPASS; ++set) {
    const int t_lo = set ? n_taps0 : 0;
    const int nt = set ? n_taps - n_taps0 : (MODE ? n_taps0 : n_taps);
    const int n_steps = ((set && skip1) || (CH_EXP & 128)) ? 0 : n_chunks * nt;
    if (!SUM2 || set == 0) {
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[rm][rn][r] = 0.0f;
            if (DUAL) acx[rm][rn % (DUAL ? RN : 1)][r] = 0.0f;
          }
    }
    if (set == 0 || ACC1) {
      const int ld_ = (set && ACC1) ? g.out_ld1 : g.out_ld0;
      fill_rows(ld_ > 0 ? ld_ : ((set && ACC1) ? g.Cout1 : g.Cout0));
    }
    const bool second = set != 0;              // which weight set (MODE 1, 2) / input (MODE 2) this pass reads

    // ---- one slot's DMA: instruction gidx of the workgroup's list (A first, then B) goes to wave gidx % 4 ----
    auto issue = [&](int buf, int c, int t) {
      const int tp = tap_word(t_lo + t);
      const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
      const int widx = (tp >> 16) & 0xff;
      unsigned voff_a = ((row_taps >> (t_lo + t)) & 1u) ? pix_off + (unsigned)((dy * g.IW + dx) * 16) : kOob;
      if ((CH_EXP & 1) && t != 0) voff_a = kOob;
      unsigned char* bufp = smem + buf * BUF_B;
#pragma unroll
      for (int k = 0; k < ((CH_EXP & 8) ? 0 : LW); ++k) {
        const int gidx = wave + 4 * k;                            // (wave-uniform: the branches below are scalar)
        if (gidx < NA) {
          const int kp = gidx / SGN;                               // kk * 4 + plane * 2 + half
          const int kk = kp >> 2, ph = kp & 3;
          const unsigned soff = (unsigned)((((c * KST + kk) * 2 + (ph & 1)) * 2) + (ph >> 1)) * a.in_slab;
          unsigned char* dst = bufp + kk * SUB_B + (ph * BM + sg * 64) * 16;
          if (SUM2 && second) dma16(rs_in1, dst, voff_a, soff);
          else dma16(rs_in0, dst, voff_a, soff);
        } else {
          const int j = gidx - NA;
          const int kp = j / NGN, ng = j - kp * NGN;
          const int kk = kp >> 2, ph = kp & 3;
          const unsigned voff_b = (CH_EXP & 2) ? kOob : (unsigned)((n0 + ng * 64 + lane) * 16);
          const unsigned soff = ((unsigned)(widx * k16_total + c * KST + kk) * 4u + (unsigned)ph) * w_plane_b;
          unsigned char* dst = bufp + kk * SUB_B + A_B + (ph * BNL + ng * 64) * 16;
          if (second) dma16(rs_w1, dst, voff_b, soff);
          else dma16(rs_w0, dst, voff_b, soff);
        }
      }
    };

    // issue cursor: (chunk, tap) of the next slot to fetch, tap fastest (the taps of a chunk re-read the same pixels)
    int ic = 0, itp = 0, issued = 0;
    auto issue_next = [&]() {
      issue(issued % NBUF, ic, itp);
      ++issued;
      if (++itp == nt) { itp = 0; ++ic; }
    };
    auto wait_slots_outstanding = [&](int k) {         // at most k slots' worth of this wave's DMA still in flight
      if (NBUF > 2 && k == NBUF - 2) wait_vmcnt<(NBUF - 2) * LW>();
      else if (NBUF > 2 && k == NBUF - 1) wait_vmcnt<(NBUF - 1) * LW>();
      else wait_vmcnt<0>();
    };
    f16x8 af[RM][2], bfr[RN][2], afn[2];
    auto read_a = [&](int slot, int kk, int rm, f16x8 (&dst)[2]) {
      if constexpr (CH_EXP & 32) { asm volatile("" : "+v"(dst[0]), "+v"(dst[1])); return; }
      const unsigned char* pa = smem + slot * BUF_B + kk * SUB_B + (lh * BM + wm * 32 * RM + rm * 32 + li) * 16;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) dst[pl] = *reinterpret_cast<const f16x8*>(pa + pl * 2 * BM * 16);
    };
    auto read_b = [&](int slot, int kk, int rn, f16x8 (&dst)[2]) {
      if constexpr (CH_EXP & 32) { asm volatile("" : "+v"(dst[0]), "+v"(dst[1])); return; }
      const unsigned char* pb = smem + slot * BUF_B + kk * SUB_B + A_B + (lh * BNL + wn * 32 * RN + rn * 32 + li) * 16;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) dst[pl] = *reinterpret_cast<const f16x8*>(pb + pl * 2 * BNL * 16);
    };
    auto block = [&](int rm, int rn) {
      if constexpr (CH_EXP & 16) {
        asm volatile("" :: "v"(af[rm][0]), "v"(af[rm][1]), "v"(bfr[rn][0]), "v"(bfr[rn][1]));
      } else if constexpr (DUAL) {
        f32x16 c = acx[rm][rn];
        c = mfma_f16(af[rm][1], bfr[rn][0], c);
        c = mfma_f16(af[rm][0], bfr[rn][1], c);
        acx[rm][rn] = c;
        acc[rm][rn] = mfma_f16(af[rm][0], bfr[rn][0], acc[rm][rn]);
      } else {
        f32x16 c = acc[rm][rn];
        c = mfma_f16(af[rm][1], bfr[rn][0], c);
        c = mfma_f16(af[rm][0], bfr[rn][1], c);
        c = mfma_f16(af[rm][0], bfr[rn][0], c);
        acc[rm][rn] = c;
      }
    };
    // the slot's one barrier: everyone has drained slot s into registers, slot s+1 has landed for everyone; refill slot s
    auto sync = [&](int s_, bool more) {
      if (more) wait_slots_outstanding(issued - s_ - 2);
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) -- the builtin, so that hipcc's own wait-count bookkeeping sees it
      __builtin_amdgcn_s_barrier();
      if (issued < n_steps) issue_next();
    };
    auto group0 = [&]() {
      if constexpr (RM == 2) {
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) block(0, rn);
      } else if constexpr (RN > 1) {
        block(0, 0);
      }
    };
    for (int p = 0; p < NBUF && p < n_steps; ++p) issue_next();
    if (n_steps > 0) {
      wait_slots_outstanding(issued - 1);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int rm = 0; rm < RM; ++rm) read_a(0, 0, rm, af[rm]);
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) read_b(0, 0, rn, bfr[rn]);
      group0();
    }
    // Rotated loop over k-groups q = (slot s, kk): [sync(s) before the last k-group's second half], second half of q, first half
    // of q+1 -- every prefetched fragment is consumed inside the iteration that fetched it (exact lgkmcnt counts from hipcc).
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
      const bool more = s + 1 < n_steps;
      const int slot = s % NBUF, nslot = (s + 1) % NBUF;
#pragma unroll
      for (int kk = 0; kk < KST; ++kk) {
        const bool last = kk == KST - 1;
        if (last) sync(s, more);
        // (the prefetches are unconditional -- after the last slot they read a stale slot and the values are dropped -- so that
        //  the iteration is one basic block)
        const int rs = last ? nslot : slot, rk = last ? 0 : kk + 1;
        if constexpr (RM == 2) {
          read_a(rs, rk, 0, af[0]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rn = 0; rn < RN; ++rn) {
            block(1, rn);
            __builtin_amdgcn_sched_barrier(0);
            read_b(rs, rk, rn, bfr[rn]);
            if (rn == RN - 1) read_a(rs, rk, 1, af[1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          static_assert(RM == 2 || RN > 1, "single-block wave tiles are not instantiated");
          read_b(rs, rk, 0, bfr[0]);
          read_a(rs, rk, 0, afn);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rn = 1; rn < RN; ++rn) {
            block(0, rn);
            __builtin_amdgcn_sched_barrier(0);
            read_b(rs, rk, rn, bfr[rn]);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) af[0][pl] = afn[pl];
        }
        if (!last || more) group0();
      }
    }
    __builtin_amdgcn_s_barrier();              // the ring may be refilled by the next pass; sRow is complete

    if (SUM2 && set == 0) {
      // second input accumulates on top: re-express this pass's sums in the second pass's units (exact: powers of two)
      const int shift = skip1 ? 0 : (ka1 + kw1) - (ka0 + kw0);
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[rm][rn][r] = __builtin_ldexpf(acc[rm][rn][r], shift);
            if (DUAL) acx[rm][rn % (DUAL ? RN : 1)][r] = __builtin_ldexpf(acx[rm][rn % (DUAL ? RN : 1)][r], shift);
          }
      continue;
    }

    // ---- epilogue (branch-free: rows beyond M carry an offset the buffer unit rejects) ----
    const int k_back = (SUM2 && skip1) ? -(ka0 + kw0) : -((set ? ka1 : ka0) + (set ? kw1 : kw0));
    const int oset = SUM2 ? 0 : set;
    float* outp = oset ? op.out1 : op.out0;
    const int cout = oset ? g.Cout1 : g.Cout0;
    double* stats = oset ? op.stats1 : op.stats0;
    const bool masked = (oset == 0) && op.mask_src != nullptr;
    const bool red = (oset == 0) && op.red_sums != nullptr;
    const bool want_mm = (oset == 0) && op.mm0 != nullptr;
    const bool want_amax = (oset == 0) && op.out0_amax != nullptr;
    const int old_ = oset ? g.out_ld1 : g.out_ld0;
    const int out_ld = old_ > 0 ? old_ : cout;
    const unsigned out_bytes = (unsigned)((((long)g.B * g.OH * g.OW - 1) * out_ld + cout) * 4);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(outp, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? op.mask_src : outp), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(red ? op.red_a : outp), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(red ? op.red_b : outp), 0, out_bytes, 0x00020000);
    float csum[RN], csq[RN], rs0[RN], rs1[RN], rs2[RN], rs3[RN], vmx[RN], vng[RN];
    const float kNegInf = __uint_as_float(0xff800000u);
    float out_amax = 0.f;
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) { csum[rn] = csq[rn] = rs0[rn] = rs1[rn] = rs2[rn] = rs3[rn] = 0.f; vmx[rn] = vng[rn] = kNegInf; }
    const int ncol0 = n0 + wn * 32 * RN;
    const unsigned col_off = (unsigned)((ncol0 + li) * 4);
#pragma unroll
    for (int rm = 0; rm < ((CH_EXP & 64) ? 0 : RM); ++rm) {
      unsigned voff[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const u32x4 e = *reinterpret_cast<const u32x4*>(sRow + wm * 32 * RM + rm * 32 + 8 * rg + 4 * lh);
        voff[4 * rg] = e.x + col_off; voff[4 * rg + 1] = e.y + col_off; voff[4 * rg + 2] = e.z + col_off; voff[4 * rg + 3] = e.w + col_off;
      }
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const int nb = ncol0 + rn * 32;              // wave-uniform: a 32-column group is in or out as a whole (cout % 32 == 0)
        if (nb < cout) {
          const int n = nb + li;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            v[r] = __builtin_ldexpf(DUAL ? acc[rm][rn][r] + acx[rm][rn % (DUAL ? RN : 1)][r] : acc[rm][rn][r], k_back);
          if (masked) {
            const float msc = op.mask_scale[n], msh = op.mask_shift[n];
            float src[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              src[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_m, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (!(fmaf(src[r], msc, msh) > 0.f)) v[r] = 0.f;
              csq[rn] = fmaf(v[r], src[r], csq[rn]);
            }
          }
          if (red) {
            const float ms = op.red_scale[n], mt = op.red_shift[n];
            float xa[16], xb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              xa[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_ra, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0));
              xb[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_rb, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0));
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
            csum[rn] += v[r];                        // rows beyond M accumulated zeros (their inputs were read as 0)
            if (!masked) csq[rn] = fmaf(v[r], v[r], csq[rn]);
          }
          if (want_mm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool ok = voff[r] < 0xFFFFF000u;
              vmx[rn] = fmaxf(vmx[rn], ok ? v[r] : kNegInf);
              vng[rn] = fmaxf(vng[rn], ok ? -v[r] : kNegInf);
            }
          }
          if (want_amax) {
#pragma unroll
            for (int r = 0; r < 16; ++r) out_amax = fmaxf(out_amax, fabsf(v[r]));     // (rows beyond M hold zeros)
          }
        }
      }
    }
    if (want_mm) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float a_ = fmaxf(vmx[rn], __shfl_xor(vmx[rn], 32, 64)), b_ = fmaxf(vng[rn], __shfl_xor(vng[rn], 32, 64));
        if (lh == 0) { sMM[(wave * 32 * RN + rn * 32 + li) * 2] = a_; sMM[(wave * 32 * RN + rn * 32 + li) * 2 + 1] = b_; }
      }
    }
    if (want_amax) {       // one look-then-atomic per wave into the workgroup's sub-slot (common.h)
      float m = wave_max(out_amax);
      if (lane == 0) {
        if (!(m == m)) m = __uint_as_float(0x7f800000u);
        unsigned* dst = reinterpret_cast<unsigned*>(op.out0_amax + (blockIdx.x % MPOSE_AMAX_SUBSLOTS) * MPOSE_AMAX_STRIDE);
        if (__float_as_uint(m) > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, __float_as_uint(m));
      }
    }
    if (red) {       // (sRed is free: red_* and stats* exclude each other) [4 waves][32*RN][4]
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float t0 = rs0[rn] + __shfl_xor(rs0[rn], 32, 64), t1 = rs1[rn] + __shfl_xor(rs1[rn], 32, 64);
        const float t2 = rs2[rn] + __shfl_xor(rs2[rn], 32, 64), t3 = rs3[rn] + __shfl_xor(rs3[rn], 32, 64);
        if (lh == 0) *reinterpret_cast<float4*>(sRed + (wave * 32 * RN + rn * 32 + li) * 4) = make_float4(t0, t1, t2, t3);
      }
    }
    if (stats != nullptr) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float s_ = csum[rn] + __shfl_xor(csum[rn], 32, 64);
        const float q_ = csq[rn] + __shfl_xor(csq[rn], 32, 64);
        if (lh == 0) {
          float* d = sRed + ((oset * 4 + wave) * 32 * RN + rn * 32 + li) * 2;
          d[0] = s_; d[1] = q_;
        }
      }
    }
    if (ACC1 && set == 0) __builtin_amdgcn_s_barrier();      // sRow is rewritten for the second output
  }
  __syncthreads();
  // cross-wave sums: column `tid` of the workgroup's BN output columns is held by the WM waves (wn_, 0..WM-1)
  const int wn_ = tid / (32 * RN), col = tid - wn_ * 32 * RN;
  const bool part = (a.flags & MPOSE_CONV_STATS_PART) != 0;      // plain stores into row blockIdx.x of fp32 partial buffers
  const int prow = a.part_row0 + (int)blockIdx.x;
  const bool hdr_writer = prow == 0 && blockIdx.y == 0;
#pragma unroll
  for (int set = 0; set < (ACC1 ? 2 : 1); ++set) {
    double* stats = set ? op.stats1 : op.stats0;
    const int cout = set ? g.Cout1 : g.Cout0;
    if (stats != nullptr && tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        const float* d = sRed + ((set * 4 + (wn_ * WM + w)) * 32 * RN + col) * 2;
        s += d[0]; q += d[1];
      }
      const int n = n0 + tid;
      if (n < cout) {
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
    float a_ = __uint_as_float(0xff800000u), b_ = a_;
#pragma unroll
    for (int w = 0; w < WM; ++w) {
      a_ = fmaxf(a_, sMM[((wn_ * WM + w) * 32 * RN + col) * 2]);
      b_ = fmaxf(b_, sMM[((wn_ * WM + w) * 32 * RN + col) * 2 + 1]);
    }
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
    for (int w = 0; w < WM; ++w) {
      const float4 d = *reinterpret_cast<const float4*>(sRed + ((wn_ * WM + w) * 32 * RN + col) * 4);
      t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
    }
    if (part) {
      float* pb = reinterpret_cast<float*>(op.red_sums);
      if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
      reinterpret_cast<float4*>(pb + kPartHdr)[(size_t)prow * g.Cout0 + n0 + tid] = t;
    } else {
      double* d = op.red_sums + (size_t)(n0 + tid) * 4;
      atomicAdd(d, (double)t.x); atomicAdd(d + 1, (double)t.y); atomicAdd(d + 2, (double)t.z); atomicAdd(d + 3, (double)t.w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_h2r_k: the stride-1 form (one class, output grid = input grid: every 3x3 / 1x1 of a regular ResidualBlock and their
// data-gradients).  What bounded conv_h2_k above is the LDS-DMA path itself: ~26 cycles per KiB and CU whatever the source
// (measured: the same with every DMA out of range, i.e. without any L2 traffic), and a 128 x 128 tile that re-stages its A tile
// for each of the nine taps moves 32 KiB per 96 MFMAs.  Here
//   * A is staged ONCE per 16-channel chunk as a halo tile: the BM + (largest - smallest tap shift) consecutive pixels
//     m0 + lo .. of both planes; a tap is an address shift of the fragment read, and a lane whose (pixel, tap) falls outside
//     the image reads one of sixteen zero rows (the one of its own bank class: the read stays conflict-free) -- A traffic / 6;
//   * the tile is tall and narrow, 256 pixels x 64 channels (4 waves stacked along the pixels, 64 x 64 each): all four waves
//     share one B tile, half the B traffic per MFMA of a 128 x 128 tile; the 192-channel layers get 3 column tiles, 288 workgroups;
//   * a step is one chunk x up to three taps (36 MFMAs per wave and barrier); B tiles of the next-but-one step and a share of the
//     next chunk's A tile are issued at each barrier (one batch in flight: plain vmcnt(0)).
// 9.2 MFMAs per KiB of DMA instead of 3.  Two workgroups per CU; two accumulators per block as above.
// ---------------------------------------------------------------------------------------------------------------------
struct ConvHRArgs {
  mpose_conv_geom g;
  mpose_conv_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gw, div_ghw;
  int M;
  int n_mtiles;
  int flags;
  int part_row0, part_rows;
  unsigned in_slab;               // bytes of one (channel octet, plane) slab of the input: B*IH*IW*16
  int lo;                         // smallest tap shift dy*IW + dx over ALL taps of the launch (pixels)
  int NG;                         // 64-row groups of one staged A plane-half: ceil((256 + largest - smallest shift) / 64)
  int a_rows;                     // rows of one (plane, half) sub-array of an A buffer: NG * 64 + 16 zero rows
};

constexpr int HR_BM = 256;
constexpr int HR_TG = 3;          // taps per step
constexpr int HR_MAXT = 9;        // taps per pass
constexpr int HR_MAXNG = 6;       // 64-row groups of an A plane-half: 256 + (largest - smallest tap shift) <= 384 pixels

// LDS layout of conv_h2r_k: two A buffers (a halo tile's (plane, k half) sub-arrays), the B ring of two slots x three taps, the
// epilogue scratch (cross-wave sums: it aliases the rings once the K loop is done; behind the parked result tiles of MODE 2) and
// the output-row table at the end.
template <int RN, int MODE, bool X1>
struct H2rLds {
  static constexpr int BN = 32 * RN, BNL = (BN + 63) / 64 * 64, NPLN = X1 ? 1 : 2;
  static constexpr int SA = (HR_MAXNG * 64 + 16) * 16, ABUF = 2 * NPLN * SA, B_SLOT = HR_TG * 2 * NPLN * BNL * 16;
  static constexpr int RING = 2 * ABUF + 2 * B_SLOT;
  static constexpr int EP_TILE = MODE == 2 ? 4 * 32 * 72 * 4 : 0;
  static constexpr int RED_OFF = 2 * ABUF > EP_TILE ? 2 * ABUF : EP_TILE;
  static constexpr int RED_END = RED_OFF + 4 * BN * 4 * 4 + 4 * BN * 2 * 4;
  static constexpr int ROW_OFF = RING > RED_END ? RING : RED_END;
  static constexpr int BYTES = ROW_OFF + HR_BM * 4;
};

// X1 (round 6): MPOSE_CONV_F16X1's arithmetic on the same planes -- the h pieces ARE the operands rounded to fp16, so the kernel
// leaves the l planes where they are: half the DMA instructions and fragment reads, one product per multiply-add, one accumulator.
template <int RN, int MODE, bool X1 = false>
__global__ __launch_bounds__(256, 2) void conv_h2r_k(ConvHRArgs a) {
  constexpr bool ACC1 = MODE == 1, SUM2 = MODE == 2;
  constexpr int NPASS = MODE ? 2 : 1;
  constexpr int BM = HR_BM, BN = 32 * RN, BNL = (BN + 63) / 64 * 64, NGN = BNL / 64;
  constexpr int TG = HR_TG;
  constexpr int NPLN = X1 ? 1 : 2;             // operand planes the kernel reads
  constexpr int B_TAP = 2 * NPLN * BNL * 16;   // one tap's B tile: [plane][k half][BNL columns][16 B]
  constexpr int B_SLOT = TG * B_TAP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SA = (HR_MAXNG * 64 + 16) * 16;        // bytes of one (plane, half) sub-array: compile-time, so that planes are immediate offsets
  using L = H2rLds<RN, MODE, X1>;
  constexpr int ABUF = 2 * NPLN * SA;
  static_assert(ABUF == L::ABUF && B_SLOT == L::B_SLOT, "one LDS layout for the kernel and its launcher");
  unsigned char* sB = smem + 2 * ABUF;                                         // [2 slots][B_SLOT]
  unsigned* sRow = reinterpret_cast<unsigned*>(smem + L::ROW_OFF);             // [BM] output row byte offsets
  float* sRed = reinterpret_cast<float*>(smem + L::RED_OFF);                   // epilogue scratch, aliases the (then idle) rings:
  float* sMM = sRed + 4 * BN * 4;                                              //   [4 waves][BN][4] and [4 waves][BN][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const mpose_conv_geom& g = a.g;
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0 && !(CH_EXP & 4)) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
  const int m0 = bid * BM;
  const int n0 = blockIdx.y * BN;
  const mpose_conv_operands& op = a.op[blockIdx.z];
  const int n_taps = g.cls[0].n_taps;
  const int IW = g.IW;
  // lane t keeps tap t ({dy, dx, widx, acc}) and its row shift inside the staged tile, (dy*IW + dx) - lo
  const int lane_tap = lane < MPOSE_MAX_TAPS ? *reinterpret_cast<const int*>(&g.cls[0].taps[lane < MPOSE_MAX_TAPS ? lane : 0]) : 0;
  const int lane_shift = (int)(signed char)(lane_tap & 0xff) * IW + (int)(signed char)((lane_tap >> 8) & 0xff) - a.lo;
  auto tap_word = [&](int t) { return __builtin_amdgcn_readlane(lane_tap, t); };
  auto tap_shift = [&](int t) { return __builtin_amdgcn_readlane(lane_shift, t); };

  // ---- per-lane state of the fragment reads: tile row of output pixel (wave, rm, li) and which taps stay inside the image ----
  unsigned fr_ok[2] = {0u, 0u};
#pragma unroll
  for (int rm = 0; rm < 2; ++rm) {
    const unsigned m = (unsigned)(m0 + wave * 64 + rm * 32 + li);
    if ((int)m < a.M) {
      const unsigned b = fdiv(m, a.div_ghw);
      const unsigned rem = m - b * (unsigned)(g.GH * g.GW);
      const int gy = (int)fdiv(rem, a.div_gw);
      const int gx = (int)rem - gy * g.GW;
      for (int t = 0; t < n_taps; ++t) {
        const int tp = tap_word(t);
        const int iy = gy + (int)(signed char)(tp & 0xff), ix = gx + (int)(signed char)((tp >> 8) & 0xff);
        if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) fr_ok[rm] |= 1u << t;
      }
    }
  }
  constexpr int zrow = HR_MAXNG * 64;          // first zero row of a sub-array
  // ---- output row table ----
  auto fill_rows = [&](int out_ld) {
    const unsigned m = (unsigned)(m0 + tid);
    sRow[tid] = (int)m < a.M ? m * (unsigned)out_ld * 4u : 0xFFFFF000u;      // (output grid = slot grid: pixel index = m)
  };
  // the zero rows of both A buffers (never written by the DMA: it covers rows 0 .. NG*64-1)
  for (int i = tid; i < 2 * 2 * NPLN * 16; i += 256) {
    const int sub = i >> 4, r = i & 15;
    *reinterpret_cast<u32x4*>(smem + sub * SA + (zrow + r) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  const int k16_total = g.Cin >> 4;
  const int npad = g.Npad0;
  const unsigned w_plane_b = (unsigned)npad * 16u;
  const long npix = (long)g.B * g.IH * g.IW;
  const __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.in), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SUM2 ? op.in1 : op.in), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.w0), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MODE ? op.w1 : op.w0), 0, 0xFFFFFF00, 0x00020000);

  const int ka0 = f16_scale_exp(amax_gather(op.in_amax));
  const int ka1 = SUM2 ? f16_scale_exp(amax_gather(op.in1_amax)) : ka0;
  const int kw0 = f16_scale_exp(*op.w0_amax);
  const int kw1 = MODE ? f16_scale_exp(*op.w1_amax) : kw0;
  const bool skip1 = SUM2 && (ka1 + kw1) - (ka0 + kw0) > 60;      // (see conv_h2_k)

  f32x16 acc[2][RN], acx[2][RN];
  int n_taps0 = 0;
  for (int t = 0; t < n_taps; ++t) n_taps0 += (((tap_word(t) >> 24) & 0xff) == 0) ? 1 : 0;
  const bool part = (a.flags & MPOSE_CONV_STATS_PART) != 0;
  const int prow = a.part_row0 + (int)blockIdx.x;
  const bool hdr_writer = prow == 0 && blockIdx.y == 0;

  // (a pass is a lambda so that the two-input form can run its two passes as two straight-line copies: as iterations of one
  //  loop the accumulators are loop-carried through both K-loop variants, and hipcc then needed 95 registers more than MODE 0)
  auto one_pass = [&](const int set, auto form_c) __attribute__((always_inline)) {
    constexpr int FORM = decltype(form_c)::value;      // which K loop a pass may need: 0 either (decided at run time), 1 the three-tap steps, 2 the plain loop
    const int t_lo = set ? n_taps0 : 0;
    const int nt = set ? n_taps - n_taps0 : (MODE ? n_taps0 : n_taps);
    const int ng = (nt + TG - 1) / TG;                               // steps per chunk
    const int n_steps = ((set && skip1) || (CH_EXP & 128)) ? 0 : k16_total * ng;
    const bool second = set != 0;
    if (!SUM2 || set == 0) {
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[rm][rn][r] = 0.0f; acx[rm][rn][r] = 0.0f; }
    }
    if (set == 0 || ACC1) {
      const int ld_ = (set && ACC1) ? g.out_ld1 : g.out_ld0;
      fill_rows(ld_ > 0 ? ld_ : ((set && ACC1) ? g.Cout1 : g.Cout0));
    }

    // ---- loop-invariant addresses of this pass ----
    const int ngw = X1 ? (2 * a.NG + 3) / 4 : a.NG;                 // A instructions per wave and halo tile (X1: the h plane's two sub-arrays only)
    const int la = (ngw + ng - 1) / ng;                             // A instructions per wave and batch
    const unsigned bvoff0 = (CH_EXP & 2) ? kOob : (unsigned)((n0 + lane) * 16);
    const int rowb0 = wave * 64 + li, rowb1 = rowb0 + 32;           // tile rows of this lane's two output pixels (before the tap shift)
    // An LDS-DMA instruction costs its wave ~25 cycles when its operands are ready and ~180 when they are computed in line (scalar
    // divisions, 64-bit compares: measured, 10 us of a 35 us workgroup), so everything that does not change from batch to batch
    // lives in lane tables, handed to the scalar unit by v_readlane:
    //   lane t (a tap of this pass):   byte offset of the tap's packed weights (chunk 0, plane 0, half 0)
    //   lane k (this wave's k-th instruction of a halo tile, id = wave + 4k -> sub-array id / NG, row group id % NG):
    //                                  LDS offset inside an A buffer, slab offset of (plane, half) inside a channel octet pair, first row
    const unsigned tab_b = (unsigned)(((lane_tap >> 16) & 0xff) * k16_total * 4) * w_plane_b;
    unsigned tab_adst, tab_asoff;
    int tab_arow;
    {
      const int id = wave + 4 * (lane < HR_MAXNG ? lane : 0);
      const int sub = id / a.NG, grp = id - sub * a.NG;
      tab_adst = (unsigned)(sub * SA + grp * 1024);
      tab_asoff = (unsigned)((sub & 1) * 2 + (sub >> 1)) * a.in_slab;
      tab_arow = grp * 64;
    }
    const unsigned q0 = (unsigned)(m0 + a.lo + lane);               // pixel staged by this lane in row group 0 (wraps below zero: out of range)
    const unsigned b_chunk = 4u * w_plane_b, a_chunk = 4u * a.in_slab;
    const unsigned lds_b0 = (unsigned)(2 * ABUF) + (unsigned)(wave * BNL * 16);

    // One DMA batch = the B tiles of step (cb, gb) into ring slot `bslot` + share `ja` of chunk ca's halo tile into A buffer ca & 1
    // (b_live / a_live, wave-uniform: the step / chunk exists; otherwise the sources are out of range and zeros land where nobody reads).
    auto issue_batch = [&](int cb, int gb, int bslot, bool b_live, int ca, int ja, bool a_live) {
      if constexpr (CH_EXP & 8) return;
      static_assert(NGN == 1, "one B instruction per wave and tap");
#pragma unroll
      for (int ti = 0; ti < TG; ++ti) {
        const int t = gb * TG + ti;
        if (t < nt && (!X1 || wave < 2)) {                         // (wave-uniform; always true for the 3x3 passes.  X1: the h plane's two k halves)
          const unsigned soff = __builtin_amdgcn_readlane(tab_b, t_lo + t) + (unsigned)cb * b_chunk + (unsigned)wave * w_plane_b;
          unsigned char* dst = smem + lds_b0 + bslot * B_SLOT + ti * B_TAP;
          if constexpr (CH_EXP & 512) {       // timing experiment: the B tile by a plain load (no LDS-DMA, results wrong)
            const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs_w0, (int)bvoff0, (int)soff, 0);
            asm volatile("" :: "v"(r));
          } else if (second) dma16(rs_w1, dst, b_live ? bvoff0 : kOob, b_live ? soff : 0u);
          else dma16(rs_w0, dst, b_live ? bvoff0 : kOob, b_live ? soff : 0u);
        }
      }
      const int k_hi = (ja + 1) * la < ngw ? (ja + 1) * la : ngw;
      for (int k = ja * la; k < k_hi; ++k) {
        if (X1 && wave + 4 * k >= 2 * a.NG) continue;      // (an odd NG: the last ids of the list are the l plane's sub-arrays, which X1 neither reads nor has room for)
        const unsigned q = q0 + (unsigned)__builtin_amdgcn_readlane(tab_arow, k);
        const unsigned vo = (a_live && q < (unsigned)npix && !(CH_EXP & 1)) ? q * 16u : kOob;
        const unsigned soff = a_live ? __builtin_amdgcn_readlane(tab_asoff, k) + (unsigned)ca * a_chunk : 0u;
        unsigned char* dst = smem + (ca & 1) * ABUF + __builtin_amdgcn_readlane(tab_adst, k);
        if (SUM2 && second) dma16(rs_in1, dst, vo, soff);
        else dma16(rs_in0, dst, vo, soff);
      }
    };

    f16x8 fa[2][2][2], fb[2][RN][2];             // [register set][row block | column block][plane]
    // fragments of tap t of the pass (B tile ti of ring slot bslot; A buffer abuf) into register set ST.  A lane whose tap leaves
    // the image reads the zero row of its own bank class.
    auto read_frags = [&](auto ST, int t, int ti, int bslot, int abuf) {
      constexpr int st = decltype(ST)::value;
      if constexpr (CH_EXP & 32) {
        asm volatile("" : "+v"(fa[st][0][0]), "+v"(fa[st][0][1]), "+v"(fa[st][1][0]), "+v"(fa[st][1][1]));
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) asm volatile("" : "+v"(fb[st][rn][0]), "+v"(fb[st][rn][1]));
        return;
      }
      const int sh = tap_shift(t_lo + t);
      const unsigned char* pa = smem + abuf * ABUF + lh * SA;
#pragma unroll
      for (int rm = 0; rm < 2; ++rm) {
        const int row = (rm ? rowb1 : rowb0) + sh;
        const bool ok = (fr_ok[rm] >> (t_lo + t)) & 1u;
        const int r = ok ? row : zrow + (row & 15);
        fa[st][rm][0] = *reinterpret_cast<const f16x8*>(pa + r * 16);
        if constexpr (!X1) fa[st][rm][1] = *reinterpret_cast<const f16x8*>(pa + r * 16 + 2 * SA);
      }
      const unsigned char* pb = sB + bslot * B_SLOT + ti * B_TAP + (lh * BNL + li) * 16;
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        fb[st][rn][0] = *reinterpret_cast<const f16x8*>(pb + rn * 32 * 16);
        if constexpr (!X1) fb[st][rn][1] = *reinterpret_cast<const f16x8*>(pb + rn * 32 * 16 + 2 * BNL * 16);
      }
    };
    auto mfma_tap = [&](auto ST) {                // products outermost: no two consecutive MFMAs on one accumulator
      constexpr int st = decltype(ST)::value;
      if constexpr (CH_EXP & 16) {
        asm volatile("" :: "v"(fa[st][0][0]), "v"(fa[st][0][1]), "v"(fa[st][1][0]), "v"(fa[st][1][1]), "v"(fb[st][0][0]), "v"(fb[st][0][1]));
        return;
      }
      if constexpr (!X1) {
#pragma unroll
        for (int rm = 0; rm < 2; ++rm)
#pragma unroll
          for (int rn = 0; rn < RN; ++rn) acx[rm][rn] = mfma_f16(fa[st][rm][1], fb[st][rn][0], acx[rm][rn]);
#pragma unroll
        for (int rm = 0; rm < 2; ++rm)
#pragma unroll
          for (int rn = 0; rn < RN; ++rn) acx[rm][rn] = mfma_f16(fa[st][rm][0], fb[st][rn][1], acx[rm][rn]);
      }
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) acc[rm][rn] = mfma_f16(fa[st][rm][0], fb[st][rn][0], acc[rm][rn]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // the batch that rides on step (c, g): B of step s+2, share (g + 1) % ng of the halo tile of the chunk after step s+1's
    auto batch_of = [&](int c, int g_, int s_) {
      int c2 = c, g2 = g_ + 2;
      if (g2 >= ng) { g2 -= ng; ++c2; }
      if (g2 >= ng) { g2 -= ng; ++c2; }
      const int nc = g_ + 1 < ng ? c : c + 1;
      const int ja = g_ + 1 < ng ? g_ + 1 : 0;
      issue_batch(c2, g2, s_ & 1, c2 < k16_total, nc + 1, ja, nc + 1 < k16_total);
    };

    if (n_steps > 0) {
      // prologue: the whole first halo tile, the first two steps' B tiles, and what the batch "before step 0" would have carried
      __builtin_amdgcn_s_barrier();              // (the previous pass's epilogue scratch / the zero rows are settled)
      for (int j = 0; j < ng; ++j) issue_batch(0, ng, 0, false, 0, j, true);       // A(0), whole (tap group ng: no B instruction)
      issue_batch(0, 0, 0, true, 1, 0, 1 < k16_total);                             // B(0), share 0 of A(1)
      issue_batch(ng > 1 ? 0 : 1, ng > 1 ? 1 : 0, 1, n_steps > 1, 0, ng, false); // B(1) (share index ng: no A instruction)
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (FORM == 1 || (FORM == 0 && nt == TG * ng)) {
        // ---- full steps of three taps (every 3x3).  The first tap's fragments of a step already sit in a register set:
        //   tap 0, 1:  MFMAs  ||  fragment reads of the next tap
        //   then everything of the step is in registers -> wait for the batch issued one step ago, s_barrier
        //   tap 2:     MFMAs  ||  the next DMA batch (B two steps ahead into the slot just drained, an A share into the buffer
        //                         the previous chunk has left)  ||  fragment reads of the NEXT step's first tap
        // so that every wait has a tap's MFMAs in flight.  Three taps flip the register-set parity: two steps per iteration.
        read_frags(I0{}, 0, 0, 0, 0);
        int c = 0, g_ = 0;
        auto advance = [&]() { if (++g_ == ng) { g_ = 0; ++c; } };
        // (CH_EXP & 256: s_memtime stamps of the even steps of workgroup 0 / wave 0 into the out1 buffer: tools/h2_stamps.py)
#if (CH_EXP & 256)
#define STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0 && lane == 0 && s_ < 48) { \
          __builtin_amdgcn_sched_barrier(0); reinterpret_cast<unsigned long long*>(op.out1)[(s_ / 2) * 8 + (i)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
        // issue order inside a tap's block: every MFMA gap takes one fragment read (and its address arithmetic) or one DMA
        // instruction of the batch; without the hints hipcc bunches the reads in front of the MFMAs and the DMA behind them
        auto spread_reads = [&]() {
          if constexpr (SCHED_HINTS && !X1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * RN - 4 > 0 ? 6 * RN - 8 : 0, 0);
          }
        };
        auto spread_batch = [&]() {
          if constexpr (SCHED_HINTS && !X1) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x004, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 6 * RN - 5; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
          }
        };
#pragma unroll 1
        for (int s_ = 0; s_ < n_steps; s_ += 2) {
          {   // step s_: sets 0, 1, 0; leaves the next step's first fragments in set 1
            const int t0 = g_ * TG;
            STAMP(0);
            read_frags(I1{}, t0 + 1, 1, s_ & 1, c & 1);
            mfma_tap(I0{});
            spread_reads();
            STAMP(1);
            read_frags(I0{}, t0 + 2, 2, s_ & 1, c & 1);
            mfma_tap(I1{});
            spread_reads();
            STAMP(2);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            STAMP(3);
            __builtin_amdgcn_s_barrier();
            STAMP(4);
            batch_of(c, g_, s_);
            advance();
            read_frags(I1{}, g_ * TG, 0, (s_ + 1) & 1, c & 1);      // (after the last step: stale data, never used)
            mfma_tap(I0{});
            spread_batch();
            STAMP(5);
          }
          if (s_ + 1 < n_steps) {   // step s_+1: sets 1, 0, 1; leaves the next step's first fragments in set 0
            const int t0 = g_ * TG, s1 = s_ + 1;
            read_frags(I0{}, t0 + 1, 1, s1 & 1, c & 1);
            mfma_tap(I1{});
            spread_reads();
            read_frags(I1{}, t0 + 2, 2, s1 & 1, c & 1);
            mfma_tap(I0{});
            spread_reads();
            wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            batch_of(c, g_, s1);
            advance();
            read_frags(I0{}, g_ * TG, 0, (s1 + 1) & 1, c & 1);
            mfma_tap(I1{});
            spread_batch();
          }
        }
      } else {
        // ---- any other tap count (the 1x1 shortcut passes: one tap per step): plain order, one register set ----
        int c = 0, g_ = 0;
#pragma unroll 1
        for (int s_ = 0; s_ < n_steps; ++s_) {
          const int ntap = (g_ + 1) * TG <= nt ? TG : nt - g_ * TG;
          for (int ti = 0; ti < ntap; ++ti) {
            read_frags(I0{}, g_ * TG + ti, ti, s_ & 1, c & 1);
            mfma_tap(I0{});
          }
          wait_vmcnt<0>();
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
          batch_of(c, g_, s_);
          if (++g_ == ng) { g_ = 0; ++c; }
        }
      }
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();              // (nobody reads the ring any more: the epilogue may use it)
    }

    if (SUM2 && set == 0) {
      const int shift = skip1 ? 0 : (ka1 + kw1) - (ka0 + kw0);
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[rm][rn][r] = __builtin_ldexpf(acc[rm][rn][r], shift);
            acx[rm][rn][r] = __builtin_ldexpf(acx[rm][rn][r], shift);
          }
      return;
    }

    // ---- epilogue (branch-free: rows beyond M carry an offset the buffer unit rejects) ----
    const int k_back = (SUM2 && skip1) ? -(ka0 + kw0) : -((set ? ka1 : ka0) + (set ? kw1 : kw0));
    const int oset = SUM2 ? 0 : set;
    float* outp = oset ? op.out1 : op.out0;
    const int cout = oset ? g.Cout1 : g.Cout0;
    // (which epilogue options a MODE may carry is fixed at compile time -- the launcher rejects the others: every option costs
    //  registers beside 128 accumulators; MODE 2 with all of them spilled)
    double* stats = SUM2 ? nullptr : (oset ? op.stats1 : op.stats0);
    const bool masked = MODE == 0 && op.mask_src != nullptr;
    const bool red = !ACC1 && (oset == 0) && op.red_sums != nullptr;
    const bool want_mm = !SUM2 && (oset == 0) && op.mm0 != nullptr;
    const bool want_amax = (oset == 0) && op.out0_amax != nullptr;
    const int old_ = oset ? g.out_ld1 : g.out_ld0;
    const int out_ld = old_ > 0 ? old_ : cout;
    const unsigned out_bytes = (unsigned)((((long)g.B * g.OH * g.OW - 1) * out_ld + cout) * 4);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(outp, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? op.mask_src : outp), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(red ? op.red_a : outp), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(red ? op.red_b : outp), 0, out_bytes, 0x00020000);
    float csum[RN], csq[RN], rs0[RN], rs1[RN], rs2[RN], rs3[RN], vmx[RN], vng[RN];
    const float kNegInf = __uint_as_float(0xff800000u);
    float out_amax = 0.f;
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) { csum[rn] = csq[rn] = rs0[rn] = rs1[rn] = rs2[rn] = rs3[rn] = 0.f; vmx[rn] = vng[rn] = kNegInf; }
    const unsigned col_off = (unsigned)((n0 + li) * 4);
    // SUM2 (round 6): the two-input data gradient's epilogue goes through LDS.  With the consumer's four BatchNorm-backward sums
    // (16 loads of two more tensors per block, in accumulator layout) beside 128 accumulator registers the register form spilled in
    // every arrangement tried.  Here a wave parks 32 rows x BN columns of results in the (idle) A buffers -- ds_write_b32 in
    // accumulator layout, row pitch 72 floats: the two row groups of a store land in different bank halves -- and reads them back
    // row-major, four consecutive channels per lane: the two extra tensors and the output move as 16-byte accesses of whole
    // 256-byte row segments, and a lane carries four columns' sums over its eight rows.
    constexpr int EP_PITCH = 72;
    float4 es0 = make_float4(0.f, 0.f, 0.f, 0.f), es1 = es0, es2 = es0, es3 = es0;
    const int e_row = lane >> 4, e_col = (lane & 15) * 4;
    const bool e_ok = e_col < BN && n0 + e_col < cout;
    if constexpr (SUM2) {
      static_assert(4 * 32 * EP_PITCH * 4 == L::EP_TILE, "the parked tiles' bytes as the LDS layout reserves them");
      float* tile = reinterpret_cast<float*>(smem) + wave * (32 * EP_PITCH);
      float4 ems = make_float4(0.f, 0.f, 0.f, 0.f), emt = ems;
      if (red && e_ok) { ems = *reinterpret_cast<const float4*>(op.red_scale + n0 + e_col); emt = *reinterpret_cast<const float4*>(op.red_shift + n0 + e_col); }
      const unsigned e_coloff = e_ok ? (unsigned)((n0 + e_col) * 4) : 0xFFFFF000u;
#pragma unroll
      for (int rm = 0; rm < ((CH_EXP & 64) ? 0 : 2); ++rm) {
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tile[(8 * (r >> 2) + 4 * lh + (r & 3)) * EP_PITCH + rn * 32 + li] = __builtin_ldexpf(acc[rm][rn][r] + acx[rm][rn][r], k_back);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS accesses execute in order)
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
          float4 v[4], xa[4], xb[4];
          unsigned vo[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int row = e_row + 4 * (h4 * 4 + k);
            const unsigned ro = sRow[wave * 64 + rm * 32 + row];
            vo[k] = ro >= 0xFFFFF000u ? ro : ro + e_coloff;       // (a row beyond M, or columns beyond the tensor: rejected by the buffer unit)
            if (e_coloff >= 0xFFFFF000u) vo[k] = 0xFFFFF000u;
            v[k] = *reinterpret_cast<const float4*>(tile + row * EP_PITCH + (e_col < BN ? e_col : 0));
            if (red) {
              const u32x4 a_ = __builtin_amdgcn_raw_buffer_load_b128(rs_ra, (int)vo[k], 0, 0), b_ = __builtin_amdgcn_raw_buffer_load_b128(rs_rb, (int)vo[k], 0, 0);
              xa[k] = make_float4(__uint_as_float(a_.x), __uint_as_float(a_.y), __uint_as_float(a_.z), __uint_as_float(a_.w));
              xb[k] = make_float4(__uint_as_float(b_.x), __uint_as_float(b_.y), __uint_as_float(b_.z), __uint_as_float(b_.w));
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (vo[k] >= 0xFFFFF000u) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);      // (idle lanes / rows beyond M add nothing)
            if (red) {
              const float gx = fmaf(xa[k].x, ems.x, emt.x) > 0.f ? v[k].x : 0.f, gy = fmaf(xa[k].y, ems.y, emt.y) > 0.f ? v[k].y : 0.f;
              const float gz = fmaf(xa[k].z, ems.z, emt.z) > 0.f ? v[k].z : 0.f, gw = fmaf(xa[k].w, ems.w, emt.w) > 0.f ? v[k].w : 0.f;
              es0.x += gx; es0.y += gy; es0.z += gz; es0.w += gw;
              es1.x = fmaf(gx, xa[k].x, es1.x); es1.y = fmaf(gy, xa[k].y, es1.y); es1.z = fmaf(gz, xa[k].z, es1.z); es1.w = fmaf(gw, xa[k].w, es1.w);
              es2.x += v[k].x; es2.y += v[k].y; es2.z += v[k].z; es2.w += v[k].w;
              es3.x = fmaf(v[k].x, xb[k].x, es3.x); es3.y = fmaf(v[k].y, xb[k].y, es3.y); es3.z = fmaf(v[k].z, xb[k].z, es3.z); es3.w = fmaf(v[k].w, xb[k].w, es3.w);
            }
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[k].x), __float_as_uint(v[k].y), __float_as_uint(v[k].z), __float_as_uint(v[k].w)},
                                                   rs_o, (int)vo[k], 0, 0);
            if (want_amax) out_amax = fmaxf(fmaxf(out_amax, fmaxf(fabsf(v[k].x), fabsf(v[k].y))), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the tile is rewritten by the next row block)
      }
    } else {
#pragma unroll
    for (int rm = 0; rm < ((CH_EXP & 64) ? 0 : 2); ++rm) {
      unsigned voff[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const u32x4 e = *reinterpret_cast<const u32x4*>(sRow + wave * 64 + rm * 32 + 8 * rg + 4 * lh);
        voff[4 * rg] = e.x + col_off; voff[4 * rg + 1] = e.y + col_off; voff[4 * rg + 2] = e.z + col_off; voff[4 * rg + 3] = e.w + col_off;
      }
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const int nb = n0 + rn * 32;                 // wave-uniform: a 32-column group is in or out as a whole (cout % 32 == 0)
        if (nb < cout) {
          const int n = nb + li;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __builtin_ldexpf(acc[rm][rn][r] + acx[rm][rn][r], k_back);
          if (masked) {
            const float msc = op.mask_scale[n], msh = op.mask_shift[n];
            float src[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              src[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_m, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (!(fmaf(src[r], msc, msh) > 0.f)) v[r] = 0.f;
              csq[rn] = fmaf(v[r], src[r], csq[rn]);
            }
          }
          if (red) {
            const float ms = op.red_scale[n], mt = op.red_shift[n];
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {       // (eight rows at a time: sixteen of both tensors beside the accumulators spilled)
              float xa[8], xb[8];
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                xa[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_ra, (int)(voff[h8 * 8 + r] + (unsigned)(rn * 128)), 0, 0));
                xb[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_rb, (int)(voff[h8 * 8 + r] + (unsigned)(rn * 128)), 0, 0));
              }
#pragma unroll
              for (int r = 0; r < 8; ++r) {       // (rows beyond M hold v = 0 and read 0: they add nothing)
                const float vv = v[h8 * 8 + r];
                const float ga = fmaf(xa[r], ms, mt) > 0.f ? vv : 0.f;
                rs0[rn] += ga;
                rs1[rn] = fmaf(ga, xa[r], rs1[rn]);
                rs2[rn] += vv;
                rs3[rn] = fmaf(vv, xb[r], rs3[rn]);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_o, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0);
            csum[rn] += v[r];                        // rows beyond M accumulated zeros (their inputs were read as 0)
            if (!masked) csq[rn] = fmaf(v[r], v[r], csq[rn]);
          }
          if (want_mm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool ok = voff[r] < 0xFFFFF000u;
              vmx[rn] = fmaxf(vmx[rn], ok ? v[r] : kNegInf);
              vng[rn] = fmaxf(vng[rn], ok ? -v[r] : kNegInf);
            }
          }
          if (want_amax) {
#pragma unroll
            for (int r = 0; r < 16; ++r) out_amax = fmaxf(out_amax, fabsf(v[r]));     // (rows beyond M hold zeros)
          }
        }
      }
    }
    }      // (!SUM2)
    if (want_amax) {       // one look-then-atomic per wave into the workgroup's sub-slot (common.h)
      float m = wave_max(out_amax);
      if (lane == 0) {
        if (!(m == m)) m = __uint_as_float(0x7f800000u);
        unsigned* dst = reinterpret_cast<unsigned*>(op.out0_amax + (blockIdx.x % MPOSE_AMAX_SUBSLOTS) * MPOSE_AMAX_STRIDE);
        if (__float_as_uint(m) > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, __float_as_uint(m));
      }
    }
    // ---- per-channel sums of the four waves through the (idle) B ring, then one row of the partial buffers / atomics ----
    if (want_mm) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float a_ = fmaxf(vmx[rn], __shfl_xor(vmx[rn], 32, 64)), b_ = fmaxf(vng[rn], __shfl_xor(vng[rn], 32, 64));
        if (lh == 0) { sMM[(wave * BN + rn * 32 + li) * 2] = a_; sMM[(wave * BN + rn * 32 + li) * 2 + 1] = b_; }
      }
    }
    if (red && SUM2) {       // the four lane groups of a wave hold the same four columns over different rows
      auto fold = [&](float x) { x += __shfl_xor(x, 16, 64); return x + __shfl_xor(x, 32, 64); };
      const float4 t0 = make_float4(fold(es0.x), fold(es0.y), fold(es0.z), fold(es0.w)), t1 = make_float4(fold(es1.x), fold(es1.y), fold(es1.z), fold(es1.w));
      const float4 t2 = make_float4(fold(es2.x), fold(es2.y), fold(es2.z), fold(es2.w)), t3 = make_float4(fold(es3.x), fold(es3.y), fold(es3.z), fold(es3.w));
      if (lane < 16 && e_col < BN) {
        float4* d = reinterpret_cast<float4*>(sRed + (wave * BN + e_col) * 4);
        d[0] = make_float4(t0.x, t1.x, t2.x, t3.x); d[1] = make_float4(t0.y, t1.y, t2.y, t3.y);
        d[2] = make_float4(t0.z, t1.z, t2.z, t3.z); d[3] = make_float4(t0.w, t1.w, t2.w, t3.w);
      }
    } else if (red) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float t0 = rs0[rn] + __shfl_xor(rs0[rn], 32, 64), t1 = rs1[rn] + __shfl_xor(rs1[rn], 32, 64);
        const float t2 = rs2[rn] + __shfl_xor(rs2[rn], 32, 64), t3 = rs3[rn] + __shfl_xor(rs3[rn], 32, 64);
        if (lh == 0) *reinterpret_cast<float4*>(sRed + (wave * BN + rn * 32 + li) * 4) = make_float4(t0, t1, t2, t3);
      }
    } else if (stats != nullptr) {
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) {
        const float s_ = csum[rn] + __shfl_xor(csum[rn], 32, 64);
        const float q_ = csq[rn] + __shfl_xor(csq[rn], 32, 64);
        if (lh == 0) { float* d = sRed + (wave * BN + rn * 32 + li) * 4; d[0] = s_; d[1] = q_; }
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < cout) {
      const int n = n0 + tid;
      if (red) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float4 d = *reinterpret_cast<const float4*>(sRed + (w * BN + tid) * 4);
          t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
        }
        if (part) {
          float* pb = reinterpret_cast<float*>(op.red_sums);
          if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
          reinterpret_cast<float4*>(pb + kPartHdr)[(size_t)prow * cout + n] = t;
        } else {
          double* d = op.red_sums + (size_t)n * 4;
          atomicAdd(d, (double)t.x); atomicAdd(d + 1, (double)t.y); atomicAdd(d + 2, (double)t.z); atomicAdd(d + 3, (double)t.w);
        }
      } else if (stats != nullptr) {
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const float* d = sRed + (w * BN + tid) * 4; s_ += d[0]; q_ += d[1]; }
        if (part) {
          float* pb = reinterpret_cast<float*>(stats);
          if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
          reinterpret_cast<float2*>(pb + kPartHdr)[(size_t)prow * cout + n] = make_float2(s_, q_);
        } else {
          atomicAdd(stats + (size_t)n * 2, (double)s_);
          atomicAdd(stats + (size_t)n * 2 + 1, (double)q_);
        }
      }
      if (want_mm) {
        float a_ = kNegInf, b_ = kNegInf;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a_ = fmaxf(a_, sMM[(w * BN + tid) * 2]); b_ = fmaxf(b_, sMM[(w * BN + tid) * 2 + 1]); }
        if (part) {
          float* pb = reinterpret_cast<float*>(op.mm0);
          if (hdr_writer && tid == 0) *reinterpret_cast<int*>(pb) = a.part_rows;
          reinterpret_cast<float2*>(pb + kPartHdr)[(size_t)prow * cout + n] = make_float2(a_, b_);
        } else {
          atomicMax(op.mm0 + (size_t)n * 2, float_key(a_));
          atomicMax(op.mm0 + (size_t)n * 2 + 1, float_key(b_));
        }
      }
    }
  };
  if constexpr (SUM2) {      // (the launcher checked: whole kernel rows from the first input, then single taps from the second)
    one_pass(0, std::integral_constant<int, 1>{});
    one_pass(1, std::integral_constant<int, 2>{});
  } else {
#pragma unroll 1
    for (int set = 0; set < NPASS; ++set) one_pass(set, std::integral_constant<int, 0>{});
  }
}

// eligibility of the stride-1 form and its tile parameters
inline bool h2r_eligible(const mpose_conv_geom& g, int* lo_out, int* hi_out) {
  if (g.n_classes != 1 || g.in_mul != 1 || g.in_mul_x != 1 || g.out_mul != 1 || g.out_mul_x != 1) return false;
  if (g.IH != g.GH || g.IW != g.GW || g.OH != g.GH || g.OW != g.GW || g.cls[0].oy || g.cls[0].ox) return false;
  if (g.cls[0].n_taps < 1) return false;
  int n0 = 0;
  for (int t = 0; t < g.cls[0].n_taps; ++t) n0 += g.cls[0].taps[t].acc == 0;
  if (n0 > HR_MAXT || g.cls[0].n_taps - n0 > HR_MAXT) return false;
  int lo = 1 << 30, hi = -(1 << 30);
  for (int t = 0; t < g.cls[0].n_taps; ++t) {
    const int sft = g.cls[0].taps[t].dy * g.IW + g.cls[0].taps[t].dx;
    lo = sft < lo ? sft : lo; hi = sft > hi ? sft : hi;
  }
  *lo_out = lo; *hi_out = hi;
  return true;
}

template <int RN, int MODE, bool X1 = false>
int launch_h2r(ConvHRArgs a, int cmax, int n_groups, hipStream_t s) {
  constexpr int BN = 32 * RN;
  constexpr int lds = H2rLds<RN, MODE, X1>::BYTES;
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  if (a.NG > HR_MAXNG) return MPOSE_ENOSYS;
  a.n_mtiles = (a.M + HR_BM - 1) / HR_BM;
  if (mpose_dry_rows) { *mpose_dry_rows += a.n_mtiles; return 0; }     // (mpose_conv_stat_rows)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h2r_k<RN, MODE, X1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MPOSE_EINVAL;
    attr_set = true;
  }
  dim3 grid(a.n_mtiles, (cmax + BN - 1) / BN, n_groups);
  a.part_row0 = mpose_part_phase.row0;
  a.part_rows = mpose_part_phase.total > 0 ? mpose_part_phase.total : (int)grid.x;
  launch(conv_h2r_k<RN, MODE, X1>, dim3(grid), dim3(256), lds, s, a);
  return launch_status();
}

template <int RN, bool X1 = false>
int launch_h2r_mode(const ConvHRArgs& a, int mode, int cmax, int n_groups, hipStream_t s) {
  if (mode == 1) return launch_h2r<RN, 1, X1>(a, cmax, n_groups, s);
  // (MODE 2, the two-input sum: its second pass restages the whole halo tile for one tap per step, and the 64-channel form
  //  spilled 36 bytes per lane beside the consumer-sums epilogue -- conv_h2_k takes it)
  if (mode == 2) {           // (conv_h2r_k<., 2>: kernel rows of three taps from the first input, fewer than three taps from the second)
    int n0 = 0;
    for (int t = 0; t < a.g.cls[0].n_taps; ++t) n0 += a.g.cls[0].taps[t].acc == 0;
    const int n1 = a.g.cls[0].n_taps - n0;
    if (n0 % HR_TG || n0 == 0 || n1 == 0 || n1 >= HR_TG) return MPOSE_ENOSYS;
    return launch_h2r<RN, 2, X1>(a, cmax, n_groups, s);
  }
  return launch_h2r<RN, 0, X1>(a, cmax, n_groups, s);
}

template <int WM, int WN, int RM, int RN, int KST, int NBUF, int MODE, bool DUAL>
int launch_h2(const ConvHArgs& a0, int n_groups, hipStream_t s) {
  constexpr int BM = 32 * RM * WM, BN = 32 * RN * WN, BNL = (BN + 63) / 64 * 64;
  constexpr int lds = NBUF * KST * (4 * BM * 16 + 4 * BNL * 16) + BM * 4 + 2 * 4 * 32 * RN * 2 * 4 + 4 * 32 * RN * 2 * 4;
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  if (a0.g.Cin % (16 * KST)) return MPOSE_EINVAL;
  if (mpose_dry_rows) { *mpose_dry_rows += ((a0.M + BM - 1) / BM) * a0.g.n_classes; return 0; }     // (mpose_conv_stat_rows)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h2_k<WM, WN, RM, RN, KST, NBUF, MODE, DUAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MPOSE_EINVAL;
    attr_set = true;
  }
  if (a0.g.Cin % (16 * KST)) return MPOSE_EINVAL;
  ConvHArgs a = a0;
  a.n_mtiles = (a.M + BM - 1) / BM;
  const int cmax = a.g.Cout1 > a.g.Cout0 && MODE == 1 ? a.g.Cout1 : a.g.Cout0;
  dim3 grid(a.n_mtiles * a.g.n_classes, (cmax + BN - 1) / BN, n_groups);
  a.part_row0 = mpose_part_phase.row0;
  a.part_rows = mpose_part_phase.total > 0 ? mpose_part_phase.total : (int)grid.x;
  launch(conv_h2_k<WM, WN, RM, RN, KST, NBUF, MODE, DUAL>, dim3(grid), dim3(256), lds, s, a);
  return launch_status();
}

template <int WM, int WN, int RM, int RN, int KST, int NBUF, bool DUAL>
int launch_h2_mode(const ConvHArgs& a, int mode, int n_groups, hipStream_t s) {
  if (mode == 1) return launch_h2<WM, WN, RM, RN, KST, NBUF, 1, DUAL>(a, n_groups, s);
  if (mode == 2) return launch_h2<WM, WN, RM, RN, KST, NBUF, 2, DUAL>(a, n_groups, s);
  return launch_h2<WM, WN, RM, RN, KST, NBUF, 0, DUAL>(a, n_groups, s);
}

// Tile choice (MPOSE_H2_TILE overrides for timing runs: see the table in the function).
int launch_h2_shape(const ConvHArgs& a, int mode, int cmax, int n_groups, hipStream_t s) {
  constexpr int forced = 0;      // (1-4: the tile shapes of round 4's timing runs, kept as the table below)
  if (cmax % 128 == 0) {
    switch (forced) {
      case 1: return launch_h2_mode<2, 2, 2, 2, 1, 4, true>(a, mode, n_groups, s);      // 128 x 128, 16-channel slots, 4 deep
      case 2: return launch_h2_mode<2, 2, 2, 2, 1, 3, true>(a, mode, n_groups, s);
      case 3: return launch_h2_mode<2, 2, 2, 2, 2, 2, false>(a, mode, n_groups, s);     // single accumulator (A/B of the chain bound)
      default: return launch_h2_mode<2, 2, 2, 2, 2, 2, true>(a, mode, n_groups, s);     // 128 x 128, 32-channel slots, 2 deep
    }
  }
  if (cmax % 192 == 0) {
    switch (forced) {
      case 1: return launch_h2_mode<2, 2, 1, 3, 1, 4, true>(a, mode, n_groups, s);
      case 4: return launch_h2_mode<2, 2, 2, 3, 1, 2, false>(a, mode, n_groups, s);     // 128 x 192, single accumulator
      default: return launch_h2_mode<2, 2, 1, 3, 2, 2, true>(a, mode, n_groups, s);     // 64 x 192
    }
  }
  if (cmax % 64 == 0) return launch_h2_mode<4, 1, 1, 2, 2, 2, true>(a, mode, n_groups, s);
  return MPOSE_ENOSYS;
}

}  // namespace
}  // namespace mpose

using namespace mpose;

// Entry used by mpose_conv_fwd when MPOSE_CONV_H2_IN is set (conv.hip validated geometry and operands).
int mpose_conv_h2_launch(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups, int flags, int mode,
                         int cmax, void* stream) {
  ConvHArgs a{};
  a.g = *geom;
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.M = geom->B * geom->GH * geom->GW;
  a.div_gw = make_fastdiv((unsigned)geom->GW);
  a.div_ghw = make_fastdiv((unsigned)(geom->GH * geom->GW));
  a.flags = flags;
  const bool x1 = (flags & MPOSE_CONV_F16X1) != 0;
  const long npix = (long)geom->B * geom->IH * geom->IW;
  const long in_bytes = npix * 16 * 2 * (geom->Cin / 8);
  if (in_bytes >= 0xFFFFFF00l - (1l << 20) || geom->in_ld > 0) return MPOSE_EINVAL;       // 32-bit buffer offsets; dense inputs only
  if ((long)geom->Npad0 * 16 * 4 * (geom->Cin / 16) * MPOSE_MAX_TAPS >= 0xFFFFFF00l) return MPOSE_EINVAL;
  a.in_slab = (unsigned)(npix * 16);
  constexpr int form = 1;
  int lo = 0, hi = 0;
  bool opts_ok = true;       // epilogue options per mode (conv_h2r_k): 0: all; 1: statistics + extremes; 2: consumer sums + output amax
  for (int i = 0; i < n_groups; ++i) {
    if (mode == 1 && (ops[i].mask_src || ops[i].red_sums)) opts_ok = false;
    if (mode == 2 && (ops[i].mask_src || ops[i].stats0 || ops[i].mm0)) opts_ok = false;
  }
  if (form && opts_ok && h2r_eligible(*geom, &lo, &hi)) {
    ConvHRArgs r{};
    r.g = a.g;
    for (int i = 0; i < n_groups; ++i) r.op[i] = ops[i];
    r.M = a.M; r.div_gw = a.div_gw; r.div_ghw = a.div_ghw; r.flags = flags; r.in_slab = a.in_slab;
    r.lo = lo;
    r.NG = (HR_BM + hi - lo + 63) / 64;
    r.a_rows = r.NG * 64 + 16;
    // (MPOSE_CONV_F16X1: the 64-channel tile of the stride-1 form only -- the engine's regular blocks)
    const int rc = x1 ? (cmax <= 32 ? MPOSE_ENOSYS : launch_h2r_mode<2, true>(r, mode, cmax, n_groups, (hipStream_t)stream))
                      : (cmax <= 32 ? launch_h2r_mode<1>(r, mode, cmax, n_groups, (hipStream_t)stream)
                                    : launch_h2r_mode<2>(r, mode, cmax, n_groups, (hipStream_t)stream));
    if (rc != MPOSE_ENOSYS || x1) return rc;
  }
  if (x1) return MPOSE_ENOSYS;
  return launch_h2_shape(a, mode, cmax, n_groups, (hipStream_t)stream);
}
