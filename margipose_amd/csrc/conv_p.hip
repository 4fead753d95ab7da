// Implicit-GEMM convolution on PRE-SPLIT operands for gfx950 ("plane" engine, round 2).
//
// Same arithmetic as conv.hip's conv_igemm_k -- an fp32 convolution as six exact bf16 x bf16 products per pair of
// operands, accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- but the activations arrive already split:
//   * activations: three bf16 planes (hi, mid, lo; x = hi + mid + lo to 2^-27) in the blocked layout
//         P8[C/8][plane][pixel][8]                      (16 bytes per pixel per (channel octet, plane))
//     written once by the elementwise producer (BatchNorm+ReLU / residual add / BatchNorm backward), instead of being
//     re-split by every (tap, output tile) that reads them -- round 1 measured that split at 11 % of every launch;
//   * weights: the same three planes, packed [widx][K/16][plane][half][Npad][8] by pack_weights_k (layout 1).
// With nothing left to do on the VALU, both operands go global -> LDS by DMA (buffer_load_dwordx4 ... lds, 1 KiB per
// wave instruction, fully contiguous: 64 consecutive pixels x 16 B, or 64 consecutive output channels x 16 B), the
// padding zeros come from the buffer unit's range check, and a workgroup shares its B tile through LDS instead of
// every wave streaming its own copy from L2 (round 1: 96 of 128 KiB per tile step).
//
// Structure: 256 threads = 4 waves as WM x WN; wave tile 32*RM pixels x 32*RN channels; TWO workgroups per CU
// (2 waves per SIMD, 256 registers each) so that one wave's LDS reads / address work / barrier waits sit under the
// other's MFMAs -- round 1's single wave per SIMD left every such cycle exposed.  A K step is one (tap, 16 input
// channels) slice: A tile BM x 16 x 3 planes, B tile 16 x BN x 3 planes, in an NBUF-deep LDS ring filled NBUF-1
// steps ahead; one s_barrier per step, counted vmcnt (the DMA of later steps stays in flight across it).
// LDS images (all ds_read_b128 conflict-free: 32 consecutive lanes read 32 consecutive 16-byte words):
//   A: [plane][k-half][BM pixels][16 B]      lane (i = l&31, h = l>>5) reads pixel i of its row block, half h
//   B: [plane][k-half][BNL columns][16 B]    lane (n = l&31, h) reads column n of its column block, half h
// Geometry (tap lists, output classes), fused shortcut (MODE 1), two-input sum (MODE 2), BatchNorm statistics /
// ReLU-mask / accumulate epilogues: identical contracts to conv_igemm_k (include/margipose_hip.h).
//
// NPL = 3: fp32-equivalent (bf16x6).  NPL = 1: single-pass bf16 x bf16 -> fp32 (BASELINE configs[4]'s reduced-precision
// convolutions): only the hi planes are read, one MFMA per fragment pair.
//
// Replaces Conv2d / ConvTranspose2d of reference src/margipose/models/margipose_model.py:33,67-82 and their
// data-gradients inside the columns.
#include <stdlib.h>
#include "common.h"

namespace mpose {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

struct ConvPArgs {
  mpose_conv_geom g;
  mpose_conv_operands op[MPOSE_MAX_GROUP];
  FastDiv div_gw, div_ghw;
  int M;                          // slots per class = B*GH*GW
  int n_mtiles;
  int flags;
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

__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int WM, int WN, int RM, int RN, int NBUF, int MODE, int NPL>
__global__ __launch_bounds__(256, 2) void conv_planes_k(ConvPArgs a) {
  static_assert(WM * WN == 4, "four waves");
  constexpr bool ACC1 = MODE == 1, SUM2 = MODE == 2;
  constexpr int NPASS = ACC1 ? 2 : 1;          // SUM2 runs as ONE pass over all taps (the acc == 1 taps read in1 / w1)
  constexpr int BM = 32 * RM * WM, BN = 32 * RN * WN, BNL = (BN + 63) / 64 * 64;
  constexpr int SGN = BM / 64, NGN = BNL / 64;                 // 64-pixel / 64-column groups = DMA instructions per (plane, half)
  static_assert(SGN == 1 || SGN == 2 || SGN == 4, "slot groups must divide the wave count");
  constexpr int A_B = NPL * 2 * BM * 16, B_B = NPL * 2 * BNL * 16, BUF_B = A_B + B_B;
  constexpr int NA = NPL * 2 * SGN, NB = NPL * 2 * NGN, TOT = NA + NB;     // DMA instructions per step (workgroup)
  constexpr int L_LO = TOT / 4, L_REM = TOT % 4;               // per wave: L_LO (+1 for waves < L_REM)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* sRow = reinterpret_cast<unsigned*>(smem + NBUF * BUF_B);                 // [BM] output row byte offsets
  unsigned* sPix = reinterpret_cast<unsigned*>(smem + NBUF * BUF_B + BM * 4);        // [BM] output pixel index (fused plane output)
  float* sRed = reinterpret_cast<float*>(smem + NBUF * BUF_B + 2 * BM * 4);          // [2 sets][4 waves][32*RN][2]
  constexpr int TILE_PITCH = 32 * RN + 4;                                            // floats per row of a wave's staged output tile
  constexpr bool TILE_FITS = 4 * 32 * RM * TILE_PITCH * 4 <= NBUF * BUF_B;            // the fused plane output stages the tile in the (free) ring

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  const mpose_conv_geom& g = a.g;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed placement, speed only), each XCD has its own L2; giving an
  // XCD a CONTIGUOUS run of pixel tiles lets neighbouring tiles share their halo rows (and a class's tiles their weights)
  // in one L2 instead of fetching them once per XCD (HBM-side reads were 1.9x the algorithmic bytes with the plain order).
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0 && !(a.flags & 0x400)) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
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
      sPix[tid] = (int)m < a.M ? pix : 0xFFFFFFFFu;
    }
  };

  const int k16_total = g.Cin >> 4;
  const int npad = g.Npad0;                                        // == Npad1 when a second weight set is used
  const unsigned w_plane_b = (unsigned)npad * 16u;                 // bytes of one (plane, half) slab of packed weights
  const __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.in), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SUM2 ? op.in1 : op.in), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.w0), 0, 0xFFFFFF00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MODE ? op.w1 : op.w0), 0, 0xFFFFFF00, 0x00020000);

  // Two accumulators per output block when NPL == 3: `acc` takes only the hi x hi products, `acs` the five cross terms
  // (<= 2^-8 of them).  Every MFMA rounds its accumulator once; with a single accumulator that is 6 roundings of the
  // full-size running sum per step (432 for a 128-channel 3x3: 1.1e-6 relative, 4x a CPU fp32 convolution), with the
  // split it is ONE (72: below the CPU's), and the cross-term accumulator's roundings are 2^-8 smaller.
  constexpr int NACS = NPL == 3 ? RN : 1;
  f32x16 acc[RM][RN], acs[RM][NACS];
  // taps with acc == 0 first, taps with acc == 1 (second weight set) last
  int n_taps0 = 0;
  for (int t = 0; t < n_taps; ++t) n_taps0 += (((tap_word(t) >> 24) & 0xff) == 0) ? 1 : 0;

#pragma unroll 1
  for (int set = 0; set < NPASS; ++set) {
    const int t_lo = set ? n_taps0 : 0;
    const int nt = set ? n_taps - n_taps0 : (ACC1 ? n_taps0 : n_taps);
    const int n_steps = k16_total * nt;
    {
#pragma unroll
      for (int rm = 0; rm < RM; ++rm)
#pragma unroll
        for (int rn = 0; rn < RN; ++rn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[rm][rn][r] = 0.0f;
            if (NPL == 3) acs[rm][rn % NACS][r] = 0.0f;
          }
    }
    if (set == 0 || ACC1) {
      const int ld_ = (set && ACC1) ? g.out_ld1 : g.out_ld0;
      fill_rows(ld_ > 0 ? ld_ : ((set && ACC1) ? g.Cout1 : g.Cout0));
    }

    // ---- one step's DMA: instruction gidx of the workgroup's list (A first, then B) goes to wave gidx % 4 ----
    auto issue = [&](int buf, int c16, int t) {
      const int tp = tap_word(t_lo + t);
      const int dy = (int)(signed char)(tp & 0xff), dx = (int)(signed char)((tp >> 8) & 0xff);
      const int widx = (tp >> 16) & 0xff;
      unsigned voff_a = ((row_taps >> (t_lo + t)) & 1u) ? pix_off + (unsigned)((dy * g.IW + dx) * 16) : kOob;
      if ((a.flags & 0x100) && t != 0) voff_a = kOob;          // timing experiments only (CP_TRAFFIC_EXP builds): no A traffic after a chunk's first tap
      const bool skip_b = (a.flags & 0x200) != 0;               //                                       no B traffic
      const unsigned w_base = (unsigned)(widx * k16_total + c16) * 6u * w_plane_b;
      unsigned char* bufp = smem + buf * BUF_B;
      const bool second = ACC1 ? set != 0 : (SUM2 && (t_lo + t) >= n_taps0);     // which input / weight set this tap reads
#pragma unroll
      for (int k = 0; k < (TOT + 3) / 4; ++k) {
        const int gidx = wave + 4 * k;
        if (gidx < NA) {
          const int ph = gidx / SGN;                               // plane * 2 + half
          const unsigned soff = (unsigned)((c16 * 2 + (ph & 1)) * 3 + (ph >> 1)) * a.in_slab;
          if (SUM2 && second) dma16(rs_in1, bufp + (ph * BM + sg * 64) * 16, voff_a, soff);
          else dma16(rs_in0, bufp + (ph * BM + sg * 64) * 16, voff_a, soff);
        } else if (gidx < TOT) {
          const int j = gidx - NA;
          const int ph = j / NGN, ng = j - ph * NGN;
          const unsigned voff_b = skip_b ? kOob : (unsigned)((n0 + ng * 64 + lane) * 16);
          if (second) dma16(rs_w1, bufp + A_B + (ph * BNL + ng * 64) * 16, voff_b, w_base + (unsigned)ph * w_plane_b);
          else dma16(rs_w0, bufp + A_B + (ph * BNL + ng * 64) * 16, voff_b, w_base + (unsigned)ph * w_plane_b);
        }
      }
    };

    // issue cursor: (chunk, tap) of the next step to fetch, tap fastest (the taps of a chunk re-read the same pixels)
    int ic = 0, itp = 0, issued = 0;
    auto issue_next = [&]() {
      issue(issued % NBUF, ic, itp);
      ++issued;
      if (++itp == nt) { itp = 0; ++ic; }
    };
    // Software pipeline with ONE fragment register set ("rolling"): the MFMA blocks of a step run in an order that retires
    // its fragments one after the other, and each retired fragment's registers are refilled at once with the same fragment
    // of step s+1 -- whose MFMAs therefore never wait for LDS, at no register cost (a second fragment set would not fit
    // beside the two accumulator sets of the 64 x 64 wave tile).  The ring slot of step s is free once all its fragments
    // sit in registers, which is where the step's one barrier goes: after the first group of blocks.
    //     RM == 2:  blocks (0,*) | sync | A(rm0) <- s+1 | (1,0) B(0) <- s+1 | (1,1) B(1) <- s+1 ... | A(rm1) <- s+1
    //     RM == 1:  block  (0,0) | sync | B(0), A' <- s+1 | (0,1) B(1) <- s+1 | ...  | A = A'   (A is used by every block: second set)
    //     sync = wait until step s+1 has landed (mine) ; lgkmcnt(0) ; s_barrier (everyone's) ; issue the DMA of step s+NBUF
    auto wait_steps_outstanding = [&](int k) {         // at most k steps' worth of this wave's DMA still in flight
      if (NBUF > 2 && k == NBUF - 2) {
        if (wave < L_REM) wait_vmcnt<(NBUF - 2) * (L_LO + 1)>(); else wait_vmcnt<(NBUF - 2) * L_LO>();
      } else if (NBUF > 2 && k == NBUF - 1) {
        if (wave < L_REM) wait_vmcnt<(NBUF - 1) * (L_LO + 1)>(); else wait_vmcnt<(NBUF - 1) * L_LO>();
      } else {
        wait_vmcnt<0>();
      }
    };
    bf16x8 af[RM][NPL], bfr[RN][NPL], afn[NPL];
    auto read_a = [&](int step, int rm, bf16x8 (&dst)[NPL]) {
      const unsigned char* pa = smem + (step % NBUF) * BUF_B + (lh * BM + wm * 32 * RM + rm * 32 + li) * 16;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) dst[pl] = *reinterpret_cast<const bf16x8*>(pa + pl * 2 * BM * 16);
    };
    auto read_b = [&](int step, int rn, bf16x8 (&dst)[NPL]) {
      const unsigned char* pb = smem + (step % NBUF) * BUF_B + A_B + (lh * BNL + wn * 32 * RN + rn * 32 + li) * 16;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) dst[pl] = *reinterpret_cast<const bf16x8*>(pb + pl * 2 * BNL * 16);
    };
    auto block = [&](int rm, int rn) {
      if constexpr (NPL == 3) {
        f32x16 c = acs[rm][rn];                  // cross terms, smallest first
        c = mfma_bf16(af[rm][2], bfr[rn][0], c);
        c = mfma_bf16(af[rm][0], bfr[rn][2], c);
        c = mfma_bf16(af[rm][1], bfr[rn][1], c);
        c = mfma_bf16(af[rm][1], bfr[rn][0], c);
        c = mfma_bf16(af[rm][0], bfr[rn][1], c);
        acs[rm][rn] = c;
      }
      acc[rm][rn] = mfma_bf16(af[rm][0], bfr[rn][0], acc[rm][rn]);
    };
    auto sync = [&](int s_, bool more) {
      if (more) wait_steps_outstanding(issued - s_ - 2);
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) -- the builtin, so that hipcc's own wait-count bookkeeping sees it
      __builtin_amdgcn_s_barrier();
      if (issued < n_steps) issue_next();
    };
    for (int p = 0; p < NBUF && p < n_steps; ++p) issue_next();
    if (n_steps > 0) {
      wait_steps_outstanding(issued - 1);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int rm = 0; rm < RM; ++rm) read_a(0, rm, af[rm]);
#pragma unroll
      for (int rn = 0; rn < RN; ++rn) read_b(0, rn, bfr[rn]);
    }
    // (the loop is rotated -- sync(s), second group of step s, first group of step s+1 -- so that every prefetched fragment is
    //  consumed inside the iteration that fetched it and hipcc can place exact lgkmcnt counts instead of a drain at the top)
    auto group0 = [&]() {
      if constexpr (RM == 2) {
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) block(0, rn);
      } else if constexpr (RN > 1) {
        block(0, 0);
      }
    };
    if (n_steps > 0) group0();
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
      const bool more = s + 1 < n_steps;
      sync(s, more);
      // (the prefetches are unconditional -- after the last step they read a stale slot and the values are dropped -- so that
      //  the iteration is one basic block and hipcc counts the LDS reads exactly)
      // (sched_barrier: hipcc's list scheduler would otherwise hoist every MFMA whose operands are ready above the LDS reads
      //  and sink the reads to the end of the step, which re-exposes exactly the latency this order is meant to hide)
      if constexpr (RM == 2) {
        read_a(s + 1, 0, af[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rn = 0; rn < RN; ++rn) {
          block(1, rn);
          __builtin_amdgcn_sched_barrier(0);
          read_b(s + 1, rn, bfr[rn]);
          if (rn == RN - 1) read_a(s + 1, 1, af[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (RN == 1) {      // a single block per step: both fragments double-buffered
        bf16x8 bfn[NPL];
        read_b(s + 1, 0, bfn);
        read_a(s + 1, 0, afn);
        __builtin_amdgcn_sched_barrier(0);
        block(0, 0);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) { af[0][pl] = afn[pl]; bfr[0][pl] = bfn[pl]; }
      } else {
        read_b(s + 1, 0, bfr[0]);
        read_a(s + 1, 0, afn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rn = 1; rn < RN; ++rn) {
          block(0, rn);
          __builtin_amdgcn_sched_barrier(0);
          read_b(s + 1, rn, bfr[rn]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) af[0][pl] = afn[pl];
      }
      if (more) group0();
    }
    __builtin_amdgcn_s_barrier();              // the ring may be refilled by the next pass; sRow is complete

    // ---- epilogue (branch-free: rows beyond M carry an offset the buffer unit rejects) ----
    const int oset = SUM2 ? 0 : set;
    float* outp = oset ? op.out1 : op.out0;
    const int cout = oset ? g.Cout1 : g.Cout0;
    double* stats = oset ? op.stats1 : op.stats0;
    const bool masked = (oset == 0) && op.mask_src != nullptr;
    const bool accumulate = (oset == 0) && (a.flags & 1);
    // fused output stage (inference): y = [relu](scale*conv + shift) [+ add_scale*add_src + add_shift] -> fp32 and/or planes
    const bool epi = (oset == 0) && op.epi_scale0 != nullptr;
    const bool epi_relu = (a.flags & 16) != 0;
    const bool epi_add = epi && op.add_src != nullptr;
    void* out_planes = (TILE_FITS && oset == 0) ? op.out0_planes : nullptr;
    const bool store_f32 = outp != nullptr;
    float* tile = reinterpret_cast<float*>(smem) + wave * (32 * RM * TILE_PITCH);     // (the ring is free: every wave passed the barrier above)
    const int old_ = oset ? g.out_ld1 : g.out_ld0;
    const int out_ld = old_ > 0 ? old_ : cout;
    const unsigned out_bytes = (unsigned)((((long)g.B * g.OH * g.OW - 1) * out_ld + cout) * 4);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(outp, 0, store_f32 ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi_add ? op.add_src : op.w0), 0, epi_add ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? op.mask_src : outp), 0, out_bytes, 0x00020000);
    float csum[RN], csq[RN];
#pragma unroll
    for (int rn = 0; rn < RN; ++rn) csum[rn] = csq[rn] = 0.f;
    const int ncol0 = n0 + wn * 32 * RN;
    const unsigned col_off = (unsigned)((ncol0 + li) * 4);
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
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
          for (int r = 0; r < 16; ++r) v[r] = NPL == 3 ? acc[rm][rn][r] + acs[rm][rn % NACS][r] : acc[rm][rn][r];
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
          if (accumulate) {
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_o, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += old[r];
          }
          if (epi) {
            const float esc = op.epi_scale0[n], esh = op.epi_shift0[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              v[r] = fmaf(v[r], esc, esh);
              if (epi_relu) v[r] = fmaxf(v[r], 0.f);
            }
            if (epi_add) {
              const float asc = op.add_scale[n], ash = op.add_shift[n];
              float src[16];
#pragma unroll
              for (int r = 0; r < 16; ++r)
                src[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0));
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] += fmaf(src[r], asc, ash);
            }
          }
          if (out_planes != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(rm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TILE_PITCH + rn * 32 + li] = v[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_o, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0);
            csum[rn] += v[r];                        // rows beyond M accumulated zeros (their inputs were read as 0)
            if (!masked) csq[rn] = fmaf(v[r], v[r], csq[rn]);
          }
        }
      }
    }
    if (out_planes != nullptr) {
      // The wave's fp32 tile sits in LDS [row][column] (pitch 32*RN + 4 floats: a 16-lane group reading 8 consecutive channels of
      // 16 consecutive rows touches every bank once).  Lane = row: 64 consecutive pixels x 16 B per (octet, plane) -> 1 KiB stores.
      __builtin_amdgcn_wave_barrier();
      const long npix_out = (long)g.B * g.OH * g.OW;
      constexpr int ROWS = 32 * RM, OCTS = 4 * RN;
#pragma unroll 1
      for (int p = lane; p < ROWS * OCTS; p += 64) {
        const int row = p % ROWS, oc = p / ROWS;
        const int nb = ncol0 + oc * 8;
        const unsigned pix = sPix[wm * ROWS + row];
        const float4 lo4 = *reinterpret_cast<const float4*>(tile + row * TILE_PITCH + oc * 8);
        const float4 hi4 = *reinterpret_cast<const float4*>(tile + row * TILE_PITCH + oc * 8 + 4);
        if (nb < cout && pix != 0xFFFFFFFFu) {
          const float v8[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
          u32x4 ph, pm, pl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x0 = v8[2 * e], x1 = v8[2 * e + 1];
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
            const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
            const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
            const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xFFFF0000u);
            const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2));
            ph[e] = h; pm[e] = m; pl[e] = l;
          }
          u32x4* d = reinterpret_cast<u32x4*>(out_planes) + (long)(nb >> 3) * 3 * npix_out + pix;
          d[0] = ph; d[npix_out] = pm; d[2 * npix_out] = pl;
        }
      }
      __builtin_amdgcn_wave_barrier();
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
#pragma unroll
  for (int set = 0; set < (ACC1 ? 2 : 1); ++set) {
    double* stats = set ? op.stats1 : op.stats0;
    const int cout = set ? g.Cout1 : g.Cout0;
    if (stats != nullptr && tid < BN) {
      const int wn_ = tid / (32 * RN), col = tid - wn_ * 32 * RN;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        const float* d = sRed + ((set * 4 + (wn_ * WM + w)) * 32 * RN + col) * 2;
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

template <int WM, int WN, int RM, int RN, int NBUF, int MODE, int NPL>
int launch_planes(const ConvPArgs& a0, int n_groups, hipStream_t s) {
  constexpr int BM = 32 * RM * WM, BN = 32 * RN * WN, BNL = (BN + 63) / 64 * 64;
  constexpr int lds = NBUF * (NPL * 2 * BM * 16 + NPL * 2 * BNL * 16) + 2 * BM * 4 + 2 * 4 * 32 * RN * 2 * 4;
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  if (mpose_dry_rows) { *mpose_dry_rows += ((a0.M + BM - 1) / BM) * a0.g.n_classes; return 0; }     // (mpose_conv_stat_rows)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_planes_k<WM, WN, RM, RN, NBUF, MODE, NPL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return MPOSE_EINVAL;
    attr_set = true;
  }
  ConvPArgs a = a0;
  a.n_mtiles = (a.M + BM - 1) / BM;
  const int cmax = a.g.Cout1 > a.g.Cout0 && MODE == 1 ? a.g.Cout1 : a.g.Cout0;
  dim3 grid(a.n_mtiles * a.g.n_classes, (cmax + BN - 1) / BN, n_groups);
  launch(conv_planes_k<WM, WN, RM, RN, NBUF, MODE, NPL>, dim3(grid), dim3(256), lds, s, a);
  return launch_status();
}

template <int WM, int WN, int RM, int RN, int NBUF, int NPL>
int launch_planes_mode(const ConvPArgs& a, int mode, int n_groups, hipStream_t s) {
  if (mode == 1) return launch_planes<WM, WN, RM, RN, NBUF, 1, NPL>(a, n_groups, s);
  if (mode == 2) return launch_planes<WM, WN, RM, RN, NBUF, 2, NPL>(a, n_groups, s);
  return launch_planes<WM, WN, RM, RN, NBUF, 0, NPL>(a, n_groups, s);
}

// Tile choice.  Two workgroups share a CU (512 resident workgroups on the chip); the wide 128-pixel tile halves the
// B traffic per MFMA, the 64-pixel tile fills the chip when a launch has few pixels (the 16x16 mid-resolution layers).
template <int NPL>
int launch_planes_shape(const ConvPArgs& a, int mode, int cmax, int n_groups, hipStream_t s) {
  const long m_nominal = 32l * a.g.GH * a.g.GW;                    // (nominal batch: the choice must not depend on B, see conv.hip)
  const long wide_wgs = ((m_nominal + 127) / 128) * a.g.n_classes * n_groups;
  const bool narrow_m = wide_wgs < 384;
  if (cmax <= 32) return launch_planes_mode<4, 1, 1, 1, 3, NPL>(a, mode, n_groups, s);
  if (cmax == 64) return launch_planes_mode<4, 1, 1, 2, 3, NPL>(a, mode, n_groups, s);
  if (cmax == 96) return launch_planes_mode<4, 1, 1, 3, 3, NPL>(a, mode, n_groups, s);
  if (cmax % 192 == 0) return launch_planes_mode<2, 2, 1, 3, 3, NPL>(a, mode, n_groups, s);     // (a 128 x 192 tile would need 2 x 96 accumulator registers)
  if (cmax % 128 == 0) {
    if (narrow_m) return launch_planes_mode<2, 2, 1, 2, 3, NPL>(a, mode, n_groups, s);
    return launch_planes_mode<2, 2, 2, 2, 3, NPL>(a, mode, n_groups, s);
  }
  return MPOSE_EINVAL;
}

}  // namespace
}  // namespace mpose

using namespace mpose;

// Entry used by mpose_conv_fwd when MPOSE_CONV_PLANES_IN is set (conv.hip validated geometry and operands).
int mpose_conv_planes_launch(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups, int flags, int mode,
                             int cmax, void* stream) {
  ConvPArgs a{};
  a.g = *geom;
  for (int i = 0; i < n_groups; ++i) a.op[i] = ops[i];
  a.M = geom->B * geom->GH * geom->GW;
  a.div_gw = make_fastdiv((unsigned)geom->GW);
  a.div_ghw = make_fastdiv((unsigned)(geom->GH * geom->GW));
  a.flags = flags;
#ifdef CP_TRAFFIC_EXP      // timing experiments, debug builds only (-DCP_TRAFFIC_EXP=<bits>): 1 = skip A re-reads, 2 = skip B (wrong results); 4 = plain tile order
  a.flags |= (CP_TRAFFIC_EXP & 7) << 8;
#endif
  const long npix = (long)geom->B * geom->IH * geom->IW;
  const long in_bytes = npix * 16 * 3 * (geom->Cin / 8);
  if (in_bytes >= 0xFFFFFF00l - (1l << 20) || geom->in_ld > 0) return MPOSE_EINVAL;       // 32-bit buffer offsets; dense inputs only
  if ((long)geom->Npad0 * 16 * 6 * (geom->Cin / 16) * MPOSE_MAX_TAPS >= 0xFFFFFF00l) return MPOSE_EINVAL;
  a.in_slab = (unsigned)(npix * 16);
  for (int i = 0; i < n_groups; ++i) {
    const mpose_conv_operands& o = ops[i];
    const bool epi = o.epi_scale0 != nullptr || o.out0_planes != nullptr || o.add_src != nullptr;
    if (!epi) continue;
    if ((flags & (MPOSE_CONV_BF16 | MPOSE_CONV_ACCUMULATE)) || o.stats0 || o.mask_src) return MPOSE_EINVAL;    // inference-only stage
    if ((o.epi_scale0 != nullptr) != (o.epi_shift0 != nullptr)) return MPOSE_EINVAL;
    if (o.add_src && (!o.epi_scale0 || !o.add_scale || !o.add_shift)) return MPOSE_EINVAL;
    if (geom->out_ld0 > 0 && geom->out_ld0 != geom->Cout0) return MPOSE_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  if (flags & MPOSE_CONV_BF16) return launch_planes_shape<1>(a, mode, cmax, n_groups, s);
  return launch_planes_shape<3>(a, mode, cmax, n_groups, s);
}
