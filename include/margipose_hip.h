/*
 * margipose_hip.h -- C ABI of libmargipose_hip.so, the MI355X (gfx950) implementation of the
 * MargiPose forward/backward hot path.
 *
 * The reference (anibali/margipose) has NO native/FFI boundary: its hot path is Python calling
 * torch.nn.  This header is therefore the boundary a maintainer would bind if the
 * reference grew one; each entry point names the reference code it replaces (paths relative to
 * src/margipose/ in the reference tree).  The Python host side (margipose_amd/) binds it with
 * ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller;
 *   - no hidden allocation, no host synchronisation; work is enqueued on `stream` (hipStream_t);
 *   - return value: 0 on success, otherwise a hipError_t value, or a negative MPOSE_E* code for
 *     invalid arguments;
 *   - activations inside the backbone are NHWC fp32 (channel stride `C`), heatmaps/logits at the
 *     model boundary are NCHW fp32 exactly as the reference's tensors.
 */
#ifndef MARGIPOSE_HIP_H
#define MARGIPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPOSE_EINVAL (-22)
#define MPOSE_ENOSYS (-38)

#define MPOSE_MAX_GROUP 3   /* the xy / zy / xz columns of one stage run as one grouped launch */
#define MPOSE_MAX_TAPS 12
#define MPOSE_MAX_CLASSES 8

int mpose_abi_version(void);
/* sizeof() of the ABI structs, for binding self-checks: which = 0 geom, 1 conv operands, 2 wgrad
 * operands, 3 pack job, 4 unpack job, 5 bn job, 6 bn coef job, 7 bn_add ops, 8 reduce ops, 9 apply ops, 10 split ops, 11 sgd job. */
int mpose_sizeof(int which);

/* ------------------------------------------------------------------------------------------
 * Soft-argmax tail (dsntnn.py, models/margipose_model.py:215-261)
 * A "row" is one (batch, joint) heatmap of n = H*W fp32 values, n % 4 == 0, n <= 4096.
 * Kernels use one 64-lane wavefront per (row, plane); planes = xy, zy, xz.
 * ------------------------------------------------------------------------------------------ */

/* flat_softmax (dsntnn.py:124-130) fused with dsnt (dsntnn.py:39-62,84-96) and, when n_planes == 3,
 * MargiPoseModel.heatmaps_to_coords (models/margipose_model.py:254-261).
 *   logits[p], heatmaps[p]: (rows, H, W) for plane p (heatmaps[p] may be NULL: coordinates only)
 *   plane_coords: (n_planes, rows, 2) = (mu_x, mu_y) per plane, may be NULL
 *   xyz: (rows, 3), only written when n_planes == 3, may be NULL
 *   io_dtype: 0 = fp32 logits/heatmaps, 1 = bf16 logits/heatmaps, 2 = fp32 logits -> bf16 heatmaps (arithmetic and the
 *   coordinates are fp32 in every mode: BASELINE configs[1], "bf16 heatmaps + fp32 soft-argmax") */
int mpose_softmax_dsnt_fwd(const void* const* logits, void* const* heatmaps, float* plane_coords,
                           float* xyz, int n_planes, int rows, int H, int W, int io_dtype,
                           void* stream);

/* dsnt alone on already-normalised heatmaps (MargiPoseModel.heatmaps_to_coords on arbitrary input). */
int mpose_dsnt_fwd(const float* const* heatmaps, float* plane_coords, float* xyz, int n_planes,
                   int rows, int H, int W, void* stream);

/* backward of dsnt: d_heatmap[r,h,w] (+)= d_mu_x[r] * x_w + d_mu_y[r] * y_h.
 * d_plane_coords: (n_planes, rows, 2).  accumulate != 0 adds into d_heatmaps. */
int mpose_dsnt_bwd(const float* d_plane_coords, float* const* d_heatmaps, int n_planes, int rows,
                   int H, int W, int accumulate, void* stream);

/* backward of flat_softmax: dlogits = p * (g - sum_k p_k g_k), g = g1 (+ g2 when non-NULL). */
int mpose_softmax_bwd(const float* const* heatmaps, const float* const* g1, const float* const* g2,
                      float* const* dlogits, int n_planes, int rows, int n, void* stream);

/* One stage of MargiPoseModel.forward_3d_losses / forward_2d_losses
 * (models/margipose_model.py:223-252): per (b,j)
 *     sum over planes of js_reg_losses (dsntnn.py:154-232, sigma in pixels)   [if pixelwise]
 *   + euclidean_losses(heatmaps_to_coords(...), target) (dsntnn.py:133-151).
 *   heatmaps: 3 planes (rows, H, W); target: (rows, 3); three_d == 0 selects the 2D variant
 *   (xy plane only).  losses: (rows); accumulate != 0 adds the stage into `losses` (the
 *   reference's `losses += ...`).  xyz_out: (rows, 3) coordinates saved for the backward. */
int mpose_stage_loss_fwd(const float* const* heatmaps, const float* target, float* losses,
                         float* xyz_out, int rows, int H, int W, float sigma, int pixelwise,
                         int three_d, int accumulate, void* stream);

/* Backward of the above w.r.t. the heatmaps (SURVEY.md §8 row a-T): g[p] = dloss[r] * dL_r/dp.
 * The Gaussian targets are regenerated in registers.  accumulate != 0 adds into g. */
int mpose_stage_loss_bwd(const float* const* heatmaps, const float* target, const float* xyz,
                         const float* dloss, float* const* g, int rows, int H, int W, float sigma,
                         int pixelwise, int three_d, int accumulate, void* stream);

/* js_reg_losses alone (dsntnn.py:220-232) for one plane: mu (rows, 2) -> js (rows). */
int mpose_js_fwd(const float* heatmaps, const float* mu, float* js, int rows, int H, int W,
                 float sigma, void* stream);
int mpose_js_bwd(const float* heatmaps, const float* mu, const float* djs, float* g, int rows,
                 int H, int W, float sigma, void* stream);

/* average_loss (dsntnn.py:99-121): out[0] = sum(l*m)/max(sum(m),1); out[1] = the denominator. */
int mpose_average_loss_fwd(const float* losses, const float* mask, float* out2, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backbone (models/margipose_model.py:25-200).  Declared in the sections below as they land.
 * ------------------------------------------------------------------------------------------ */

/* Geometry of one implicit-GEMM convolution launch (all Conv2d / ConvTranspose2d of
 * models/margipose_model.py:33,67-82 and their data-gradients are instances):
 * the launch enumerates a grid of GH x GW "slots" per image and per class; slot (gy,gx) of class
 * c reads input pixels (gy*in_mul + dy_t, gx*in_mul_x + dx_t) for every tap t of the class and
 * writes output pixel (gy*out_mul + oy_c, gx*out_mul_x + ox_c).  in_mul_x / out_mul_x = 0 mean "as along y" (the
 * square strides of margipose_model.py); the dilated, one-axis-strided Conv2d / ConvTranspose2d of
 * models/chatterbox_model.py:88-125 set them. */
typedef struct {
  int8_t dy, dx;      /* input offset of the tap */
  int8_t widx;        /* which packed weight slice the tap multiplies with */
  int8_t acc;         /* 0: main accumulator (output 0), 1: second accumulator (output 1) */
} mpose_tap;

typedef struct {
  int n_taps;
  int oy, ox;                          /* output phase of the class */
  mpose_tap taps[MPOSE_MAX_TAPS];
} mpose_tap_class;

typedef struct {
  int B, IH, IW, Cin;                  /* input  (B, IH, IW, Cin) NHWC, Cin % 32 == 0 */
  int OH, OW, Cout0, Cout1;            /* outputs (B, OH, OW, CoutX) NHWC (storage channel counts) */
  int GH, GW;                          /* slot grid per image and class */
  int in_mul, out_mul;                 /* 1 .. 8 */
  int n_classes;
  int Npad0, Npad1;                    /* padded N of the packed weights for acc 0 / acc 1 */
  int in_ld;                           /* pixel stride (floats) of the input tensor, 0 = Cin (dense); a larger value */
  int out_ld0, out_ld1;                /*   reads/writes a channel slice of a wider (concatenated) NHWC tensor       */
  int in_mul_x, out_mul_x;             /* 0: in_mul / out_mul along x too */
  mpose_tap_class cls[MPOSE_MAX_CLASSES];
} mpose_conv_geom;

/* Per-group (column) operands of a conv launch. */
typedef struct {
  const float* in;                     /* input activations */
  const float* in_scale;               /* optional per-channel prologue a*x+b then ReLU (BN+ReLU of */
  const float* in_shift;               /*   models/margipose_model.py:31-32,34-35); NULL = identity;
                                        *   all groups of a launch alike; not with a fused shortcut   */
  const float* w0;                     /* packed weights (mpose_pack_weights layout: three bf16 planes) */
  const float* w1;                     /* packed weights of the second accumulator, or NULL */
  float* out0;
  float* out1;
  double* stats0;                      /* optional (Cout0, 2) sum / sum-of-squares accumulators */
  double* stats1;
  const float* mask_src;               /* epilogue ReLU-mask source (pre-BN activations), or NULL */
  const float* mask_scale;
  const float* mask_shift;
  const float* in1;                    /* MPOSE_CONV_SUM_INPUTS: input of the taps with acc == 1 (same shape as `in`) */
  /* Fused output stage (inference: BatchNorm with running statistics is a per-channel affine map, so a ResidualBlock needs
   * no elementwise pass at all):
   *     y0 = [relu](epi_scale0 * conv0 + epi_shift0) [+ add_scale * add_src + add_shift]      (MPOSE_CONV_EPI_RELU0)
   * y0 is written as fp32 NHWC to out0; the launch can also accumulate max |y0| into the amax slot out0_amax (what the next
   * MPOSE_CONV_F16X3 convolution needs).  Not combined with stats0 / mask_src / MPOSE_CONV_ACCUMULATE. */
  const float* epi_scale0;
  const float* epi_shift0;
  const float* add_src;                /* fp32 NHWC, same shape as out0 (the shortcut branch), or NULL */
  const float* add_scale;
  const float* add_shift;
  void* out0_planes;                   /* reserved (round 2's bf16 plane engine, removed in round 6): must be NULL */
  /* MPOSE_CONV_F16X3: largest magnitudes of the tensors as the K loop sees them -- `in` AFTER the in_scale / in_shift / ReLU
   * prologue and `in1` as activation amax SLOTS (see mpose_absmax), the torch-layout weights behind w0 / w1 as one float each
   * (mpose_weights_absmax).  A value below the true maximum overflows fp16 (inf / NaN results). */
  const float* in_amax;
  const float* in1_amax;
  const float* w0_amax;
  const float* w1_amax;
  float* out0_amax;                    /* optional, with the fused output stage of conv.hip's engine: amax slot of y0 */
  /* BatchNorm-backward sums of the CONSUMER of out0, taken while out0 is stored (conv.hip's engine; out0 = g is the gradient
   * w.r.t. relu(bn_a(red_a)) + bn_b(red_b), models/margipose_model.py:34-40): red_sums[n][4] += (sum g*m, sum g*m*red_a, sum g,
   * sum g*red_b) over the launch's pixels, m = [red_scale[n] * red_a + red_shift[n] > 0] -- what mpose_bn_bwd_reduce computes in
   * a pass of its own.  red_a / red_b: fp32 NHWC tensors of out0's shape.  Not combined with stats0 / stats1. */
  const float* red_a;
  const float* red_b;
  const float* red_scale;
  const float* red_shift;
  double* red_sums;
  /* Per-channel extremes of out0 (conv.hip's engine, with stats0): mm0[2n] / mm0[2n+1] accumulate (atomic max) the sortable
   * keys (mpose_float_key) of max out0[.., n] and of max -out0[.., n] over the launch's pixels; zero the array first (key 0 is
   * below every float).  mpose_bn_finalize turns them into the EXACT largest magnitude of relu(scale * out0 + shift) -- the
   * operand of the next MPOSE_CONV_F16X3 convolution -- without a measuring pass (an affine map followed by ReLU is monotone). */
  unsigned* mm0;
  /* Train-mode BatchNorm of out0 / out1 finished by the launch itself (conv.hip's engine, with stats0 / stats1): the LAST
   * workgroup of this group's launch to pass its statistics on runs the mpose_bn_finalize job(s) fin0 / fin1 point at
   * (`const mpose_bn_job*` in device memory; scale / shift / mean / invstd, the running-statistics update and, with mm0, the
   * amax of relu(bn(out0))) -- the next launch finds them done, no mpose_bn_finalize launch in between.  fin_count: one
   * zero-initialised unsigned per (launch, group) in device memory; it is zero again when the launch ends. */
  const void* fin0;
  const void* fin1;
  unsigned* fin_count;
  float fin_eps, fin_momentum;
} mpose_conv_operands;

#define MPOSE_CONV_ACCUMULATE 1   /* out0 += result */
#define MPOSE_CONV_PLANES_IN 4    /* reserved: round 2's plane engine on three bf16 planes (conv_p.hip) left the library in round 6 --
                                   * MPOSE_CONV_H2_IN is its successor on two fp16 planes; the flag is rejected (MPOSE_EINVAL) */
#define MPOSE_CONV_EPI_RELU0 16    /* with the fused output stage of mpose_conv_operands: ReLU after epi_scale0 / epi_shift0 */
#define MPOSE_CONV_BF16 8         /* reserved (the plane engine's single-pass bf16 mode): rejected; configs[4]'s mode is MPOSE_CONV_F16X1 */
#define MPOSE_CONV_F16X3 32       /* fp32 convolution as THREE fp16 MFMA products instead of six bf16 ones (conv.hip engine only).
                                   * x * 2^k = h + l with two fp16 values (11 significant bits each, |x - (h+l) 2^-k| <= 2^-22 |x|),
                                   * k = 141 - biased_exponent(amax) (clamped to [-113, 114]) so that the tensor's largest
                                   * magnitude lands in [2^14, 2^15); a*b ~= ah*bh + ah*bl + al*bh (products exact in the fp32
                                   * accumulator, the dropped al*bl <= 2^-22 |ab|); the accumulator is scaled back by
                                   * 2^-(ka+kb) (exact).  gfx950's fp16 MFMA honours subnormals (tools/probe/f16_probe), so elements
                                   * far below amax degrade gradually (absolute error 2^-25 * 2^-k).  Measured against fp64 the
                                   * result is as close as the six-product bf16 form (tests/test_conv_gpu.py) at half the matrix
                                   * work.  Needs ops[i].in_amax / w0_amax (+ in1_amax / w1_amax with a second input / weight set)
                                   * and weights packed with layout 2. */
#define MPOSE_CONV_F16X1 64       /* with MPOSE_CONV_F16X3 (same operands, amax slots, packed weights and scales): multiply the h pieces
                                   * only -- operands ROUNDED to fp16 (11 significant bits) after the per-tensor power-of-two scale,
                                   * one MFMA product per multiply-add, fp32 accumulation.  The reduced-precision mode of BASELINE
                                   * configs[4] ("fp16 convs with MFMA"), NOT fp32-equivalent; BatchNorm, losses, soft-argmax stay fp32. */
#define MPOSE_CONV_H2_IN 128      /* with MPOSE_CONV_F16X3 (same arithmetic, scales and amax operands): `in` / `in1` are PRODUCER-SPLIT
                                   * activations -- the two fp16 planes (h, l) of x * 2^k, k = the scale exponent of the tensor's
                                   * amax slot, in the blocked layout H8[Cin/8][plane 2][B*IH*IW][8] (4 bytes per element;
                                   * mpose_split_h2 and the *_h2 outputs of the BatchNorm kernels write it) -- and `w0` / `w1` are
                                   * packed with layout 3; in_scale must be NULL, in_ld 0.  Runs conv_h.hip's engine: both operands
                                   * reach LDS by DMA, two workgroups per CU, no operand arithmetic in the K loop, two accumulators
                                   * per output block (h x h products / cross products) instead of a split-K.  The slot of such a
                                   * tensor holds a BOUND on its largest magnitude that its producer and every consumer read (never
                                   * a measured maximum: the producer needs the scale before it writes).  A bound B >= amax costs
                                   * no precision while B / amax < 2^12 or so: an element keeps 22 significant bits down to 2^-18
                                   * of B and an absolute error of 2^-40 B below that.
                                   * Epilogues: stats0 / stats1, mm0, mask_src, red_*, out0_amax (max |out0| as stored, also
                                   * without the fused output stage); not: epi_*, add_*, out0_planes, fin*, MPOSE_CONV_ACCUMULATE. */
#define MPOSE_CONV_STATS_PART 256  /* stats0 / stats1 / red_sums / mm0 are per-WORKGROUP partial buffers of fp32, WRITTEN with plain stores
                                   * instead of accumulated with fp64 atomics (768 workgroups x 256 device-scope atomics cost a
                                   * 128-channel launch 20 us of its ~110; round 4):
                                   *     stats0 / stats1 -> float [4 + rows*Cout*2]:  header, then [rows][Cout][2] (sum, sum of squares |
                                   *                                                  sum d, sum d*mask_src)
                                   *     red_sums        -> float [4 + rows*Cout0*4]: header, then [rows][Cout0][4]
                                   *     mm0             -> float [4 + rows*Cout0*2]: header, then [rows][Cout0][2] (max v, max -v:
                                   *                                                  plain floats, -inf = no pixel)
                                   * The launch writes `rows` (an int) into the first header word; rows = mpose_conv_stat_rows() of
                                   * the same call = its workgroups along the pixel axis (<= ceil(B*GH*GW / 64) * n_classes: size
                                   * the buffers with that, 16-byte aligned).  Every row of every channel the launch covers is
                                   * written (nothing to zero).  mpose_bn_finalize / mpose_bn_bwd_coef add the rows in a fixed order
                                   * (jobs' `part` fields): deterministic.  Not with fin* or MPOSE_CONV_PLANES_IN. */
#define MPOSE_CONV_SUM_INPUTS 2   /* taps with acc == 1 read `in1` through `w1` and add into out0 (one pass, one
                                   * output): the data-gradient of a ResidualBlock's input, dX = conv_in^T(dC1) +
                                   * shortcut^T(dSC), models/margipose_model.py:39 */

/* Conv forward / data-gradient.  flags: MPOSE_CONV_* bits; bit0 = accumulate into out0 (out0 += result);
 * when mask_src != NULL the result is multiplied by [mask_scale*mask_src+mask_shift > 0] before it
 * is stored, and stats0 receives (sum d, sum d*mask_src) instead of (sum, sum of squares). */
int mpose_conv_fwd(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups,
                   int flags, void* stream);
/* Rows of the MPOSE_CONV_STATS_PART buffers that exactly this call of mpose_conv_fwd would write (nothing is launched;
 * depends on the tile the library picks for the geometry, operands and flags), or a negative error code. */
int mpose_conv_stat_rows(const mpose_conv_geom* geom, const mpose_conv_operands* ops, int n_groups, int flags);

/* Weight-gradient of the same family: for every tap slice `widx`,
 *   dWp[split][widx][Cin/4][Npad][4] = sum over the split's slots of in(tap-shifted)^T * gout.
 * `gout` is indexed like an output of the forward geometry (acc 0 taps use gout0, acc 1 gout1). */
typedef struct {
  const float* in;
  const float* in_scale;
  const float* in_shift;
  const float* gout0;
  const float* gout1;
  float* dw0;                          /* (n_split, n_widx0, Cin/4, Npad0, 4) partial sums */
  float* dw1;
  /* When in_amax != NULL (all groups alike) the launch runs the three-product fp16 form (see MPOSE_CONV_F16X3): amax slots
   * (see mpose_absmax) of `in` (after the prologue), gout0 and gout1. */
  const float* in_amax;
  const float* gout0_amax;
  const float* gout1_amax;
  int single_product;                  /* with in_amax (all groups alike): MPOSE_CONV_F16X1's arithmetic -- the h x h product only */
  int planes_in;                       /* with in_amax (all groups alike): `in`, gout0 and gout1 are H8 plane tensors (see MPOSE_CONV_H2_IN:
                                        * [C/8][plane 2][pixel][8] fp16 of x * 2^k, k from the tensor's amax slot as its producer read it)
                                        * instead of fp32 NHWC -- stride-1 geometries of whole 32-channel groups, GW % 16 == 0, no
                                        * prologue (the producer applied it); anything else is MPOSE_EINVAL */
} mpose_wgrad_operands;

/* Number of (tap, input-channel tile, output-channel tile) work units of one group's weight-gradient launch;
 * the launch runs units * n_groups * n_split workgroups, one per CU at a time (the caller sizes n_split so that
 * this is close to a multiple of 256, and the partial-sum buffer as n_split * packed-fp32 weight size). */
int mpose_conv_wgrad_tiles(const mpose_conv_geom* geom);
/* Workgroups of that launch that share a CU (1, or 3 for the narrow tiles of the row-of-taps kernel): a round of the launch is
 * 256 * this many workgroups. */
int mpose_conv_wgrad_occupancy(const mpose_conv_geom* geom);
/* Waves per workgroup of that launch: 4, or 1 / 2 for the 32-channel tiles of the row-of-taps kernel (32 -> 32, 32 -> 64: the
 * feature extractor's first layers) -- a workgroup there takes a quarter / half of a wide one's footprint, so the caller may
 * split the pixels 4 / 2 times further for the same share of the chip. */
int mpose_conv_wgrad_waves(const mpose_conv_geom* geom);

/* d > 1: the geometry is a stride-1 convolution dilated by d along x (every tap's dx a multiple of d), whose weight gradient
 * mpose_conv_wgrad computes as d launches over the residues of x mod d -- IF n_split is a multiple of d; each residue then
 * writes n_split / d of the n_split partials, and mpose_conv_wgrad_tiles counts the units of ONE residue's launch over
 * B * GH * GW / d slots.  1: no such decomposition (size n_split as usual). */
int mpose_conv_wgrad_phases(const mpose_conv_geom* geom);

int mpose_conv_wgrad(const mpose_conv_geom* geom, const mpose_wgrad_operands* ops, int n_groups,
                     int n_split, void* stream);

/* Batched weight (re)packing and gradient un-packing; jobs live in device memory. */
typedef struct {
  const float* src;                    /* torch-layout weight */
  float* dst;                          /* packed 16-bit planes, at most 1.5 floats per element:
                                        *   layout 0: [T][Kpad/16][3 (hi,mid,lo)][Npad][2][8] bf16  (conv.hip: fragments from L2)
                                        *   layout 1: [T][Kpad/16][3][2 (k half)][Npad][8] bf16     (conv_p.hip: B tiles by DMA)
                                        *   layout 2: [T][Kpad/16][2 (h,l)][Npad][2][8] fp16 of w * 2^k(*amax)   (MPOSE_CONV_F16X3)
                                        *   layout 3: [T][Kpad/16][2 (h,l)][2 (k half)][Npad][8] fp16, same values  (MPOSE_CONV_H2_IN) */
  int N, K, T, Npad, Kpad;
  int64_t sn, sk, st;                  /* element strides of n, k, tap in src */
  int layout;
  float* amax;                         /* layout 2: max |src| (N*K*T contiguous floats), written by mpose_weights_absmax */
} mpose_pack_job;

/* *jobs[i].amax = max |jobs[i].src[0 .. N*K*T)| for every job with amax != NULL (one workgroup per job; before mpose_pack_weights). */
int mpose_weights_absmax(const mpose_pack_job* jobs_dev, int n_jobs, void* stream);

/* Largest magnitude of up to MPOSE_ABSMAX_MAX activation tensors (npix x C fp32, NHWC dense) as a convolution's K loop will
 * see them: max |[relu](scale[c] * src + shift[c])|, scale NULL = identity.
 * An activation amax SLOT is MPOSE_AMAX_SUBSLOTS floats spaced MPOSE_AMAX_STRIDE floats apart (4 KiB per slot); the tensor's
 * largest magnitude is the maximum over the sub-slots (consumers take it).  Producers accumulate with an atomic max on the
 * float's bit pattern, workgroup b into sub-slot b % MPOSE_AMAX_SUBSLOTS (thousands of same-address atomics serialise in L2):
 * zero the slot before the first launch that targets it.  The *_amax outputs of mpose_bn_add_fwd / mpose_bn_bwd_apply are such
 * slots too. */
#define MPOSE_AMAX_SUBSLOTS 16
#define MPOSE_AMAX_STRIDE 64
#define MPOSE_ABSMAX_MAX 6
typedef struct {
  const float* src;
  const float* scale;
  const float* shift;
  float* dst;
} mpose_absmax_operands;
int mpose_absmax(const mpose_absmax_operands* ops, int n_tensors, int64_t npix, int C, int relu, void* stream);

int mpose_pack_weights(const mpose_pack_job* jobs_dev, int n_jobs, int max_elems_per_job,
                       void* stream);

typedef struct {
  const float* src;                    /* packed partial sums (n_split, T, Kpad/4, Npad, 4), 16-byte aligned */
  float* dst;                          /* torch-layout gradient */
  int N, K, T, Npad, Kpad, n_split;
  int64_t sn, sk, st;
  int accumulate;                      /* dst += (like autograd) or dst = */
} mpose_unpack_job;

int mpose_unpack_wgrads(const mpose_unpack_job* jobs_dev, int n_jobs, int max_elems_per_job,
                        void* stream);

/* Producer-split activations for MPOSE_CONV_H2_IN (csrc/split.hip): planes = the two fp16 pieces of
 * [relu](scale * src + shift) * 2^k, k = f16 scale exponent of max over the sub-slots of `amax` (a BOUND the caller guarantees),
 * layout H8[C/8][2][npix][8]; mpose_h2_bytes gives the buffer size (= the fp32 tensor's). */
typedef struct {
  const float* src;
  const float* scale;
  const float* shift;
  void* planes;
  const float* amax;
} mpose_split_h2_operands;
int64_t mpose_h2_bytes(int64_t npix, int C);
int mpose_split_h2(const mpose_split_h2_operands* ops, int n_groups, int64_t npix, int C, int relu, void* stream);

/* BatchNorm pieces (models/margipose_model.py:31,34,37; train = batch statistics, biased variance,
 * eps 1e-5; running update momentum 0.1 with unbiased variance). */
typedef struct {
  const double* stats;                 /* (C, 2) sum, sumsq from the producing conv, or NULL in eval */
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* scale;                        /* out: gamma * invstd */
  float* shift;                        /* out: beta - mean * scale */
  float* mean;                         /* out (train): batch mean */
  float* invstd;                       /* out (train): 1/sqrt(var + eps) */
  int C;
  int count;                           /* B*H*W of the normalised tensor */
  const float* conv_bias;              /* optional bias of the producing conv (folded: the conv kernels are bias-free) */
  float eps;                           /* 0 = the launch-wide default */
  int pad_;
  const unsigned* minmax;              /* optional (C, 2) extremes of the normalised tensor (mpose_conv_operands.mm0) ...           */
  float* amax_out;                     /* ... -> max over channels of max(0, scale*max + shift, scale*min + shift), accumulated     */
                                       /*     (atomic max) into sub-slot 0 of this activation amax slot (see mpose_absmax)          */
  /* MPOSE_CONV_STATS_PART: when part != NULL the statistics are the sums over the rows of that buffer (header + [rows][part_ld][2],
   * at most n_part rows) in place of `stats`, and when mm_part != NULL the extremes are the maxima over the rows of that buffer
   * in place of `minmax` (amax_out is ACCUMULATED with an atomic max like everywhere else -- a job's channels may be split over
   * several workgroups: zero the slot first). */
  const float* part;
  const float* mm_part;
  int n_part, part_ld;
  /* MPOSE_CONV_H2_IN producers need a tensor's scale BEFORE they write it: when bound_out != NULL (train only) the job also writes
   *     max over channels of (|gamma| + |gamma2|) * sqrt(count) + |beta| + |beta2|          (bound_gamma2 / bound_beta2 may be NULL)
   * into sub-slot 0 of the amax slot bound_out (accumulated with an atomic max: zero the slot first) -- a bound on
   * |[relu](bn(x))| + |bn2(y)|, the ResidualBlock's output sum, that needs no look at the data: a value normalised with its
   * own batch statistics is at most sqrt(count - 1) standard deviations from the mean. */
  const float* bound_gamma2;
  const float* bound_beta2;
  float* bound_out;
} mpose_bn_job;

/* For every job: derive scale/shift (+ mean/invstd); train bit 0: batch statistics (also updates the running stats);
 * train bit 1: the jobs carry MPOSE_CONV_STATS_PART rows (`part`; the launch then runs 1024 threads per job);
 * train bit 2: write the jobs' bounds (bound_out). */
int mpose_bn_finalize(const mpose_bn_job* jobs_dev, int n_jobs, int train, float eps,
                      float momentum, void* stream);

/* out = relu(a_scale*a + a_shift) + (b_scale*b + b_shift): the tail of ResidualBlock.forward
 * (models/margipose_model.py:34-40 -- second BN + ReLU of the main branch, BN of the shortcut, add).  layout: 0 = NHWC out with C channels, 1 = NCHW out keeping
 * the first `c_keep` channels (the column's logits), 2 = NHWC without the ReLU on branch a (both branches plain BatchNorm:
 * a torchvision ResNet block with a downsample path, before its post-add ReLU). */
typedef struct {
  const float* a; const float* a_scale; const float* a_shift;
  const float* b; const float* b_scale; const float* b_shift;
  float* out;
  float* out_amax;                     /* optional (layouts 0 and 2): amax slot (see mpose_absmax) accumulating max |out|, for MPOSE_CONV_F16X3 */
} mpose_bn_add_operands;

int mpose_bn_add_fwd(const mpose_bn_add_operands* ops, int n_groups, int pixels_per_image, int B,
                     int C, int layout, int c_keep, void* stream);
/* mpose_bn_add_fwd (layout 0) that also writes the sum as the two fp16 planes of MPOSE_CONV_H2_IN, h2[i], scaled as the amax
 * slot ops[i].out_amax prescribes -- which is READ here (a bound somebody else wrote, see mpose_bn_job.bound_out), never
 * measured.  ops[i].out may be NULL (planes only). */
int mpose_bn_add_h2(const mpose_bn_add_operands* ops, void* const* h2, int n_groups, int64_t npix, int C, void* stream);

/* The last ResidualBlock's residual sum (mpose_bn_add_fwd layout 1: models/margipose_model.py:34-40) fused with flat_softmax + dsnt:
 * heatmaps[g] (B, J, H, W) = flat_softmax(relu(a_scale*a + a_shift) + (b_scale*b + b_shift)) over the first J of the C NHWC channels
 * of ops[g].a / ops[g].b, plane_coords (n_groups, B*J, 2) = dsnt(heatmaps) (may be NULL); the logits never reach memory, and the
 * heatmaps are bit-identical to the two-launch path's.  io_dtype: 0 fp32 heatmaps, 2 bf16 heatmaps.  H*W <= 4096, W % 4 == 0, C % 4 == 0.
 * Two forms, chosen by the launcher: one workgroup per (image, group, four joints) while the inputs are small (they are re-read
 * from the Infinity Cache by the five joint groups), one per (image, group) with all J rows in LDS -- every 128-byte channel line
 * read once, eight lanes per line -- beyond 64 MB of inputs when C == 32 and J * (H*W + 4) * 4 bytes fit in 144 KB of LDS. */
int mpose_bn_add_softmax_fwd(const mpose_bn_add_operands* ops, void* const* heatmaps, float* plane_coords, int n_groups,
                             int B, int H, int W, int C, int J, int io_dtype, void* stream);
/* xyz (rows, 3) from plane_coords (3, rows, 2): MargiPoseModel.heatmaps_to_coords' merge, z = (zy.x + xz.y) / 2
 * (models/margipose_model.py:254-261). */
int mpose_coords_merge(const float* plane_coords, float* xyz, int rows, void* stream);


/* out = relu(x*scale + shift) over an NHWC tensor of n elements with C channels (the stem's BN+ReLU,
 * materialised because every column of every stage reads it), and the ReLU backward gm = g*[y>0]. */
int mpose_bn_relu_fwd(const float* x, const float* scale, const float* shift, float* out, int64_t n,
                      int C, void* stream);
int mpose_relu_bwd(const float* g, const float* y, float* gm, int64_t n, void* stream);

/* Backward reductions/applications of BatchNorm, see csrc/bn.hip for the algebra. */
typedef struct {
  const float* g;                      /* upstream gradient, NHWC */
  const float* a;                      /* pre-BN activation of branch a */
  const float* b;                      /* pre-BN activation of branch b, or NULL */
  const float* a_scale;                /* when non-NULL, branch a sees g * [a_scale*a + a_shift > 0] */
  const float* a_shift;                /*   (the ReLU that follows branch a's BatchNorm) */
  double* sums;                        /* (C, 4): sum ga, sum ga*a, sum g, sum g*b */
} mpose_bn_bwd_reduce_operands;

int mpose_bn_bwd_reduce(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image,
                        int B, int C, int layout, int c_keep, void* stream);
/* The same sums WRITTEN (not accumulated) through per-workgroup partial sums in a caller-provided workspace (no atomics,
 * deterministic order): what the engine uses. */
int64_t mpose_bn_bwd_reduce_ws_bytes(int n_groups, int pixels_per_image, int B, int C);
int mpose_bn_bwd_reduce_ws(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C,
                           void* workspace, int64_t workspace_bytes, void* stream);

typedef struct {
  const float* g;
  const float* a; const float* b;
  const float* coef_a;                 /* (4, C): da = ca0*ga + ca1*(a - ca3) + ca2, ga = g * relu-mask (see below); ca3 = mean */
  const float* coef_b;                 /* (4, C): db = cb0*g + cb1*(b - cb3) + cb2 */
  const float* a_scale;                /* optional ReLU mask of branch a, as in the reduce step */
  const float* a_shift;
  float* da; float* db;
  float* da_amax; float* db_amax;      /* optional: amax slots (see mpose_absmax) accumulating max |da| / |db|, for MPOSE_CONV_F16X3 */
} mpose_bn_bwd_apply_operands;

int mpose_bn_bwd_apply(const mpose_bn_bwd_apply_operands* ops, int n_groups, int pixels_per_image,
                       int B, int C, int layout, int c_keep, void* stream);
/* mpose_bn_bwd_apply that writes da as the two fp16 planes of MPOSE_CONV_H2_IN, da_h2[i], scaled as the amax slot
 * ops[i].da_amax prescribes (READ: the bound of mpose_bn_bwd_coef_job.bound_out); ops[i].da (fp32) may be NULL.
 * db_h2 == NULL: db is written as fp32 and its largest magnitude accumulated into ops[i].db_amax as mpose_bn_bwd_apply does.
 * db_h2 != NULL (all groups alike): db is written as planes too, db_h2[i], scaled as ops[i].db_amax prescribes (READ: a bound,
 * nothing is accumulated); ops[i].db (fp32) may then be NULL. */
int mpose_bn_bwd_apply_h2(const mpose_bn_bwd_apply_operands* ops, void* const* da_h2, void* const* db_h2, int n_groups, int64_t npix,
                          int C, void* stream);

typedef struct {
  const double* sums;                  /* (Cs, 3) from mpose_bn_bwd_reduce, or (Cs, 2) from a conv epilogue */
  const float* gamma; const float* mean; const float* invstd;
  float* coef;                         /* out (4, c_stride): c0, c1, c2, mean */
  float* dgamma; float* dbeta;         /* out (written, not accumulated), may be NULL */
  int sums_stride;                     /* 4 (mpose_bn_bwd_reduce) or 2 (conv epilogue) */
  int which;                           /* column of `sums` holding sum g*x for this BN */
  int C;                               /* logical channels */
  int c_stride;                        /* storage channels (row length of coef) */
  int count;
  int sg_col;                          /* column of `sums` holding sum g for this BN */
  float* dconv_bias;                   /* optional: gradient of the producing conv's bias (zero in train mode, where the
                                        * batch mean removes the bias; gamma*invstd*sum g with frozen statistics) */
  /* MPOSE_CONV_STATS_PART: when part != NULL (and bit 1 of mode is clear) the sums are the sums over the rows of that buffer
   * (header + [rows][part_ld][sums_stride], at most n_part rows), same columns as `sums`. */
  const float* part;
  int n_part, part_ld;
  /* MPOSE_CONV_H2_IN: when bound_out != NULL the job also accumulates (atomic max, sub-slot 0: zero the slot first)
   *     max over channels of |c0| * (gmax + |mean g| + sqrt(count) * |mean(g * xhat)|),  gmax = the maximum of the amax slot g_amax,
   * a bound on |c0*g + c1*(x - mean) + c2| -- the gradient mpose_bn_bwd_apply is about to write -- into the amax slot bound_out. */
  const float* g_amax;
  float* bound_out;
} mpose_bn_bwd_coef_job;

/* mode: bit 0 = eval_mode (running statistics were constants), bit 1 = read `sums` even where a job has `part`,
 * bit 2 = 1024 threads per job (launches whose jobs carry partial rows), bit 3 = write the jobs' bounds (bound_out). */
int mpose_bn_bwd_coef(const mpose_bn_bwd_coef_job* jobs_dev, int n_jobs, int mode, void* stream);
/* mpose_bn_bwd_reduce_ws followed by mpose_bn_bwd_coef(jobs, n_coef_jobs, coef_mode) -- coef_mode: bit 0 = eval_mode only; jobs with
 * sums_stride 4 whose `sums` point into the sums of one of the reduction's groups -- as two launches instead of three: the
 * reduction's finishing pass runs the coefficient jobs of the channels whose sums it has just completed (same results). */
int mpose_bn_bwd_reduce_coef_ws(const mpose_bn_bwd_reduce_operands* ops, int n_groups, int pixels_per_image, int B, int C,
                                void* workspace, int64_t workspace_bytes, const mpose_bn_bwd_coef_job* coef_jobs_dev,
                                int n_coef_jobs, int coef_mode, void* stream);

/* 3x3 pooling over NHWC with the producer's BN+ReLU applied on the fly (scale/shift may be NULL = identity).
 * kind 0: max pool, stride 2, pad 1 (the reference rewrites MaxPool2d padding to k//2, models/margipose_model.py:111-117);
 * kind 1: average pool, stride 1, pad 1, count_include_pad = False.  Output goes to a channel slice (out_ld). */
int mpose_pool3_fwd(const float* in, const float* scale, const float* shift, float* out, int B, int IH, int IW, int C,
                    int out_ld, int kind, void* stream);
/* Backward: adds into d_in (gradient w.r.t. the ACTIVATED input); g is read from a channel slice (g_ld). */
int mpose_pool3_bwd(const float* in, const float* scale, const float* shift, const float* g, float* d_in, int B, int IH,
                    int IW, int C, int g_ld, int kind, void* stream);
/* The max-pool case (kind 0) of the above in two passes through a caller-provided workspace of >= B*OH*OW*C bytes
 * (OH = (IH-1)/2+1): window arg-max positions are computed once per output element instead of up to four times per
 * input element.  Same result bit for bit. */
int mpose_maxpool3_bwd_ws(const float* in, const float* scale, const float* shift, const float* g, float* d_in,
                          void* workspace, long workspace_bytes, int B, int IH, int IW, int C, int g_ld, void* stream);
/* The same with the arg-max positions kept from the FORWARD pass: mpose_maxpool3_fwd_arg is mpose_pool3_fwd(kind 0) that also
 * writes one byte per pooled element (>= B*OH*OW*C bytes; first maximum in row-major order), mpose_maxpool3_bwd_arg the second
 * pass of mpose_maxpool3_bwd_ws reading them -- the backward pass no longer re-reads the pre-pool tensor.  Same results. */
int mpose_maxpool3_fwd_arg(const float* in, const float* scale, const float* shift, float* out, void* arg_out, long arg_bytes,
                           int B, int IH, int IW, int C, int out_ld, void* stream);
int mpose_maxpool3_bwd_arg(const float* g, const void* arg, long arg_bytes, float* d_in, int B, int IH, int IW, int C, int g_ld,
                           void* stream);
/* NCHW (B, C, H, W) -> NHWC (B, H, W, Cpad) zero padded, and the reverse gather for the input gradient. */
int mpose_image_to_nhwc(const float* x, float* out, int B, int C, int H, int W, int Cpad, void* stream);
int mpose_nhwc_to_image(const float* g, float* dx, int B, int C, int H, int W, int Cpad, void* stream);
/* uint8 RGB frames (B,3,H,W) -> (x/255 - mean[c]) / std[c] in fp32 (`ImageSpecs.convert`, data_specs.py:6-13,38-39; mean3 /
 * std3 are HOST arrays of 3 floats): NHWC zero-padded to Cpad channels (Cpad % 4 == 0, the InceptionV4 stem's first load),
 * or NCHW when Cpad == 0. */
int mpose_frames_u8(const unsigned char* frames, const float* mean3, const float* std3, float* out, int B, int H, int W,
                    int Cpad, void* stream);
/* First layer of the InceptionV4 stem (Conv2d(3, 32, 3, stride 2, padding 1), pretrainedmodels InceptionV4 features[0] with
 * the reference's forced k//2 padding, models/margipose_model.py:111-117) as a gather + 1x1 convolution: patches
 * (B, H/2, W/2, 32) with channel q = c*9 + ky*3 + kx (27..31 zero) from an fp32 NCHW image or (is_u8) uint8 frames
 * normalised as in mpose_frames_u8; and the gradient of the gather, patches-gradient -> image gradient (B, 3, H, W). */
int mpose_im2col_k3s2(const void* img, int is_u8, const float* mean3, const float* std3, float* out, int B, int H, int W,
                      void* stream);
int mpose_col2im_k3s2(const float* dpatches, float* dx, int B, int H, int W, void* stream);
/* The same gather for any odd k (stride 2, padding k/2) into Cpad >= 3*k*k channels (Cpad % 4 == 0), q = (c*k + ky)*k + kx:
 * torchvision ResNet's conv1 = Conv2d(3, 64, 7, stride 2, padding 3) (models/margipose_model.py:119-137) as a 1x1
 * convolution with K = 147 (+13 zero). */
int mpose_im2col_s2(const void* img, int is_u8, const float* mean3, const float* std3, float* out, int B, int H, int W,
                    int k, int Cpad, void* stream);
int mpose_col2im_s2(const float* dpatches, float* dx, int B, int H, int W, int k, int Cpad, void* stream);

/* Layout / glue kernels. */
/* NCHW image (B,3,S,S) -> NHWC space-to-depth (B, S/8, S/8, 192) for the patch8 stem, and back. */
int mpose_space_to_depth8(const float* x, float* out, int B, int S, void* stream);
int mpose_depth_to_space8(const float* g, float* dx, int B, int S, void* stream);

/* The column's axis permutation (models/margipose_model.py:91-97) on NHWC (B,S,S,C) tensors.
 * space: 1 = zy, 2 = xz.  The permutation is an involution, so backward = forward. */
int mpose_axis_permute(const float* const* in, float* const* out, const int* spaces, int n_groups,
                       int B, int S, int C, void* stream);

/* HeatmapCombiner (models/margipose_model.py:142-150) + the cumulative add (:195):
 *   out[b,y,x,c] = inp[b,y,x,c] + sum_{p,j} W[c, p*J+j] * hm[p][b,j,y,x]   (hm NCHW, inp/out NHWC) */
int mpose_combiner_fwd(const float* const* hm, const float* w, const float* inp, float* out, int B,
                       int J, int HW, int C, void* stream);
/* The same with bf16 heatmaps (inference storage mode, mpose_softmax_dsnt_fwd io_dtype 2). */
int mpose_combiner_fwd_bf16(const void* const* hm, const float* w, const float* inp, float* out, int B,
                            int J, int HW, int C, void* stream);
/* d_hm[p][b,j,y,x] = sum_c W[c,p*J+j] * g[b,y,x,c];  dw partial sums (n_blocks, C, 3J). */
int mpose_combiner_bwd(const float* const* hm, const float* w, const float* g, float* const* d_hm,
                       float* dw_partial, int n_partial, int B, int J, int HW, int C, void* stream);

/* out = a + b elementwise (gradient fan-in), n % 4 == 0. */
int mpose_add(const float* a, const float* b, float* out, int64_t n, void* stream);

/* (B, J, P) NCHW -> (B, P, Cpad) NHWC with channels >= J zero-filled (logits gradient entering the
 * last ResidualBlock's backward). */
int mpose_nchw_to_nhwc_pad(const float* const* in, float* const* out, int n_groups, int B, int J,
                           int P, int Cpad, void* stream);

/* dst[i] (+)= sum_p src[p*n + i]  (second stage of deterministic block-partial reductions). */
int mpose_reduce_partials(const float* src, float* dst, int n_partial, int64_t n, int accumulate,
                          void* stream);

/* torch.optim.SGD(lr, momentum).step() over all parameters in one launch (bin/train_3d.py:186,339), hyper-parameters read
 * from DEVICE memory: hyper_dev = {lr, momentum, first} (first != 0: momentum buffers are initialised with the gradient, as
 * torch does on its first step).  p, g, buf: 16-byte aligned device arrays of n floats; jobs live in device memory. */
typedef struct {
  float* p;
  const float* g;
  float* buf;
  int64_t n;
} mpose_sgd_job;
int mpose_sgd_step(const mpose_sgd_job* jobs_dev, int n_jobs, int64_t max_n, const float* hyper_dev, void* stream);
/* dst[0..3] = {a, b, c, d} on the stream (how the host hands the next step's hyper-parameters to a replayed graph). */
int mpose_set4(float* dst, float a, float b, float c, float d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Launch plans (csrc/plan.hip): the launches of a training iteration (reference bin/train_3d.py:154-186) recorded once and
 * re-issued from one C loop -- the eager schedule (same kernels, argument values, streams and cross-stream dependencies)
 * without its host cost.  Every launch of this library passes through one wrapper; between mpose_plan_begin and mpose_plan_end
 * it is also recorded.  `streams`: the HIP streams the iteration uses, streams[0] the main one; a launch or wait on any other
 * stream makes mpose_plan_end fail.  Recording is process-wide (autograd issues the backward pass from its own thread); one
 * recording at a time.  The recorded iteration executes normally.  A replay is valid while every buffer the recorded launches
 * touch lives at its recorded address, and only for what the LIBRARY launched: kernels of other libraries are not recorded.
 * mpose_plan_replay issues ops [first_op, ...) on `streams` (same count as recorded) until the end or the next break
 * (mpose_plan_break: a point where the host must act, e.g. issue a collective) and stores the index to continue from.
 * mpose_stream_wait(waiter, signaler): `waiter` waits for everything enqueued on `signaler` so far (event record + stream
 * wait); recorded like a launch.  None of these calls synchronises with the device. */
int mpose_plan_begin(void* const* streams, int n_streams);
int mpose_plan_recording(void);
int mpose_plan_end(void** plan_out);
int mpose_plan_abort(void);
int mpose_plan_break(void);
int mpose_plan_size(void* plan, int* n_launch, int* n_wait, int* n_break);
int mpose_plan_replay(void* plan, void* const* streams, int n_streams, int first_op, int* next_op);
int mpose_plan_destroy(void* plan);
int mpose_stream_wait(void* waiter, void* signaler);
/* What an iteration otherwise takes from the tensor library, as recordable launches: a 32-bit fill and a copy (n_bytes % 4 == 0),
 * p[i] += v over n device int64s (the BatchNorms' num_batches_tracked), out = a + b over n floats (a == NULL: 0 + b: the stage-loss sum of
 * models/margipose_model.py:238-252), and average_loss's backward d_losses = mask * (grad[0] / out2[1]) (dsntnn.py:99-121; out2 =
 * mpose_average_loss_fwd's {mean, denominator}; mask may be NULL). */
int mpose_fill_u32(void* dst, unsigned value, int64_t n_bytes, void* stream);
int mpose_copy_bytes(const void* src, void* dst, int64_t n_bytes, void* stream);
int mpose_add_i64(int64_t* p, int64_t v, int64_t n, void* stream);
int mpose_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
int mpose_copy_div_f32(const float* src, float* dst, float divisor, int64_t n, void* stream);      /* dst = src / divisor */
int mpose_average_loss_bwd(const float* grad, const float* out2, const float* mask, float* d_losses, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MARGIPOSE_HIP_H */
