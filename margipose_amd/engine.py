"""Launch sequencing of the MargiPose backbone on MI355X: the host-side "runtime" that turns
MargiPoseModelInner.forward (reference models/margipose_model.py:179-200) and its autograd backward
into a fixed sequence of gfx950 kernel launches through the C ABI (include/margipose_hip.h).

What lives here (and nowhere in torch): which kernel runs when, on which buffers; the packed-weight,
BatchNorm-statistics and coefficient arenas; the device-resident job tables; the convolution
geometries (tap lists).  PyTorch supplies device memory, the stream, and the autograd edge.

Data layout in HBM
  * activations: NHWC fp32, the three columns (xy, zy, xz) of a stage are separate tensors processed by
    one grouped launch (blockIdx.z = column);
  * 17-channel tensors of the last ResidualBlock are stored with 32 channels (zero padded);
  * weights: torch layout in the nn.Parameters (state_dict contract) + a packed arena
    [widx][K/4][Npad][4] rebuilt by ONE launch per step (fwd and dgrad flavours);
  * BatchNorm: per layer (sum, sumsq) fp64 accumulators filled by the producing conv's epilogue, and a
    float arena holding scale / shift / mean / invstd / backward coefficients.
"""
import ctypes
import os
import weakref

import numpy as np
import torch
import torch.distributed

from . import _lib
from ._lib import (BnAddOperands, BnBwdApplyOperands, BnBwdReduceOperands, AbsmaxOperands, ConvGeom, ConvOperands, SplitH2Operands, WgradOperands, c_int,
                   c_int64, c_void_p, check, lib, ptr, ptr_array, stream_ptr)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
TAPS3 = [(ky, kx) for ky in range(3) for kx in range(3)]


def _rup(a, b):
    return (a + b - 1) // b * b


# numpy mirrors of the device-resident job structs (checked against mpose_sizeof at start-up)
PACK_DT = np.dtype([('src', 'u8'), ('dst', 'u8'), ('N', 'i4'), ('K', 'i4'), ('T', 'i4'), ('Npad', 'i4'), ('Kpad', 'i4'),
                    ('sn', 'i8'), ('sk', 'i8'), ('st', 'i8'), ('layout', 'i4'), ('amax', 'u8')], align=True)
UNPACK_DT = np.dtype([('src', 'u8'), ('dst', 'u8'), ('N', 'i4'), ('K', 'i4'), ('T', 'i4'), ('Npad', 'i4'), ('Kpad', 'i4'),
                      ('n_split', 'i4'), ('sn', 'i8'), ('sk', 'i8'), ('st', 'i8'), ('accumulate', 'i4')], align=True)
BN_DT = np.dtype([('stats', 'u8'), ('gamma', 'u8'), ('beta', 'u8'), ('running_mean', 'u8'), ('running_var', 'u8'),
                  ('scale', 'u8'), ('shift', 'u8'), ('mean', 'u8'), ('invstd', 'u8'), ('C', 'i4'), ('count', 'i4'),
                  ('conv_bias', 'u8'), ('eps', 'f4'), ('pad_', 'i4'), ('minmax', 'u8'), ('amax_out', 'u8'),
                  ('part', 'u8'), ('mm_part', 'u8'), ('n_part', 'i4'), ('part_ld', 'i4'),
                  ('bound_gamma2', 'u8'), ('bound_beta2', 'u8'), ('bound_out', 'u8')], align=True)
COEF_DT = np.dtype([('sums', 'u8'), ('gamma', 'u8'), ('mean', 'u8'), ('invstd', 'u8'), ('coef', 'u8'), ('dgamma', 'u8'),
                    ('dbeta', 'u8'), ('sums_stride', 'i4'), ('which', 'i4'), ('C', 'i4'), ('c_stride', 'i4'), ('count', 'i4'),
                    ('sg_col', 'i4'), ('dconv_bias', 'u8'), ('part', 'u8'), ('n_part', 'i4'), ('part_ld', 'i4'),
                    ('g_amax', 'u8'), ('bound_out', 'u8')], align=True)

# MPOSE_GFLAT_FILL=2: the backward pass poisons the flat gradient buffer with NaNs instead of skipping its zero fill
# (tests/test_model_gpu.py::test_every_gradient_element_is_written); 1: always zero it.
_GFLAT_FILL = int(os.environ.get('MPOSE_GFLAT_FILL', '0'))
_WG_FIXED = 3.0              # fixed per-workgroup cost of a weight-gradient launch, in 128-pixel row units (Engine._n_split)
AMAX_SLOT = 16 * 64          # floats per activation amax slot (MPOSE_AMAX_SUBSLOTS * MPOSE_AMAX_STRIDE)
_SIZES_CHECKED = False


def _check_struct_sizes():
    global _SIZES_CHECKED
    if _SIZES_CHECKED:
        return
    L = lib()
    expect = {0: ctypes.sizeof(ConvGeom), 1: ctypes.sizeof(ConvOperands), 2: ctypes.sizeof(WgradOperands),
              3: PACK_DT.itemsize, 4: UNPACK_DT.itemsize, 5: BN_DT.itemsize, 6: COEF_DT.itemsize,
              7: ctypes.sizeof(BnAddOperands), 8: ctypes.sizeof(BnBwdReduceOperands), 9: ctypes.sizeof(BnBwdApplyOperands),
              10: ctypes.sizeof(SplitH2Operands), 12: ctypes.sizeof(AbsmaxOperands)}
    for which, size in expect.items():
        got = L.mpose_sizeof(which)
        if got != size:
            raise _lib.MposeError('ABI struct %d: library says %d bytes, binding says %d' % (which, got, size))
    _SIZES_CHECKED = True


# ---------------------------------------------------------------------------------------------
# geometry builders (see include/margipose_hip.h: mpose_conv_geom)
# ---------------------------------------------------------------------------------------------
def _geom(B, IH, Cin, OH, Cout0, Cout1, GH, in_mul, out_mul, classes, Npad0, Npad1=0):
    return _geom2(B, (IH, IH), Cin, (OH, OH), Cout0, Cout1, (GH, GH), (in_mul, in_mul), (out_mul, out_mul), classes, Npad0, Npad1)


def _geom2(B, in_hw, Cin, out_hw, Cout0, Cout1, grid_hw, in_mul, out_mul, classes, Npad0, Npad1=0):
    """mpose_conv_geom with separate (y, x) sizes and slot strides (include/margipose_hip.h)."""
    g = ConvGeom()
    g.B, g.IH, g.IW, g.Cin = B, in_hw[0], in_hw[1], Cin
    g.OH, g.OW, g.Cout0, g.Cout1 = out_hw[0], out_hw[1], Cout0, Cout1
    g.GH, g.GW = grid_hw
    g.in_mul, g.in_mul_x = in_mul
    g.out_mul, g.out_mul_x = out_mul
    g.n_classes = len(classes)
    g.Npad0, g.Npad1 = Npad0, Npad1
    for ci, (oy, ox, taps) in enumerate(classes):
        c = g.cls[ci]
        c.n_taps, c.oy, c.ox = len(taps), oy, ox
        for ti, (dy, dx, widx, acc) in enumerate(taps):
            c.taps[ti].dy, c.taps[ti].dx, c.taps[ti].widx, c.taps[ti].acc = dy, dx, widx, acc
    return g


def axis_taps(gather, k, stride, dilation, padding):
    """One axis of a convolution as slot classes: (in_mul, out_mul, [(output phase, [(input offset, kernel index), ...]), ...]).
    gather=True: the slots are the outputs of a strided Conv (out[o] = sum_k in[o*stride + k*dilation - padding] w[k]);
    gather=False: the slots are output positions of one phase of its transpose (ConvTranspose forward / Conv data-gradient):
    out[stride*q + r] = sum over the k with (r + padding - k*dilation) % stride == 0 of in[q + (r + padding - k*dilation)/stride] w[k]."""
    if gather:
        return stride, 1, [(0, [(i * dilation - padding, i) for i in range(k)])]
    classes = []
    for r in range(stride):
        classes.append((r, [((r + padding - i * dilation) // stride, i) for i in range(k) if (r + padding - i * dilation) % stride == 0]))
    return 1, stride, classes


def conv_classes(gather, kernel, stride, dilation, padding):
    """2-D product of axis_taps: ((in_mul_y, in_mul_x), (out_mul_y, out_mul_x), classes in mpose_conv_geom form).  Kernel index
    ky*kw + kx is the packed weight slice in both directions (a ConvTranspose2d's weight is stored (Cin, Cout, kh, kw))."""
    imy, omy, cy = axis_taps(gather, kernel[0], stride[0], dilation[0], padding[0])
    imx, omx, cx = axis_taps(gather, kernel[1], stride[1], dilation[1], padding[1])
    classes = [(py, px, [(dy, dx, iy * kernel[1] + ix, 0) for dy, iy in ty for dx, ix in tx]) for py, ty in cy for px, tx in cx]
    return (imy, imx), (omy, omx), classes


def conv_geom(kind, transposed, B, src_hw, cin_s, dst_hw, cout_s, kernel, stride, dilation, padding, npad):
    """Launch geometry of a Conv2d / ConvTranspose2d (src -> dst) with per-axis kernel, stride, dilation and padding:
    kind 'f' = forward (also the weight-gradient geometry: it pairs src pixels with dst pixels), 'd' = data-gradient
    (reads the gradient w.r.t. dst, writes the one w.r.t. src).  The slots of a Conv2d's forward and of a ConvTranspose2d's
    data-gradient are the SMALL side's pixels, gathering stride-spaced pixels of the large side; the other two launches
    enumerate the large side by output phase."""
    gather = (kind == 'f') != transposed
    in_mul, out_mul, classes = conv_classes(gather, kernel, stride, dilation, padding)
    small, large = (src_hw, dst_hw) if transposed else (dst_hw, src_hw)
    if not all(large[a] == small[a] * stride[a] for a in (0, 1)):
        raise _lib.MposeError('unsupported convolution extent %s -> %s (stride %s)' % (src_hw, dst_hw, stride))
    if kind == 'f':
        return _geom2(B, src_hw, cin_s, dst_hw, cout_s, 0, small, in_mul, out_mul, classes, npad)
    return _geom2(B, dst_hw, cout_s, src_hw, cin_s, 0, small, in_mul, out_mul, classes, npad)


def _up_classes(with_shortcut, single_tap=False):
    """Output-parity classes of a stride-2 transposed 3x3 (pad 1, output_padding 1):
    out[2y'+py] receives kernel rows ky with ky = (py+1) mod 2; input row y' + (py+1-ky)/2."""
    classes = []
    for py in range(2):
        for px in range(2):
            taps = []
            if not single_tap:
                for ky, kx in TAPS3:
                    if (py + 1 - ky) % 2 == 0 and (px + 1 - kx) % 2 == 0:
                        taps.append(((py + 1 - ky) // 2, (px + 1 - kx) // 2, ky * 3 + kx, 0))
            elif py == 0 and px == 0:
                taps.append((0, 0, 0, 0))
            if with_shortcut and py == 0 and px == 0:
                taps.append((0, 0, 0, 1))
            classes.append((py, px, taps))
    return classes


class _Conv:
    """One convolution's weights: torch-layout parameter + slots in the packed arena."""

    def __init__(self, param, transposed, k, cin, cout, cin_s, cout_s, stem=False):
        self.param, self.transposed, self.k = param, transposed, k
        self.cin, self.cout, self.cin_s, self.cout_s = cin, cout, cin_s, cout_s
        self.T = 1 if stem else k * k
        self.stem = stem
        self.npad_f = _rup(cout, 64)             # forward pack: N = cout, K = cin_s
        self.npad_d = _rup(cin, 64)              # dgrad pack:   N = cin,  K = cout_s
        # arena sized for the largest packing: three bf16 planes (hi, mid, lo: the patch8 stem's six-product form) = 6 bytes per
        # element = 1.5 floats; the three-product form packs two fp16 planes (4 bytes per element) into the same slot
        self.size_f = self.T * cin_s * self.npad_f * 3 // 2
        self.size_d = self.T * cout_s * self.npad_d * 3 // 2
        self.size_g = self.T * cin_s * self.npad_f          # one split-K partial of the weight gradient (fp32)
        self.off_f = self.off_d = self.off_g = -1
        self.column = False                      # a convolution of the stages' columns (not the feature extractor's)

    def strides(self, dgrad):
        """(sn, sk, st): element strides of (n, k, tap) in the torch-layout weight."""
        kk = getattr(self, 'kk', self.k * self.k)
        if self.stem:                            # (128, 3, 8, 8) seen as (128, 192) 1x1
            return (192, 1, 0) if not dgrad else (1, 192, 0)
        if not self.transposed:                  # (Cout, Cin, k, k)
            return (self.cin * kk, kk, 1) if not dgrad else (kk, self.cin * kk, 1)
        return (kk, self.cout * kk, 1) if not dgrad else (self.cout * kk, kk, 1)   # (Cin, Cout, k, k)


class _BN:
    def __init__(self, module, C, Cs):
        self.m, self.C, self.Cs = module, C, Cs
        self.f_off = self.s_off = -1             # offsets into the float / double arenas


_H2_CHANNELS = (128,)        # regular blocks of these widths run on the H2 engine (192 -> 192 at 16 x 16: 288 tiles on 512 slots, slower: DESIGN 4.5)


class _Block:
    """One ResidualBlock (reference models/margipose_model.py:25-40) of one column."""

    def __init__(self, rb, kind, cin, cout):
        self.kind = kind
        self.cin, self.cout = cin, cout
        self.cin_s, self.cout_s = _rup(cin, 32), _rup(cout, 32)
        tr = kind == 'up'
        self.conv_in = _Conv(rb.module[0].weight, tr, 3, cin, cout, self.cin_s, self.cout_s)
        self.conv2 = _Conv(rb.module[3].weight, False, 3, cout, cout, self.cout_s, self.cout_s)
        self.conv_sc = _Conv(rb.shortcut[0].weight, tr, 1, cin, cout, self.cin_s, self.cout_s)
        self.bn1 = _BN(rb.module[1], cout, self.cout_s)
        self.bn2 = _BN(rb.module[4], cout, self.cout_s)
        self.bns = _BN(rb.shortcut[1], cout, self.cout_s)
        # the H2 engine (csrc/conv_h.hip: producer-split fp16 planes) takes the 128-channel regular blocks: see Engine.h2
        self.h2 = kind == 'regular' and cin == cout and cin in _H2_CHANNELS


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are enqueued on (torch's current stream).
    Used by bench.py to measure per-kernel average durations inside the timed region."""

    def __init__(self, only=None):
        self.records = []          # (tag, start, end, work)
        self.only = None if only is None else set(only)      # selective: bracket the launches with these tags, leave the schedule alone

    @property
    def selective(self):
        return self.only is not None

    def start(self, tag=None):
        if self.only is not None and tag not in self.only:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, tag, start, work):
        if start is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((tag, start, ev, work))

    def calibrate(self, n=64):
        """Cost of an empty start/stop bracket (two event packets back to back on a busy stream), in ms.  REPORTED beside the
        averages (bench.py), not subtracted from them: a launch's duration is what the events say."""
        evs = []
        busy = torch.zeros(1 << 20, device='cuda')
        for _ in range(n):
            busy.add_(1.0)                       # keep the queue non-empty, as it is during a step
            a = torch.cuda.Event(enable_timing=True); a.record()
            b = torch.cuda.Event(enable_timing=True); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        self.bracket_cal_ms = ts[len(ts) // 2]
        return self.bracket_cal_ms

    def summary(self):
        """tag -> dict(n, total_ms, avg_us, work_per_launch).  Call after torch.cuda.synchronize()."""
        out = {}
        off = 0.0
        for tag, a, b, work in self.records:
            d = out.setdefault(tag, {'n': 0, 'total_ms': 0.0, 'work': 0.0})
            d['n'] += 1
            d['total_ms'] += max(0.0, a.elapsed_time(b) - off)
            d['work'] += work
        for d in out.values():
            d['avg_us'] = 1e3 * d['total_ms'] / d['n']
            d['work_per_launch'] = d['work'] / d['n']
        return out


def _geom_flops(g):
    """Algorithmic FLOPs of one group of a conv launch: 2 * slots * taps * Cin * Cout."""
    f = 0
    for ci in range(g.n_classes):
        for ti in range(g.cls[ci].n_taps):
            cout = g.Cout1 if g.cls[ci].taps[ti].acc else g.Cout0
            f += 2 * g.B * g.GH * g.GW * g.Cin * cout
    return f


class _Ctx(dict):
    """Per-forward context handed to the autograd node (a dict that can be weakly referenced)."""
    __slots__ = ('__weakref__',)


def _first_same(tensors, c):
    """Index of the first tensor that shares tensors[c]'s storage."""
    return next(k for k in range(len(tensors)) if tensors[k].data_ptr() == tensors[c].data_ptr())


def _jobs_to_device(arr, device):
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


class Engine:
    """Owns the launch plan of one MargiPoseModelInner instance."""
    COEF_ITEMSIZE = COEF_DT.itemsize

    def __init__(self, inner):
        self.inner = inner
        self.T = inner.n_stages
        self.J = inner.n_joints
        self.spaces = inner.spaces                 # e.g. (0, 1, 2): xy, zy, xz
        self.device = None
        self.stage_blocks = []                     # [stage][block index 0..9] -> list of 3 _Block
        cols = (inner.xy_hm_cnns, inner.zy_hm_cnns, inner.xz_hm_cnns)
        kinds = ['regular', 'regular', 'down', 'regular', 'regular', 'regular', 'regular', 'up', 'regular', 'regular']
        chans = [(128, 128), (128, 128), (128, 192), (192, 192), (192, 192), (192, 192), (192, 192), (192, 128), (128, 128),
                 (128, self.J)]
        for t in range(self.T):
            blocks = []
            for i in range(10):
                grp = []
                for c in range(3):
                    col = cols[c][t]
                    rb = col.down_layers[i] if i < 5 else col.up_layers[i - 5]
                    grp.append(_Block(rb, kinds[i], chans[i][0], chans[i][1]))
                blocks.append(grp)
            self.stage_blocks.append(blocks)
        self.combiners = [m.conv.weight for m in inner.hm_combiners]
        self._all_blocks = [b for st in self.stage_blocks for grp in st for b in grp]
        block_convs = [c for b in self._all_blocks for c in (b.conv_in, b.conv2, b.conv_sc)]
        # One arithmetic everywhere: an fp32 multiply-add as THREE fp16 products of two-way split, per-tensor-scaled operands
        # (MPOSE_CONV_F16X3 in include/margipose_hip.h), fp32 accumulation.  Two kernels serve it: conv_igemm_k (csrc/conv.hip)
        # reads fp32 activations and splits them in its K loop -- the feature extractor, the stride-2 / 192-channel / joint blocks,
        # inference with its fused epilogues -- and conv_h2r_k (csrc/conv_h.hip) reads operands PRE-split into fp16 planes by their
        # producers: the regular 128-channel blocks in training (see conv_mode_for).
        self.conv_f16x1 = False      # every convolution on fp16-ROUNDED operands, one product (MPOSE_CONV_F16X1; conv_dtype = float16)
        for c in block_convs:
            c.column = True
        block_bns = [n for b in self._all_blocks for n in (b.bn1, b.bn2, b.bns)]
        self.stem = None
        fe_name = getattr(inner, 'feature_extractor_name', 'patch8')
        if fe_name != 'patch8':
            from .stem import ChatterboxGraph, InceptionV4Stem, ResNetStem
            self.stem_conv = self.stem_bn = None
            self._convs, self._bns = [], block_bns           # filled right below (the stem needs `self` first)
            if fe_name == 'chatterbox':                      # the whole ChatterboxModel is one graph (no stages)
                self.stem = ChatterboxGraph(self, inner)
            else:
                self.stem = (InceptionV4Stem if fe_name == 'inceptionv4' else ResNetStem)(self, inner.in_cnn)
            self._convs = self.stem.convs + block_convs
        else:
            self.stem_conv = _Conv(inner.in_cnn[0].weight, False, 1, 192, 128, 192, 128, stem=True)
            self.stem_bn = _BN(inner.in_cnn[1], 128, 128)
            self._convs = [self.stem_conv] + block_convs
            self._bns = [self.stem_bn] + block_bns
        self._pack_epoch = 0         # counts pack_weights() calls (PlannedInference(frozen_weights=True) notices a repack by someone else)
        self.pack_frozen = False     # True: a forward whose engine mode the packed arena already holds does not repack (PlannedInference)
        self._geoms = {}
        self._tables = {}
        self._arena_key = None
        self._plist = None
        self._key_tensors = None
        self._bn_bound = []
        self.timer = None            # optional KernelTimer (bench.py)
        self._dp_works = []          # in-flight gradient all-reduces of the current backward pass (data parallel)
        self.side_stream = None      # weight-gradient GEMMs run here, off the data-gradient critical path
        # True: the backward pass hands autograd views of the flat gradient buffer itself instead of a per-step copy of it (51 us for
        # 178 MB at B = 32) -- the gradients are then valid until the NEXT backward pass overwrites them.  PlannedTrainStep sets it
        # while it records: its iteration reads the gradients (the optimiser) before it runs the next backward.
        self.grad_views = False
        self._side_keep = []         # tensors the side stream still reads (released at the next bucket boundary)
        # BatchNorm statistics (forward sums, channel extremes, backward sums) leave the convolution launches as per-workgroup
        # partial rows written with plain stores (MPOSE_CONV_STATS_PART) and are added up in a fixed order by the finalize /
        # coefficient kernels (round 4: the fp64 device-scope atomics they replace cost a 128-channel launch 20 us of ~110).
        #
        # Training / differentiable forwards run the regular 128-channel blocks on conv_h2r_k: forward, BOTH data gradients and
        # (round 6) both weight gradients read fp16 planes that the elementwise pass producing the tensor wrote ONCE -- the residual
        # sum of the block before (mpose_bn_add_h2), BatchNorm + ReLU (mpose_split_h2), the BatchNorm-backward applications
        # (mpose_bn_bwd_apply_h2: planes only, no fp32 copy) -- scaled by a BOUND on the tensor's magnitude that the finalize /
        # coefficient kernels derive before the pass runs (mpose_bn_job.bound_out, mpose_bn_bwd_coef_job.bound_out): no measuring,
        # no split pass, no operand arithmetic in any K loop.  (Batch statistics bound a normalised value; running statistics do
        # not: eval-mode forwards that will be differentiated measure and split.)
        # MPOSE_H2_PLANES=0: round 5's backward for those blocks (fp32 gradients beside the planes, the two-input data gradient and
        # the weight gradients on the fp32 kernels) -- same-box A/B runs only.
        self.h2_planes = os.environ.get('MPOSE_H2_PLANES', '1') != '0'
        # the last ResidualBlock's residual sum, flat_softmax and dsnt as ONE launch per stage (mpose_bn_add_softmax_fwd: an image's
        # logits stay in LDS); heatmaps and coordinates are bit-identical to the two-launch path (tail_fuse = False)
        self.tail_fuse = True
        # Training forward of the feature extractor: a node whose channels all come out of a BatchNorm takes its amax slot from
        # the finalize kernel's a-priori bound max_c(|gamma_c| sqrt(N) + |beta_c|) instead of a measuring pass (mpose_absmax): 12 of
        # the 16 passes per step.  The bound is 2^5-2^7 above the true maximum at these sizes; a two-piece fp16 operand keeps its 22
        # bits down to 2^-18 of the bound, and below that its ABSOLUTE error (bound * 2^-39) is far under fp32's at the tensor's
        # typical magnitude.  False: measure every node.
        self.stem_bounds = True
        # split-K partials summed right behind each weight-gradient launch, on its stream (unpack_after): the launch's 50 MB of
        # partials are read back out of the Infinity Cache instead of 0.65 GB per stage out of HBM on the main stream (round 5:
        # 21.96 -> 21.67 ms).  False: one unpack launch per stage bucket (bit-identical).
        self.inline_unpack = True
        # the coefficient jobs that read a BatchNorm-backward reduction's sums run in its finishing pass (bn_bwd_reduce): 22 launches
        # of pure latency per step fewer, bit-identical results
        self.fuse_coef = True
        self.overlap_wgrad = True    # +2.3 % step rate, bit-identical results; launches bracketed by a KernelTimer stay serial
        self.dp = None               # optional (process_group, world_size): gradient all-reduce after backward
        self.input_norm = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])    # uint8 frames: ImageSpecs mean / stddev
        # Several forwards may be alive before their backwards run (plain autograd allows it): the BatchNorm float
        # arenas (scale / shift / mean / invstd) belong to ONE forward at a time.  `_gen` counts forwards, `_arena_gen`
        # says whose values the arenas hold; a forward that is about to overwrite them first snapshots them into the
        # still-pending context, and that context's backward restores them (see _snapshot_pending / _restore_arenas).
        self._gen = 0
        self._arena_gen = 0
        self._pending = None         # weakref to the newest context that saved activations and has not run backward

    # ------------------------------------------------------------------ parameters / arenas
    def param_list(self):
        """Every learnable tensor, in the order grads are returned by the autograd Function."""
        if self._plist is not None:
            return self._plist
        ps = []
        for c in self._convs:
            ps.append(c.param)
        for n in self._bns:
            ps += [n.m.weight, n.m.bias]
        if self.stem is not None:
            for m in self.stem.bn_modules:
                ps += [m.weight, m.bias]
            ps += self.stem.extra_params
        ps += self.combiners
        if self._arena_key is not None:          # (the stem adds its parameters while the engine is being built: cache afterwards)
            self._plist = ps
        return ps

    def invalidate(self):
        self._arena_key = None
        self._plist = None
        self._key_tensors = None
        self._tables = {}

    def _bn_buffers(self):
        mods = [n.m for n in self._bns] + (self.stem.bn_modules if self.stem is not None else [])
        return [t for m in mods for t in (m.running_mean, m.running_var)]

    def _bn_bound_now(self):
        """The tensor objects bound to the BatchNorm modules right now (module dicts directly, not Module.__getattr__)."""
        out = []
        for m in [n.m for n in self._bns] + (self.stem.bn_modules if self.stem is not None else []):
            p, b = m._parameters, m._buffers
            out += [p.get('weight'), p.get('bias'), b.get('running_mean'), b.get('running_var')]
        return out

    def grad_layout(self):
        """Layout of the flat fp32 gradient buffer (pure host logic: tests/test_parallel_cpu.py checks it without a GPU).
        Returns (offsets aligned with param_list(), buckets as [lo, hi) float ranges, total floats).
        Buckets: the parameters of stage T-1 first (their gradients are complete first in the backward pass), ...,
        stage 0, then the stem: each bucket is one contiguous slice = one all-reduce that overlaps the rest of the backward pass."""
        order, buckets = [], []
        for t in reversed(range(self.T)):
            start = len(order)
            for grp in self.stage_blocks[t]:
                for b in grp:
                    order += [b.conv_in.param, b.conv2.param, b.conv_sc.param]
                    for n in (b.bn1, b.bn2, b.bns):
                        order += [n.m.weight, n.m.bias]
            if t > 0:
                order.append(self.combiners[t - 1])
            buckets.append((start, len(order)))
        seen = set(id(p) for p in order)
        start = len(order)
        order += [p for p in self.param_list() if id(p) not in seen]          # stem (+ anything not stage-owned)
        buckets.append((start, len(order)))
        goff, off_of, bounds = 0, {}, []
        for i, p in enumerate(order):
            off_of[id(p)] = goff
            goff += _rup(p.numel(), 4)
            bounds.append(goff)
        buckets = [((bounds[a - 1] if a > 0 else 0), (bounds[b - 1] if b > 0 else 0)) for a, b in buckets]
        return [off_of[id(p)] for p in self.param_list()], buckets, goff

    def _ensure_arenas(self, device):
        # every address baked into the device-resident job tables takes part in the key: a parameter or buffer that was
        # re-bound outside Module._apply (load_state_dict(assign=True), p.data = ..., swap_tensors) rebuilds the tables
        # (the tensor OBJECTS are cached; a BatchNorm parameter or buffer bound to a new object -- bn.running_mean = t,
        # load_state_dict(assign=True) -- is caught by the identity walk below: dict lookups, ~40 us per call)
        if self._key_tensors is not None:
            bound = self._bn_bound_now()
            if len(bound) != len(self._bn_bound) or any(a is not b for a, b in zip(bound, self._bn_bound)):
                self._plist = None
                self._key_tensors = None
        if self._key_tensors is None:
            self._key_tensors = self.param_list() + self._bn_buffers()
            self._bn_bound = self._bn_bound_now()
        key = (str(device), hash(tuple([t.data_ptr() for t in self._key_tensors])))
        if self._arena_key == key:
            return
        self._plist = None
        _check_struct_sizes()
        for p in self.param_list():
            _lib.dev_f32(p.data, 'parameter')
        self.device = device
        # packed weights (fwd + dgrad) and wgrad partial offsets
        off = 0
        for c in self._convs:
            c.off_f = off; off += c.size_f
            c.off_d = off; off += c.size_d
        self.wpack = torch.zeros(off, dtype=torch.float32, device=device)
        # float arena per BN: scale, shift, mean, invstd, coef[4] -> 8*Cs ; double arena: fwd stats 2*Cs + bwd sums 4*Cs
        foff = soff = 0
        for n in self._bns:
            n.f_off = foff; foff += 8 * n.Cs
            n.s_off = soff; soff += 7 * n.Cs         # (+ Cs doubles = 2*Cs sortable keys: the channel extremes of the forward)
        self.bnf = torch.zeros(foff, dtype=torch.float32, device=device)
        self.stat_arena = torch.zeros(soff, dtype=torch.float64, device=device)
        self._grad_offsets, self._buckets, self._grad_total = self.grad_layout()
        goff = self._grad_total
        self.gflat = torch.zeros(goff, dtype=torch.float32, device=device)
        # num_batches_tracked of every BN becomes a view of one int64 vector: one increment per step
        bn_mods = [n.m for n in self._bns] + (self.stem.bn_modules if self.stem is not None else [])
        self._nbt = torch.stack([m.num_batches_tracked.to(device) for m in bn_mods]).contiguous()
        for i, m in enumerate(bn_mods):
            m.num_batches_tracked = self._nbt[i]
        if self.stem is not None:
            self.stem.setup(device)
        # pack job table
        jobs = np.zeros(2 * len(self._convs), dtype=PACK_DT)
        mx = 0
        base = self.wpack.data_ptr()
        for i, c in enumerate(self._convs):
            for d in (0, 1):
                j = jobs[2 * i + d]
                sn, sk, st = c.strides(bool(d))
                j['src'] = c.param.data_ptr()
                j['dst'] = base + 4 * (c.off_d if d else c.off_f)
                j['N'], j['K'] = (c.cin, c.cout) if d else (c.cout, c.cin)
                j['T'] = c.T
                j['Npad'] = c.npad_d if d else c.npad_f
                j['Kpad'] = c.cout_s if d else c.cin_s
                j['sn'], j['sk'], j['st'] = sn, sk, st
                j['layout'] = 0
                mx = max(mx, int(j['T']) * int(j['Kpad']) * int(j['Npad']))
        # largest magnitudes (MPOSE_CONV_F16X3): one float per convolution weight, per block (cur[3], a1[3]) for the forward
        # operands -- kept until the backward pass, the weight gradients read the same tensors -- and (d_c2[3], d_c1[3], d_sc[3])
        # for the gradients
        self.wamax = torch.zeros(len(self._convs), dtype=torch.float32, device=device)
        # (an activation slot is 16 sub-slots 64 floats apart = 1024 floats: include/margipose_hip.h, mpose_absmax)
        self.amax_f = torch.zeros(self.T * 10 * 6 * AMAX_SLOT, dtype=torch.float32, device=device)
        self.amax_b = torch.zeros(self.T * 10 * 15 * AMAX_SLOT, dtype=torch.float32, device=device)      # (+ 3 per block: the gradient w.r.t. its output; + 3: d_a1)
        for i, c in enumerate(self._convs):
            c.amax_ptr = self.wamax.data_ptr() + 4 * i
            if c.column or getattr(c, 'generic', False):      # columns and the feature extractor's graph (not the patch8 stem)
                jobs[2 * i]['layout'] = jobs[2 * i + 1]['layout'] = 2
                jobs[2 * i]['amax'] = jobs[2 * i + 1]['amax'] = c.amax_ptr
        # [3]: [2] with layout 3 (conv_h.hip's B tiles) for what the H2 engine runs: forward of conv_in / shortcut / conv2 and the
        # data-gradients of the H2 blocks
        jobs_h2 = jobs.copy()
        h2_convs = {}
        for b in self._all_blocks:
            if b.h2:                 # (h2_planes: the two-input data gradient runs on the H2 engine too)
                h2_convs[id(b.conv_in)] = h2_convs[id(b.conv_sc)] = (0, 1) if self.h2_planes else (0,)
                h2_convs[id(b.conv2)] = (0, 1)
        for i, c in enumerate(self._convs):
            for d in h2_convs.get(id(c), ()):
                jobs_h2[2 * i + d]['layout'] = 3
        # pack jobs by engine mode (conv_mode_for): 2 = conv_igemm_k everywhere, 3 = the same + the H2 engine's blocks in layout 3
        self._pack_jobs = {2: _jobs_to_device(jobs, device), 3: _jobs_to_device(jobs_h2, device)}
        # (the forward and the data-gradient job of a convolution read the same weight and share its amax slot: measured once)
        self._amax_jobs = {2: _jobs_to_device(np.ascontiguousarray(jobs[0::2]), device), 3: _jobs_to_device(np.ascontiguousarray(jobs_h2[0::2]), device)}
        self._packed_for = None      # which of the three the packed arena currently holds
        self._pack_max = mx
        self._tables = {}
        self._arena_key = key

    # views into the BN arenas --------------------------------------------------------------
    def _bnf_ptr(self, n, slot):
        """slot: 0 scale, 1 shift, 2 mean, 3 invstd, 4 coef(4*Cs: c0, c1, c2, mean)."""
        return self.bnf.data_ptr() + 4 * (n.f_off + slot * n.Cs)

    def _stats_ptr(self, n, bwd=False):
        return self.stat_arena.data_ptr() + 8 * (n.s_off + (2 * n.Cs if bwd else 0))

    def _mm_ptr(self, n):
        """(Cs, 2) sortable keys of max / max(-x) per channel of the tensor this BatchNorm normalises (mpose_conv_operands.mm0)."""
        return self.stat_arena.data_ptr() + 8 * (n.s_off + 6 * n.Cs)

    # ------------------------------------------------------------------ job tables per batch size
    def _tables_for(self, B, F):
        """Device-resident job tables + persistent workspaces for one (batch, heatmap size).  Everything they
        point at (parameters, arenas, the flat gradient buffer, the wgrad partial arena) has a fixed address,
        so a whole step is replayable from a hipGraph."""
        key = (B, F)
        if key in self._tables:
            return self._tables[key]
        S = F // 2
        dev = self.device
        tb = {}

        def hw_out(i):
            return F if (i < 2 or i >= 7) else S

        def hw_in(i):
            return F if (i <= 2 or i >= 8) else S

        def bn_job(j, n, count):
            j['stats'] = self._stats_ptr(n)
            j['gamma'] = n.m.weight.data_ptr(); j['beta'] = n.m.bias.data_ptr()
            j['running_mean'] = n.m.running_mean.data_ptr(); j['running_var'] = n.m.running_var.data_ptr()
            j['scale'] = self._bnf_ptr(n, 0); j['shift'] = self._bnf_ptr(n, 1)
            j['mean'] = self._bnf_ptr(n, 2); j['invstd'] = self._bnf_ptr(n, 3)
            j['C'] = n.C; j['count'] = count

        gbase = self.gflat.data_ptr()
        goff = dict((id(p), o) for p, o in zip(self.param_list(), self._grad_offsets))

        def coef_job(j, n, sums_from, stride, sg_col, which, count):
            j['sums'] = self._stats_ptr(sums_from, True)
            j['sg_col'] = sg_col
            j['gamma'] = n.m.weight.data_ptr(); j['mean'] = self._bnf_ptr(n, 2); j['invstd'] = self._bnf_ptr(n, 3)
            j['coef'] = self._bnf_ptr(n, 4)
            j['dgamma'] = gbase + 4 * goff[id(n.m.weight)]; j['dbeta'] = gbase + 4 * goff[id(n.m.bias)]
            j['sums_stride'], j['which'], j['C'], j['c_stride'], j['count'] = stride, which, n.C, n.Cs, count

        # forward finalize jobs: [stem] then per (stage, block): bn1 x3, bns x3, bn2 x3
        fj = np.zeros(1 + self.T * 90, dtype=BN_DT)
        # backward coefficient jobs: per (stage, block): bn2 x3, bns x3, bn1 x3 ; stem last
        cj = np.zeros(self.T * 90 + 1, dtype=COEF_DT)
        # wgrad partial arena + unpack jobs: per (stage, block, column): conv2, conv_in, conv_sc ; stem last
        n_stem_convs = len(self.stem.convs) if self.stem is not None else 1
        uj = np.zeros(self.T * 90 + n_stem_convs, dtype=UNPACK_DT)
        if self.stem is None:
            bn_job(fj[0], self.stem_bn, B * F * F)
        part_off = 0
        part_offs = {}
        mx = 0

        def unpack_job(j, conv, nsp):
            nonlocal part_off, mx
            sn, sk, stt = conv.strides(False)
            j['src'] = part_off            # patched to an absolute address below
            j['dst'] = gbase + 4 * goff[id(conv.param)]
            j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['n_split'] = conv.cout, conv.cin, conv.T, conv.npad_f, conv.cin_s, nsp
            j['sn'], j['sk'], j['st'], j['accumulate'] = sn, sk, stt, 0
            part_offs[id(conv)] = part_off
            part_off += nsp * conv.size_g
            mx = max(mx, conv.cout * conv.cin * conv.T)

        # MPOSE_CONV_STATS_PART buffers (header + rows): per BatchNorm the forward sums and, for bn1, the channel extremes; per
        # block the backward sums of bn1 (2 per channel, from the second 3x3's data-gradient) and of bn2 + shortcut (4 per channel,
        # from the next block's data-gradient).  Sized for the smallest pixel tile any engine uses (64).
        sp_off = 0
        sp = {}

        def part_buf(key, g, Cs, k):
            """(g: the geometry of the launch that WRITES the rows, or several when more than one kind of launch may: the largest counts)"""
            nonlocal sp_off
            rows = max(-(-(g_.B * g_.GH * g_.GW) // 64) * g_.n_classes for g_ in (g if isinstance(g, (list, tuple)) else [g]))
            sp[key] = (sp_off, rows)
            sp_off += _rup(4 + rows * Cs * k, 4)

        k = 1
        for t in range(self.T):
            for i in range(10):
                cnt = B * hw_out(i) * hw_out(i)
                grp = self.stage_blocks[t][i]
                base = (t * 10 + i) * 9
                for c, b in enumerate(grp):
                    gname = {'regular': 'f_in_regular', 'down': 'f_in_down', 'up': 'f_in_up'}[b.kind]
                    g_in, g_c2 = self.geom(gname, B, hw_in(i), b), self.geom('f_conv2', B, hw_out(i), b)
                    part_buf((id(b.bn1), 'f'), g_in, b.cout_s, 2)
                    part_buf((id(b.bn1), 'mm'), g_in, b.cout_s, 2)
                    part_buf((id(b.bns), 'f'), g_in, b.cout_s, 2)
                    part_buf((id(b.bn2), 'f'), g_c2, b.cout_s, 2)
                    part_buf((id(b.bn1), 'b'), g_c2, b.cout_s, 2)            # (d_conv2: the grid of f_conv2)
                    # (block i+1's two-input data gradient writes these rows, so ITS launch geometry sizes the buffer -- a
                    #  d_in_down launch has four output-parity classes of ceil(B*S*S / 64) rows each, more than f_conv2's single class
                    #  whenever B*S*S is not a multiple of 64)
                    g_wr = [g_c2]
                    if i + 1 < 10:
                        nb = self.stage_blocks[t][i + 1][c]
                        g_wr.append(self.geom({'regular': 'd_in_regular', 'down': 'd_in_down', 'up': 'd_in_up'}[nb.kind], B, hw_out(i + 1), nb))
                    part_buf((id(b.bn2), 'b'), g_wr, b.cout_s, 4)
                    bn_job(fj[1 + base + c], b.bn1, cnt)
                    # largest relu(bn1(c1)) -- the operand of the block's second convolution -- from c1's channel extremes
                    fj[1 + base + c]['minmax'], fj[1 + base + c]['amax_out'] = self._mm_ptr(b.bn1), self._amax_f(t, i, 1, c)
                    bn_job(fj[1 + base + 3 + c], b.bns, cnt)
                    bn_job(fj[1 + base + 6 + c], b.bn2, cnt)
                    coef_job(cj[base + c], b.bn2, b.bn2, 4, 0, 1, cnt)
                    coef_job(cj[base + 3 + c], b.bns, b.bn2, 4, 2, 3, cnt)
                    coef_job(cj[base + 6 + c], b.bn1, b.bn1, 2, 0, 1, cnt)
                    slots_in = B * (hw_in(i) if b.kind == 'up' else hw_out(i)) ** 2
                    gname = {'regular': 'f_in_regular', 'down': 'f_in_down', 'up': 'f_in_up'}[b.kind]
                    nsp_in = self.wg_n_split(self.geom(gname, B, hw_in(i), b))
                    # (the jobs of ONE weight-gradient launch are contiguous: conv2 of the three columns, then conv_in + shortcut)
                    unpack_job(uj[base + c], b.conv2, self.wg_n_split(self.geom('f_conv2', B, hw_out(i), b)))
                    unpack_job(uj[base + 3 + c], b.conv_in, nsp_in)
                    unpack_job(uj[base + 6 + c], b.conv_sc, nsp_in)
        if self.stem is None:
            coef_job(cj[self.T * 90], self.stem_bn, self.stem_bn, 4, 0, 1, B * F * F)
            unpack_job(uj[self.T * 90], self.stem_conv, self._n_split(B * F * F, 2, 1))
        else:
            for i, op in enumerate([o for o in self.stem.ops if hasattr(o, 'conv')]):
                unpack_job(uj[self.T * 90 + i], op.conv, self.stem_n_split(B, 8 * F, op))
        tb['stat_part'] = torch.zeros(max(sp_off, 4), dtype=torch.float32, device=dev)
        spb = tb['stat_part'].data_ptr()
        tb['sp_ptr'] = dict((k_, spb + 4 * v[0]) for k_, v in sp.items())
        for t in range(self.T):              # bounds for the H2 engine's fused producers (used when the launch asks for them)
            for i in range(10):
                base = (t * 10 + i) * 9
                for c, b in enumerate(self.stage_blocks[t][i]):
                    if self.h2_next(t, i):       # |relu(bn2(c2)) + bn_s(sc)| <= (|g2| + |gs|) sqrt(n) + |b2| + |bs|: the next block's input
                        j = fj[1 + base + 6 + c]
                        j['bound_gamma2'], j['bound_beta2'] = b.bns.m.weight.data_ptr(), b.bns.m.bias.data_ptr()
                        j['bound_out'] = self._amax_f(t, i + 1, 0, c)
                    if b.h2:                     # the BatchNorm-backward application of bn2: d_c2, the data-gradient's operand
                        cj[base + c]['g_amax'], cj[base + c]['bound_out'] = self._amax_b(t, i, 3, c), self._amax_b(t, i, 0, c)
                        if self.h2_planes:       # ... of the shortcut's BatchNorm (d_sc) and of bn1 (d_c1, from the largest |d_a1|) too
                            cj[base + 3 + c]['g_amax'], cj[base + 3 + c]['bound_out'] = self._amax_b(t, i, 3, c), self._amax_b(t, i, 2, c)
                            cj[base + 6 + c]['g_amax'], cj[base + 6 + c]['bound_out'] = self._amax_b(t, i, 4, c), self._amax_b(t, i, 1, c)
        for t in range(self.T):
            for i in range(10):
                base = (t * 10 + i) * 9
                for c, b in enumerate(self.stage_blocks[t][i]):
                    for jn, bn in ((1 + base + c, b.bn1), (1 + base + 3 + c, b.bns), (1 + base + 6 + c, b.bn2)):
                        fj[jn]['part'], fj[jn]['n_part'], fj[jn]['part_ld'] = tb['sp_ptr'][(id(bn), 'f')], sp[(id(bn), 'f')][1], b.cout_s
                    fj[1 + base + c]['mm_part'] = tb['sp_ptr'][(id(b.bn1), 'mm')]
                    for jn, pk in ((base + c, (id(b.bn2), 'b')), (base + 3 + c, (id(b.bn2), 'b')), (base + 6 + c, (id(b.bn1), 'b'))):
                        cj[jn]['part'], cj[jn]['n_part'], cj[jn]['part_ld'] = tb['sp_ptr'][pk], sp[pk][1], b.cout_s
        tb['n_unpack'] = self.T * 90 + n_stem_convs
        tb['partials'] = torch.empty(part_off, dtype=torch.float32, device=dev)
        pbase = tb['partials'].data_ptr()
        uj['src'] = pbase + 4 * uj['src']
        tb['part_ptr'] = dict((k_, pbase + 4 * v) for k_, v in part_offs.items())
        tb['fin'] = _jobs_to_device(fj, dev)
        tb['fin_count'] = torch.zeros(len(fj), dtype=torch.int32, device=dev)     # tickets of the launches that finalise their own BatchNorm
        tb['coef'] = _jobs_to_device(cj, dev)
        tb['unpack'] = _jobs_to_device(uj, dev)
        tb['unpack_max'] = mx
        # index of every convolution's job: the unpack of a weight-gradient launch follows it on the same stream (wgrad_async)
        idx = dict((int(uj[k_]['dst']), k_) for k_ in range(len(uj)))
        tb['unpack_idx'] = dict((id(c), idx[gbase + 4 * goff[id(c.param)]]) for c in self._convs)
        self._tables[key] = tb
        return tb

    def fin_index(self, t, i, which):
        """Index of the first of the 3 finalize jobs: which = 0 (bn1), 1 (bns), 2 (bn2)."""
        return 1 + (t * 10 + i) * 9 + which * 3

    # ------------------------------------------------------------------ geometries
    def geom(self, name, B, H, blk=None):
        key = (name, B, H, None if blk is None else (blk.cin_s, blk.cout_s))
        g = self._geoms.get(key)
        if g is not None:
            return g
        ci, co = (blk.cin_s, blk.cout_s) if blk is not None else (0, 0)
        np_f = _rup(blk.cout, 64) if blk is not None else 0      # forward N padding
        np_d = _rup(blk.cin, 64) if blk is not None else 0       # dgrad N padding (N = cin)
        t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in TAPS3]
        if name == 'f_in_regular':       # 3x3 + fused 1x1 shortcut
            g = _geom(B, H, ci, H, co, co, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], np_f, np_f)
        elif name == 'f_conv2':
            g = _geom(B, H, co, H, co, 0, H, 1, 1, [(0, 0, t9)], np_f)
        elif name == 'f_in3_regular':    # the 3x3 of f_in_regular alone / its 1x1 shortcut alone (kernel benchmarks)
            g = _geom(B, H, ci, H, co, 0, H, 1, 1, [(0, 0, t9)], np_f)
        elif name == 'f_in1_regular':
            g = _geom(B, H, ci, H, co, 0, H, 1, 1, [(0, 0, [(0, 0, 0, 0)])], np_f)
        elif name == 'f_in_down':        # H = input size
            g = _geom(B, H, ci, H // 2, co, co, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], np_f, np_f)
        elif name == 'f_in_up':          # H = input size
            g = _geom(B, H, ci, 2 * H, co, co, H, 1, 2, _up_classes(True), np_f, np_f)
        elif name == 'd_conv2':          # dgrad of the block's second 3x3: K = N = cout
            g = _geom(B, H, co, H, co, 0, H, 1, 1, [(0, 0, [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky, kx in TAPS3])], np_f)
        elif name == 'd_in3_regular':    # dgrad of conv_in: K = cout_s, N = cin
            g = _geom(B, H, co, H, ci, 0, H, 1, 1, [(0, 0, [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky, kx in TAPS3])], np_d)
        elif name == 'd_in_regular':     # dX = conv_in^T(dC1) + shortcut^T(dSC) in one pass (MPOSE_CONV_SUM_INPUTS)
            g = _geom(B, H, co, H, ci, ci, H, 1, 1, [(0, 0, [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky, kx in TAPS3] + [(0, 0, 0, 1)])], np_d, np_d)
        elif name == 'd_in_down':
            g = _geom(B, H, co, 2 * H, ci, ci, H, 1, 2, _up_classes(True), np_d, np_d)
        elif name == 'd_in_up':
            g = _geom(B, H, co, H // 2, ci, ci, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], np_d, np_d)
        elif name == 'd_in1_regular':
            g = _geom(B, H, co, H, ci, 0, H, 1, 1, [(0, 0, [(0, 0, 0, 0)])], np_d)
        elif name == 'd_in3_down':       # H = size of the conv OUTPUT (gradient input); result is 2H
            g = _geom(B, H, co, 2 * H, ci, 0, H, 1, 2, _up_classes(False), np_d)
        elif name == 'd_in1_down':
            g = _geom(B, H, co, 2 * H, ci, 0, H, 1, 2, _up_classes(False, single_tap=True), np_d)
        elif name == 'd_in3_up':         # H = size of the convT OUTPUT (gradient input); result is H/2
            g = _geom(B, H, co, H // 2, ci, 0, H // 2, 2, 1, [(0, 0, t9)], np_d)
        elif name == 'd_in1_up':
            g = _geom(B, H, co, H // 2, ci, 0, H // 2, 2, 1, [(0, 0, [(0, 0, 0, 0)])], np_d)
        elif name == 'f_stem':
            g = _geom(B, H, 192, H, 128, 0, H, 1, 1, [(0, 0, [(0, 0, 0, 0)])], 128)
        elif name == 'd_stem':
            g = _geom(B, H, 128, H, 192, 0, H, 1, 1, [(0, 0, [(0, 0, 0, 0)])], 192)
        else:
            raise KeyError(name)
        g._name = '%s/%dx%d/%d->%d' % (name, g.IH, g.IW, g.Cin, g.Cout0)
        g._flops = _geom_flops(g)
        self._geoms[key] = g
        return g

    # ------------------------------------------------------------------ launch helpers
    def _wptr(self, conv, dgrad=False):
        return self.wpack.data_ptr() + 4 * (conv.off_d if dgrad else conv.off_f)

    def conv(self, g, ops, flags=0):
        arr = (ConvOperands * 3)(*ops)
        t0 = self.timer.start('conv:' + g._name) if self.timer is not None else None
        check(lib().mpose_conv_fwd(ctypes.byref(g), arr, len(ops), flags, stream_ptr()), 'mpose_conv_fwd')
        if t0 is not None:
            self.timer.stop('conv:' + g._name, t0, g._flops * len(ops))

    # ---- pre-split activations (csrc/split.hip) ----
    def split_h2(self, srcs, slots, npix, C, scales=None, shifts=None, relu=False):
        """fp32 NHWC tensors -> two fp16 planes of [relu](scale*x + shift) * 2^k (mpose_split_h2; k from each tensor's amax slot);
        tensors that share storage are split once."""
        outs, ops, done = [], [], {}
        for i, src in enumerate(srcs):
            key = (src.data_ptr(), slots[i])
            if key in done:
                outs.append(done[key])
                continue
            pl = torch.empty(npix * C, dtype=torch.float32, device=self.device)
            so = SplitH2Operands()
            so.src, so.planes, so.amax = src.data_ptr(), pl.data_ptr(), slots[i]
            if scales is not None:
                so.scale, so.shift = scales[i], shifts[i]
            ops.append(so)
            done[key] = pl
            outs.append(pl)
        check(lib().mpose_split_h2((SplitH2Operands * 3)(*ops), len(ops), c_int64(npix), C, int(relu), stream_ptr()), 'mpose_split_h2')
        return outs

    def bn_bwd_reduce(self, rops, pixels_per_image, B, C, coef=None):
        """BatchNorm-backward sums of one grouped launch (mpose_bn_bwd_reduce_ws: per-workgroup partials, no atomics).
        coef = (device address of the first mpose_bn_bwd_coef_job, jobs, eval_mode): the coefficient jobs that read these sums
        run in the reduction's finishing pass (mpose_bn_bwd_reduce_coef_ws: one launch fewer, same results); returns True when
        they did (fuse_coef = False, or an empty batch: the caller launches mpose_bn_bwd_coef itself)."""
        L = lib()
        n = len(rops)
        need = int(L.mpose_bn_bwd_reduce_ws_bytes(n, pixels_per_image, B, C))
        ws = getattr(self, '_reduce_ws', None)
        if ws is None or ws.numel() < need or ws.device != self.device:
            self._reduce_ws = ws = torch.empty(max(need, 8 << 20), dtype=torch.uint8, device=self.device)
        if coef is not None and self.fuse_coef and B * pixels_per_image > 0:
            check(L.mpose_bn_bwd_reduce_coef_ws((BnBwdReduceOperands * 3)(*rops), n, pixels_per_image, B, C, ptr(ws), c_int64(ws.numel()),
                                                c_void_p(coef[0]), coef[1], int(coef[2]), stream_ptr()), 'mpose_bn_bwd_reduce_coef_ws')
            return True
        check(L.mpose_bn_bwd_reduce_ws((BnBwdReduceOperands * 3)(*rops), n, pixels_per_image, B, C, ptr(ws), c_int64(ws.numel()), stream_ptr()),
              'mpose_bn_bwd_reduce_ws')
        return False

    def conv_mode_for(self, train, save):
        """2: conv_igemm_k everywhere (inference with its fused epilogues); 3: the same with the H2 engine (conv_h2r_k on
        producer-split fp16 planes) on the regular 128-channel blocks -- training and differentiable forwards, in the fp32-equivalent
        arithmetic and (round 6) in the fp16-rounded mode, where the kernels read the h planes alone."""
        if (train or save) and self.stages_have_h2():
            return 3
        return 2

    def conv_flags(self, cmode):
        """mpose_conv_fwd flags of the arithmetic: MPOSE_CONV_F16X3 (+ MPOSE_CONV_F16X1 in the fp16-rounded mode)."""
        return 32 | (64 if self.conv_f16x1 else 0)

    def stages_have_h2(self):
        return any(b.h2 for b in self._all_blocks)

    def absmax(self, tensors, slots, C, scales=None, shifts=None, relu=False):
        """Largest magnitude of each NHWC tensor (after an optional per-channel affine map + ReLU) into its device slot; tensors
        that share storage share one pass."""
        ops, seen = [], set()
        for i, t in enumerate(tensors):
            key = (t.data_ptr(), slots[i])
            if key in seen:
                continue
            seen.add(key)
            ao = AbsmaxOperands()
            ao.src, ao.dst = t.data_ptr(), slots[i]
            if scales is not None:
                ao.scale, ao.shift = scales[i], shifts[i]
            ops.append(ao)
        npix = tensors[0].numel() // C
        check(lib().mpose_absmax((AbsmaxOperands * len(ops))(*ops), len(ops), c_int64(npix), C, int(relu), stream_ptr()), 'mpose_absmax')

    def _amax_f(self, t, i, which, c):
        """which: 0 block input, 1 relu(bn1(c1))."""
        return self.amax_f.data_ptr() + 4 * AMAX_SLOT * (((t * 10 + i) * 2 + which) * 3 + c)

    def _amax_b(self, t, i, which, c):
        """which: 0 d_c2, 1 d_c1, 2 d_sc, 3 the gradient w.r.t. the block's output, 4 d_a1 (the masked data gradient of the second 3x3)."""
        return self.amax_b.data_ptr() + 4 * AMAX_SLOT * (((t * 10 + i) * 5 + which) * 3 + c)

    def h2_next(self, t, i):
        """Block i's residual sum feeds an H2 block directly (no axis permutation in between): it can write that block's planes."""
        return i + 1 < 10 and self.stage_blocks[t][i + 1][0].h2 and not (i == 4 and any(sp != 0 for sp in self.spaces))

    def wgrad(self, g, ops, n_split):
        arr = (WgradOperands * 3)(*ops)
        t0 = self.timer.start('wgrad:' + g._name) if self.timer is not None else None
        check(lib().mpose_conv_wgrad(ctypes.byref(g), arr, len(ops), n_split, stream_ptr()), 'mpose_conv_wgrad')
        if t0 is not None:
            self.timer.stop('wgrad:' + g._name, t0, g._flops * len(ops))

    def dp_overlap(self):
        e = os.environ.get('MPOSE_DP_OVERLAP')
        if e is not None:
            return e != '0'
        try:
            return torch.distributed.get_backend(self.dp[0]) == 'nccl'
        except Exception:
            return False

    def unpack_after(self, tb, convs):
        """(table, first job, jobs, largest job) for wgrad_async: the split-K partials of `convs` (one launch's, contiguous jobs)
        are summed into the flat gradient buffer right behind the launch, on its stream, while they are still in the
        Infinity Cache -- instead of one pass per stage over 0.7 GB of partials on the main stream at the bucket's end."""
        if not self.inline_unpack:
            return None
        ks = sorted(tb['unpack_idx'][id(c)] for c in convs)
        assert ks == list(range(ks[0], ks[0] + len(ks)))
        return tb['unpack'].data_ptr() + ks[0] * UNPACK_DT.itemsize, len(ks), max(c.cout * c.cin * c.T for c in convs)

    def _unpack_now(self, unpack):
        if unpack is not None:
            check(lib().mpose_unpack_wgrads(c_void_p(unpack[0]), unpack[1], unpack[2], stream_ptr()), 'mpose_unpack_wgrads')

    def wgrad_async(self, g, ops, n_split, tensors, unpack=None):
        """Weight-gradient launch on the side stream: it only feeds the final unpack, so it overlaps the next
        block's data-gradient chain and the small BatchNorm kernels.  `tensors` are the buffers it reads: their
        memory must not be recycled by the caching allocator before the side stream is done with them."""
        # Serial when a KernelTimer brackets the launches (clean durations).  Under data parallelism the side stream stays on
        # with the RCCL backend (same stream logic as single-GPU: the bucket's all-reduce is issued after the main stream has
        # waited for the side stream and unpacked the partials) and off with gloo, whose host-staged collectives made a shared-GPU
        # step 2-6x slower with it (188-568 vs 95 ms); MPOSE_DP_OVERLAP=0/1 overrides.  (The RCCL combination has not run on
        # hardware: the pool has no multi-GPU node.  tests/test_model_gpu.py runs the schedule under gloo, functionally.)
        if not self.overlap_wgrad or (self.timer is not None and not self.timer.selective) or (self.dp is not None and not self.dp_overlap()):
            self.wgrad(g, ops, n_split)
            self._unpack_now(unpack)
            return
        main = torch.cuda.current_stream()
        if self.side_stream is None or self.side_stream.device != main.device:
            self.side_stream = torch.cuda.Stream(device=main.device)      # (a lower queue priority was tried in round 5: no gain)
        side = self.side_stream
        _lib.stream_wait(side, main)
        with torch.cuda.stream(side):
            self.wgrad(g, ops, n_split)
            self._unpack_now(unpack)
        # keep the operands alive until the main stream has waited for the side stream (_finish_bucket): unlike
        # Tensor.record_stream this costs no allocator head-room (reserved memory 8.3 GB instead of 30 GB at B=32)
        self._side_keep.extend(tensors)

    def part_stats(self):
        """Statistics leave the convolution launches as per-workgroup partial rows (MPOSE_CONV_STATS_PART), always."""
        return True

    def finalize_table(self, table, first, n, train, part=False, bounds=False):
        base = table.data_ptr() + first * BN_DT.itemsize
        check(lib().mpose_bn_finalize(c_void_p(base), n, int(train) | (2 if (part and train) else 0) | (4 if (bounds and train) else 0),
                                      ctypes.c_float(BN_EPS), ctypes.c_float(BN_MOMENTUM), stream_ptr()), 'mpose_bn_finalize')

    def part_ptr(self, B, S, conv):
        return self._tables_for(B, S // 8)['part_ptr'][id(conv)]

    def stem_n_split(self, B, S, op):
        return self.wg_n_split(self.stem.geom(op, B, S, 'f'), 1, width32=True)

    def wg_n_split(self, g, groups=3, width32=False):
        """Split-K factor of one weight-gradient launch: the library says how many workgroups one pixel split of one column
        takes for this geometry (mpose_conv_wgrad_tiles: it knows which kernel will run), _n_split fills the chip with them."""
        nsp = getattr(g, '_n_split', None)
        if nsp is None:
            tiles = int(lib().mpose_conv_wgrad_tiles(ctypes.byref(g)))
            if tiles <= 0:
                raise _lib.MposeError('mpose_conv_wgrad_tiles rejected the geometry %s' % getattr(g, '_name', '?'))
            slots = g.B * g.GH * (32 if width32 else g.GW)
            d = max(1, int(lib().mpose_conv_wgrad_phases(ctypes.byref(g))))       # x-dilated kernels: d launches, n_split / d each
            occ = 1
            if tiles * groups <= 2:
                # a one-tile launch of conv_wgrad_k -- the image's 27 -> 32 channel layer, the LAST weight gradient of a step, which the
                # main stream waits for with nothing left to run: 64 pixel splits were 64 workgroups (122 us)
                occ = 4
            waves = int(lib().mpose_conv_wgrad_waves(ctypes.byref(g)))
            if 0 < waves < 4:
                occ = max(occ, 4 // waves)     # one- / two-wave workgroups (the 32-channel tiles): 4 / 2 of them are one wide workgroup's footprint
            nsp = g._n_split = d * self._n_split(slots // d, tiles, groups, occ)
        return nsp

    def finalize(self, tb, first, n, train, part=False, bounds=False):
        base = tb['fin'].data_ptr() + first * BN_DT.itemsize
        check(lib().mpose_bn_finalize(c_void_p(base), n, int(train) | (2 if (part and train) else 0) | (4 if (bounds and train) else 0),
                                      ctypes.c_float(BN_EPS), ctypes.c_float(BN_MOMENTUM), stream_ptr()), 'mpose_bn_finalize')

    def pack_weights(self, cmode):
        """Measure and pack every convolution weight for engine mode `cmode` (conv_mode_for): one launch each."""
        n = len(self._convs)
        check(lib().mpose_weights_absmax(c_void_p(self._amax_jobs[cmode].data_ptr()), n, stream_ptr()), 'mpose_weights_absmax')
        check(lib().mpose_pack_weights(c_void_p(self._pack_jobs[cmode].data_ptr()), 2 * n, self._pack_max, stream_ptr()), 'mpose_pack_weights')
        self._packed_for = cmode
        self._pack_epoch += 1

    # ------------------------------------------------------------------ forward
    def forward(self, x, train, save, hm_bf16=False, features=None):
        """x: (B, 3, S, S) NCHW device tensor.  Returns (heatmap lists [3][T], xyz of last stage, ctx).
        hm_bf16 (inference only): heatmaps are stored as bf16; the soft-argmax coordinates stay fp32.
        features (forward only): a (B, 128, F, F) NCHW feature tensor fed to the stages in place of the feature extractor's
        output (`x` is ignored) -- how the reference's per-column fixtures drive a HeatmapColumn on its own."""
        if hm_bf16 and (train or save):
            raise _lib.MposeError('bf16 heatmaps are an inference storage mode (eval, no autograd)')
        L = lib()
        if features is not None:
            if save:
                raise _lib.MposeError('features= is a forward-only entry')
            features = _lib.dev_f32(features.contiguous(), 'features')
            B, C3, S, S2 = features.shape[0], 3, 8 * features.shape[2], 8 * features.shape[3]
            if features.shape[1] != 128:
                raise _lib.MposeError('features must have 128 channels')
            x = features
        elif isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and x.is_cuda:
            x = x.contiguous()         # raw RGB frames: normalised on the fly by the stem's first load (mpose_frames_u8)
            B, C3, S, S2 = x.shape
        else:
            x = _lib.dev_f32(x.contiguous(), 'input')
            B, C3, S, S2 = x.shape
        if C3 != 3 or S != S2 or S % 16 != 0:
            raise _lib.MposeError('expected a (B, 3, S, S) input with S %% 16 == 0, got %s' % (tuple(x.shape),))
        F = S // 8
        Sm = F // 2
        # (the fused tail reads one float4 of each pixel's channel line per workgroup: fine while the two inputs of the three columns
        #  stay in the 256 MB Infinity Cache, 8x over-fetch from HBM beyond -- measured 1.57 ms at B = 2048)
        # (beyond 128 MB the library takes the kernel's all-joints form -- every line read once -- when the J rows fit in LDS)
        tail_fused = (self.tail_fuse and (F & 3) == 0 and F * F <= 4096 and
                      (6 * B * F * F * 32 * 4 <= (128 << 20) or self.J * (F * F + 4) * 4 <= 144 * 1024))
        if 192 % Sm != 0 or (Sm % 4) != 0 or (F * F) % 64 != 0:
            raise _lib.MposeError('unsupported input size %d (mid size %d must divide 192 and be a multiple of 4)' % (S, Sm))
        if save and Sm % 8 != 0:       # the weight-gradient kernel walks slot rows in octets (mpose_conv_wgrad: GW % 8 == 0)
            raise _lib.MposeError('input size %d is inference-only: training needs a mid size (%d = S/16) that is a multiple of 8, '
                                  'e.g. 128, 256, 384' % (S, Sm))
        dev = x.device
        self._ensure_arenas(dev)
        tb = self._tables_for(B, F)
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=dev)
        ctx = _Ctx({'B': B, 'F': F, 'train': train, 'blocks': [], 'x_shape': tuple(x.shape)})
        self._snapshot_pending()       # an earlier forward whose backward is still to come keeps its BatchNorm vectors
        self._gen += 1
        self._arena_gen = ctx['gen'] = self._gen
        if save:
            self._pending = weakref.ref(ctx)
            ctx['param_versions'] = [p._version for p in self.param_list()]

        # (repacked on every forward, 0.33 ms: a cache keyed on the parameters' version counters would miss `p.data` updates and
        #  anything a replayed graph or a raw kernel such as DeviceSGD writes)
        cmode = ctx['cmode'] = self.conv_mode_for(train, save)
        h2 = cmode == 3
        h2f = ctx['h2f'] = bool(h2 and train)       # the producers of the H2 blocks' operands write the planes themselves (a-priori bounds)
        # statistics as per-workgroup partial rows (MPOSE_CONV_STATS_PART): deterministic, no atomics
        spart = ctx['spart'] = bool(train)
        sp = tb['sp_ptr']
        if not (self.pack_frozen and self._packed_for == cmode):
            self.pack_weights(cmode)
        _lib.fill_zero(self.amax_f)        # (fills through the library: a launch plan records them, csrc/plan.hip)
        if train:
            _lib.fill_zero(self.stat_arena)
        elif self.stem is None:
            self.finalize(tb, 0, 1 + self.T * 90, False)     # scale/shift from the running statistics
        else:
            self.finalize(tb, 1, self.T * 90, False)

        if features is not None:
            inp = features.permute(0, 2, 3, 1).contiguous()
        elif self.stem is not None:
            # ---- InceptionV4 feature extractor (stem.py) ----
            inp, ctx['stem_ctx'] = self.stem.forward(x, train, save)
        else:
            # ---- patch8 stem: space-to-depth + 1x1 conv (192->128) + BN + ReLU ----
            if x.dtype == torch.uint8:
                xf = torch.empty(B, 3, S, S, **f32)
                check(L.mpose_frames_u8(c_void_p(x.data_ptr()), (ctypes.c_float * 3)(*self.input_norm[0]), (ctypes.c_float * 3)(*self.input_norm[1]),
                                        ptr(xf), B, S, S, 0, st()), 'mpose_frames_u8')
                x = xf
            s2d = torch.empty(B, F, F, 192, **f32)
            check(L.mpose_space_to_depth8(ptr(x), ptr(s2d), B, S, st()), 'mpose_space_to_depth8')
            stem_raw = torch.empty(B, F, F, 128, **f32)
            op = ConvOperands()
            op.in_, op.w0, op.out0 = s2d.data_ptr(), self._wptr(self.stem_conv), stem_raw.data_ptr()
            if train:
                op.stats0 = self._stats_ptr(self.stem_bn)
            self.conv(self.geom('f_stem', B, F), [op])
            if train:
                self.finalize(tb, 0, 1, True)
            inp = torch.empty(B, F, F, 128, **f32)
            check(L.mpose_bn_relu_fwd(ptr(stem_raw), c_void_p(self._bnf_ptr(self.stem_bn, 0)), c_void_p(self._bnf_ptr(self.stem_bn, 1)),
                                      ptr(inp), c_int64(inp.numel()), 128, st()), 'mpose_bn_relu_fwd')
            ctx['s2d'], ctx['stem_raw'], ctx['stem_out'] = s2d, stem_raw, inp
        ctx['inps'] = []

        hms = [[], [], []]
        xyz = None
        for t in range(self.T):
            if t > 0:
                new_inp = torch.empty_like(inp)
                check((L.mpose_combiner_fwd_bf16 if hm_bf16 else L.mpose_combiner_fwd)(
                    ptr_array([hms[p][t - 1] for p in range(3)]), ptr(self.combiners[t - 1]), ptr(inp), ptr(new_inp), B, self.J, F * F, 128,
                    st()), 'mpose_combiner_fwd')
                inp = new_inp
            ctx['inps'].append(inp)
            cur = [inp, inp, inp]
            pflags = ctx['pflags'] = self.conv_flags(cmode)
            stage_saved = []
            nxt_h = None
            for i in range(10):
                grp = self.stage_blocks[t][i]
                b0 = grp[0]
                Hin = F if (i <= 2 or i >= 8) else Sm          # input spatial size of block i
                Hout = F if (i < 2 or i >= 7) else Sm
                if i == 5 and any(sp != 0 for sp in self.spaces):
                    outs = [cur[c] if self.spaces[c] == 0 else torch.empty_like(cur[c]) for c in range(3)]
                    spaces = (c_int * 3)(*self.spaces)
                    check(L.mpose_axis_permute(ptr_array(cur), ptr_array(outs), spaces, 3, B, Sm, 192, st()), 'mpose_axis_permute')
                    cur = outs
                gname = {'regular': 'f_in_regular', 'down': 'f_in_down', 'up': 'f_in_up'}[b0.kind]
                g1 = self.geom(gname, B, Hin, b0)
                last = i == 9
                npix_o = B * Hout * Hout
                # Inference: BatchNorm with running statistics is a per-channel affine map, so it (and the ReLU, and the block's
                # residual sum) run in the convolutions' epilogues -- the first launch stores relu(bn1(.)) (the second reads a plain
                # tensor: no prologue), the second relu(bn2(.)) + bn_s(shortcut) -- and both measure max |.| for the next
                # convolution's scale as they store: two launches per block, no elementwise or amax pass (reference :31-40).
                fused_h = not train and not save
                c1 = [torch.empty(B, Hout, Hout, b0.cout_s, **f32) for _ in range(3)]
                sc = [torch.empty(B, Hout, Hout, b0.cout_s, **f32) for _ in range(3)]
                # (columns reading one tensor -- the stage input -- share its slot)
                cur_slot = [self._amax_f(t, i, 0, _first_same(cur, c)) for c in range(3)]
                if i == 0:               # later blocks: the previous block's residual add measured its output as it wrote it
                    self.absmax(cur, cur_slot, b0.cin_s)
                blk_h2 = h2 and b0.h2        # this block's forward convolutions read producer-split fp16 planes (conv_h.hip)
                if blk_h2:
                    if nxt_h is not None:    # (the block before wrote them with its residual sum)
                        cur_h, nxt_h = nxt_h, None
                    else:
                        cur_h = self.split_h2(cur, cur_slot, B * Hin * Hin, b0.cin_s)
                ops = []
                for c, b in enumerate(grp):
                    op = ConvOperands()
                    op.in_ = (cur_h[c] if blk_h2 else cur[c]).data_ptr()
                    op.w0, op.w1 = self._wptr(b.conv_in), self._wptr(b.conv_sc)
                    op.in_amax, op.w0_amax, op.w1_amax = cur_slot[c], b.conv_in.amax_ptr, b.conv_sc.amax_ptr
                    op.out1 = sc[c].data_ptr()
                    op.out0 = c1[c].data_ptr()
                    if fused_h:          # c1 holds relu(bn1(conv)) here
                        op.epi_scale0, op.epi_shift0 = self._bnf_ptr(b.bn1, 0), self._bnf_ptr(b.bn1, 1)
                        op.out0_amax = self._amax_f(t, i, 1, c)
                    if train:
                        op.stats0, op.stats1 = sp[(id(b.bn1), 'f')], sp[(id(b.bns), 'f')]
                        op.mm0 = sp[(id(b.bn1), 'mm')]
                    ops.append(op)
                self.conv(g1, ops, pflags | (16 if fused_h else 0) | (256 if spart else 0) | (128 if blk_h2 else 0))
                if train:
                    self.finalize(tb, self.fin_index(t, i, 0), 6, True, spart)
                # largest relu(bn1(c1)): what the next K loop (and its weight gradient) will split.  In training bn_finalize just
                # derived it from c1's channel extremes (the convolution's epilogue took them); otherwise it is measured
                if not fused_h and not train:
                    self.absmax(c1, [self._amax_f(t, i, 1, c) for c in range(3)], b0.cout_s, [self._bnf_ptr(b.bn1, 0) for b in grp],
                                [self._bnf_ptr(b.bn1, 1) for b in grp], relu=True)
                if blk_h2:                   # relu(bn1(c1)) * 2^k as two fp16 planes, once (its amax slot: exact, from c1's channel extremes)
                    a1_h = self.split_h2(c1, [self._amax_f(t, i, 1, c) for c in range(3)], npix_o, b0.cout_s,
                                         [self._bnf_ptr(b.bn1, 0) for b in grp], [self._bnf_ptr(b.bn1, 1) for b in grp], relu=True)
                fuse2_h = fused_h and not last
                c2 = None if fuse2_h else [torch.empty(B, Hout, Hout, b0.cout_s, **f32) for _ in range(3)]
                if last:
                    outs = [None] * 3 if tail_fused else [torch.empty(B, self.J, F, F, **f32) for _ in range(3)]
                else:
                    outs = [torch.empty(B, Hout, Hout, b0.cout_s, **f32) for _ in range(3)]
                ops = []
                for c, b in enumerate(grp):
                    op = ConvOperands()
                    op.w0 = self._wptr(b.conv2)
                    op.in_ = (a1_h[c] if blk_h2 else c1[c]).data_ptr()
                    if not fused_h and not blk_h2:
                        op.in_scale, op.in_shift = self._bnf_ptr(b.bn1, 0), self._bnf_ptr(b.bn1, 1)
                    op.in_amax, op.w0_amax = self._amax_f(t, i, 1, c), b.conv2.amax_ptr
                    if fuse2_h:
                        op.out0 = outs[c].data_ptr()
                        op.epi_scale0, op.epi_shift0 = self._bnf_ptr(b.bn2, 0), self._bnf_ptr(b.bn2, 1)
                        op.add_src, op.add_scale, op.add_shift = sc[c].data_ptr(), self._bnf_ptr(b.bns, 0), self._bnf_ptr(b.bns, 1)
                        op.out0_amax = self._amax_f(t, i + 1, 0, c)           # (the axis permutation after block 4 keeps the maximum)
                    else:
                        op.out0 = c2[c].data_ptr()
                    if train:
                        op.stats0 = sp[(id(b.bn2), 'f')]
                    ops.append(op)
                self.conv(self.geom('f_conv2', B, Hout, b0), ops, pflags | (16 if fuse2_h else 0) | (256 if spart else 0) | (128 if blk_h2 else 0))
                add_h2 = h2f and self.h2_next(t, i)      # this block's sum is the next (H2) block's input: planes written here
                if train:
                    self.finalize(tb, self.fin_index(t, i, 2), 3, True, spart, bounds=add_h2)
                if not fuse2_h:
                    aops = []
                    for c, b in enumerate(grp):
                        ao = BnAddOperands()
                        ao.a, ao.a_scale, ao.a_shift = c2[c].data_ptr(), self._bnf_ptr(b.bn2, 0), self._bnf_ptr(b.bn2, 1)
                        ao.b, ao.b_scale, ao.b_shift = sc[c].data_ptr(), self._bnf_ptr(b.bns, 0), self._bnf_ptr(b.bns, 1)
                        ao.out = outs[c].data_ptr() if outs[c] is not None else None
                        if not last:
                            ao.out_amax = self._amax_f(t, i + 1, 0, c)       # (the axis permutation after block 4 keeps the maximum)
                        aops.append(ao)
                    if add_h2:               # (out_amax is READ there: the bound bn_finalize just wrote)
                        nxt_h = [torch.empty(npix_o * b0.cout_s, **f32) for _ in range(3)]
                        check(L.mpose_bn_add_h2((BnAddOperands * 3)(*aops), ptr_array(nxt_h), 3, c_int64(npix_o), b0.cout_s, st()),
                              'mpose_bn_add_h2')
                    elif last and tail_fused:       # residual sum + flat_softmax + dsnt in one launch: no logits in memory
                        heat = [torch.empty(B, self.J, F, F, dtype=torch.bfloat16 if hm_bf16 else torch.float32, device=x.device)
                                for _ in range(3)]
                        pc = torch.empty(3, B * self.J, 2, **f32) if t == self.T - 1 else None
                        t0 = self.timer.start('tail:bn_add_softmax_fwd') if self.timer is not None else None
                        check(L.mpose_bn_add_softmax_fwd((BnAddOperands * 3)(*aops), ptr_array(heat), ptr(pc) if pc is not None else None,
                                                         3, B, F, F, b0.cout_s, self.J, 2 if hm_bf16 else 0, st()), 'mpose_bn_add_softmax_fwd')
                        if t0 is not None:    # algorithmic bytes: the joint channels of both inputs once, the heatmaps once
                            self.timer.stop('tail:bn_add_softmax_fwd', t0, 3 * B * self.J * F * F * (10 if hm_bf16 else 12))
                        if pc is not None:
                            xyz = torch.empty(B, self.J, 3, **f32)
                            check(L.mpose_coords_merge(ptr(pc), ptr(xyz), B * self.J, st()), 'mpose_coords_merge')
                    else:
                        check(L.mpose_bn_add_fwd((BnAddOperands * 3)(*aops), 3, Hout * Hout, B, b0.cout_s, 1 if last else 0, self.J, st()),
                              'mpose_bn_add_fwd')
                if save:
                    stage_saved.append({'x': cur, 'c1': c1, 'sc': sc, 'c2': c2})
                    if blk_h2 and h2f and self.h2_planes:      # the weight gradients read the planes (Engine.h2_planes)
                        stage_saved[-1]['x_h'], stage_saved[-1]['a1_h'] = cur_h, a1_h
                cur = outs
            if not tail_fused:
                logits = cur
                heat = [torch.empty_like(l, dtype=torch.bfloat16 if hm_bf16 else torch.float32) for l in logits]
                want_xyz = t == self.T - 1
                if want_xyz:
                    xyz = torch.empty(B, self.J, 3, **f32)
                t0 = self.timer.start('tail:softmax_dsnt_fwd') if self.timer is not None else None
                check(L.mpose_softmax_dsnt_fwd(ptr_array(logits), ptr_array(heat), None, ptr(xyz) if want_xyz else None, 3, B * self.J, F,
                                               F, 2 if hm_bf16 else 0, st()), 'mpose_softmax_dsnt_fwd')
                if t0 is not None:      # algorithmic bytes: read logits once, write heatmaps once (+ coords)
                    self.timer.stop('tail:softmax_dsnt_fwd', t0, 3 * B * self.J * F * F * (6 if hm_bf16 else 8) + (B * self.J * 12 if want_xyz else 0))
            for p in range(3):
                hms[p].append(heat[p])
            ctx['blocks'].append(stage_saved)
        if train:
            check(L.mpose_add_i64(c_void_p(self._nbt.data_ptr()), c_int64(1), c_int64(self._nbt.numel()), st()), 'mpose_add_i64')
        return hms, xyz, ctx

    # ------------------------------------------------------------------ several forwards in flight
    # ------------------------------------------------------------------ graph-only models (ChatterboxModel)
    def graph_forward(self, x, train, save, features=None):
        """A model that is ONE stem.py graph with several raw outputs (no stages): returns (list of NHWC outputs, ctx).
        features (forward only): a (B, 128, 32, 32) tensor in place of the feature extractor's output (`x` is ignored) -- how the
        reference's _ChatterboxCnn fixtures drive the heads on their own."""
        if features is not None:
            if save:
                raise _lib.MposeError('features= is a forward-only entry')
            x = features = _lib.dev_f32(features.contiguous(), 'features')
            B, S = features.shape[0], self.stem.INPUT_SIZE
        else:
            if isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and x.is_cuda:
                x = x.contiguous()
            else:
                x = _lib.dev_f32(x.contiguous(), 'input')
            B, C3, S, S2 = x.shape
            if C3 != 3 or S != S2 or S != self.stem.INPUT_SIZE:
                raise _lib.MposeError('expected a (B, 3, %d, %d) input, got %s' % (self.stem.INPUT_SIZE, self.stem.INPUT_SIZE, tuple(x.shape)))
        self._ensure_arenas(x.device)
        ctx = _Ctx({'B': B, 'F': S // 8, 'train': train, 'x_shape': tuple(x.shape)})
        self._snapshot_pending()
        self._gen += 1
        self._arena_gen = ctx['gen'] = self._gen
        if save:
            self._pending = weakref.ref(ctx)
            ctx['param_versions'] = [p._version for p in self.param_list()]
        cmode = ctx['cmode'] = self.conv_mode_for(train, save)
        self.pack_weights(cmode)
        outs, ctx['stem_ctx'] = self.stem.forward(x, train, save, features=features)
        if train and features is None:
            check(lib().mpose_add_i64(c_void_p(self._nbt.data_ptr()), c_int64(1), c_int64(self._nbt.numel()), stream_ptr()), 'mpose_add_i64')
        return outs, ctx

    def graph_backward(self, ctx, grads, need_dx):
        """grads: gradient w.r.t. each raw output (or None).  Returns (flat gradient buffer, dx or None)."""
        for p, v in zip(self.param_list(), ctx['param_versions']):
            if p._version != v:
                raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace '
                                   'operation: a parameter of the model changed between forward and backward')
        self._restore_arenas(ctx)
        tb = self._tables_for(ctx['B'], ctx['F'])
        _lib.fill_zero(self.gflat)
        if self._packed_for != ctx['cmode']:
            self.pack_weights(ctx['cmode'])
        works = self._dp_works
        del works[:]
        dx = self.stem.backward(ctx['stem_ctx'], grads, need_dx)
        self._finish_bucket(tb, 0, 0, tb['n_unpack'], works)
        if self.dp is not None:
            _lib.plan_host(self._wait_collectives)
        ctx['done'] = True
        return self.gflat, dx

    def _arena_tensors(self):
        return [self.bnf, self.amax_f] + ([self.stem.f_arena, self.stem.amax_f] if self.stem is not None else [])

    # ------------------------------------------------------------------ launch plans (train_helpers.PlannedTrainStep / PlannedInference)
    def plan_stamp(self, table_keys=None, backward=True):
        """Everything a recorded launch plan (csrc/plan.hip) bakes in BY ADDRESS or by value but that lives outside the plan's
        private allocator pool: the parameters and BatchNorm buffers as bound right now, the engine's arenas, the per-(B, F) job
        tables, the BatchNorm-backward reduction workspace (regrown by a larger eager batch), and the switches that decide which
        launches an iteration consists of.  A replay compares it with the stamp taken at recording time: any difference means
        the plan would read or write through stale pointers.  `table_keys`: the job tables the recording used (default:
        all that exist now); tables built later for other batch sizes do not invalidate a plan."""
        if self._arena_key is None:
            return None
        bound = [t for t in self._bn_bound_now() if t is not None] + [c.param for c in self._convs] + list(self.combiners)
        if self.stem is not None:
            bound += list(self.stem.extra_params)
        ws = getattr(self, '_reduce_ws', None) if backward else None      # (a forward-only plan never touches the reduction workspace)
        arenas = [self.wpack, self.bnf, self.stat_arena, self.gflat, self.wamax, self.amax_f, self.amax_b, self._nbt]
        if self.stem is not None:
            arenas += [t for t in (getattr(self.stem, 'f_arena', None), getattr(self.stem, 'amax_f', None)) if t is not None]
        tables = tuple((k, id(self._tables.get(k))) for k in (sorted(self._tables) if table_keys is None else table_keys))
        flags = (self.h2_planes, self.inline_unpack, self.overlap_wgrad, self.fuse_coef, self.tail_fuse, self.stem_bounds, self.conv_f16x1,
                 self.dp is not None)
        return (self._arena_key[0], hash(tuple(t.data_ptr() for t in bound)), tuple(t.data_ptr() for t in arenas),
                (ws.data_ptr(), ws.numel()) if ws is not None else None, tables, flags)

    def before_replay(self):
        """A launch-plan replay overwrites the BatchNorm arenas like a forward does: an eager forward whose backward is still to
        come keeps its values (snapshot), and the arenas then belong to nobody's saved context (the replayed iteration runs its
        own backward)."""
        self._snapshot_pending()
        self._gen += 1
        self._arena_gen = self._gen
        self._pending = None

    def _snapshot_pending(self):
        """Called before a forward overwrites the BatchNorm arenas: if the previous saved forward has not run its backward
        yet, it gets a private copy of them (no cost in the ordinary forward-backward-forward-backward loop)."""
        prev = self._pending() if self._pending is not None else None
        if prev is not None and not prev.get('done') and 'arena_snap' not in prev and prev['gen'] == self._arena_gen:
            prev['arena_snap'] = [t.clone() for t in self._arena_tensors()]

    def _restore_arenas(self, ctx):
        if ctx['gen'] == self._arena_gen:
            return
        self._snapshot_pending()                 # the newer forward may still want its own values back
        snap = ctx.get('arena_snap')
        if snap is None:
            raise _lib.MposeError('the BatchNorm state of this forward pass is gone (internal error: no snapshot was taken)')
        for dst, src in zip(self._arena_tensors(), snap):
            dst.copy_(src)
        self._arena_gen = ctx['gen']

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, hms, g_hms, need_dx):
        """hms[p][t]: heatmaps of the forward; g_hms[p][t]: gradient w.r.t. them (or None).
        Returns (persistent flat gradient buffer, dx or None).
        Eval-mode forwards are differentiable too (running statistics are constants: dx = gamma*invstd*g)."""
        L = lib()
        for p, v in zip(self.param_list(), ctx['param_versions']):
            if p._version != v:      # the packed weights follow the parameters: same rule (and wording) as autograd's own check
                raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace '
                                   'operation: a parameter of MargiPoseModel changed between forward and backward')
        self._restore_arenas(ctx)
        eval_bn = 0 if ctx['train'] else 1
        B, F = ctx['B'], ctx['F']
        Sm = F // 2
        dev = self.device
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=dev)
        J = self.J
        tb = self._tables_for(B, F)
        # When the last stage's heatmaps carry a gradient every stage runs, and every element of the flat gradient buffer a parameter
        # owns is WRITTEN by this pass (weights: mpose_unpack_wgrads with accumulate = 0; BatchNorm: the coefficient kernels;
        # combiners: mpose_reduce_partials): no zero fill then (24 us for 178 MB).  A stage without any gradient is skipped and
        # its parameters keep the fill's zeros.  MPOSE_GFLAT_FILL=1 always fills, =2 fills with NaNs instead of skipping
        # (test_every_gradient_element_is_written: no gradient may keep one)
        # (a frozen parameter -- requires_grad False -- may change what the pass writes: the fill runs, 24 us)
        if _GFLAT_FILL == 1 or all(g_hms[p][self.T - 1] is None for p in range(3)) or not all(p.requires_grad for p in self.param_list()):
            _lib.fill_zero(self.gflat)
        elif _GFLAT_FILL == 2:
            check(lib().mpose_fill_u32(c_void_p(self.gflat.data_ptr()), 0x7fc00000, c_int64(4 * self.gflat.numel()), stream_ptr()), 'mpose_fill_u32')
        _lib.fill_zero(self.stat_arena)        # forward sums are consumed (mean/invstd live in the float arena)
        goff = dict((id(p), o) for p, o in zip(self.param_list(), self._grad_offsets))
        coef_base = tb['coef'].data_ptr()
        cmode = ctx['cmode']
        h2 = cmode == 3
        pflags = ctx['pflags']         # (the forward's convolution engine and precision)
        x1 = bool(pflags & 64)
        if self._packed_for != cmode:      # a forward on another engine ran in between: the parameters are unchanged (checked
            self.pack_weights(cmode)       # above), so this restores exactly the packing of this context's forward
        _lib.fill_zero(self.amax_b)

        # (the backward's sums as partial rows too; an eval-mode forward has ctx['spart'] False but its backward may still use them)
        spart = True
        sp = tb['sp_ptr']

        h2f = bool(h2 and ctx.get('h2f', False))

        def run_coef(first, n, from_sums=False, bounds=False):
            # mode: bit 0 eval, bit 1 the jobs' `sums` were written by a reduction pass (not partial rows), bit 2: 1024 threads,
            # bit 3: the jobs' bounds (mpose_bn_bwd_coef_job.bound_out)
            mode = eval_bn | (0 if (spart and not from_sums) else 2) | (4 if (spart and not from_sums) else 0) | (8 if bounds else 0)
            check(L.mpose_bn_bwd_coef(c_void_p(coef_base + first * COEF_DT.itemsize), n, mode, st()), 'mpose_bn_bwd_coef')

        works = self._dp_works          # in-flight gradient all-reduces (data parallel), one per finished bucket
        del works[:]
        D = None            # gradient w.r.t. the stage input, cumulative over later stages (:195 is `inp = inp + ...`)
        g_comb = None
        for t in reversed(range(self.T)):
            heat = [hms[p][t] for p in range(3)]
            g1 = [g_hms[p][t] for p in range(3)]
            if all(g is None for g in g1) and g_comb is None and D is None:
                continue
            g1 = [_lib.dev_f32(g.contiguous(), 'grad') if g is not None else _lib.fill_zero(torch.empty_like(heat[p])) for p, g in enumerate(g1)]
            dlog = [torch.empty_like(h) for h in heat]
            check(L.mpose_softmax_bwd(ptr_array(heat), ptr_array(g1), ptr_array(g_comb) if g_comb is not None else None,
                                      ptr_array(dlog), 3, B * J, F * F, st()), 'mpose_softmax_bwd')
            g = [torch.empty(B, F, F, 32, **f32) for _ in range(3)]
            check(L.mpose_nchw_to_nhwc_pad(ptr_array(dlog), ptr_array(g), 3, B, J, F * F, 32, st()), 'mpose_nchw_to_nhwc_pad')

            saved = ctx['blocks'][t]
            for i in reversed(range(10)):
                grp = self.stage_blocks[t][i]
                b0 = grp[0]
                sv = saved[i]
                Hin = F if (i <= 2 or i >= 8) else Sm
                Hout = F if (i < 2 or i >= 7) else Sm
                cnt = B * Hout * Hout
                Cs = b0.cout_s
                jb = (t * 10 + i) * 9
                if i == 9:
                    sums_done = False        # (g comes from the tail)
                # this block's data-gradient can take block i-1's sums unless the axis permutation sits between them (after
                # block 4: it mixes channels and pixels)
                fuse_sums = i >= 1 and not (i == 5 and any(sp != 0 for sp in self.spaces))
                # (1) sums of g, g*c2, g*sc -> BN2 / BN_shortcut backward coefficients, dgamma, dbeta -- already taken by the
                #     data-gradient that produced g (the epilogue of the next block's step (6)) where that was possible
                if not sums_done:
                    rops = []
                    for c, b in enumerate(grp):
                        ro = BnBwdReduceOperands()
                        ro.g, ro.a, ro.b = g[c].data_ptr(), sv['c2'][c].data_ptr(), sv['sc'][c].data_ptr()
                        ro.a_scale, ro.a_shift = self._bnf_ptr(b.bn2, 0), self._bnf_ptr(b.bn2, 1)     # ReLU after the second BN
                        ro.sums = self._stats_ptr(b.bn2, True)
                        rops.append(ro)
                blk_h2 = h2 and b0.h2
                app_h2 = h2f and blk_h2     # bn2's backward application writes d_c2 as planes too, scaled by the coefficient kernel's bound
                pl_bwd = app_h2 and self.h2_planes and 'x_h' in sv        # planes end to end: no fp32 d_c2 / d_sc / d_c1 at all
                coef_done = False
                if not sums_done:
                    # (the six coefficient jobs of bn2 / bn_s read these sums: they run in the reduction's finishing pass unless they
                    #  also have to produce the bound of an H2 block's planes)
                    coef_done = self.bn_bwd_reduce(rops, Hout * Hout, B, Cs,
                                                   None if app_h2 else (coef_base + jb * COEF_DT.itemsize, 6, eval_bn))
                if not coef_done:
                    run_coef(jb, 6, from_sums=not sums_done, bounds=app_h2)
                d_c2 = None if pl_bwd else [torch.empty(B, Hout, Hout, Cs, **f32) for _ in range(3)]
                d_sc = None if pl_bwd else [torch.empty(B, Hout, Hout, Cs, **f32) for _ in range(3)]
                aops = []
                for c, b in enumerate(grp):
                    ao = BnBwdApplyOperands()
                    ao.g, ao.a, ao.b = g[c].data_ptr(), sv['c2'][c].data_ptr(), sv['sc'][c].data_ptr()
                    ao.coef_a, ao.coef_b = self._bnf_ptr(b.bn2, 4), self._bnf_ptr(b.bns, 4)
                    ao.a_scale, ao.a_shift = self._bnf_ptr(b.bn2, 0), self._bnf_ptr(b.bn2, 1)
                    if not pl_bwd:
                        ao.da, ao.db = d_c2[c].data_ptr(), d_sc[c].data_ptr()
                    ao.da_amax, ao.db_amax = self._amax_b(t, i, 0, c), self._amax_b(t, i, 2, c)
                    aops.append(ao)
                if app_h2:           # (da_amax is READ there; with pl_bwd db_amax too: both hold the coefficient kernel's bounds)
                    d_c2_h = [torch.empty(cnt * Cs, **f32) for _ in range(3)]
                    d_sc_h = [torch.empty(cnt * Cs, **f32) for _ in range(3)] if pl_bwd else None
                    check(L.mpose_bn_bwd_apply_h2((BnBwdApplyOperands * 3)(*aops), ptr_array(d_c2_h), ptr_array(d_sc_h) if pl_bwd else None,
                                                  3, c_int64(cnt), Cs, st()), 'mpose_bn_bwd_apply_h2')
                else:
                    check(L.mpose_bn_bwd_apply((BnBwdApplyOperands * 3)(*aops), 3, Hout * Hout, B, Cs, 0, 0, st()), 'mpose_bn_bwd_apply')
                # (2) dgrad of the second 3x3; ReLU mask and the BN1-backward sums happen in its epilogue
                d_a1 = [torch.empty(B, Hout, Hout, Cs, **f32) for _ in range(3)]
                if blk_h2 and not app_h2:    # the second 3x3's data-gradient on conv_h.hip: its operand as two fp16 planes
                    d_c2_h = self.split_h2(d_c2, [self._amax_b(t, i, 0, c) for c in range(3)], cnt, Cs)
                ops = []
                for c, b in enumerate(grp):
                    op = ConvOperands()
                    op.in_ = (d_c2_h[c] if blk_h2 else d_c2[c]).data_ptr()
                    op.w0, op.out0 = self._wptr(b.conv2, True), d_a1[c].data_ptr()
                    op.in_amax, op.w0_amax = self._amax_b(t, i, 0, c), b.conv2.amax_ptr
                    op.mask_src = sv['c1'][c].data_ptr()
                    op.mask_scale, op.mask_shift = self._bnf_ptr(b.bn1, 0), self._bnf_ptr(b.bn1, 1)
                    op.stats0 = sp[(id(b.bn1), 'b')] if spart else self._stats_ptr(b.bn1, True)
                    if pl_bwd:               # the largest |d_a1|: what bn1's coefficient kernel bounds d_c1 with
                        op.out0_amax = self._amax_b(t, i, 4, c)
                    ops.append(op)
                self.conv(self.geom('d_conv2', B, Hout, b0), ops, pflags | (256 if spart else 0) | (128 if blk_h2 else 0))
                # (3) wgrad of the second 3x3 (its input relu(bn1(c1)) is recomputed while staging; pl_bwd: the forward's planes of it)
                wops = []
                for c, b in enumerate(grp):
                    wo = WgradOperands()
                    if pl_bwd:
                        wo.in_, wo.gout0, wo.planes_in = sv['a1_h'][c].data_ptr(), d_c2_h[c].data_ptr(), 1
                    else:
                        wo.in_, wo.in_scale, wo.in_shift = sv['c1'][c].data_ptr(), self._bnf_ptr(b.bn1, 0), self._bnf_ptr(b.bn1, 1)
                        wo.gout0 = d_c2[c].data_ptr()
                    wo.dw0 = tb['part_ptr'][id(b.conv2)]
                    wo.in_amax, wo.gout0_amax = self._amax_f(t, i, 1, c), self._amax_b(t, i, 0, c)
                    wo.single_product = int(x1)
                    wops.append(wo)
                g_w2 = self.geom('f_conv2', B, Hout, b0)
                self.wgrad_async(g_w2, wops, self.wg_n_split(g_w2), (sv['a1_h'] + d_c2_h) if pl_bwd else (sv['c1'] + d_c2),
                                 self.unpack_after(tb, [b.conv2 for b in grp]))
                # (4) BN1 backward
                run_coef(jb + 6, 3, bounds=pl_bwd)
                d_c1 = None if pl_bwd else [torch.empty(B, Hout, Hout, Cs, **f32) for _ in range(3)]
                aops = []
                for c, b in enumerate(grp):
                    ao = BnBwdApplyOperands()
                    ao.g, ao.a, ao.coef_a = d_a1[c].data_ptr(), sv['c1'][c].data_ptr(), self._bnf_ptr(b.bn1, 4)
                    if not pl_bwd:
                        ao.da = d_c1[c].data_ptr()
                    ao.da_amax = self._amax_b(t, i, 1, c)
                    aops.append(ao)
                if pl_bwd:           # d_c1 as planes only, scaled by the bound the coefficient kernel just wrote
                    d_c1_h = [torch.empty(cnt * Cs, **f32) for _ in range(3)]
                    check(L.mpose_bn_bwd_apply_h2((BnBwdApplyOperands * 3)(*aops), ptr_array(d_c1_h), None, 3, c_int64(cnt), Cs, st()),
                          'mpose_bn_bwd_apply_h2')
                else:
                    check(L.mpose_bn_bwd_apply((BnBwdApplyOperands * 3)(*aops), 3, Hout * Hout, B, Cs, 0, 0, st()), 'mpose_bn_bwd_apply')
                # (5) wgrad of conv_in + shortcut: one launch over the fused forward geometry
                gname = {'regular': 'f_in_regular', 'down': 'f_in_down', 'up': 'f_in_up'}[b0.kind]
                slots = B * (Hin if b0.kind == 'up' else Hout) ** 2
                wops = []
                for c, b in enumerate(grp):
                    wo = WgradOperands()
                    if pl_bwd:
                        wo.in_, wo.gout0, wo.gout1, wo.planes_in = sv['x_h'][c].data_ptr(), d_c1_h[c].data_ptr(), d_sc_h[c].data_ptr(), 1
                    else:
                        wo.in_ = sv['x'][c].data_ptr()
                        wo.gout0, wo.gout1 = d_c1[c].data_ptr(), d_sc[c].data_ptr()
                    wo.dw0, wo.dw1 = tb['part_ptr'][id(b.conv_in)], tb['part_ptr'][id(b.conv_sc)]
                    wo.in_amax = self._amax_f(t, i, 0, _first_same(sv['x'], c))
                    wo.gout0_amax, wo.gout1_amax = self._amax_b(t, i, 1, c), self._amax_b(t, i, 2, c)
                    wo.single_product = int(x1)
                    wops.append(wo)
                g_w1 = self.geom(gname, B, Hin, b0)
                self.wgrad_async(g_w1, wops, self.wg_n_split(g_w1), (list(sv['x_h']) + d_c1_h + d_sc_h) if pl_bwd else (list(sv['x']) + d_c1 + d_sc),
                                 self.unpack_after(tb, [b.conv_in for b in grp] + [b.conv_sc for b in grp]))
                # (6) dgrad of conv_in + the shortcut's dgrad: one launch, the shortcut as a tap on a second input
                din_h2 = blk_h2 and self.h2_planes       # on conv_h2r_k (its weights are packed for it): both inputs as planes
                if din_h2 and not pl_bwd:                # (an eval-mode backward: measured and split like d_c2 above)
                    d_c1_h = self.split_h2(d_c1, [self._amax_b(t, i, 1, c) for c in range(3)], cnt, Cs)
                    d_sc_h = self.split_h2(d_sc, [self._amax_b(t, i, 2, c) for c in range(3)], cnt, Cs)
                d_x = [torch.empty(B, Hin, Hin, b0.cin_s, **f32) for _ in range(3)]
                kd = {'regular': 'd_in_regular', 'down': 'd_in_down', 'up': 'd_in_up'}[b0.kind]
                ops = []
                for c, b in enumerate(grp):
                    op = ConvOperands()
                    op.in_ = (d_c1_h[c] if din_h2 else d_c1[c]).data_ptr()
                    op.in1 = (d_sc_h[c] if din_h2 else d_sc[c]).data_ptr()
                    op.w0, op.out0 = self._wptr(b.conv_in, True), d_x[c].data_ptr()
                    op.w1 = self._wptr(b.conv_sc, True)
                    op.in_amax, op.in1_amax = self._amax_b(t, i, 1, c), self._amax_b(t, i, 2, c)
                    op.w0_amax, op.w1_amax = b.conv_in.amax_ptr, b.conv_sc.amax_ptr
                    if fuse_sums:            # d_x is the previous block's g: its BatchNorm-backward sums while it is stored
                        pb, psv = self.stage_blocks[t][i - 1][c], saved[i - 1]
                        op.red_a, op.red_b = psv['c2'][c].data_ptr(), psv['sc'][c].data_ptr()
                        op.red_scale, op.red_shift = self._bnf_ptr(pb.bn2, 0), self._bnf_ptr(pb.bn2, 1)
                        op.red_sums = sp[(id(pb.bn2), 'b')] if spart else self._stats_ptr(pb.bn2, True)
                    if h2f and i >= 1 and self.stage_blocks[t][i - 1][0].h2:     # d_x is an H2 block's g: its largest magnitude for that block's bound
                        op.out0_amax = self._amax_b(t, i - 1, 3, c)
                    ops.append(op)
                self.conv(self.geom(kd, B, Hout, b0), ops, 2 | pflags | (256 if (spart and fuse_sums) else 0) | (128 if din_h2 else 0))
                sums_done = fuse_sums
                g = d_x
                if i == 5 and any(sp != 0 for sp in self.spaces):     # the permutation is an involution
                    outs = [g[c] if self.spaces[c] == 0 else torch.empty_like(g[c]) for c in range(3)]
                    spaces = (c_int * 3)(*self.spaces)
                    check(L.mpose_axis_permute(ptr_array(g), ptr_array(outs), spaces, 3, B, Sm, 192, st()), 'mpose_axis_permute')
                    g = outs
            # fan-in of the three columns (+ the cumulative gradient from later stages)
            tot = torch.empty_like(g[0])
            check(L.mpose_add(ptr(g[0]), ptr(g[1]), ptr(tot), c_int64(tot.numel()), st()), 'mpose_add')
            check(L.mpose_add(ptr(tot), ptr(g[2]), ptr(tot), c_int64(tot.numel()), st()), 'mpose_add')
            if D is not None:
                check(L.mpose_add(ptr(tot), ptr(D), ptr(tot), c_int64(tot.numel()), st()), 'mpose_add')
            D = tot
            if t > 0:
                n_part = 512             # (one 64-pixel tile per workgroup at B = 32, two workgroups per CU: 97 -> 68 us)
                w = self.combiners[t - 1]
                dwp = torch.empty(n_part * w.numel(), **f32)
                g_comb = [torch.empty_like(heat[0]) for _ in range(3)]
                prev = [hms[p][t - 1] for p in range(3)]
                check(L.mpose_combiner_bwd(ptr_array(prev), ptr(w), ptr(D), ptr_array(g_comb), ptr(dwp), n_part, B, J, F * F, 128, st()),
                      'mpose_combiner_bwd')
                check(L.mpose_reduce_partials(ptr(dwp), c_void_p(self.gflat.data_ptr() + 4 * goff[id(w)]), n_part, c_int64(w.numel()),
                                              0, st()), 'mpose_reduce_partials')
            else:
                g_comb = None
            self._finish_bucket(tb, self.T - 1 - t, t * 90, 90, works)

        dx = None
        if D is not None:
            if self.stem is not None:
                dx = self.stem.backward(ctx['stem_ctx'], D, need_dx)
            else:
                # ---- stem backward: ReLU mask, BN backward, wgrad (+ dgrad when the input wants a gradient) ----
                n = self.stem_bn
                gm = torch.empty_like(D)
                check(L.mpose_relu_bwd(ptr(D), ptr(ctx['stem_out']), ptr(gm), c_int64(gm.numel()), st()), 'mpose_relu_bwd')
                ro = BnBwdReduceOperands()
                ro.g, ro.a, ro.sums = gm.data_ptr(), ctx['stem_raw'].data_ptr(), self._stats_ptr(n, True)
                if not self.bn_bwd_reduce([ro], F * F, B, 128, (coef_base + self.T * 90 * COEF_DT.itemsize, 1, eval_bn)):
                    run_coef(self.T * 90, 1)
                d_raw = torch.empty_like(D)
                ao = BnBwdApplyOperands()
                ao.g, ao.a, ao.coef_a, ao.da = gm.data_ptr(), ctx['stem_raw'].data_ptr(), self._bnf_ptr(n, 4), d_raw.data_ptr()
                check(L.mpose_bn_bwd_apply((BnBwdApplyOperands * 3)(ao), 1, F * F, B, 128, 0, 0, st()), 'mpose_bn_bwd_apply')
                wo = WgradOperands()
                wo.in_, wo.gout0, wo.dw0 = ctx['s2d'].data_ptr(), d_raw.data_ptr(), tb['part_ptr'][id(self.stem_conv)]
                self.wgrad_async(self.geom('f_stem', B, F), [wo], self._n_split(B * F * F, 2, 1), [ctx['s2d'], d_raw],
                                 self.unpack_after(tb, [self.stem_conv]))
                if need_dx:
                    d_s2d = torch.empty_like(ctx['s2d'])
                    op = ConvOperands()
                    op.in_, op.w0, op.out0 = d_raw.data_ptr(), self._wptr(self.stem_conv, True), d_s2d.data_ptr()
                    self.conv(self.geom('d_stem', B, F), [op])
                    dx = torch.empty(ctx['x_shape'], **f32)
                    check(L.mpose_depth_to_space8(ptr(d_s2d), ptr(dx), B, ctx['x_shape'][2], st()), 'mpose_depth_to_space8')
            # ---- the stem's packed partial sums -> torch-layout gradients; last bucket ----
            self._finish_bucket(tb, self.T, self.T * 90, tb['n_unpack'] - self.T * 90, works)
        if self.dp is not None:
            _lib.plan_host(self._wait_collectives)                   # (stream-ordered for RCCL: the current stream waits for the collective)
        ctx['done'] = True
        return self.gflat, dx

    def _wait_collectives(self):
        for w in self._dp_works:
            w.wait()                   # (stream-ordered for RCCL: the current stream waits for the collective)
        del self._dp_works[:]

    def _finish_bucket(self, tb, bucket, first_job, n_jobs, works):
        """Stage `bucket`'s weight-gradient partials -> flat gradient slice (one unpack launch over its job range), then,
        under data parallelism, one asynchronous all-reduce of that slice that overlaps the rest of the backward."""
        if self.overlap_wgrad and self.side_stream is not None:
            _lib.stream_wait(torch.cuda.current_stream(), self.side_stream)
        self._side_keep.clear()            # (their memory may now be recycled by main-stream allocations)
        if n_jobs > 0 and not self.inline_unpack:
            check(lib().mpose_unpack_wgrads(c_void_p(tb['unpack'].data_ptr() + first_job * UNPACK_DT.itemsize), n_jobs,
                                            tb['unpack_max'], stream_ptr()), 'mpose_unpack_wgrads')
        if self.dp is not None:
            lo, hi = self._buckets[bucket]
            if hi > lo:
                sl, grp = self.gflat[lo:hi], self.dp[0]
                # (a host action: a launch plan marks the point and re-issues the collective there on every replay)
                _lib.plan_host(lambda: works.append(torch.distributed.all_reduce(sl, op=torch.distributed.ReduceOp.SUM, group=grp, async_op=True)))

    @staticmethod
    def _n_split(slots, tiles=27, groups=3, occ=1):
        """Split-K factor of the weight-gradient GEMMs.  `tiles` = (tap entries) x (Cin tiles) x (Cout tiles) of ONE
        column (mpose_conv_wgrad_tiles); the launch has `groups` columns and runs tiles*groups*n workgroups, one
        per CU at a time.  Pick the n whose last round of 256 is fullest, discounted by the fixed per-workgroup cost
        (LDS reduction, pipeline fill) so that short row ranges are not split further than they pay for."""
        rows = max(1, slots // 32)
        best, best_score = 1, -1.0
        per_round = 256 * occ          # (`occ` workgroups share a CU: the narrow tiles of the row-of-taps kernel, mpose_conv_wgrad_occupancy)
        for n in range(1, 64 * occ + 1):
            if n > 1 and rows < 8 * n:
                break
            wgs = tiles * groups * n
            eff = wgs / (float(per_round) * ((wgs + per_round - 1) // per_round))
            work = rows / (4.0 * n)
            score = eff * work / (work + _WG_FIXED)
            if score > best_score + 1e-9:
                best, best_score = n, score
        return best

    @staticmethod
    def _wg_blocks(c):
        return 4 if c % 128 == 0 else (3 if c % 96 == 0 else (2 if c % 64 == 0 else 1))

    def grads_from_flat(self, flat):
        out = []
        for p, o in zip(self.param_list(), self._grad_offsets):
            out.append(flat[o:o + p.numel()].view(p.shape))
        return out
