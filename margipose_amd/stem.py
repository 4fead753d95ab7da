"""Image feature extractors of the reference (models/margipose_model.py:103-139), executed with the same gfx950
kernels as the columns:
  * 'inceptionv4' (the default, :104-118): `inceptionv4().features[0:7]` with every Conv2d / MaxPool2d padding
    rewritten to k//2, then Conv2d(384,128,1) + BatchNorm2d + ReLU;
  * 'resnet18' / 'resnet34' / 'resnet50' (:119-137): torchvision's conv1, bn1, relu, maxpool, layer1, layer2, plus
    Conv2d(512,128,1) + BatchNorm2d + ReLU for resnet50 (layer2 of the BasicBlock nets already has 128 channels).

The layer definitions come from the third-party package pretrainedmodels==0.6.0, which is NOT in the reference
tree: they are restated from SURVEY.md Appendix B (shape-checked there: 972,896 parameters, 2.18 GMAC/image) and
CANNOT be pinned against the original here -> stem parity is "unpinned" (checked only against this repo's own
oracle restatement, oracle/model_ref.py::inceptionv4_stem).  ImageNet weights cannot be downloaded either: the
stem starts from PyTorch's default initialisation.

Execution model: a small static graph.  Every tensor ("node") is stored RAW (pre-BatchNorm) in NHWC together with
per-channel (scale, shift) vectors; consumers apply relu(scale*x+shift) while staging their input (convs) or
reading it (pools).  Concatenations are channel slices of one wider node (convs write with a leading dimension).
Max/avg-pool outputs are "identity" channel ranges (scale 1, shift 0).  Backward walks the graph in reverse:
masked BatchNorm backward over a whole node, then data-/weight-gradients of the producing convs.
ResNet's residual sums are identity nodes too: s_raw = bn2(c2) + relu(prev) (or + bn_d(downsample)) is materialised
by mpose_bn_add_fwd, and the block's post-add ReLU is the consumers' relu(1*s_raw + 0); the conv outputs feeding the
sum (BatchNorm WITHOUT ReLU) are nodes with relu=False, whose BatchNorm backward runs unmasked.

torchvision (0.3.0, requirements.txt:7) is not in the reference tree either: the ResNet layer definitions are
restated from the published architecture (He et al. 2015; torchvision's v1.5 Bottleneck with the stride on the 3x3),
and, like the InceptionV4 stem, can only be checked against this repo's oracle restatement (parity unpinned).
"""
import ctypes
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import (BnAddOperands, BnBwdApplyOperands, BnBwdReduceOperands, ConvOperands, WgradOperands, c_int64, c_void_p, check, lib, ptr,
                   stream_ptr)

_POOL_ARG = True             # the max pools' window choices are kept from the forward pass (round 5: -0.10 ms per step against recomputing them)
_FIRST_WRITES = True         # a node gradient's first contribution WRITES where its launch covers the node (round 5: -0.21 ms against zero fills)

BN_EPS_STEM = 1e-3        # BasicConv2d's BatchNorm2d(eps=0.001)


# ---------------------------------------------------------------------------------------------
# parameter holders (state_dict key layout of pretrainedmodels' InceptionV4 blocks)
# ---------------------------------------------------------------------------------------------
class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride=1):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.conv = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k[0] // 2, k[1] // 2), bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=BN_EPS_STEM, momentum=0.1, affine=True)
        self.relu = nn.Identity()


class Mixed_3a(nn.Module):
    def __init__(self):
        super().__init__()
        self.maxpool = nn.Identity()
        self.conv = BasicConv2d(64, 96, 3, stride=2)


class Mixed_4a(nn.Module):
    def __init__(self):
        super().__init__()
        self.branch0 = nn.Sequential(BasicConv2d(160, 64, 1), BasicConv2d(64, 96, 3))
        self.branch1 = nn.Sequential(BasicConv2d(160, 64, 1), BasicConv2d(64, 64, (1, 7)), BasicConv2d(64, 64, (7, 1)),
                                     BasicConv2d(64, 96, 3))


class Mixed_5a(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = BasicConv2d(192, 192, 3, stride=2)
        self.maxpool = nn.Identity()


class Inception_A(nn.Module):
    def __init__(self):
        super().__init__()
        self.branch0 = BasicConv2d(384, 96, 1)
        self.branch1 = nn.Sequential(BasicConv2d(384, 64, 1), BasicConv2d(64, 96, 3))
        self.branch2 = nn.Sequential(BasicConv2d(384, 64, 1), BasicConv2d(64, 96, 3), BasicConv2d(96, 96, 3))
        self.branch3 = nn.Sequential(nn.Identity(), BasicConv2d(384, 96, 1))


def make_inceptionv4_stem_modules():
    return nn.Sequential(BasicConv2d(3, 32, 3, stride=2), BasicConv2d(32, 32, 3), BasicConv2d(32, 64, 3), Mixed_3a(), Mixed_4a(),
                         Mixed_5a(), Inception_A(), nn.Conv2d(384, 128, 1), nn.BatchNorm2d(128), nn.Identity())


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock's parameters (conv1, bn1, conv2, bn2, downsample.{0,1})."""
    expansion = 1

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.stride = stride
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck's parameters (stride on conv2, "v1.5")."""
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))


RESNETS = {'resnet18': (BasicBlock, (2, 2)), 'resnet34': (BasicBlock, (3, 4)), 'resnet50': (Bottleneck, (3, 4))}


def make_resnet_stem_modules(name):
    """nn.Sequential(conv1, bn1, relu, maxpool, layer1, layer2[, conv, bn, relu]) with torchvision's module names and
    initialisation (kaiming_normal_ fan_out for convs, BN weight 1 / bias 0); the reference loads ImageNet weights
    (pretrained=True, :120), which cannot be downloaded here."""
    block, (n1, n2) = RESNETS[name]
    e = block.expansion
    layer1 = nn.Sequential(*[block(64 if i == 0 else 64 * e, 64, 1) for i in range(n1)])
    layer2 = nn.Sequential(*[block(64 * e if i == 0 else 128 * e, 128, 2 if i == 0 else 1) for i in range(n2)])
    mods = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.Identity(), nn.Identity(), layer1, layer2]
    for m in mods[:1] + list(layer1.modules()) + list(layer2.modules()):
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
    if 128 * e != 128:
        mods += [nn.Conv2d(128 * e, 128, 1), nn.BatchNorm2d(128), nn.Identity()]
    return nn.Sequential(*mods)


# ---------------------------------------------------------------------------------------------
# graph description
# ---------------------------------------------------------------------------------------------
class _Node:
    def __init__(self, name, C, div, relu=True):
        self.name, self.C, self.div = name, C, div      # spatial size = input size // div; div = (div_y, div_x) when they differ
        self.relu = relu                                 # consumers see relu(scale*x+shift) (False: scale*x+shift)
        self.parts = []                                  # (c0, c1, bn_module or None, eps, conv_bias)
        self.producers = []
        self.f_off = self.s_off = -1
        self.is_image = False

    def hw(self, S):
        dy, dx = self.div if isinstance(self.div, tuple) else (self.div, self.div)
        assert S % dy == 0 and S % dx == 0
        return S // dy, S // dx


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class _ConvOp:
    """Conv2d (padding k//2 * dilation unless given) or ConvTranspose2d; stride / dilation / padding may differ per axis.
    flat=True relabels a (1, k) kernel that spans the whole width of its input (models/chatterbox_model.py:104,112): the
    NHWC tensors are read as one row of H*W pixels, the kernel walks it with stride k (a one-pixel-wide slot grid would not
    satisfy the weight-gradient kernel's GW % 8 == 0)."""

    def __init__(self, src, dst, c0, basic, stride=1, bn=None, eps=None, bias=None, dilation=1, padding=None, flat=False):
        self.src, self.dst, self.c0 = src, dst, c0
        self.stride, self.dilation, self.flat = _pair(stride), _pair(dilation), flat
        if isinstance(basic, (nn.Conv2d, nn.ConvTranspose2d)):   # plain convolution: the BatchNorm that follows is given explicitly
            self.weight, self.bn, self.eps, self.bias = basic.weight, bn, (0.0 if eps is None else eps), bias
        else:                                             # pretrainedmodels' BasicConv2d
            self.weight, self.bn, self.eps, self.bias = basic.conv.weight, basic.bn, BN_EPS_STEM, None
        self.transposed = isinstance(basic, nn.ConvTranspose2d)
        if self.transposed:
            self.cin, self.cout, self.kh, self.kw = self.weight.shape
        else:
            self.cout, self.cin, self.kh, self.kw = self.weight.shape
        self.padding = _pair(padding) if padding is not None else (self.kh // 2 * self.dilation[0], self.kw // 2 * self.dilation[1])
        self.cout_s = -(-self.cout // 32) * 32            # storage channels of the output slice (heatmap convolutions: 17 -> 32)
        self.conv = None                                  # engine._Conv (packing slots)


class _FirstConvOp(_ConvOp):
    """Conv2d(3, C, k, stride 2, padding k//2) on the image, run as a 1x1 convolution over gathered k x k x 3 patches
    (mpose_im2col_k3s2 / mpose_im2col_s2): K = 27 (+5 zero) or 147 (+13 zero) instead of k*k taps x 32 padded channels."""

    def __init__(self, src, dst, c0, basic, **kw):
        super().__init__(src, dst, c0, basic, 1, **kw)
        k = self.weight.shape[2]
        assert tuple(self.weight.shape[1:]) == (3, k, k)
        self.cin, self.kh, self.kw = 3 * k * k, 1, 1
        self.padding = (0, 0)


class _AddOp:
    """dst_raw = [relu](affine(a)) + affine(b): the residual sum of a ResNet block (post-add ReLU is the consumers')."""

    def __init__(self, a, b, dst, relu_a):
        self.a, self.b, self.dst, self.relu_a = a, b, dst, relu_a
        self.src, self.c0 = b, 0


class _PoolOp:
    def __init__(self, src, dst, c0, kind):
        self.src, self.dst, self.c0, self.kind = src, dst, c0, kind


class _GraphStem:
    """Owns a stem's graph, arenas and job tables; driven by engine.Engine.  Subclasses build `ops` / `nodes`."""

    IMG_C = 32          # the image enters as gathered k x k x 3 patches zero padded to IMG_C channels (Cin % 32 == 0)
    IMG_K = 3

    def _finish(self, engine, seq, ops, nodes, out_node, extra_params):
        from .engine import _Conv
        self.engine, self.seq = engine, seq
        self.ops, self.nodes, self.out_node = ops, nodes, out_node
        for op in ops:
            op.dst.producers.append(op)
            if isinstance(op, _ConvOp):
                op.dst.parts.append((op.c0, op.c0 + (op.cout if op.bn is not None else op.cout_s), op.bn, op.eps, op.bias))
                cin_s = self.IMG_C if op.src.is_image else op.cin
                op.conv = _Conv(op.weight, op.transposed, 1, op.cin, op.cout, cin_s, op.cout_s)
                op.conv.T = op.kh * op.kw
                op.conv.kk = op.kh * op.kw
                op.conv.size_g = op.conv.T * cin_s * op.conv.npad_f
                op.conv.size_f = op.conv.size_g * 3 // 2
                op.conv.size_d = op.conv.T * op.cout_s * op.conv.npad_d * 3 // 2
                op.conv.generic = True
            elif isinstance(op, _AddOp):
                op.dst.parts.append((0, op.dst.C, None, 0.0, None))
            else:
                op.dst.parts.append((op.c0, op.c0 + op.src.C, None, 0.0, None))
        order = dict((id(n), i) for i, n in enumerate(nodes))
        for op in ops:                                     # the backward pass walks `nodes` in reverse: sources before destinations
            for t in ([op.a, op.b] if isinstance(op, _AddOp) else [op.src]):
                assert order[id(t)] < order[id(op.dst)], 'nodes must be created in topological order (%s -> %s)' % (t.name, op.dst.name)
        self.convs = [op.conv for op in ops if isinstance(op, _ConvOp)]
        self.out_nodes = None                              # several RAW outputs instead of one activated one (ChatterboxGraph)
        self.bn_modules = [p[2] for n in self.nodes for p in n.parts if p[2] is not None]
        self.extra_params = extra_params
        self._tables = {}
        self._geoms = {}

    @staticmethod
    def _node_factory():
        N = []

        def node(name, C, div, relu=True):
            N.append(_Node(name, C, div, relu))
            return N[-1]
        return N, node

    # ------------------------------------------------------------------ arenas
    def setup(self, device):
        foff = soff = 0
        for n in self.nodes:
            n.f_off, n.s_off = foff, soff
            foff += 8 * n.C
            soff += 6 * n.C
        self.f_arena = torch.zeros(foff, dtype=torch.float32, device=device)
        self.s_arena = torch.zeros(soff, dtype=torch.float64, device=device)
        # MPOSE_CONV_F16X3: one amax slot per node for what its consumers read (relu(bn(raw)); the image patches as they are) and
        # one for the gradient w.r.t. its raw values
        from .engine import AMAX_SLOT
        self.amax_f = torch.zeros(len(self.nodes) * AMAX_SLOT, dtype=torch.float32, device=device)
        self.amax_b = torch.zeros(len(self.nodes) * AMAX_SLOT, dtype=torch.float32, device=device)
        for i, n in enumerate(self.nodes):
            n.amax_f = self.amax_f.data_ptr() + 4 * AMAX_SLOT * i
            n.amax_b = self.amax_b.data_ptr() + 4 * AMAX_SLOT * i
        for n in self.nodes:       # identity parts: scale 1, shift 0, backward coefficient c0 = 1
            for c0, c1, bn, _, _ in n.parts:
                if bn is None:
                    self.f_arena[n.f_off + c0:n.f_off + c1] = 1.0
                    self.f_arena[n.f_off + 4 * n.C + c0:n.f_off + 4 * n.C + c1] = 1.0
        self._tables = {}

    def fptr(self, n, slot, c0=0):
        """slot: 0 scale, 1 shift, 2 mean, 3 invstd, 4 coef (4*C)."""
        return self.f_arena.data_ptr() + 4 * (n.f_off + slot * n.C + c0)

    def sptr(self, n, bwd=False, c0=0):
        return self.s_arena.data_ptr() + 8 * (n.s_off + (2 * n.C if bwd else 0) + c0 * (4 if bwd else 2))

    def tables(self, B, S):
        from .engine import BN_DT, COEF_DT, _jobs_to_device
        key = (B, S)
        if key in self._tables:
            return self._tables[key]
        eng = self.engine
        goff = dict((id(p), o) for p, o in zip(eng.param_list(), eng._grad_offsets))
        gbase = eng.gflat.data_ptr()
        fin, coef = [], []
        tb = {'fin_range': {}, 'coef_range': {}, 'fin_job': {}}
        for n in self.nodes:
            f0, c0_ = len(fin), len(coef)
            for (a, b, bn, eps, bias) in n.parts:
                if bn is None:
                    continue
                H, W = n.hw(S)
                j = np.zeros(1, dtype=BN_DT)[0]
                j['stats'] = self.sptr(n, False, a)
                j['gamma'] = bn.weight.data_ptr(); j['beta'] = bn.bias.data_ptr()
                j['running_mean'] = bn.running_mean.data_ptr(); j['running_var'] = bn.running_var.data_ptr()
                j['scale'] = self.fptr(n, 0, a); j['shift'] = self.fptr(n, 1, a)
                j['mean'] = self.fptr(n, 2, a); j['invstd'] = self.fptr(n, 3, a)
                j['C'] = b - a; j['count'] = B * H * W
                j['conv_bias'] = bias.data_ptr() if bias is not None else 0
                j['eps'] = eps
                if all(pp[2] is not None for pp in n.parts):      # every channel of the node is a BatchNorm output: an a-priori bound exists
                    j['bound_out'] = n.amax_f
                tb['fin_job'][(n.name, a)] = len(fin)          # (the job of the convolution that writes channels a.. of this node)
                fin.append(j)
                k = np.zeros(1, dtype=COEF_DT)[0]
                k['sums'] = self.sptr(n, True, a)
                k['gamma'] = bn.weight.data_ptr(); k['mean'] = self.fptr(n, 2, a); k['invstd'] = self.fptr(n, 3, a)
                k['coef'] = self.fptr(n, 4, a)
                k['dgamma'] = gbase + 4 * goff[id(bn.weight)]; k['dbeta'] = gbase + 4 * goff[id(bn.bias)]
                k['sums_stride'], k['which'], k['C'], k['c_stride'], k['count'], k['sg_col'] = 4, 1, b - a, n.C, B * H * W, 0
                k['dconv_bias'] = gbase + 4 * goff[id(bias)] if bias is not None else 0
                coef.append(k)
            tb['fin_range'][n.name] = (f0, len(fin) - f0)
            tb['coef_range'][n.name] = (c0_, len(coef) - c0_)
        # MPOSE_CONV_STATS_PART: one buffer of per-workgroup partial rows (header + [rows][cout][2]) per convolution that feeds a BatchNorm
        sp_off, sp = 0, {}
        for op in self.ops:
            pk = (op.dst.name, getattr(op, 'c0', 0))
            if isinstance(op, _ConvOp) and pk in tb['fin_job']:
                g = self.geom(op, B, S, 'f')
                rows = -(-(g.B * g.GH * g.GW) // 64) * g.n_classes + 8        # (smallest pixel tile; slack for x-dilated residue launches)
                sp[pk] = (sp_off, rows, op.cout_s)
                sp_off += (4 + rows * op.cout_s * 2 + 3) // 4 * 4
        tb['stat_part'] = torch.zeros(max(sp_off, 4), dtype=torch.float32, device=eng.device)
        tb['sp_ptr'] = dict((k_, tb['stat_part'].data_ptr() + 4 * v[0]) for k_, v in sp.items())
        for pk, (_, rows, ld) in sp.items():
            j = fin[tb['fin_job'][pk]]
            j['part'], j['n_part'], j['part_ld'] = tb['sp_ptr'][pk], rows, ld
        tb['fin'] = _jobs_to_device(np.array(fin, dtype=BN_DT), eng.device)
        tb['coef'] = _jobs_to_device(np.array(coef, dtype=COEF_DT), eng.device)
        tb['n_fin'] = len(fin)
        tb['fin_count'] = torch.zeros(max(1, len(fin)), dtype=torch.int32, device=eng.device)
        self._tables[key] = tb
        return tb

    # ------------------------------------------------------------------ geometries
    def geom(self, op, B, S, kind):
        """kind: 'f' forward (also the weight-gradient geometry), 'd' data-gradient."""
        from .engine import _geom_flops, conv_geom
        key = (id(op), B, S, kind)
        g = self._geoms.get(key)
        if g is not None:
            return g
        src_hw, dst_hw = op.src.hw(S), op.dst.hw(S)
        if op.flat:
            src_hw, dst_hw = (1, src_hw[0] * src_hw[1]), (1, dst_hw[0] * dst_hw[1])
        cin_s = self.IMG_C if op.src.is_image else op.cin
        g = conv_geom(kind, op.transposed, B, src_hw, cin_s, dst_hw, op.cout_s, (op.kh, op.kw), op.stride, op.dilation, op.padding,
                      op.conv.npad_f if kind == 'f' else op.conv.npad_d)
        if kind == 'f':
            g.in_ld, g.out_ld0 = op.src.C, op.dst.C
        else:
            g.in_ld, g.out_ld0 = op.dst.C, op.src.C
        g._name = 'stem_%s/%s->%s/%dx%d' % (kind, op.src.name, op.dst.name, op.kh, op.kw)
        g._flops = _geom_flops(g)
        self._geoms[key] = g
        return g

    @staticmethod
    def _d_covers(op):
        """True when the data-gradient launch of `op` stores every element of the source node's gradient (so that the first
        contribution needs no zero fill): a plain Conv2d whose kernel is at least as wide as its stride on both axes."""
        return (not op.transposed and not op.flat and tuple(op.dilation) == (1, 1) and op.kh >= op.stride[0] and op.kw >= op.stride[1]
                and op.cin == op.src.C)

    # ------------------------------------------------------------------ forward
    def forward(self, x, train, save, f16=True, features=None):
        """f16: the convolutions run the three-product fp16 form (engine.py); every node's largest consumer-side magnitude is
        measured once, when its BatchNorm vectors are final.
        features (forward only, graphs that name a `feat_node`): a (B, C, H, W) tensor that takes the place of that node -- what
        its consumers read through their ReLU -- so that the heads can be driven on their own (`x` is not read)."""
        eng, L = self.engine, lib()
        if features is not None:
            B, S = features.shape[0], self.INPUT_SIZE
        else:
            B, _, S, _ = x.shape
        dev = (x if features is None else features).device
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=dev)
        tb = self.tables(B, S)
        if train:
            _lib.fill_zero(self.s_arena)       # (fills / copies through the library: a launch plan records them)
        else:
            eng.finalize_table(tb['fin'], 0, tb['n_fin'], False)
        if f16:
            _lib.fill_zero(self.amax_f)
        cflags = eng.conv_flags(2) if f16 else 0       # (three-product form, or its single-product reduced-precision variant)
        measured = set()
        raw = {}
        first_op = 0
        if features is not None:
            fn_ = self.feat_node
            assert not save and tuple(features.shape[1:]) == (fn_.C,) + fn_.hw(S)
            raw[fn_.name] = features.permute(0, 2, 3, 1).contiguous()
            first_op = 1 + max(i for i, op in enumerate(self.ops) if op.dst is fn_)
            return self._run_ops(raw, first_op, B, S, train, save, f16, cflags, measured, tb)
        img = self.nodes[0]
        raw[img.name] = torch.empty(B, S // 2, S // 2, self.IMG_C, **f32)
        is_u8 = x.dtype == torch.uint8     # raw RGB frames: to_tensor + normalisation fused into the gather
        mean3 = (ctypes.c_float * 3)(*eng.input_norm[0]) if is_u8 else None
        std3 = (ctypes.c_float * 3)(*eng.input_norm[1]) if is_u8 else None
        if self.IMG_K == 3:
            check(L.mpose_im2col_k3s2(ctypes.c_void_p(x.data_ptr()), int(is_u8), mean3, std3, ptr(raw[img.name]), B, S, S, st()),
                  'mpose_im2col_k3s2')
        else:
            check(L.mpose_im2col_s2(ctypes.c_void_p(x.data_ptr()), int(is_u8), mean3, std3, ptr(raw[img.name]), B, S, S, self.IMG_K,
                                    self.IMG_C, st()), 'mpose_im2col_s2')
        return self._run_ops(raw, first_op, B, S, train, save, f16, cflags, measured, tb)

    def _run_ops(self, raw, first_op, B, S, train, save, f16, cflags, measured, tb):
        eng, L = self.engine, lib()
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=next(iter(raw.values())).device)
        done = set(id(op) for op in self.ops[:first_op])
        pool_args = {}
        for op in self.ops[first_op:]:
            n = op.dst
            if n.name not in raw:
                H, W = n.hw(S)
                raw[n.name] = torch.empty(B, H, W, n.C, **f32)
            src = op.src
            sc = None if src.is_image else self.fptr(src, 0)
            sh = None if src.is_image else self.fptr(src, 1)
            if isinstance(op, _ConvOp):
                o = ConvOperands()
                o.in_, o.w0 = raw[src.name].data_ptr(), eng._wptr(op.conv)
                o.in_scale, o.in_shift = sc, sh
                o.out0 = raw[n.name].data_ptr() + 4 * op.c0
                spart = train and eng.part_stats() and (n.name, op.c0) in tb['sp_ptr']
                if spart:           # statistics as per-workgroup partial rows (no fp64 atomics); the finalize kernel adds them up
                    o.stats0 = tb['sp_ptr'][(n.name, op.c0)]
                elif train:          # (a producer without a partial-row buffer: fp64 atomics into the statistics arena)
                    o.stats0 = self.sptr(n, False, op.c0)
                if f16:
                    if src.name not in measured:       # (all of the node's channels: a bound for any channel slice of it)
                        eng.absmax([raw[src.name]], [src.amax_f], src.C, None if sc is None else [sc], None if sc is None else [sh],
                                   relu=sc is not None)
                        measured.add(src.name)
                    o.in_amax, o.w0_amax = src.amax_f, op.conv.amax_ptr
                eng.conv(self.geom(op, B, S, 'f'), [o], cflags | (256 if spart else 0))
            elif isinstance(op, _AddOp):
                ao = BnAddOperands()
                ao.a, ao.a_scale, ao.a_shift = raw[op.a.name].data_ptr(), self.fptr(op.a, 0), self.fptr(op.a, 1)
                ao.b, ao.b_scale, ao.b_shift = raw[op.b.name].data_ptr(), self.fptr(op.b, 0), self.fptr(op.b, 1)
                ao.out = raw[n.name].data_ptr()
                H, W = n.hw(S)
                check(L.mpose_bn_add_fwd((BnAddOperands * 3)(ao), 1, H * W, B, n.C, 0 if op.relu_a else 2, 0, st()), 'mpose_bn_add_fwd')
            else:
                Hs = src.hw(S)[0]       # (pools: square maps only)
                if save and op.kind == 0 and _POOL_ARG:
                    # (a differentiable forward keeps the max pool's window choices, a byte per pooled element: its backward then
                    #  needs neither the pre-pool tensor nor a pass over it)
                    arg = torch.empty(B * n.hw(S)[0] * n.hw(S)[1] * src.C, dtype=torch.uint8, device=raw[src.name].device)
                    check(L.mpose_maxpool3_fwd_arg(ptr(raw[src.name]), c_void_p(sc), c_void_p(sh), c_void_p(raw[n.name].data_ptr() + 4 * op.c0),
                                                   ptr(arg), ctypes.c_long(arg.numel()), B, Hs, Hs, src.C, n.C, st()), 'mpose_maxpool3_fwd_arg')
                    pool_args[id(op)] = arg
                else:
                    check(L.mpose_pool3_fwd(ptr(raw[src.name]), c_void_p(sc), c_void_p(sh), c_void_p(raw[n.name].data_ptr() + 4 * op.c0),
                                            B, Hs, Hs, src.C, n.C, op.kind, st()), 'mpose_pool3_fwd')
            done.add(id(op))
            if train and all(id(p) in done for p in n.producers):
                f0, nf = tb['fin_range'][n.name]
                if nf:
                    bounded = f16 and eng.stem_bounds and all(pp[2] is not None for pp in n.parts)
                    eng.finalize_table(tb['fin'], f0, nf, True, eng.part_stats(), bounds=bounded)
                    if bounded:          # the node's amax slot now holds max_c(|gamma_c| sqrt(N) + |beta_c|): no measuring pass
                        measured.add(n.name)
        if self.out_nodes is not None:
            out = [raw[n.name] for n in self.out_nodes]
        else:
            n7 = self.out_node
            out = torch.empty(B, S // n7.div, S // n7.div, n7.C, **f32)
            check(L.mpose_bn_relu_fwd(ptr(raw[n7.name]), c_void_p(self.fptr(n7, 0)), c_void_p(self.fptr(n7, 1)), ptr(out),
                                      c_int64(out.numel()), n7.C, st()), 'mpose_bn_relu_fwd')
        ctx = {'raw': raw, 'B': B, 'S': S, 'train': train, 'f16': f16, 'cflags': cflags, 'measured': measured, 'pool_args': pool_args} if save else None
        return out, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, D, need_dx):
        """D: gradient w.r.t. the activated stem output (B, F, F, 128) -- or, with `out_nodes`, the list of gradients w.r.t.
        the raw output nodes (None where an output took no part in the loss).  Returns dx (NCHW) or None."""
        eng, L = self.engine, lib()
        B, S, raw = ctx['B'], ctx['S'], ctx['raw']
        dev = next(iter(raw.values())).device
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=dev)
        tb = self.tables(B, S)
        _lib.fill_zero(self.s_arena)
        f16 = ctx.get('f16', False)
        cflags = ctx.get('cflags', 32 if f16 else 0)
        if f16:
            _lib.fill_zero(self.amax_b)
        fresh = set()         # nodes whose gradient tensor is allocated but not yet written
        if self.out_nodes is not None:
            dact = dict((n.name, g) for n, g in zip(self.out_nodes, D) if g is not None)
        else:
            dact = {self.out_node.name: D}
        for n in reversed(self.nodes):
            if n.is_image or n.name not in dact:
                continue
            H, W = n.hw(S)
            g = dact[n.name]
            # masked BatchNorm backward over the whole node (identity channel ranges: plain ReLU mask)
            ro = BnBwdReduceOperands()
            ro.g, ro.a, ro.sums = g.data_ptr(), raw[n.name].data_ptr(), self.sptr(n, True)
            if n.relu:
                ro.a_scale, ro.a_shift = self.fptr(n, 0), self.fptr(n, 1)
            c0_, nc = tb['coef_range'][n.name]
            eval_bn = 0 if ctx.get('train', True) else 1
            # (the node's coefficient jobs run in the reduction's finishing pass: one launch fewer per node)
            if not eng.bn_bwd_reduce([ro], H * W, B, n.C, (tb['coef'].data_ptr() + c0_ * eng.COEF_ITEMSIZE, nc, eval_bn) if nc else None) and nc:
                check(L.mpose_bn_bwd_coef(c_void_p(tb['coef'].data_ptr() + c0_ * eng.COEF_ITEMSIZE), nc, eval_bn, st()), 'mpose_bn_bwd_coef')
            d_raw = torch.empty(B, H, W, n.C, **f32)
            ao = BnBwdApplyOperands()
            ao.g, ao.a, ao.coef_a, ao.da = g.data_ptr(), raw[n.name].data_ptr(), self.fptr(n, 4), d_raw.data_ptr()
            if n.relu:
                ao.a_scale, ao.a_shift = self.fptr(n, 0), self.fptr(n, 1)
            if f16:
                ao.da_amax = n.amax_b
            check(L.mpose_bn_bwd_apply((BnBwdApplyOperands * 3)(ao), 1, H * W, B, n.C, 0, 0, st()), 'mpose_bn_bwd_apply')
            # (convolutions whose data-gradient can WRITE a source node's gradient go first: a pool reading the same source -- the
            #  max-pool / strided-convolution pairs of Mixed_3a and Mixed_5a -- then accumulates into it, and nobody zero-fills)
            prods = sorted(n.producers, key=lambda op: 0 if (_FIRST_WRITES and isinstance(op, _ConvOp) and self._d_covers(op)) else 1)
            for op in prods:
                if isinstance(op, _AddOp):     # both addends receive d_raw (w.r.t. their affine / activated values)
                    for t in (op.a, op.b):
                        if t.name not in dact:       # (the identity path has other consumers that accumulate into it later)
                            dact[t.name] = _lib.copy_into(torch.empty_like(d_raw), d_raw) if (t is op.a and op.relu_a) else d_raw
                        else:
                            acc = dact[t.name]
                            check(L.mpose_add(ptr(acc), ptr(d_raw), ptr(acc), c_int64(acc.numel()), st()), 'mpose_add')
                    continue
                src = op.src
                Hs, Ws = src.hw(S)
                want_dsrc = (not src.is_image) or need_dx
                if want_dsrc and src.name not in dact:
                    # the gradient of a node is the sum of its consumers' contributions: the first one WRITES it when it is a
                    # convolution whose data-gradient launch covers every pixel of the source (every output phase has a tap:
                    # kernel >= stride); anything else (pools accumulate; the image) starts from zeros
                    dact[src.name] = torch.empty(B, Hs, Ws, src.C, **f32)
                    if _FIRST_WRITES and isinstance(op, _ConvOp) and not src.is_image and self._d_covers(op):
                        fresh.add(src.name)
                    else:
                        _lib.fill_zero(dact[src.name])
                sc = None if src.is_image else self.fptr(src, 0)
                sh = None if src.is_image else self.fptr(src, 1)
                if isinstance(op, _ConvOp):
                    wo = WgradOperands()
                    wo.in_, wo.in_scale, wo.in_shift = raw[src.name].data_ptr(), sc, sh
                    wo.gout0 = d_raw.data_ptr() + 4 * op.c0
                    wo.dw0 = eng.part_ptr(B, S, op.conv)
                    if f16:
                        wo.in_amax, wo.gout0_amax = src.amax_f, n.amax_b
                        wo.single_product = int(bool(cflags & 64))
                    eng.wgrad_async(self.geom(op, B, S, 'f'), [wo], eng.stem_n_split(B, S, op), [raw[src.name], d_raw],
                                    eng.unpack_after(eng._tables_for(B, S // 8), [op.conv]))
                    if want_dsrc:
                        o = ConvOperands()
                        o.in_, o.w0 = d_raw.data_ptr() + 4 * op.c0, eng._wptr(op.conv, True)
                        o.out0 = dact[src.name].data_ptr()
                        if f16:
                            o.in_amax, o.w0_amax = n.amax_b, op.conv.amax_ptr
                        eng.conv(self.geom(op, B, S, 'd'), [o], (0 if src.name in fresh else 1) | cflags)       # (first contribution: write; later ones accumulate)
                        fresh.discard(src.name)
                elif want_dsrc and op.kind == 0 and id(op) in ctx.get('pool_args', {}):
                    arg = ctx['pool_args'][id(op)]
                    check(L.mpose_maxpool3_bwd_arg(c_void_p(d_raw.data_ptr() + 4 * op.c0), ptr(arg), ctypes.c_long(arg.numel()), ptr(dact[src.name]),
                                                   B, Hs, Hs, src.C, n.C, st()), 'mpose_maxpool3_bwd_arg')
                elif want_dsrc and op.kind == 0:
                    ws = torch.empty(B * H * H * src.C, dtype=torch.uint8, device=dev)       # window arg-max positions
                    check(L.mpose_maxpool3_bwd_ws(ptr(raw[src.name]), c_void_p(sc), c_void_p(sh), c_void_p(d_raw.data_ptr() + 4 * op.c0),
                                                  ptr(dact[src.name]), ptr(ws), ctypes.c_long(ws.numel()), B, Hs, Hs, src.C, n.C, st()),
                          'mpose_maxpool3_bwd_ws')
                elif want_dsrc:
                    check(L.mpose_pool3_bwd(ptr(raw[src.name]), c_void_p(sc), c_void_p(sh), c_void_p(d_raw.data_ptr() + 4 * op.c0),
                                            ptr(dact[src.name]), B, Hs, Hs, src.C, n.C, op.kind, st()), 'mpose_pool3_bwd')
        if need_dx:
            dx = torch.empty(B, 3, S, S, **f32)
            if self.IMG_K == 3:
                check(L.mpose_col2im_k3s2(ptr(dact['img']), ptr(dx), B, S, S, st()), 'mpose_col2im_k3s2')
            else:
                check(L.mpose_col2im_s2(ptr(dact['img']), ptr(dx), B, S, S, self.IMG_K, self.IMG_C, st()), 'mpose_col2im_s2')
            return dx
        return None


class InceptionV4Stem(_GraphStem):
    def __init__(self, engine, seq):
        nodes, node = self._node_factory()
        img = node('img', self.IMG_C, 2); img.is_image = True         # 3x3x3 patches of the image at half resolution
        n0, n1, n2 = node('n0', 32, 2), node('n1', 32, 2), node('n2', 64, 2)
        n3 = node('n3', 160, 4)
        a4, b4, c4, d4, n4 = node('a4', 64, 4), node('b4', 64, 4), node('c4', 64, 4), node('d4', 64, 4), node('n4', 192, 4)
        n5 = node('n5', 384, 8)
        e6, f6, g6, p6, n6 = node('e6', 64, 8), node('f6', 64, 8), node('g6', 96, 8), node('p6', 384, 8), node('n6', 384, 8)
        n7 = node('n7', 128, 8)
        m3, m4, m5, m6 = seq[3], seq[4], seq[5], seq[6]
        ops = [
            _FirstConvOp(img, n0, 0, seq[0]), _ConvOp(n0, n1, 0, seq[1]), _ConvOp(n1, n2, 0, seq[2]),
            _PoolOp(n2, n3, 0, 0), _ConvOp(n2, n3, 64, m3.conv, 2),
            _ConvOp(n3, a4, 0, m4.branch0[0]), _ConvOp(a4, n4, 0, m4.branch0[1]),
            _ConvOp(n3, b4, 0, m4.branch1[0]), _ConvOp(b4, c4, 0, m4.branch1[1]), _ConvOp(c4, d4, 0, m4.branch1[2]),
            _ConvOp(d4, n4, 96, m4.branch1[3]),
            _ConvOp(n4, n5, 0, m5.conv, 2), _PoolOp(n4, n5, 192, 0),
            _ConvOp(n5, n6, 0, m6.branch0),
            _ConvOp(n5, e6, 0, m6.branch1[0]), _ConvOp(e6, n6, 96, m6.branch1[1]),
            _ConvOp(n5, f6, 0, m6.branch2[0]), _ConvOp(f6, g6, 0, m6.branch2[1]), _ConvOp(g6, n6, 192, m6.branch2[2]),
            _PoolOp(n5, p6, 0, 1), _ConvOp(p6, n6, 288, m6.branch3[1]),
            _ConvOp(n6, n7, 0, seq[7], bn=seq[8], bias=seq[7].bias),
        ]
        self._finish(engine, seq, ops, nodes, n7, [seq[7].bias])


class ResNetStem(_GraphStem):
    """conv1 (7x7/2 as a 1x1 over gathered patches) - bn1 - relu - maxpool(3, 2, 1) - layer1 - layer2 [- conv1x1+bn+relu]."""

    IMG_C = 160         # 7*7*3 = 147 patch values zero padded to 160
    IMG_K = 7

    def __init__(self, engine, seq):
        nodes, node = self._node_factory()
        img = node('img', self.IMG_C, 2); img.is_image = True
        c1 = node('c1', 64, 2)
        ops = [_FirstConvOp(img, c1, 0, seq[0], bn=seq[1])]
        cur = node('p1', 64, 4)
        ops.append(_PoolOp(c1, cur, 0, 0))
        div = 4
        for li, layer in ((1, seq[4]), (2, seq[5])):
            for bi, blk in enumerate(layer):
                tag = 'l%db%d' % (li, bi)
                odiv = div * blk.stride
                if isinstance(blk, BasicBlock):
                    planes = blk.conv1.weight.shape[0]
                    a = node(tag + '_c1', planes, odiv)
                    b = node(tag + '_c2', planes, odiv, relu=False)
                    ops += [_ConvOp(cur, a, 0, blk.conv1, blk.stride, bn=blk.bn1), _ConvOp(a, b, 0, blk.conv2, 1, bn=blk.bn2)]
                    cout = planes
                else:
                    planes = blk.conv1.weight.shape[0]
                    a = node(tag + '_c1', planes, div)
                    a2 = node(tag + '_c2', planes, odiv)
                    b = node(tag + '_c3', planes * 4, odiv, relu=False)
                    ops += [_ConvOp(cur, a, 0, blk.conv1, 1, bn=blk.bn1), _ConvOp(a, a2, 0, blk.conv2, blk.stride, bn=blk.bn2),
                            _ConvOp(a2, b, 0, blk.conv3, 1, bn=blk.bn3)]
                    cout = planes * 4
                if hasattr(blk, 'downsample'):
                    d = node(tag + '_d', cout, odiv, relu=False)
                    ops.append(_ConvOp(cur, d, 0, blk.downsample[0], blk.stride, bn=blk.downsample[1]))
                    s_ = node(tag + '_s', cout, odiv)
                    ops.append(_AddOp(d, b, s_, False))
                else:
                    s_ = node(tag + '_s', cout, odiv)
                    ops.append(_AddOp(cur, b, s_, True))
                cur, div = s_, odiv
        extra = []
        if len(seq) > 6:
            out = node('out', 128, div)
            ops.append(_ConvOp(cur, out, 0, seq[6], bn=seq[7], bias=seq[6].bias))
            extra = [seq[6].bias]
            cur = out
        self._finish(engine, seq, ops, nodes, cur, extra)


# ---------------------------------------------------------------------------------------------
# ChatterboxModel (reference models/chatterbox_model.py): the whole network as one graph
# ---------------------------------------------------------------------------------------------
class ChatterboxBlock(nn.Module):
    """Parameters of _ChatterboxCnn._DownBlock / _UpBlock (models/chatterbox_model.py:132-214): conv1, bn1, conv2, bn2 and,
    when the block changes stride or width, resample.{0,1}.  `up`: conv1 / resample.0 are ConvTranspose2d."""

    def __init__(self, up, cin, cout, stride=(1, 1), dilation=(1, 1), dilation_in=None, output_padding=(0, 0)):
        super().__init__()
        dilation_in = dilation if dilation_in is None else dilation_in
        self.up, self.stride, self.dilation, self.dilation_in = up, tuple(stride), tuple(dilation), tuple(dilation_in)
        if self.stride != (1, 1) or cin != cout:
            if up:
                rs = nn.ConvTranspose2d(cin, cout, kernel_size=1, stride=stride, output_padding=output_padding, bias=False)
            else:
                rs = nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)
            self.resample = nn.Sequential(rs, nn.BatchNorm2d(cout))
        else:
            self.resample = None
        if up:
            self.conv1 = nn.ConvTranspose2d(cin, cout, 3, stride=stride, padding=dilation_in, dilation=dilation_in,
                                            output_padding=output_padding, bias=False)
        else:
            self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=dilation_in, dilation=dilation_in, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)


def make_chatterbox_cnn_modules(n_joints, shrink_width):
    """(down_convs, up_convs) of _ChatterboxCnn (models/chatterbox_model.py:87-126), same module indices."""
    def f(a, b):
        return (a, b) if shrink_width else (b, a)
    D, U = (lambda *a, **k: ChatterboxBlock(False, *a, **k)), (lambda *a, **k: ChatterboxBlock(True, *a, **k))
    down = nn.Sequential(
        D(128, 256, stride=f(1, 2), dilation=f(2, 1), dilation_in=f(1, 1)), D(256, 256, dilation=f(2, 1)),
        D(256, 512, stride=f(1, 2), dilation=f(4, 1), dilation_in=f(2, 1)), D(512, 512, dilation=f(4, 1)),
        nn.Conv2d(512, 1024, kernel_size=f(1, 8), bias=False), nn.BatchNorm2d(1024), nn.Identity())
    up = nn.Sequential(
        nn.ConvTranspose2d(1024, 512, kernel_size=f(1, 8), bias=False), nn.BatchNorm2d(512), nn.Identity(),
        U(512, 512, dilation=f(4, 1)),
        U(512, 256, stride=f(1, 2), dilation=f(2, 1), dilation_in=f(4, 1), output_padding=f(0, 1)),
        U(256, 256, dilation=f(2, 1)),
        U(256, 128, stride=f(1, 2), dilation=f(1, 1), dilation_in=f(2, 1), output_padding=f(0, 1)),
        nn.Conv2d(128, n_joints, kernel_size=1, bias=False))
    return down, up


class ChatterboxGraph(_GraphStem):
    """in_cnn (ResNet-34 conv1 .. layer2, :37-54) -> three heads on its 128 x 32 x 32 output:
      xy_hm_cnn (:57-84): ResNet-34 layer3 / layer4 with their strides removed and the 3x3 convolutions dilated 2 / 4 (not the
                 first one of each layer: the reference's `elif` leaves the formerly strided convolution undilated), then a 1x1;
      zy_hm_cnn / xz_hm_cnn (:87-221): residual blocks that halve ONE axis twice (stride (1,2) / (2,1), the other axis dilated
                 instead), a kernel over the remaining 8 pixels of that axis, and the mirror-image ConvTranspose2d path back.
    Outputs: the three heads' RAW heatmap logits (B, 32, 32, 32-channel storage of which n_joints are used), NHWC."""

    IMG_C = 160
    IMG_K = 7
    INPUT_SIZE = 256         # ImageSpecs(256) (:227): the heads' fixed 32 -> 16 -> 8 -> 1 ladder needs a 32 x 32 feature map

    def __init__(self, engine, model):
        nodes, node = self._node_factory()
        fe = model.in_cnn
        img = node('img', self.IMG_C, 2); img.is_image = True
        c1 = node('c1', 64, 2)
        ops = [_FirstConvOp(img, c1, 0, fe.conv1, bn=fe.bn1)]
        cur = node('p1', 64, 4)
        ops.append(_PoolOp(c1, cur, 0, 0))

        def basic(tag, cur, blk, div, stride, dil1, dil2):
            """torchvision BasicBlock (conv1 may be strided, conv2 never is)."""
            planes = blk.conv1.weight.shape[0]
            a = node(tag + '_c1', planes, div)
            b = node(tag + '_c2', planes, div, relu=False)
            ops.extend([_ConvOp(cur, a, 0, blk.conv1, stride, bn=blk.bn1, dilation=dil1),
                        _ConvOp(a, b, 0, blk.conv2, 1, bn=blk.bn2, dilation=dil2)])
            if hasattr(blk, 'downsample'):                  # (nodes in topological order: the backward pass walks them in reverse)
                d = node(tag + '_d', planes, div, relu=False)
                ops.append(_ConvOp(cur, d, 0, blk.downsample[0], stride, bn=blk.downsample[1]))
                s_ = node(tag + '_s', planes, div)
                ops.append(_AddOp(d, b, s_, False))
            else:
                s_ = node(tag + '_s', planes, div)
                ops.append(_AddOp(cur, b, s_, True))
            return s_

        div = 4
        for li, layer in ((1, fe.layer1), (2, fe.layer2)):
            for bi, blk in enumerate(layer):
                div *= blk.stride
                cur = basic('l%db%d' % (li, bi), cur, blk, div, blk.stride, 1, 1)
        feat = cur                                          # 128 x 32 x 32
        # ---- xy head ----
        cur = feat
        for li, (layer, dil) in enumerate(((model.xy_hm_cnn.layer1, 2), (model.xy_hm_cnn.layer2, 4))):
            for bi, blk in enumerate(layer):
                cur = basic('xy%db%d' % (li, bi), cur, blk, 8, 1, 1 if bi == 0 else dil, dil)
        xy = node('xy_hm', 32, 8, relu=False)
        ops.append(_ConvOp(cur, xy, 0, model.xy_hm_cnn.hm_conv, 1))
        outs = [xy]
        # ---- zy / xz heads ----
        for name, cnn, sw in (('zy', model.zy_hm_cnn, True), ('xz', model.xz_hm_cnn, False)):
            cur, div = feat, (8, 8)
            mods = list(cnn.down_convs) + list(cnn.up_convs)
            i = 0
            while i < len(mods):
                m = mods[i]
                tag = '%s%d' % (name, i)
                if isinstance(m, ChatterboxBlock):
                    if m.up:
                        assert all(div[a] % m.stride[a] == 0 for a in (0, 1))
                        odiv = (div[0] // m.stride[0], div[1] // m.stride[1])
                    else:
                        odiv = (div[0] * m.stride[0], div[1] * m.stride[1])
                    planes = m.bn1.weight.shape[0]
                    a = node(tag + '_c1', planes, odiv)
                    b = node(tag + '_c2', planes, odiv, relu=False)
                    ops.append(_ConvOp(cur, a, 0, m.conv1, m.stride, bn=m.bn1, dilation=m.dilation_in, padding=m.dilation_in))
                    ops.append(_ConvOp(a, b, 0, m.conv2, 1, bn=m.bn2, dilation=m.dilation, padding=m.dilation))
                    if m.resample is not None:
                        d = node(tag + '_d', planes, odiv, relu=False)
                        ops.append(_ConvOp(cur, d, 0, m.resample[0], m.stride, bn=m.resample[1], padding=0))
                        s_ = node(tag + '_s', planes, odiv)
                        ops.append(_AddOp(d, b, s_, False))
                    else:
                        s_ = node(tag + '_s', planes, odiv)
                        ops.append(_AddOp(cur, b, s_, True))
                    cur, div = s_, odiv
                    i += 1
                elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and m.kernel_size != (1, 1):
                    # the kernel over the whole short axis (8 pixels): stride = kernel describes the same single position
                    k = m.kernel_size
                    if isinstance(m, nn.ConvTranspose2d):
                        odiv = (div[0] // k[0], div[1] // k[1])
                    else:
                        odiv = (div[0] * k[0], div[1] * k[1])
                    n_ = node(tag + '_k8', m.weight.shape[1] if isinstance(m, nn.ConvTranspose2d) else m.weight.shape[0], odiv)
                    ops.append(_ConvOp(cur, n_, 0, m, k, bn=mods[i + 1], padding=0, flat=sw))
                    cur, div = n_, odiv
                    i += 3                                  # (BatchNorm2d, ReLU)
                else:                                       # the final 1x1 to the heatmaps
                    h = node(name + '_hm', 32, div, relu=False)
                    ops.append(_ConvOp(cur, h, 0, m, 1))
                    outs.append(h)
                    i += 1
        self._finish(engine, model, ops, nodes, None, [])
        self.out_nodes = outs
        self.feat_node = feat
