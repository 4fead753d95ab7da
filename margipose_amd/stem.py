"""InceptionV4 feature extractor of the reference's default model (reference models/margipose_model.py:103-118:
`inceptionv4().features[0:7]` with every Conv2d / MaxPool2d padding rewritten to k//2, then Conv2d(384,128,1) +
BatchNorm2d + ReLU), executed with the same gfx950 kernels as the columns.

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
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _lib
from ._lib import (BnBwdApplyOperands, BnBwdReduceOperands, ConvOperands, WgradOperands, c_int64, c_void_p, check, lib, ptr,
                   stream_ptr)

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


# ---------------------------------------------------------------------------------------------
# graph description
# ---------------------------------------------------------------------------------------------
class _Node:
    def __init__(self, name, C, div):
        self.name, self.C, self.div = name, C, div      # spatial size = input size // div
        self.parts = []                                  # (c0, c1, bn_module or None, eps, conv_bias)
        self.producers = []
        self.f_off = self.s_off = -1
        self.is_image = False


class _ConvOp:
    def __init__(self, src, dst, c0, basic, stride=1):
        self.src, self.dst, self.c0, self.stride = src, dst, c0, stride
        self.weight = basic.weight if isinstance(basic, nn.Conv2d) else basic.conv.weight
        self.cout, self.cin, self.kh, self.kw = self.weight.shape
        self.conv = None                                  # engine._Conv (packing slots)


class _FirstConvOp(_ConvOp):
    """Conv2d(3, 32, 3, stride 2, padding 1) on the image, run as a 1x1 convolution over gathered 3x3x3 patches
    (mpose_im2col_k3s2): K = 27 (+5 zero) instead of 9 taps x 32 padded channels."""

    def __init__(self, src, dst, c0, basic):
        super().__init__(src, dst, c0, basic, 1)
        assert tuple(self.weight.shape[1:]) == (3, 3, 3)
        self.cin, self.kh, self.kw = 27, 1, 1


class _PoolOp:
    def __init__(self, src, dst, c0, kind):
        self.src, self.dst, self.c0, self.kind = src, dst, c0, kind


class InceptionV4Stem:
    """Owns the stem's graph, arenas and job tables; driven by engine.Engine."""

    IMG_C = 32          # the image enters as gathered 3x3x3 patches, 27 values zero padded to 32 (Cin % 32 == 0)

    def __init__(self, engine, seq):
        from .engine import _Conv
        self.engine = engine
        self.seq = seq
        N = {}

        def node(name, C, div):
            N[name] = _Node(name, C, div)
            return N[name]
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
            _ConvOp(n6, n7, 0, seq[7]),
        ]
        self.ops = ops
        self.nodes = [img, n0, n1, n2, n3, a4, b4, c4, d4, n4, n5, e6, f6, g6, p6, n6, n7]
        self.out_node = n7
        bn_of = {}
        for op in ops:
            op.dst.producers.append(op)
            if isinstance(op, _ConvOp):
                if op.weight is seq[7].weight:
                    op.dst.parts.append((0, 128, seq[8], 0.0, seq[7].bias))
                else:
                    owner = [m for m in seq.modules() if isinstance(m, BasicConv2d) and m.conv.weight is op.weight][0]
                    op.dst.parts.append((op.c0, op.c0 + op.cout, owner.bn, BN_EPS_STEM, None))
                cin_s = self.IMG_C if op.src.is_image else op.cin
                op.conv = _Conv(op.weight, False, 1, op.cin, op.cout, cin_s, op.cout)
                op.conv.T = op.kh * op.kw
                op.conv.kk = op.kh * op.kw
                op.conv.size_g = op.conv.T * cin_s * op.conv.npad_f
                op.conv.size_f = op.conv.size_g * 3 // 2
                op.conv.size_d = op.conv.T * op.cout * op.conv.npad_d * 3 // 2
                op.conv.generic = True
            else:
                op.dst.parts.append((op.c0, op.c0 + op.src.C, None, 0.0, None))
        self.convs = [op.conv for op in ops if isinstance(op, _ConvOp)]
        self.bn_modules = [p[2] for n in self.nodes for p in n.parts if p[2] is not None]
        self.extra_params = [seq[7].bias]
        self._tables = {}
        self._geoms = {}

    # ------------------------------------------------------------------ arenas
    def setup(self, device):
        foff = soff = 0
        for n in self.nodes:
            n.f_off, n.s_off = foff, soff
            foff += 7 * n.C
            soff += 6 * n.C
        self.f_arena = torch.zeros(foff, dtype=torch.float32, device=device)
        self.s_arena = torch.zeros(soff, dtype=torch.float64, device=device)
        for n in self.nodes:       # identity parts: scale 1, shift 0, backward coefficient c0 = 1
            for c0, c1, bn, _, _ in n.parts:
                if bn is None:
                    self.f_arena[n.f_off + c0:n.f_off + c1] = 1.0
                    self.f_arena[n.f_off + 4 * n.C + c0:n.f_off + 4 * n.C + c1] = 1.0
        self._tables = {}

    def fptr(self, n, slot, c0=0):
        """slot: 0 scale, 1 shift, 2 mean, 3 invstd, 4 coef (3*C)."""
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
        tb = {'fin_range': {}, 'coef_range': {}}
        for n in self.nodes:
            f0, c0_ = len(fin), len(coef)
            for (a, b, bn, eps, bias) in n.parts:
                if bn is None:
                    continue
                H = S // n.div
                j = np.zeros(1, dtype=BN_DT)[0]
                j['stats'] = self.sptr(n, False, a)
                j['gamma'] = bn.weight.data_ptr(); j['beta'] = bn.bias.data_ptr()
                j['running_mean'] = bn.running_mean.data_ptr(); j['running_var'] = bn.running_var.data_ptr()
                j['scale'] = self.fptr(n, 0, a); j['shift'] = self.fptr(n, 1, a)
                j['mean'] = self.fptr(n, 2, a); j['invstd'] = self.fptr(n, 3, a)
                j['C'] = b - a; j['count'] = B * H * H
                j['conv_bias'] = bias.data_ptr() if bias is not None else 0
                j['eps'] = eps
                fin.append(j)
                k = np.zeros(1, dtype=COEF_DT)[0]
                k['sums'] = self.sptr(n, True, a)
                k['gamma'] = bn.weight.data_ptr(); k['mean'] = self.fptr(n, 2, a); k['invstd'] = self.fptr(n, 3, a)
                k['coef'] = self.fptr(n, 4, a)
                k['dgamma'] = gbase + 4 * goff[id(bn.weight)]; k['dbeta'] = gbase + 4 * goff[id(bn.bias)]
                k['sums_stride'], k['which'], k['C'], k['c_stride'], k['count'], k['sg_col'] = 4, 1, b - a, n.C, B * H * H, 0
                coef.append(k)
            tb['fin_range'][n.name] = (f0, len(fin) - f0)
            tb['coef_range'][n.name] = (c0_, len(coef) - c0_)
        tb['fin'] = _jobs_to_device(np.array(fin, dtype=BN_DT), eng.device)
        tb['coef'] = _jobs_to_device(np.array(coef, dtype=COEF_DT), eng.device)
        tb['n_fin'] = len(fin)
        self._tables[key] = tb
        return tb

    # ------------------------------------------------------------------ geometries
    def geom(self, op, B, S, kind):
        """kind: 'f' forward (also the weight-gradient geometry), 'd' data-gradient."""
        from .engine import _geom, _up_classes
        key = (id(op), B, S, kind)
        g = self._geoms.get(key)
        if g is not None:
            return g
        Hin, Hout = S // op.src.div, S // op.dst.div
        cin_s = self.IMG_C if op.src.is_image else op.cin
        taps = [(ky - op.kh // 2, kx - op.kw // 2, ky * op.kw + kx) for ky in range(op.kh) for kx in range(op.kw)]
        if kind == 'f':
            g = _geom(B, Hin, cin_s, Hout, op.cout, 0, Hout, op.stride, 1, [(0, 0, [(dy, dx, w, 0) for dy, dx, w in taps])],
                      op.conv.npad_f)
            g.in_ld, g.out_ld0 = op.src.C, op.dst.C
        else:
            if op.stride == 1:
                g = _geom(B, Hout, op.cout, Hin, cin_s, 0, Hin, 1, 1, [(0, 0, [(-dy, -dx, w, 0) for dy, dx, w in taps])],
                          op.conv.npad_d)
            else:       # gradient of a stride-2 3x3 (pad 1): the transposed-conv parity classes
                assert (op.kh, op.kw, op.stride) == (3, 3, 2)
                g = _geom(B, Hout, op.cout, Hin, cin_s, 0, Hout, 1, 2, _up_classes(False), op.conv.npad_d)
            g.in_ld, g.out_ld0 = op.dst.C, op.src.C
        g._name = 'stem_%s/%s->%s/%dx%d' % (kind, op.src.name, op.dst.name, op.kh, op.kw)
        from .engine import _geom_flops
        g._flops = _geom_flops(g)
        self._geoms[key] = g
        return g

    # ------------------------------------------------------------------ forward
    def forward(self, x, train, save):
        eng, L = self.engine, lib()
        B, _, S, _ = x.shape
        dev = x.device
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=dev)
        tb = self.tables(B, S)
        if train:
            self.s_arena.zero_()
        else:
            eng.finalize_table(tb['fin'], 0, tb['n_fin'], False)
        raw = {}
        img = self.nodes[0]
        raw[img.name] = torch.empty(B, S // 2, S // 2, self.IMG_C, **f32)
        if x.dtype == torch.uint8:     # raw RGB frames: to_tensor + normalisation fused into the gather
            mean, std = eng.input_norm
            check(L.mpose_im2col_k3s2(ctypes.c_void_p(x.data_ptr()), 1, (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std),
                                      ptr(raw[img.name]), B, S, S, st()), 'mpose_im2col_k3s2')
        else:
            check(L.mpose_im2col_k3s2(ctypes.c_void_p(x.data_ptr()), 0, None, None, ptr(raw[img.name]), B, S, S, st()), 'mpose_im2col_k3s2')
        done = set()
        for op in self.ops:
            n = op.dst
            if n.name not in raw:
                H = S // n.div
                raw[n.name] = torch.empty(B, H, H, n.C, **f32)
            src = op.src
            sc = None if src.is_image else self.fptr(src, 0)
            sh = None if src.is_image else self.fptr(src, 1)
            if isinstance(op, _ConvOp):
                o = ConvOperands()
                o.in_, o.w0 = raw[src.name].data_ptr(), eng._wptr(op.conv)
                o.in_scale, o.in_shift = sc, sh
                o.out0 = raw[n.name].data_ptr() + 4 * op.c0
                if train:
                    o.stats0 = self.sptr(n, False, op.c0)
                eng.conv(self.geom(op, B, S, 'f'), [o])
            else:
                Hs = S // src.div
                check(L.mpose_pool3_fwd(ptr(raw[src.name]), c_void_p(sc), c_void_p(sh), c_void_p(raw[n.name].data_ptr() + 4 * op.c0),
                                        B, Hs, Hs, src.C, n.C, op.kind, st()), 'mpose_pool3_fwd')
            done.add(id(op))
            if train and all(id(p) in done for p in n.producers):
                f0, nf = tb['fin_range'][n.name]
                if nf:
                    eng.finalize_table(tb['fin'], f0, nf, True)
        out = torch.empty(B, S // 8, S // 8, 128, **f32)
        n7 = self.out_node
        check(L.mpose_bn_relu_fwd(ptr(raw[n7.name]), c_void_p(self.fptr(n7, 0)), c_void_p(self.fptr(n7, 1)), ptr(out),
                                  c_int64(out.numel()), 128, st()), 'mpose_bn_relu_fwd')
        ctx = {'raw': raw, 'B': B, 'S': S} if save else None
        return out, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, D, need_dx):
        """D: gradient w.r.t. the activated stem output (B, F, F, 128).  Returns dx (NCHW) or None."""
        eng, L = self.engine, lib()
        B, S, raw = ctx['B'], ctx['S'], ctx['raw']
        dev = D.device
        st = stream_ptr
        f32 = dict(dtype=torch.float32, device=dev)
        tb = self.tables(B, S)
        self.s_arena.zero_()
        dact = {self.out_node.name: D}
        for n in reversed(self.nodes):
            if n.is_image or n.name not in dact:
                continue
            H = S // n.div
            g = dact[n.name]
            # masked BatchNorm backward over the whole node (identity channel ranges: plain ReLU mask)
            ro = BnBwdReduceOperands()
            ro.g, ro.a, ro.sums = g.data_ptr(), raw[n.name].data_ptr(), self.sptr(n, True)
            ro.a_scale, ro.a_shift = self.fptr(n, 0), self.fptr(n, 1)
            check(L.mpose_bn_bwd_reduce((BnBwdReduceOperands * 3)(ro), 1, H * H, B, n.C, 0, 0, st()), 'mpose_bn_bwd_reduce')
            c0_, nc = tb['coef_range'][n.name]
            if nc:
                check(L.mpose_bn_bwd_coef(c_void_p(tb['coef'].data_ptr() + c0_ * eng.COEF_ITEMSIZE), nc, st()), 'mpose_bn_bwd_coef')
            d_raw = torch.empty(B, H, H, n.C, **f32)
            ao = BnBwdApplyOperands()
            ao.g, ao.a, ao.coef_a, ao.da = g.data_ptr(), raw[n.name].data_ptr(), self.fptr(n, 4), d_raw.data_ptr()
            ao.a_scale, ao.a_shift = self.fptr(n, 0), self.fptr(n, 1)
            check(L.mpose_bn_bwd_apply((BnBwdApplyOperands * 3)(ao), 1, H * H, B, n.C, 0, 0, st()), 'mpose_bn_bwd_apply')
            for op in n.producers:
                src = op.src
                Hs = S // src.div
                want_dsrc = (not src.is_image) or need_dx
                if want_dsrc and src.name not in dact:
                    dact[src.name] = torch.zeros(B, Hs, Hs, src.C, **f32)
                sc = None if src.is_image else self.fptr(src, 0)
                sh = None if src.is_image else self.fptr(src, 1)
                if isinstance(op, _ConvOp):
                    wo = WgradOperands()
                    wo.in_, wo.in_scale, wo.in_shift = raw[src.name].data_ptr(), sc, sh
                    wo.gout0 = d_raw.data_ptr() + 4 * op.c0
                    wo.dw0 = eng.part_ptr(B, S, op.conv)
                    eng.wgrad(self.geom(op, B, S, 'f'), [wo], eng.stem_n_split(B, S, op))
                    if want_dsrc:
                        o = ConvOperands()
                        o.in_, o.w0 = d_raw.data_ptr() + 4 * op.c0, eng._wptr(op.conv, True)
                        o.out0 = dact[src.name].data_ptr()
                        eng.conv(self.geom(op, B, S, 'd'), [o], 1)       # accumulate
                elif want_dsrc:
                    check(L.mpose_pool3_bwd(ptr(raw[src.name]), c_void_p(sc), c_void_p(sh), c_void_p(d_raw.data_ptr() + 4 * op.c0),
                                            ptr(dact[src.name]), B, Hs, Hs, src.C, n.C, op.kind, st()), 'mpose_pool3_bwd')
        if need_dx:
            dx = torch.empty(B, 3, S, S, **f32)
            check(L.mpose_col2im_k3s2(ptr(dact['img']), ptr(dx), B, S, S, st()), 'mpose_col2im_k3s2')
            return dx
        return None
