#!/usr/bin/env python
"""conv_h2_k (csrc/conv_h.hip: producer-split fp16 planes, DMA-fed LDS tiles, two workgroups per CU) against conv_igemm_k
(csrc/conv.hip: fp32 operands split inside the K loop) on the four launch flavours a column block runs in training, same data,
same arithmetic (three fp16 products):
   f_conv2  3x3 whose input is relu(bn1(c1)) (conv_igemm_k: prologue; conv_h2_k: planes written by mpose_split_h2)
   f_in     3x3 + fused 1x1 shortcut        d_conv2  data-gradient with the ReLU mask + BatchNorm-backward sums
   d_in     two-input data-gradient with the consumer's BatchNorm-backward sums
Prints microseconds per launch for both engines, the largest difference between their outputs relative to the output's
largest magnitude, and (CHECK=1, small batch) both engines' errors against an fp64 convolution."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from margipose_amd import _lib, engine as eng
from margipose_amd._lib import AbsmaxOperands, ConvOperands, SplitH2Operands, stream_ptr

L = _lib.lib()
B = int(os.environ.get('B', '32'))
CHECK = int(os.environ.get('CHECK', '0'))
NOSTATS = int(os.environ.get('NOSTATS', '0'))      # 1: no BatchNorm statistics, 2: also no channel extremes / masks / consumer sums
SLOT = 16 * 64
F16X3, H2 = 32, 128


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def amax(tensors, C, scale=None, shift=None, relu=False):
    slots = torch.zeros(len(tensors), SLOT, dtype=torch.float32, device='cuda')
    ops = []
    for i, t in enumerate(tensors):
        ao = AbsmaxOperands()
        ao.src, ao.dst = t.data_ptr(), slots[i].data_ptr()
        if scale is not None:
            ao.scale, ao.shift = scale.data_ptr(), shift.data_ptr()
        ops.append(ao)
    _lib.check(L.mpose_absmax((AbsmaxOperands * len(ops))(*ops), len(ops), ctypes.c_int64(tensors[0].numel() // C), C, int(relu), stream_ptr()), 'absmax')
    return slots


def split_h2(tensors, slots, C, scale=None, shift=None, relu=False):
    npix = tensors[0].numel() // C
    planes = [torch.empty(npix * C, dtype=torch.float32, device='cuda') for _ in tensors]
    ops = []
    for t, pl, sl in zip(tensors, planes, slots):
        so = SplitH2Operands()
        so.src, so.planes, so.amax = t.data_ptr(), pl.data_ptr(), sl.data_ptr()
        if scale is not None:
            so.scale, so.shift = scale.data_ptr(), shift.data_ptr()
        ops.append(so)
    _lib.check(L.mpose_split_h2((SplitH2Operands * len(ops))(*ops), len(ops), ctypes.c_int64(npix), C, int(relu), stream_ptr()), 'split_h2')
    return planes


def pack(w, cout, cin, T, layout):
    npad = (cout + 63) // 64 * 64
    packed = torch.zeros(T * cin * npad * 3 // 2, dtype=torch.float32, device='cuda')
    am = torch.zeros(1, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'], j['amax'] = w.data_ptr(), packed.data_ptr(), am.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = cout, cin, T, npad, cin, layout
    j['sn'], j['sk'], j['st'] = cin * T, T, 1
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_weights_absmax(_lib.ptr(dev), 1, stream_ptr()), 'weights_absmax')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * cin * npad, stream_ptr()), 'pack')
    return packed, am, npad


def run(H, C):
    rng = np.random.default_rng(H + C)
    rnd = lambda *s: torch.from_numpy(rng.standard_normal(s)).float().cuda()
    G = 3
    xs = [rnd(B, H, H, C) for _ in range(G)]
    x1 = [rnd(B, H, H, C) * 0.1 for _ in range(G)]
    aux_a = [rnd(B, H, H, C) for _ in range(G)]
    aux_b = [rnd(B, H, H, C) for _ in range(G)]
    w3 = torch.randn(C, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    w1 = torch.randn(C, C, 1, 1, device='cuda') * (2.0 / C) ** 0.5
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda') * 0.3
    ax, ax1 = amax(xs, C), amax(x1, C)
    axp = amax(xs, C, sc, sh, relu=True)
    xs_h, x1_h, xsp_h = split_h2(xs, ax, C), split_h2(x1, ax1, C), split_h2(xs, axp, C, sc, sh, relu=True)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    t9d = [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    res = {}
    for engine in ('igemm', 'h2'):
        h2 = engine == 'h2'
        p3, a3, npad = pack(w3, C, C, 9, 3 if h2 else 2)
        p1, a1, _ = pack(w1, C, C, 1, 3 if h2 else 2)
        fl = F16X3 | (H2 if h2 else 0)
        out0 = [torch.zeros(B, H, H, C, device='cuda') for _ in range(G)]
        out1 = [torch.zeros(B, H, H, C, device='cuda') for _ in range(G)]
        stats = [torch.zeros(C, 8 + 2 * 64, dtype=torch.float64, device='cuda') for _ in range(G)]
        flavours = {}
        g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], npad)
        ops = []
        for c in range(G):
            o = ConvOperands(); o.w0, o.out0 = p3.data_ptr(), out0[c].data_ptr()
            if h2:
                o.in_ = xsp_h[c].data_ptr()
            else:
                o.in_, o.in_scale, o.in_shift = xs[c].data_ptr(), sc.data_ptr(), sh.data_ptr()
            o.in_amax, o.w0_amax = axp[c].data_ptr(), a3.data_ptr()
            if not NOSTATS:
                o.stats0 = stats[c].data_ptr(); o.mm0 = stats[c].data_ptr() + 8 * 6 * C
            elif NOSTATS == 3:
                o.stats0 = stats[c].data_ptr()
            ops.append(o)
        flavours['f_conv2'] = (g, ops, fl)
        g = eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
        ops = []
        for c in range(G):
            o = ConvOperands(); o.in_ = (xs_h if h2 else xs)[c].data_ptr()
            o.w0, o.w1, o.out0, o.out1 = p3.data_ptr(), p1.data_ptr(), out0[c].data_ptr(), out1[c].data_ptr()
            o.in_amax, o.w0_amax, o.w1_amax = ax[c].data_ptr(), a3.data_ptr(), a1.data_ptr()
            o.stats0, o.stats1 = stats[c].data_ptr(), stats[c].data_ptr() + 8 * 2 * C; o.mm0 = stats[c].data_ptr() + 8 * 6 * C
            ops.append(o)
        flavours['f_in'] = (g, ops, fl)
        g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9d)], npad)
        ops = []
        for c in range(G):
            o = ConvOperands(); o.in_, o.w0, o.out0 = (xs_h if h2 else xs)[c].data_ptr(), p3.data_ptr(), out0[c].data_ptr()
            o.in_amax, o.w0_amax = ax[c].data_ptr(), a3.data_ptr()
            if NOSTATS < 2:
                o.mask_src, o.mask_scale, o.mask_shift = aux_a[c].data_ptr(), sc.data_ptr(), sh.data_ptr()
            if not NOSTATS:
                o.stats0 = stats[c].data_ptr()
            ops.append(o)
        flavours['d_conv2'] = (g, ops, fl)
        g = eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, t9d + [(0, 0, 0, 1)])], npad, npad)
        ops = []
        for c in range(G):
            o = ConvOperands(); o.in_, o.in1 = (xs_h if h2 else xs)[c].data_ptr(), (x1_h if h2 else x1)[c].data_ptr()
            o.w0, o.w1, o.out0 = p3.data_ptr(), p1.data_ptr(), out0[c].data_ptr()
            o.in_amax, o.in1_amax, o.w0_amax, o.w1_amax = ax[c].data_ptr(), ax1[c].data_ptr(), a3.data_ptr(), a1.data_ptr()
            o.red_a, o.red_b, o.red_scale, o.red_shift, o.red_sums = aux_a[c].data_ptr(), aux_b[c].data_ptr(), sc.data_ptr(), sh.data_ptr(), stats[c].data_ptr()
            ops.append(o)
        flavours['d_in'] = (g, ops, fl | 2)
        for name, (g, ops, flags) in flavours.items():
            arr = (ConvOperands * G)(*ops)
            call = lambda: _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, G, flags, stream_ptr()), name)
            for s_ in stats:
                s_.zero_()
            call(); torch.cuda.synchronize()
            snap = (out0[0].clone(), out1[0].clone(), stats[0].clone())
            us = timeit(call)
            res.setdefault(name, {})[engine] = (us, snap, eng._geom_flops(g) * G)
    for name, r in res.items():
        (ui, si, fl), (uh, sh_, _) = r['igemm'], r['h2']
        d0 = float((si[0] - sh_[0]).abs().max() / si[0].abs().max())
        d1 = float((si[1] - sh_[1]).abs().max() / si[1].abs().max()) if name == 'f_in' else 0.0
        ds = float((si[2] - sh_[2]).abs().max() / si[2].abs().max())
        print('%-8s %dx%d %d ch : igemm %7.1f us (%5.1f TF)   h2 %7.1f us (%5.1f TF)   x%.2f   max diff out0 %.1e out1 %.1e sums %.1e'
              % (name, H, H, C, ui, fl / ui / 1e6, uh, fl / uh / 1e6, ui / uh, d0, d1, ds), flush=True)
    if CHECK:      # fp64 reference of the plain 3x3 (f_in's out0) and the 1x1
        x64 = xs[0].double().permute(0, 3, 1, 2)
        ref0 = torch.nn.functional.conv2d(x64, w3.double(), padding=1).permute(0, 2, 3, 1)
        ref1 = torch.nn.functional.conv2d(x64, w1.double()).permute(0, 2, 3, 1)
        t32 = torch.nn.functional.conv2d(xs[0].permute(0, 3, 1, 2), w3, padding=1).permute(0, 2, 3, 1)
        for engine in ('igemm', 'h2'):
            o0, o1, _ = res['f_in'][engine][1]
            print('  %-6s f_in vs fp64: 3x3 max %.2e (rel. to max |out|), 1x1 %.2e   [torch fp32 conv: %.2e]' % (
                engine, float((o0.double() - ref0).abs().max() / ref0.abs().max()), float((o1.double() - ref1).abs().max() / ref1.abs().max()),
                float((t32.double() - ref0).abs().max() / ref0.abs().max())), flush=True)


if __name__ == '__main__':
    shapes = ((32, 128), (16, 192)) if not os.environ.get('SHAPES') else tuple(tuple(int(v) for v in s.split('x')) for s in os.environ['SHAPES'].split(','))
    for H, C in shapes:
        run(H, C)
