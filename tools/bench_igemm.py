#!/usr/bin/env python
"""Micro-benchmark of conv_igemm_k (three-product fp16 form) on the four launch flavours a column block runs in training, at the
model's dominant shapes (B=32, 3 column groups):
   f_conv2      3x3 with the BatchNorm+ReLU prologue, forward statistics + channel extremes
   f_in         3x3 + fused 1x1 shortcut (two outputs, two sets of statistics)
   d_conv2      data-gradient of the second 3x3: ReLU mask + BatchNorm-backward sums in the epilogue
   d_in         two-input data-gradient (3x3 of d_c1 + 1x1 of d_sc) with the consumer's BatchNorm-backward sums
MPOSE_LIB=... python tools/with_lib.py tools/bench_igemm.py   runs another build of the library."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from margipose_amd import _lib, engine as eng
from margipose_amd._lib import AbsmaxOperands, ConvOperands, stream_ptr

L = _lib.lib()
B = int(os.environ.get('B', '32'))
SLOT = 16 * 64
F16X3 = 32


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


def pack(w, cout, cin, T):
    npad = (cout + 63) // 64 * 64
    packed = torch.zeros(T * cin * npad * 3 // 2, dtype=torch.float32, device='cuda')
    am = torch.zeros(1, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'], j['amax'] = w.data_ptr(), packed.data_ptr(), am.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = cout, cin, T, npad, cin, 2
    j['sn'], j['sk'], j['st'] = cin * T, T, 1
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_weights_absmax(_lib.ptr(dev), 1, stream_ptr()), 'weights_absmax')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * cin * npad, stream_ptr()), 'pack')
    return packed, am, npad


def run(H, C):
    rng = np.random.default_rng(H + C)
    rnd = lambda *s: torch.from_numpy(rng.standard_normal(s)).float().cuda()
    xs = [rnd(B, H, H, C) for _ in range(3)]
    x1 = [rnd(B, H, H, C) * 0.1 for _ in range(3)]
    aux_a = [rnd(B, H, H, C) for _ in range(3)]
    aux_b = [rnd(B, H, H, C) for _ in range(3)]
    out0 = [torch.empty(B, H, H, C, device='cuda') for _ in range(3)]
    out1 = [torch.empty(B, H, H, C, device='cuda') for _ in range(3)]
    w3 = torch.randn(C, C, 3, 3, device='cuda') * (2.0 / (9 * C)) ** 0.5
    w1 = torch.randn(C, C, 1, 1, device='cuda') * (2.0 / C) ** 0.5
    p3, a3, npad = pack(w3, C, C, 9)
    p1, a1, _ = pack(w1, C, C, 1)
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda') * 0.3
    ax, ax1 = amax(xs, C), amax(x1, C)
    axp = amax(xs, C, sc, sh, relu=True)
    stats = [torch.zeros(C, 8, dtype=torch.float64, device='cuda') for _ in range(3)]
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    t9d = [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    flavours = {}
    g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], npad)
    ops = []
    for c in range(3):
        o = ConvOperands(); o.in_, o.w0, o.out0 = xs[c].data_ptr(), p3.data_ptr(), out0[c].data_ptr()
        o.in_scale, o.in_shift, o.in_amax, o.w0_amax = sc.data_ptr(), sh.data_ptr(), axp[c].data_ptr(), a3.data_ptr()
        o.stats0 = stats[c].data_ptr(); o.mm0 = stats[c].data_ptr() + 8 * 6 * C
        ops.append(o)
    flavours['f_conv2'] = (g, ops, F16X3)
    g = eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    ops = []
    for c in range(3):
        o = ConvOperands(); o.in_, o.w0, o.w1, o.out0, o.out1 = xs[c].data_ptr(), p3.data_ptr(), p1.data_ptr(), out0[c].data_ptr(), out1[c].data_ptr()
        o.in_amax, o.w0_amax, o.w1_amax = ax[c].data_ptr(), a3.data_ptr(), a1.data_ptr()
        o.stats0, o.stats1 = stats[c].data_ptr(), stats[c].data_ptr() + 8 * 2 * C; o.mm0 = stats[c].data_ptr() + 8 * 6 * C
        ops.append(o)
    flavours['f_in'] = (g, ops, F16X3)
    g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9d)], npad)
    ops = []
    for c in range(3):
        o = ConvOperands(); o.in_, o.w0, o.out0 = xs[c].data_ptr(), p3.data_ptr(), out0[c].data_ptr()
        o.in_amax, o.w0_amax = ax[c].data_ptr(), a3.data_ptr()
        o.mask_src, o.mask_scale, o.mask_shift, o.stats0 = aux_a[c].data_ptr(), sc.data_ptr(), sh.data_ptr(), stats[c].data_ptr()
        ops.append(o)
    flavours['d_conv2'] = (g, ops, F16X3)
    g = eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, t9d + [(0, 0, 0, 1)])], npad, npad)
    ops = []
    for c in range(3):
        o = ConvOperands(); o.in_, o.in1, o.w0, o.w1, o.out0 = xs[c].data_ptr(), x1[c].data_ptr(), p3.data_ptr(), p1.data_ptr(), out0[c].data_ptr()
        o.in_amax, o.in1_amax, o.w0_amax, o.w1_amax = ax[c].data_ptr(), ax1[c].data_ptr(), a3.data_ptr(), a1.data_ptr()
        o.red_a, o.red_b, o.red_scale, o.red_shift, o.red_sums = aux_a[c].data_ptr(), aux_b[c].data_ptr(), sc.data_ptr(), sh.data_ptr(), stats[c].data_ptr()
        ops.append(o)
    flavours['d_in'] = (g, ops, F16X3 | 2)
    for name, (g, ops, flags) in flavours.items():
        arr = (ConvOperands * 3)(*ops)
        us = timeit(lambda: _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, 3, flags, stream_ptr()), name))
        fl = eng._geom_flops(g) * 3
        print('%-8s %dx%d %d ch : %7.1f us  %6.1f TFLOP/s fp32-equivalent' % (name, H, H, C, us, fl / us / 1e6), flush=True)


if __name__ == '__main__':
    for H, C in ((32, 128), (16, 192)):
        run(H, C)
