#!/usr/bin/env python
"""Micro-benchmark + quick check of the weight-gradient kernels on the model's dominant shapes (B=32, 3 column groups, the
three-product fp16 form): wgrad.hip's row-of-taps kernel where it takes the geometry, conv.hip's conv_wgrad_k otherwise.

    python tools/bench_wgrad.py [--check]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from margipose_amd import _lib, engine as eng
from margipose_amd._lib import AbsmaxOperands, WgradOperands, stream_ptr

L = _lib.lib()
B = int(os.environ.get('B', '32'))
SLOT = 16 * 64


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
    npix = tensors[0].numel() // C
    _lib.check(L.mpose_absmax((AbsmaxOperands * len(ops))(*ops), len(ops), ctypes.c_int64(npix), C, int(relu), stream_ptr()), 'absmax')
    return slots


def run(H, cin, cout, pro, shortcut, check, groups=3, batch=None):
    b = batch or B
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    npad = (cout + 63) // 64 * 64
    kpad = (cin + 31) // 32 * 32
    taps = t9 + ([(0, 0, 0, 1)] if shortcut else [])
    g = eng._geom(b, H, cin, H, cout, cout if shortcut else 0, H, 1, 1, [(0, 0, taps)], npad, npad if shortcut else 0)
    tiles = L.mpose_conv_wgrad_tiles(ctypes.byref(g))
    nsp = int(os.environ.get('NSPLIT', '0')) or eng.Engine._n_split(b * H * H, tiles, groups)
    rng = np.random.default_rng(H * 1000 + cin)
    xs = [torch.from_numpy(rng.standard_normal((b, H, H, cin))).float().cuda() for _ in range(groups)]
    gs = [torch.from_numpy(rng.standard_normal((b, H, H, cout)) * 1e-3).float().cuda() for _ in range(groups)]
    g1 = [torch.from_numpy(rng.standard_normal((b, H, H, cout)) * 3.0).float().cuda() for _ in range(groups)]
    sc = torch.rand(cin, device='cuda') + 0.5; sh = torch.randn(cin, device='cuda') * 0.3
    parts = [torch.full((nsp * 9 * kpad * npad,), float('nan'), device='cuda') for _ in range(groups)]
    parts1 = [torch.full((nsp * kpad * npad,), float('nan'), device='cuda') for _ in range(groups)]
    ax = amax(xs, cin, sc if pro else None, sh if pro else None, relu=pro)
    ag, ag1 = amax(gs, cout), amax(g1, cout)
    wops = []
    for c in range(groups):
        wo = WgradOperands()
        wo.in_, wo.gout0, wo.dw0 = xs[c].data_ptr(), gs[c].data_ptr(), parts[c].data_ptr()
        wo.in_amax, wo.gout0_amax = ax[c].data_ptr(), ag[c].data_ptr()
        if shortcut:
            wo.gout1, wo.dw1, wo.gout1_amax = g1[c].data_ptr(), parts1[c].data_ptr(), ag1[c].data_ptr()
        if pro:
            wo.in_scale, wo.in_shift = sc.data_ptr(), sh.data_ptr()
        wops.append(wo)
    warr = (WgradOperands * groups)(*wops)
    call = lambda: _lib.check(L.mpose_conv_wgrad(ctypes.byref(g), warr, groups, nsp, stream_ptr()), 'wgrad')
    flops = eng._geom_flops(g) * groups
    us = timeit(call)
    print('wgrad %dx%d %d->%d pro=%d sc=%d nsplit=%d tiles=%d : %7.1f us  %6.1f TFLOP/s fp32-equivalent' %
          (H, H, cin, cout, pro, shortcut, nsp, tiles, us, flops / us / 1e6), flush=True)
    if check:
        torch.cuda.synchronize()
        c = groups - 1
        x = xs[c].double()
        if pro:
            x = torch.relu(x * sc.double() + sh.double())
        xn = x.permute(0, 3, 1, 2)
        w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
        F.conv2d(xn, w, padding=1).backward(gs[c].double().permute(0, 3, 1, 2))
        got = parts[c].view(nsp, 9, kpad // 4, npad, 4).double().sum(0)            # [tap][k/4][n][4]
        got = got.permute(2, 1, 3, 0).reshape(npad, kpad, 9)[:cout, :cin].reshape(cout, cin, 3, 3)
        err = float((got - w.grad).abs().max() / w.grad.abs().max())
        msg = 'rel err vs fp64: conv %.2e' % err
        if shortcut:
            w1 = torch.zeros(cout, cin, 1, 1, dtype=torch.float64, device='cuda', requires_grad=True)
            F.conv2d(xn, w1).backward(g1[c].double().permute(0, 3, 1, 2))
            got1 = parts1[c].view(nsp, 1, kpad // 4, npad, 4).double().sum(0).permute(2, 1, 3, 0).reshape(npad, kpad)[:cout, :cin]
            err1 = float((got1 - w1.grad.view(cout, cin)).abs().max() / w1.grad.abs().max())
            msg += ', shortcut %.2e' % err1
            err = max(err, err1)
        print('   ' + msg, flush=True)
        if err >= 5e-6:
            print('   FAILED', flush=True)


if __name__ == '__main__':
    check = '--check' in sys.argv
    if check:
        for (H, cin, cout, pro, scut, b) in ((8, 32, 32, 1, 0, 3), (16, 64, 96, 0, 1, 1), (24, 128, 64, 0, 0, 2), (16, 96, 128, 1, 1, 2),
                                             (16, 192, 192, 1, 0, 4), (32, 128, 128, 0, 1, 2), (32, 128, 32, 0, 1, 2)):
            run(H, cin, cout, pro, scut, True, groups=2, batch=b)
    for H, C in ((32, 128), (16, 192)):
        for pro, scut in ((0, 0), (1, 0), (0, 1)):
            run(H, C, C, pro, scut, check)
