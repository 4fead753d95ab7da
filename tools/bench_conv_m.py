#!/usr/bin/env python
"""Per-round vs per-launch cost of the igemm kernel: 3x3 conv 32x32, 128->128, x3 groups, batch swept."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import _lib
from margipose_amd.engine import _geom, TAPS3, _geom_flops
from margipose_amd._lib import ConvOperands, stream_ptr
L = _lib.lib()
H, C = 32, int(os.environ.get('C', '128'))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (1, 5, 10, 11, 21, 32, 64, 128):
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in TAPS3]
    g = _geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], C)
    flops = _geom_flops(g) * 3
    xs = [torch.randn(B, H, H, C, device='cuda') for _ in range(3)]
    ws = [(torch.randn(9 * C * C * 3, device='cuda') * 0.05).to(torch.bfloat16).view(torch.float32) for _ in range(3)]
    outs = [torch.empty(B, H, H, C, device='cuda') for _ in range(3)]
    ops = []
    for c in range(3):
        op = ConvOperands(); op.in_, op.w0, op.out0 = xs[c].data_ptr(), ws[c].data_ptr(), outs[c].data_ptr()
        ops.append(op)
    arr = (ConvOperands * 3)(*ops)
    us = timeit(lambda: _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, 3, 0, stream_ptr()), 'conv'))
    wgs = (B * H * H + 127) // 128 * 3
    print('B=%3d WGs=%5d rounds=%5.2f : %7.1f us  %6.1f TFLOP/s' % (B, wgs, wgs / 256.0, us, flops / us / 1e6))
