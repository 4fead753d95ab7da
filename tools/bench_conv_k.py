#!/usr/bin/env python
"""Fixed-overhead vs per-K cost of the igemm kernel: 3x3 conv, 32x32x(B=32) x3 groups, Cout=128, Cin swept."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import _lib
from margipose_amd.engine import _geom, TAPS3, _geom_flops
from margipose_amd._lib import ConvOperands, stream_ptr
L = _lib.lib()
B, H, CO = 32, int(os.environ.get('H', '32')), int(os.environ.get('CO', '128'))


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


for CI in (32, 64, 128, 256, 512):
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in TAPS3]
    g = _geom(B, H, CI, H, CO, 0, H, 1, 1, [(0, 0, t9)], CO)
    flops = _geom_flops(g) * 3
    xs = [torch.randn(B, H, H, CI, device='cuda') for _ in range(3)]
    ws = [(torch.randn(9 * CI * CO * 3, device='cuda') * 0.05).to(torch.bfloat16).view(torch.float32) for _ in range(3)]
    outs = [torch.empty(B, H, H, CO, device='cuda') for _ in range(3)]
    ops = []
    for c in range(3):
        op = ConvOperands(); op.in_, op.w0, op.out0 = xs[c].data_ptr(), ws[c].data_ptr(), outs[c].data_ptr()
        ops.append(op)
    arr = (ConvOperands * 3)(*ops)
    us = timeit(lambda: _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, 3, 0, stream_ptr()), 'conv'))
    print('Cin=%4d n_iter=%3d : %7.1f us  %6.1f TFLOP/s' % (CI, 9 * CI // 32, us, flops / us / 1e6))
