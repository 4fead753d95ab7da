#!/usr/bin/env python
"""Timing ablation of the conv kernel on one representative launch (128->128 3x3 @32x32, B=32, 3 groups).
Uses tools/libmargipose_ablate.so (conv.hip built with -DMPOSE_ABLATE); results of ablated runs are garbage,
only the timings matter."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import _lib
from margipose_amd.engine import _geom, TAPS3, _geom_flops
from margipose_amd._lib import ConvOperands, stream_ptr

L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmargipose_ablate.so'))
B, H, C = 32, 32, 128
t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in TAPS3]
g = _geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], 128)
flops = _geom_flops(g) * 3
xs = [torch.randn(B, H, H, C, device='cuda') for _ in range(3)]
ws = [torch.randn(9 * C * 128, device='cuda') * 0.05 for _ in range(3)]
outs = [torch.empty(B, H, H, C, device='cuda') for _ in range(3)]
ops = []
for c in range(3):
    op = ConvOperands(); op.in_, op.w0, op.out0 = xs[c].data_ptr(), ws[c].data_ptr(), outs[c].data_ptr(); ops.append(op)
arr = (ConvOperands * 3)(*ops)


def run():
    rc = L.mpose_conv_fwd(ctypes.byref(g), arr, 3, 0, stream_ptr())
    assert rc == 0, rc


for name, val in (('full', 0), ('full+setprio', 16), ('full+stagger', 32), ('full+both', 48), ('no_barrier', 4), ('no_barrier+setprio', 20), ('mfma_only', 7)):
    os.environ['MPOSE_ABLATE'] = str(val)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print('%-20s %8.1f us  %6.1f TFLOP/s-equivalent' % (name, us, flops / us / 1e6))
