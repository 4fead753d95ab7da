#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the model's dominant shapes (B=32, 3 column groups)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import _lib
import margipose_amd.build as _b
if os.environ.get('MPOSE_LIB'):      # ablation builds (tools/experiments/ablate_conv.py)
    _b.LIB_PATH = os.path.abspath(os.environ['MPOSE_LIB'])
from margipose_amd.engine import _geom, TAPS3, _geom_flops
from margipose_amd._lib import ConvOperands, WgradOperands, stream_ptr

L = _lib.lib()
B = int(os.environ.get('B', '32'))


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


def run(H, C, pro):
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in TAPS3]
    g = _geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], C)
    flops = _geom_flops(g) * 3
    xs = [torch.randn(B, H, H, C, device='cuda') for _ in range(3)]
    # packed weights: three bf16 planes per element (values only matter for speed through DVFS)
    ws = [(torch.randn(9 * C * C * 3, device='cuda') * 0.05).to(torch.bfloat16).view(torch.float32) for _ in range(3)]
    outs = [torch.empty(B, H, H, C, device='cuda') for _ in range(3)]
    sc = torch.rand(C, device='cuda') + 0.5; sh = torch.randn(C, device='cuda') * 0.1
    ops = []
    for c in range(3):
        op = ConvOperands(); op.in_, op.w0, op.out0 = xs[c].data_ptr(), ws[c].data_ptr(), outs[c].data_ptr()
        if pro:
            op.in_scale, op.in_shift = sc.data_ptr(), sh.data_ptr()
        ops.append(op)
    arr = (ConvOperands * 3)(*ops)
    us = timeit(lambda: _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, 3, 0, stream_ptr()), 'conv'))
    print('conv  %dx%d %d->%d pro=%d : %7.1f us  %6.1f TFLOP/s' % (H, H, C, C, pro, us, flops / us / 1e6))
    from margipose_amd.engine import Engine
    nt = C // (32 * Engine._wg_blocks(C))
    nsp = int(os.environ.get('NSPLIT', '0')) or Engine._n_split(B * H * H, 9 * nt * nt)
    parts = [torch.empty(nsp * 9 * C * C, device='cuda') for _ in range(3)]
    wops = []
    for c in range(3):
        wo = WgradOperands(); wo.in_, wo.gout0, wo.dw0 = xs[c].data_ptr(), outs[c].data_ptr(), parts[c].data_ptr()
        if pro:
            wo.in_scale, wo.in_shift = sc.data_ptr(), sh.data_ptr()
        wops.append(wo)
    warr = (WgradOperands * 3)(*wops)
    us = timeit(lambda: _lib.check(L.mpose_conv_wgrad(ctypes.byref(g), warr, 3, nsp, stream_ptr()), 'wgrad'))
    print('wgrad %dx%d %d->%d pro=%d nsplit=%d : %7.1f us  %6.1f TFLOP/s' % (H, H, C, C, pro, nsp, us, flops / us / 1e6))


for H, C in ((32, 128), (16, 192)):
    for pro in (0, 1):
        run(H, C, pro)
