#!/usr/bin/env python
"""A/B micro-benchmark: round-1 conv_igemm_k (fp32 NHWC in, operands split in the K loop) vs the round-2 plane engine
(conv_p.hip: pre-split bf16 planes by DMA, 2 workgroups/CU) on the columns' dominant shapes (B=32, 3 column groups),
plus the cost of the elementwise pass that produces the planes.  One process, interleaved rounds (min and median)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from margipose_amd import _lib
from margipose_amd.engine import _geom, TAPS3, _geom_flops
from margipose_amd._lib import ConvOperands, SplitOperands, stream_ptr

L = _lib.lib()
B = int(os.environ.get('B', '32'))
ROUNDS = int(os.environ.get('ROUNDS', '5'))


def timeit(fn, n=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def shape(H, cin, cout, taps, bf16=False, relu_sparse=True):
    g = _geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, taps)], (cout + 63) // 64 * 64)
    flops = _geom_flops(g) * 3
    xs = [torch.randn(B, H, H, cin, device='cuda') for _ in range(3)]
    if relu_sparse:
        xs = [x.relu_() for x in xs]
    npix = B * H * H
    T = len(taps)
    npad = (cout + 63) // 64 * 64
    ws = [(torch.randn(T * cin * npad * 3, device='cuda') * 0.05).to(torch.bfloat16).view(torch.float32) for _ in range(3)]
    outs = [torch.empty(B, H, H, cout, device='cuda') for _ in range(3)]
    planes = [torch.empty(int(L.mpose_planes_bytes(npix, cin)), dtype=torch.uint8, device='cuda') for _ in range(3)]
    sops = []
    for c in range(3):
        so = SplitOperands(); so.src, so.planes = xs[c].data_ptr(), planes[c].data_ptr(); sops.append(so)
    sarr = (SplitOperands * 3)(*sops)
    split = lambda: _lib.check(L.mpose_split_planes(sarr, 3, ctypes.c_int64(npix), cin, 0, stream_ptr()), 'split')
    split()

    def conv_call(planes_in, flags):
        ops = []
        for c in range(3):
            op = ConvOperands(); op.in_ = (planes[c] if planes_in else xs[c]).data_ptr(); op.w0, op.out0 = ws[c].data_ptr(), outs[c].data_ptr()
            ops.append(op)
        arr = (ConvOperands * 3)(*ops)
        return lambda: _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, 3, flags, stream_ptr()), 'conv')
    variants = [('round1 igemm', conv_call(False, 0)), ('planes', conv_call(True, 4)), ('split pass', split)]
    if bf16:
        variants.append(('planes bf16', conv_call(True, 12)))
    res = {k: [] for k, _ in variants}
    for _ in range(ROUNDS):
        for k, fn in variants:
            res[k].append(timeit(fn))
    print('%dx%d %d->%d taps=%d (%.1f GFLOP/launch):' % (H, H, cin, cout, T, flops / 1e9))
    for k, _ in variants:
        v = np.array(res[k])
        tf = flops / v.min() / 1e6
        print('   %-14s min %7.1f us  median %7.1f us   %s' % (k, v.min(), np.median(v), ('%6.1f TFLOP/s fp32-equiv (%.2f of 416.7)' % (tf, tf / 416.7)) if 'split' not in k and 'bf16' not in k
                                                                  else ('%6.1f TFLOP/s bf16 (%.2f of 2500)' % (tf, tf / 2500) if 'bf16' in k else '')))


t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in TAPS3]
shape(32, 128, 128, t9, bf16=True)
shape(16, 192, 192, t9, bf16=True)
shape(32, 128, 128, [(0, 0, 0, 0)])
shape(32, 128, 32, t9)
shape(32, 32, 32, t9)
