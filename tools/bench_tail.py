#!/usr/bin/env python
"""Micro-benchmark of the soft-argmax tail kernels: achieved GB/s vs algorithmic bytes."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import _lib
from margipose_amd._lib import ptr, ptr_array, stream_ptr, c_float


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    L = _lib.lib()
    res = []
    fwd_only = '--fwd-only' in sys.argv          # (the forward kernel at the cache-defeating sizes only)
    cases = ((2048, 32, False), (2048, 32, True), (910, 48, False), (512, 64, False)) if fwd_only else \
        ((64, 32, False), (64, 32, True), (32, 32, False), (2048, 32, False), (2048, 32, True), (512, 64, False))
    for B, F, bf16 in cases:
        rows = B * 17; n = F * F; E = rows * 3 * n
        dt = torch.bfloat16 if bf16 else torch.float32
        lg = [(torch.randn(B, 17, F, F, device='cuda') * 4).to(dt) for _ in range(3)]
        hm = [torch.empty_like(l) for l in lg]
        xyz = torch.empty(B, 17, 3, device='cuda')
        t = timeit(lambda: L.mpose_softmax_dsnt_fwd(ptr_array(lg), ptr_array(hm), None, ptr(xyz), 3, rows, F, F, int(bf16), stream_ptr()),
                   iters=200 if fwd_only else 50)
        bpe = 2 if bf16 else 4
        res.append(dict(kernel='softmax_dsnt_fwd', B=B, F=F, dtype=str(dt), us=t * 1e6, GBps=E * 2 * bpe / t / 1e9,
                        frac_of_8TBps=E * 2 * bpe / t / 8e12))
        if bf16 or fwd_only:
            continue
        tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1
        losses = torch.empty(B, 17, device='cuda'); dl = torch.ones(B, 17, device='cuda') / (B * 17)
        g = [torch.empty_like(h) for h in hm]; d = [torch.empty_like(h) for h in hm]
        t = timeit(lambda: L.mpose_stage_loss_fwd(ptr_array(hm), ptr(tgt), ptr(losses), ptr(xyz), rows, F, F, c_float(1.0), 1, 1, 0, stream_ptr()))
        res.append(dict(kernel='stage_loss_fwd', B=B, F=F, us=t * 1e6, GBps=E * 4 / t / 1e9))
        t = timeit(lambda: L.mpose_stage_loss_bwd(ptr_array(hm), ptr(tgt), ptr(xyz), ptr(dl), ptr_array(g), rows, F, F, c_float(1.0), 1, 1, 0, stream_ptr()))
        res.append(dict(kernel='stage_loss_bwd', B=B, F=F, us=t * 1e6, GBps=E * 8 / t / 1e9))
        t = timeit(lambda: L.mpose_softmax_bwd(ptr_array(hm), ptr_array(g), None, ptr_array(d), 3, rows, n, stream_ptr()))
        res.append(dict(kernel='softmax_bwd', B=B, F=F, us=t * 1e6, GBps=E * 12 / t / 1e9))
    for r in res:
        print(json.dumps(r))


if __name__ == '__main__':
    main()
