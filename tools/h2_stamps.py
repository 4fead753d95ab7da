#!/usr/bin/env python
"""Cycle stamps of conv_h2r_k's K loop (a CH_EXP=256 build: _ab/ch_exp256.so): one f_conv2 launch at B=4, then the deltas between
the stamps of wave 0 of workgroup 0 for the even steps.  MPOSE_LIB=_ab/ch_exp256.so python tools/with_lib.py tools/h2_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ.setdefault('B', '4')
import tools.bench_h2 as bh
from margipose_amd import _lib, engine as eng
from margipose_amd._lib import ConvOperands, stream_ptr
L = _lib.lib()
B, H, C, G = int(os.environ['B']), 32, 128, 3
rng = np.random.default_rng(0)
xs = [torch.from_numpy(rng.standard_normal((B, H, H, C))).float().cuda() for _ in range(G)]
w3 = torch.randn(C, C, 3, 3, device='cuda') * 0.03
ax = bh.amax(xs, C)
xh = bh.split_h2(xs, ax, C)
p3, a3, npad = bh.pack(w3, C, C, 9, 3)
out = [torch.zeros(B, H, H, C, device='cuda') for _ in range(G)]
dbg = torch.zeros(4096, dtype=torch.int64, device='cuda')
st = torch.zeros(C * 8, dtype=torch.float64, device='cuda')
t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], npad)
ops = []
for c in range(G):
    o = ConvOperands(); o.in_, o.w0, o.out0 = xh[c].data_ptr(), p3.data_ptr(), out[c].data_ptr()
    o.in_amax, o.w0_amax, o.stats0, o.out1 = ax[c].data_ptr(), a3.data_ptr(), st.data_ptr(), dbg.data_ptr()
    ops.append(o)
arr = (ConvOperands * G)(*ops)
for _ in range(3):
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, G, 32 | 128, stream_ptr()), 'conv')
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 8)[:12]
names = ['tap0 (mfma || read tap1)', 'tap1 (mfma || read tap2)', 'wait vmcnt/lgkm', 'barrier', 'tap2 (mfma || batch || read next)']
print('cycles per phase of the even steps (wave 0, workgroup 0):')
for r in d:
    print('  ' + '  '.join('%s %5d' % (n.split(' ')[0], r[i + 1] - r[i]) for i, n in enumerate(names)))
print('step-to-step (two steps): ', [int(d[i + 1][0] - d[i][0]) for i in range(len(d) - 1)])
