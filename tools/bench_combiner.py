"""Times mpose_combiner_bwd at the configuration's size (B=32, 17 joints, 32x32, 128 channels)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from margipose_amd._lib import check, lib, ptr, ptr_array, stream_ptr
L = lib()
B, J, HW, C = 32, 17, 1024, 128
hm = [torch.rand(B, J, HW, device='cuda') for _ in range(3)]
g = torch.randn(B, HW, C, device='cuda')
w = torch.randn(C, 3 * J, device='cuda')
d = [torch.empty_like(h) for h in hm]
n_part = int(os.environ.get("NPART", "256"))
dwp = torch.empty(n_part * w.numel(), device='cuda')
dw = torch.empty_like(w)
import ctypes
def run():
    check(L.mpose_reduce_partials(ptr(dwp), ptr(dw), n_part, ctypes.c_int64(w.numel()), 0, stream_ptr()), 'red')
    check(L.mpose_combiner_bwd(ptr_array(hm), ptr(w), ptr(g), ptr_array(d), ptr(dwp), n_part, B, J, HW, C, stream_ptr()), 'comb')
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print('combiner_bwd + reduce_partials %.1f us' % (1e3 * e0.elapsed_time(e1) / 50))
