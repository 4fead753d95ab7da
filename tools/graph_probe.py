#!/usr/bin/env python
"""Does capturing the whole training step in a HIP graph (no host work at replay) make it faster?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
B = 32
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, fused=True)
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')
def step():
    opt.zero_grad(set_to_none=True)
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    loss.backward()
    opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print('eager   %.2f ms/step' % ((time.perf_counter() - t0) / 10 * 1e3))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): g.replay()
torch.cuda.synchronize()
print('graphed %.2f ms/step  loss %.6f' % ((time.perf_counter() - t0) / 10 * 1e3, float(loss)))
