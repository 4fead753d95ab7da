#!/usr/bin/env python
"""Host-side enqueue time of a training step vs its GPU time (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
B = int(os.environ.get('B', '32'))
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, fused=True)
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')
def step(parts=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = m(x)
    t1 = time.perf_counter()
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if parts is not None:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): parts[i] += d
for _ in range(3): step()
torch.cuda.synchronize()
N = 10
parts = [0.0] * 4
t0 = time.perf_counter()
for _ in range(N): step(parts)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host enqueue %.2f ms/step (fwd %.2f, loss %.2f, bwd %.2f, opt %.2f); wall %.2f ms/step' % (
    t_enq / N * 1e3, parts[0] / N * 1e3, parts[1] / N * 1e3, parts[2] / N * 1e3, parts[3] / N * 1e3, t_all / N * 1e3))
