#!/usr/bin/env python
"""Host-side enqueue time of one training step vs its GPU time (is the step CPU-bound?)."""
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


def step():
    opt.zero_grad(set_to_none=True)
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print('B=%d: host enqueue %.2f ms/step (python + HIP launches, GPU idle at start), step incl. drain %.2f ms' % (B, 1e3 * sorted(enq)[len(enq) // 2], 1e3 * sorted(tot)[len(tot) // 2]))
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print('back-to-back %.2f ms/step' % ((time.perf_counter() - t0) / 10 * 1e3))
