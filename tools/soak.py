#!/usr/bin/env python
"""Soak: N training steps (default schedule), step time and allocator high-water marks every 100 steps (leak / drift check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
N = int(os.environ.get('N', '400'))
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
m.inner.engine().overlap_wgrad = os.environ.get('OVERLAP', '1') != '0'
PLAN = os.environ.get('PLAN', '0') != '0'      # PLAN=1: the default dispatch of bench.py (DeviceSGD + PlannedTrainStep), lr moving every step
x = torch.randn(32, 3, 256, 256, device='cuda'); tgt = torch.rand(32, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(32, 17, device='cuda')
if PLAN:
    from margipose_amd.train_helpers import DeviceSGD, PlannedTrainStep
    opt = DeviceSGD(m.parameters(), lr=1e-2, momentum=0.9)
    planned = PlannedTrainStep(m, opt, x, tgt, mask)
else:
    opt = torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, fused=True)
t0 = time.perf_counter()
for i in range(1, N + 1):
    if PLAN:
        opt.param_groups[0]['lr'] = 1e-2 * (0.5 + 0.5 * (i % 50) / 50.0)
        loss = planned()[1]
    else:
        opt.zero_grad(set_to_none=True)
        loss = dsntnn.average_loss(m.forward_3d_losses(m(x), tgt), mask)
        loss.backward()
        opt.step()
    if i % int(os.environ.get('EVERY', '100')) == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print('step %4d  %.2f ms/step  loss %.4f  allocated %.2f GB (max %.2f)  reserved %.2f GB' % (
            i, (t1 - t0) / int(os.environ.get('EVERY', '100')) * 1e3, float(loss.detach()), torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9,
            torch.cuda.memory_reserved() / 1e9), flush=True)
        assert torch.isfinite(loss.detach())
        t0 = time.perf_counter()
