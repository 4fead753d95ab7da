#!/usr/bin/env python
"""Where do the device-to-device copies of a training step come from?  (torch.profiler, one step, B=32, T=3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel

torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, fused=True)
x = torch.randn(32, 3, 256, 256, device='cuda'); tgt = torch.rand(32, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(32, 17, device='cuda')


def step():
    opt.zero_grad()
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'aten::empty', 'aten::add_', 'aten::mul', 'aten::sum', 'aten::div_', 'aten::stack', 'aten::cat')]
from collections import Counter
c = Counter()
for e in evs:
    st = [s for s in (e.stack or []) if 'margipose_amd' in s or 'bench' in s or 'find_copies' in s]
    c[(e.name, tuple(str(s) for s in (e.input_shapes or []))[:2] and str(e.input_shapes)[:60], st[0] if st else '<autograd/c++>')] += 1
for k, n in c.most_common(40):
    print(n, k)
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25))
