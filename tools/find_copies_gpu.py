#!/usr/bin/env python
"""Which host-side ops issue device-to-device copies in one training step (torch profiler with stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from margipose_amd.train_helpers import DeviceSGD
B = 8
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
opt = DeviceSGD(m.parameters(), lr=1e-3, momentum=0.9)
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')
def step():
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
names = collections.Counter(e.name for e in prof.events())
print([(k, v) for k, v in names.most_common(40)])
c = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::fill_', 'aten::zero_', 'aten::empty', 'aten::empty_like', 'aten::zeros_like'):
        st = [s for s in e.stack if 'margipose_amd' in s or 'tools/' in s]
        c[(e.name, st[0] if st else (e.stack[0] if e.stack else '?'), str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(40):
    print(v, k)
