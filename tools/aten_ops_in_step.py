#!/usr/bin/env python
"""Which ATen operators (not mpose_* launches) does one eager training iteration run?  (What a launch plan cannot record.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from margipose_amd.train_helpers import DeviceSGD

B = 8
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, sys.argv[1] if len(sys.argv) > 1 else 'inceptionv4', 'jsd').cuda().train()
opt = DeviceSGD(m.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')


def step():
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::')]
skip = {'aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::view', 'aten::as_strided', 'aten::detach', 'aten::alias', 'aten::narrow',
        'aten::slice', 'aten::select', 'aten::reshape', 'aten::permute', 'aten::_unsafe_view', 'aten::unsqueeze', 'aten::squeeze', 'aten::expand',
        'aten::to', 'aten::lift_fresh', 'aten::result_type', 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense', 'aten::contiguous', 'aten::t', 'aten::transpose'}
n = 0
for e in evs:
    if e.name in skip:
        continue
    n += 1
    st = [s for s in (e.stack or []) if 'margipose_amd' in s or 'tools/' in s]
    print('%-28s %-40s %s' % (e.name, str(e.input_shapes)[:40] if e.input_shapes else '', ' <- '.join(s.split('/')[-1] for s in st[:3])))
print('total', n)
kn = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and 'mpose' not in e.name:
        kn[e.name] = kn.get(e.name, 0) + 1
for k, v in sorted(kn.items(), key=lambda kv: -kv[1]):
    print(v, k[:150])
