#!/usr/bin/env python
"""Full per-label GPU time of one training step (HIP events around every launch), grouped by kernel family."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import dsntnn
from margipose_amd.engine import KernelTimer
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel

B, T = int(os.environ.get('B', '32')), int(os.environ.get('T', '3'))
stem = os.environ.get('STEM', 'inceptionv4')
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd').cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9)
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')


def step():
    opt.zero_grad(set_to_none=True)
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
timer = KernelTimer(); m.inner.engine().timer = timer
torch.cuda.synchronize()
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
summ = timer.summary()
tot = sum(v['total_ms'] for v in summ.values()) / N
fam = collections.defaultdict(float)
for k, v in summ.items():
    fam[k.split(':')[0] + (':stem' if 'stem' in k else '')] += v['total_ms'] / N
print('total timed ms/step %.2f' % tot)
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print('%-24s %7.3f' % (k, v))
print('--- all labels')
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['total_ms']):
    w = v.get('work_per_launch', 0) or 0
    print('%-46s n=%4d %8.3f ms/step  avg %8.1f us  %s' % (k, v['n'] // N, v['total_ms'] / N, v['avg_us'],
          ('%.1f TF' % (w / v['avg_us'] / 1e6)) if k.startswith(('conv', 'wgrad')) else ('%.0f GB/s' % (w / v['avg_us'] / 1e3) if w else '')))
