import sys, os
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'patch8', 'jsd').cuda().train()
opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
x = torch.randn(32, 3, 256, 256, device='cuda'); t = torch.rand(32, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(32, 17, device='cuda')
def step():
    opt.zero_grad(set_to_none=True)
    out = m(x); loss = dsntnn.average_loss(m.forward_3d_losses(out, t), mask); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)[:22]
for e in rows:
    print('%-60s calls=%5d self_cpu=%8.1fus cuda=%8.1fus' % (e.key[:60], e.count, e.self_cpu_time_total, e.device_time_total))
print('---- copies')
for e in ka:
    if 'copy' in e.key.lower() or 'clone' in e.key.lower() or 'Memcpy' in e.key:
        print('%-60s calls=%5d self_cpu=%8.1fus cuda=%8.1fus' % (e.key[:60], e.count, e.self_cpu_time_total, e.device_time_total))
