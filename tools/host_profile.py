#!/usr/bin/env python
"""Where the host time of an eager training step goes (cProfile), and whether anything in it synchronises with the GPU
(torch's sync debug mode).  Run on a GPU box: python tools/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile, io, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from margipose_amd.train_helpers import DeviceSGD
B = int(os.environ.get('B', '32'))
torch.manual_seed(0)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
opt = DeviceSGD(m.parameters(), lr=1e-3, momentum=0.9)
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')


def step():
    out = m(x)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, tgt), mask)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print('host enqueue %.2f ms/step (GPU idle at start), step incl. drain %.2f ms' % (1e3 * sorted(enq)[4], 1e3 * sorted(tot)[4]))
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print('back-to-back %.2f ms/step' % ((time.perf_counter() - t0) / 20 * 1e3))
if os.environ.get('HOST_PROFILE_SHORT'):       # (tools/host_profile8.sh: eight of these side by side)
    sys.exit(0)
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    step()
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
print('synchronising calls in one step: %d' % len(w))
for i in w[:10]:
    print('  ', i.filename, i.lineno, str(i.message)[:100])
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
