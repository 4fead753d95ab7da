"""Timing of the eval-mode forward at B=64 (bench.py's inference leg) and of one training step; A/B of MPOSE_SLIM."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
torch.manual_seed(1)
m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().eval()
x = torch.randn(64, 3, 256, 256, device='cuda')
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print('MPOSE_SLIM=%s eval forward B=64: %.3f ms  %.0f images/s' % (os.environ.get('MPOSE_SLIM', 'default'), 1e3 * dt, 64 / dt))
