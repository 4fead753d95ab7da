#!/usr/bin/env python
"""How well do concurrent streams of smaller launches pack against one stream of large grouped launches?  Two independent models
at batch B/2, each replaying its launch plan on its own pair of streams, against one model at batch B (images per second)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from margipose_amd.train_helpers import DeviceSGD, PlannedTrainStep

B = int(os.environ.get('B', '32'))
N = int(os.environ.get('N', '2'))          # concurrent replicas
steps = 20


def make(b, seed):
    torch.manual_seed(seed)
    m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'inceptionv4', 'jsd').cuda().train()
    opt = DeviceSGD(m.parameters(), lr=0.01, momentum=0.9)
    x = torch.randn(b, 3, 256, 256, device='cuda'); tgt = torch.rand(b, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(b, 17, device='cuda')
    return m, opt, x, tgt, mask


def run(n, b):
    streams = [torch.cuda.Stream() for _ in range(n)]
    plans = []
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            m, opt, x, tgt, mask = make(b, i)
            plans.append(PlannedTrainStep(m, opt, x, tgt, mask))
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(steps):
            for s, p in zip(streams, plans):
                with torch.cuda.stream(s):
                    p()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print('%d concurrent replicas x batch %d: %.2f ms per round, %.0f images/s' % (n, b, 1e3 * dt, n * b / dt))


run(1, B)
run(N, B // N)
run(1, B // N)
