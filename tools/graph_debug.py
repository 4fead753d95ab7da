#!/usr/bin/env python
"""Debug: eager vs graphed iterations, DeviceSGD vs torch SGD (loss curves over a few steps)."""
import copy, os, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import model_ref as R, weights as W
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from margipose_amd.train_helpers import DeviceSGD, GraphedTrainStep, training_step

T, seed, B, n_iter = 1, 55, 2, 4
x, target, mask = W.seeded_inputs(seed, B)
batches = [W.seeded_inputs(seed + 1 + i, B) for i in range(n_iter)]
sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64), x.double(), T)


def model():
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, 'patch8', 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    return m.cuda().train()


class Sch:
    def __init__(self, o): self.optimizer = o


def eager(opt_factory, overlap=True):
    m = model(); m.inner.engine().overlap_wgrad = overlap
    sch = Sch(opt_factory(m))
    out = []
    for xb, tb, mb in batches:
        _, loss = training_step(m, sch, xb.cuda(), tb.cuda(), mb.cuda(), [1] * B)
        out.append(float(loss.detach()))
    return out, m


def graphed(overlap=True):
    m = model(); m.inner.engine().overlap_wgrad = overlap
    opt = DeviceSGD(m.parameters(), lr=0.05, momentum=0.9)
    state = copy.deepcopy(m.state_dict())
    step = GraphedTrainStep(m, opt, x.cuda(), target.cuda(), mask.cuda())
    m.load_state_dict(state); opt._bufs.zero_(); opt._steps = 0
    out = []
    for xb, tb, mb in batches:
        _, loss = step(xb.cuda(), tb.cuda(), mb.cuda())
        out.append(float(loss))
    return out, m


a, ma = eager(lambda m: DeviceSGD(m.parameters(), lr=0.05, momentum=0.9))
b, mb_ = eager(lambda m: torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9))
a2, _ = eager(lambda m: DeviceSGD(m.parameters(), lr=0.05, momentum=0.9), overlap=False)
c, mc = graphed()
d, md = graphed(overlap=False)
print('eager  DeviceSGD          ', a)
print('eager  DeviceSGD no-ovlp  ', a2)
print('eager  torch SGD          ', b)
print('graph  DeviceSGD          ', c)
print('graph  DeviceSGD no-ovlp  ', d)
for name, m2 in (('eager torch', mb_), ('graph', mc), ('graph no-ovlp', md)):
    worst = max((float((p1 - p2).abs().max() / (p1.abs().max() + 1e-30)), k) for (k, p1), (_, p2) in zip(ma.state_dict().items(), m2.state_dict().items()) if p1.is_floating_point())
    print('max rel weight diff vs eager DeviceSGD:', name, worst)
