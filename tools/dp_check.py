#!/usr/bin/env python
"""Functional check of the data-parallel path on ONE GPU box: 2 ranks share cuda:0 over gloo (RCCL needs distinct devices).
   MPOSE_DIST_BACKEND=gloo MPOSE_SINGLE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 tools/dp_check.py [stem]

Parity definition: the averaged gradients equal the MEAN over the shards of a single-device fwd/bwd on each
shard.  Checked twice: against this engine's own single-process gradients of every shard (must agree to fp32 rounding of the
sum), and against the ORACLE (oracle/model_ref.py, fp64 on the host CPU) run on every shard and averaged -- with the fp32
oracle's own deviation from fp64 as the bar, as in tests/test_grad_parity_gpu.py.  Also: the gradient buckets partition the
flat buffer exactly once, in the order the backward pass completes them; parameters AND BatchNorm buffers are broadcast."""
import copy, os, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from margipose_amd import dsntnn, parallel
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from oracle import model_ref as R, weights as W

stem = sys.argv[1] if len(sys.argv) > 1 else 'patch8'
T, B, seed = 2, 2, 4100
rank, world, local = parallel.init_from_env()
dev = torch.device('cuda', local)
x0, _, _ = W.seeded_inputs(seed, B)
sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64, stem=stem), x0.double(), T)
m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd')
m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
m = m.to(dev).train()
if rank != 0:                          # replicas start different on purpose: the broadcast must fix parameters and buffers
    with torch.no_grad():
        for t in list(m.parameters()) + [b for b in m.buffers() if b.is_floating_point()]:
            t.mul_(1.5)
parallel.broadcast_parameters(m)
w0 = torch.cat([t.detach().flatten().float() for t in list(m.parameters()) + [b for b in m.buffers() if b.is_floating_point()]])
gathered = [torch.empty_like(w0) for _ in range(world)]
dist.all_gather(gathered, w0)
assert all(torch.equal(gathered[0], g) for g in gathered), 'broadcast failed'


def shard(r):
    return W.seeded_inputs(seed + 10 + r, B)


def grads(model, r):
    x, t, mk = shard(r)
    model.zero_grad(set_to_none=True)
    loss = dsntnn.average_loss(model.forward_3d_losses(model(x.to(dev)), t.to(dev)), mk.to(dev))
    loss.backward()
    return OrderedDict((k, p.grad.detach().clone()) for k, p in model.named_parameters())


state = copy.deepcopy(m.state_dict())
local_g = []
for r in range(world):                 # every rank computes every shard's plain gradient (no DP)
    m.load_state_dict(state)
    local_g.append(grads(m, r))
m.load_state_dict(state)
parallel.attach(m)
dp_g = grads(m, rank)
eng = m.inner.engine()
# bucket layout: contiguous, disjoint, covering [0, total), stage T-1 first and the stem last
edges = [b for b in eng._buckets]
assert edges[0][0] == 0 and edges[-1][1] == eng._grad_total and all(a[1] == b[0] for a, b in zip(edges, edges[1:])), edges
assert len(edges) == T + 1
offs = dict(zip((id(p) for p in eng.param_list()), eng._grad_offsets))
last_stage_w = m.inner.xy_hm_cnns[T - 1].down_layers[0].module[0].weight
assert edges[0][0] <= offs[id(last_stage_w)] < edges[0][1], 'stage T-1 must be the first bucket'
stem_w = next(m.inner.in_cnn.parameters())
assert edges[-1][0] <= offs[id(stem_w)] < edges[-1][1], 'the stem must be the last bucket'

# (1) vs the engine's own per-shard gradients
worst = 0.0
for k in dp_g:
    ref = sum(g[k] for g in local_g) / world
    n = float(ref.norm())
    if n > 0:
        worst = max(worst, float((dp_g[k] - ref).norm()) / n)
print('rank %d: DP-averaged gradient vs mean of this engine\'s shard gradients: worst tensor rel err %.2e' % (rank, worst))
assert worst < 1e-5, worst


# (2) vs the oracle's mean of shards (fp64), bar = the fp32 oracle's own deviation
def oracle_mean(dtype):
    acc = None
    for r in range(world):
        s = OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())
        params = OrderedDict((k, v.requires_grad_(True)) for k, v in s.items() if v.is_floating_point() and 'running' not in k)
        x, t, mk = shard(r)
        xy, zy, xz = R.inner_forward(s, x.to(dtype), T, True)
        R.average_loss(R.forward_3d_losses(xy, zy, xz, t.to(dtype)), mk.to(dtype)).backward()
        g = OrderedDict((k, p.grad.double()) for k, p in params.items())
        acc = g if acc is None else OrderedDict((k, acc[k] + g[k]) for k in g)
    return OrderedDict((k, v / world) for k, v in acc.items())


if rank == 0:
    g64, g32 = oracle_mean(torch.float64), oracle_mean(torch.float32)
    typical = float(np.median([float(v.norm()) for v in g64.values()]))
    keys = [k for k in g64 if float(g64[k].norm()) > 1e-9 * typical]
    e_gpu = np.array([float((dp_g[k].cpu().double() - g64[k]).norm() / g64[k].norm()) for k in keys])
    e_ref = np.array([float((g32[k] - g64[k]).norm() / g64[k].norm()) for k in keys])
    print('DP gradients vs oracle mean-of-shards (fp64): median %.2e p99 %.2e | fp32 oracle: median %.2e p99 %.2e'
          % (np.median(e_gpu), np.quantile(e_gpu, 0.99), np.median(e_ref), np.quantile(e_ref, 0.99)))
    # free running (each implementation on its own ReLU piece): the rule of tests/test_grad_parity_gpu.py -- within FREE_RATIO = 3 x
    # the reference's own fp32 deviation (which sites flip is luck; check (1) above is the exact statement about the averaging)
    assert np.median(e_gpu) <= max(1e-4, 3.0 * np.median(e_ref)) and np.quantile(e_gpu, 0.99) <= max(1e-4, 3.0 * np.quantile(e_ref, 0.99))
# (3) the same data-parallel step from a launch plan (train_helpers.PlannedTrainStep): the bucket all-reduces are host actions the
#     plan breaks at; a replay must reproduce the eager data-parallel gradients bit for bit
from margipose_amd.train_helpers import DeviceSGD, PlannedTrainStep
m.load_state_dict(state)
xs, ts, ms = [t.to(dev) for t in shard(rank)]
opt0 = DeviceSGD(m.parameters(), lr=0.0, momentum=0.0)
pstep = PlannedTrainStep(m, opt0, xs, ts, ms, warmup=1)
assert len(pstep._host_ops) == T + 2, len(pstep._host_ops)
pstep(xs, ts, ms)
torch.cuda.synchronize()
for k, p in m.named_parameters():
    assert torch.equal(p.grad, dp_g[k]), (k, float((p.grad - dp_g[k]).abs().max()))
print('rank %d: planned data-parallel step == eager data-parallel step (%d launches, %d host actions)' % (rank, pstep.n_launches, len(pstep._host_ops)))
del pstep
dist.barrier()
if rank == 0:
    print('DP_CHECK_OK')
dist.destroy_process_group()
