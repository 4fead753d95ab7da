#!/usr/bin/env python
"""Functional check of the data-parallel path on ONE GPU box: 2 ranks share cuda:0 over gloo.
   MPOSE_DIST_BACKEND=gloo MPOSE_SINGLE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 tools/dp_check.py
Checks: averaged gradients == mean of the two replicas' single-process gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from margipose_amd import dsntnn, parallel
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel

rank, world, local = parallel.init_from_env()
dev = torch.device('cuda', local)
torch.manual_seed(1 + rank)          # different init per rank on purpose: broadcast must fix it
m = MargiPoseModel(CanonicalSkeletonDesc, 1, True, 'patch8', 'jsd').to(dev).train()
parallel.broadcast_parameters(m)
w0 = torch.cat([p.detach().flatten() for p in m.parameters()])
gathered = [torch.empty_like(w0) for _ in range(world)]
dist.all_gather(gathered, w0)
assert all(torch.equal(gathered[0], g) for g in gathered), 'broadcast failed'

def grads(model, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(4, 3, 256, 256, generator=g).to(dev)
    t = (torch.rand(4, 17, 3, generator=g) * 2 - 1).to(dev)
    model.zero_grad(set_to_none=True)
    loss = dsntnn.average_loss(model.forward_3d_losses(model(x), t), torch.ones(4, 17, device=dev))
    loss.backward()
    return torch.cat([p.grad.flatten() for p in model.parameters()]).clone()

import copy
state = copy.deepcopy(m.state_dict())
local_g = [None] * world
for r in range(world):          # every rank computes every shard's plain gradient (no DP) as the reference
    m.load_state_dict(state)
    local_g[r] = grads(m, 100 + r)
m.load_state_dict(state)
parallel.attach(m)
dp_g = grads(m, 100 + rank)
ref = sum(local_g) / world
err = float((dp_g - ref).norm() / ref.norm())
print('rank %d: DP-averaged gradient vs mean of shards: rel err %.2e' % (rank, err))
assert err < 1e-5, err
dist.barrier()
if rank == 0:
    print('DP_CHECK_OK')
dist.destroy_process_group()
