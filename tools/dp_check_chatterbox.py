#!/usr/bin/env python
"""Data-parallel path of ChatterboxModel on ONE GPU box: 2 ranks share cuda:0 over gloo (RCCL needs distinct devices).
   MPOSE_DIST_BACKEND=gloo MPOSE_SINGLE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 tools/dp_check_chatterbox.py

Parity definition: the averaged gradients equal the MEAN over the shards of a single-device fwd/bwd on each
shard -- checked against this engine's own single-process gradients of every shard (fp32 rounding of the sum); the model's
single gradient bucket covers the flat buffer; parameters and BatchNorm buffers are broadcast.  (The engine's shard gradients
themselves are checked against the oracle in tests/test_chatterbox_gpu.py.)"""
import copy, os, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from margipose_amd import dsntnn, parallel
from margipose_amd.models import CanonicalSkeletonDesc, ChatterboxModel
from oracle import weights as W

B, seed = 2, 4200
rank, world, local = parallel.init_from_env()
dev = torch.device('cuda', local)
torch.manual_seed(seed)
m = ChatterboxModel(CanonicalSkeletonDesc, 'jsd').to(dev).train()
if rank != 0:                          # replicas start different on purpose: the broadcast must fix parameters and buffers
    with torch.no_grad():
        for t in list(m.parameters()) + [b for b in m.buffers() if b.is_floating_point()]:
            t.mul_(1.5)
parallel.broadcast_parameters(m)
w0 = torch.cat([t.detach().flatten().float() for t in list(m.parameters()) + [b for b in m.buffers() if b.is_floating_point()]])
gathered = [torch.empty_like(w0) for _ in range(world)]
dist.all_gather(gathered, w0)
assert all(torch.equal(gathered[0], g) for g in gathered), 'broadcast failed'


def grads(model, r):
    x, t, mk = W.seeded_inputs(seed + 10 + r, B)
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
assert m.engine().dp is not None
dp_g = grads(m, rank)
eng = m.engine()
assert eng._buckets == [(0, eng._grad_total)], eng._buckets
worst = 0.0
for k in dp_g:
    ref = sum(g[k] for g in local_g) / world
    n = float(ref.norm())
    if n > 0:
        worst = max(worst, float((dp_g[k] - ref).norm()) / n)
print('rank %d: DP-averaged gradient vs mean of the shard gradients: worst tensor rel err %.2e' % (rank, worst))
assert worst < 1e-5, worst
dist.barrier()
if rank == 0:
    print('DP_CHECK_OK')
dist.destroy_process_group()
