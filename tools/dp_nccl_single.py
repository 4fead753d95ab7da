#!/usr/bin/env python
"""The RCCL code path of the data-parallel step on the ONE GPU a test box has: a single rank initialises the `nccl` backend
(= RCCL on ROCm) with world_size 1, data parallelism is forced on, and a 2-stage training step runs the real schedule --
weight gradients on the side stream (Engine.dp_overlap() is true under nccl), one asynchronous ncclAllReduce per gradient
bucket issued from Engine._finish_bucket while the backward pass continues, work.wait() at the end.  With one rank the sum over
replicas is the identity, so the gradients must equal the plain single-device step bit for bit; the collective itself, the
stream ordering around it and (second half) its capture into a HIP graph (MPOSE_DP_GRAPH=1 in bench.py) are what runs here.

    python tools/dp_nccl_single.py            prints DP_NCCL_SINGLE_OK"""
import os, sys
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from margipose_amd import dsntnn
from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
from margipose_amd.train_helpers import DeviceSGD, GraphedTrainStep

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(29700 + os.getpid() % 200))
torch.cuda.set_device(0)
dist.init_process_group(backend='nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
torch.manual_seed(5)
T, B = 2, 4
m = MargiPoseModel(CanonicalSkeletonDesc, T, True, 'patch8', 'jsd').cuda().train()
x = torch.randn(B, 3, 256, 256, device='cuda'); tgt = torch.rand(B, 17, 3, device='cuda') * 2 - 1; mask = torch.ones(B, 17, device='cuda')
state = {k: v.clone() for k, v in m.state_dict().items()}


def grads():
    m.load_state_dict(state)
    m.zero_grad(set_to_none=True)
    loss = dsntnn.average_loss(m.forward_3d_losses(m(x), tgt), mask)
    loss.backward()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in m.parameters()], float(loss)


eng = m.inner.engine()
g_plain, l_plain = grads()
eng.dp = (None, 1)                       # (parallel.attach() leaves world size 1 alone: force the collective path)
assert eng.dp_overlap(), 'the side-stream schedule must be on under the nccl backend'
calls = []
orig = dist.all_reduce
def counting(*a, **k):
    calls.append(a[0].numel())
    return orig(*a, **k)
dist.all_reduce = counting
g_dp, l_dp = grads()
dist.all_reduce = orig
assert len(calls) == T + 1 and sum(calls) == eng._grad_total, (calls, eng._grad_total)      # one bucket per stage + the stem
assert l_dp == l_plain
for a, b in zip(g_plain, g_dp):
    assert torch.equal(a, b), float((a - b).abs().max())
print('DP_NCCL_SINGLE_OK buckets=%s (eager schedule)' % calls, flush=True)
# the same step from a LAUNCH PLAN (train_helpers.PlannedTrainStep): the plan breaks at every bucket's all-reduce and at the final
# wait, a replay issues the collectives again from the host between two segments of recorded launches
from margipose_amd.train_helpers import PlannedTrainStep
m.load_state_dict(state)
opt0 = DeviceSGD(m.parameters(), lr=0.0, momentum=0.0)          # (lr 0: the iterations leave the weights where they are)
pstep = PlannedTrainStep(m, opt0, x, tgt, mask, warmup=1)
assert len(pstep._host_ops) == T + 2, len(pstep._host_ops)       # T + 1 buckets and the wait for them
calls = []
dist.all_reduce = counting
for _ in range(2):
    out, loss = pstep(x, tgt, mask)
dist.all_reduce = orig
torch.cuda.synchronize()
assert len(calls) == 2 * (T + 1), calls
assert float(loss) == l_plain, (float(loss), l_plain)
for a, p in zip(g_plain, m.parameters()):
    assert torch.equal(a, p.grad), float((a - p.grad).abs().max())
print('DP_NCCL_PLAN_OK launches=%d waits=%d host_ops=%d' % (pstep.n_launches, pstep.n_waits, len(pstep._host_ops)), flush=True)
del pstep
if os.environ.get('MPOSE_DP_GRAPH') != '1':
    os._exit(0)                          # (skip the process-group teardown: nothing to synchronise with)
# MPOSE_DP_GRAPH=1: the same step, collectives included, captured as ONE HIP graph and replayed (a runtime whose RCCL cannot be
# captured aborts the process here: the caller treats this half as informational)
m.load_state_dict(state)
opt = DeviceSGD(m.parameters(), lr=0.0, momentum=0.0)           # (lr 0: the replays leave the weights where they are)
graph = 'ok'
try:
    step = GraphedTrainStep(m, opt, x, tgt, mask, warmup=1)
except Exception as e:                   # (a runtime that cannot capture the collective: reported, the eager path above stands)
    graph = 'capture failed: %s: %s' % (type(e).__name__, str(e)[:200])
    torch.cuda.synchronize()
if graph == 'ok':
    out, loss = step(x, tgt, mask)
    torch.cuda.synchronize()
    assert abs(float(loss) - l_plain) <= 1e-6 * abs(l_plain), (float(loss), l_plain)
    for a, p in zip(g_plain, m.parameters()):
        assert torch.equal(a, p.grad), float((a - p.grad).abs().max())
print('DP_NCCL_GRAPH %s' % graph, flush=True)
os._exit(0)                              # (skip the process-group teardown: nothing to synchronise with)
