"""Single-node data parallelism for the MargiPose hot path (net-new: the reference is single-device,
SURVEY.md §8e): one process per GPU, a full replica each, LOCAL BatchNorm (each replica is exactly the
reference computation on its shard), and per-stage buckets of the flat fp32 gradient buffer, each
summed by one asynchronous all-reduce (RCCL over xGMI; `nccl` backend on ROCm) that overlaps the rest of the backward pass
(engine.Engine._finish_bucket).  No activation is ever exchanged.
"""
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (the only mode the host driver supports); before HIP starts

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('MPOSE_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if os.environ.get('MPOSE_SINGLE_DEVICE'):        # functional testing: several ranks share GPU 0 (gloo only)
            local_rank = 0
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if os.environ.get('MPOSE_SINGLE_DEVICE'):
        local_rank = 0
    return rank, world, local_rank


def shard_range(global_batch, rank, world):
    """Contiguous shard [lo, hi) of a global batch for `rank` (global 256 -> 8 x 32)."""
    if global_batch % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def allreduce_mean_(flat, group, world):
    """In-place mean over replicas of a flat gradient buffer.  One collective; returns the same tensor."""
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat


def attach(model, group=None):
    """Turn on gradient averaging for a margipose_amd MargiPoseModel / ChatterboxModel (no-op for world size 1)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    owner = model.inner if hasattr(model, 'inner') else model        # (ChatterboxModel owns its engine itself)
    owner.engine().dp = (group, world) if world > 1 else None
    return model


def broadcast_parameters(model, src=0, group=None):
    """Replicas start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
