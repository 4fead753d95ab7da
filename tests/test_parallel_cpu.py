"""CPU, world_size 2, gloo: the data-parallel plumbing (sharding + the single flat-gradient all-reduce)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from margipose_amd import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_range(8, rank, world)
    assert hi - lo == 4 and lo == rank * 4
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(1003, generator=g)
    ref = sum(torch.randn(1003, generator=torch.Generator().manual_seed(100 + k)) for k in range(world)) / world
    parallel.allreduce_mean_(flat, None, world)
    ok = torch.allclose(flat, ref, rtol=1e-6, atol=1e-7)
    # parameters broadcast from rank 0
    lin = torch.nn.Linear(3, 2)
    torch.manual_seed(rank)
    torch.nn.init.normal_(lin.weight)
    parallel.broadcast_parameters(lin)
    gathered = [torch.zeros_like(lin.weight) for _ in range(world)]
    dist.all_gather(gathered, lin.weight.data)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    out.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gloo_world2_allreduce_and_sharding():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_range_errors():
    from margipose_amd import parallel
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_range(10, 0, 4)
    assert parallel.shard_range(256, 7, 8) == (224, 256)
