"""CPU, world_size 2, gloo: the data-parallel plumbing (sharding + the single flat-gradient all-reduce)."""
import os
import sys

import pytest

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


@pytest.mark.parametrize('T,stem', [(1, 'patch8'), (3, 'inceptionv4'), (2, 'resnet18')])
def test_gradient_bucket_layout(T, stem):
    """The flat gradient buffer's buckets (engine.Engine.grad_layout, what _finish_bucket all-reduces): contiguous,
    disjoint, covering the buffer, stage T-1 first ... stage 0, the feature extractor last; every parameter of the model owns
    exactly one 16-byte aligned range inside the bucket of its stage, the combiner feeding stage t sits in stage t's bucket
    (its gradient is complete when that stage's backward is).  Pure host logic: runs without a GPU, so a regression of the layout
    shows up before any lease."""
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd')
    eng = m.inner.engine()
    offs, buckets, total = eng.grad_layout()
    params = eng.param_list()
    named = {id(p): k for k, p in m.named_parameters()}
    assert len(params) == len(offs) == len(named) == len(set(id(p) for p in params))          # every parameter once
    assert len(buckets) == T + 1 and buckets[0][0] == 0 and buckets[-1][1] == total
    for (a, b), (c, d) in zip(buckets, buckets[1:]):
        assert a < b == c < d                                                                  # contiguous, disjoint, non-empty
    spans = sorted((o, o + p.numel(), named[id(p)]) for p, o in zip(params, offs))
    for (a0, a1, _), (b0, b1, _) in zip(spans, spans[1:]):
        assert a1 <= b0 and b0 % 4 == 0 and b0 - a1 < 4                                        # disjoint, aligned, padding < 4 floats
    assert spans[0][0] == 0 and total - spans[-1][1] < 4

    def bucket_of(off):
        return next(i for i, (lo, hi) in enumerate(buckets) if lo <= off < hi)

    for o, _, name in spans:
        bi = bucket_of(o)
        if '_hm_cnns.' in name:
            t = int(name.split('_hm_cnns.')[1].split('.')[0])
            assert bi == T - 1 - t, (name, bi)
        elif 'hm_combiners.' in name:
            t = int(name.split('hm_combiners.')[1].split('.')[0]) + 1                          # combiner t-1 feeds stage t
            assert bi == T - 1 - t, (name, bi)
        else:
            assert name.startswith('inner.in_cnn.') and bi == T, (name, bi)
