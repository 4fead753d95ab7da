"""The reference's training iteration (src/margipose/bin/train_3d.py:154-186 with the 1cycle policy of
hyperparam_scheduler.py:6-42) as ONE replayed HIP graph: margipose_amd.train_helpers.GraphedTrainStep + DeviceSGD.

  * DeviceSGD (csrc/optim.hip, hyper-parameters in device memory) follows torch.optim.SGD to rounding (fused multiply-adds
    against torch's separately rounded products: ~1 ulp per step) while lr and momentum change every step;
  * a replayed graph produces the same weights, bit for bit, as the same iterations run eagerly."""
import copy
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import model_ref as R
from oracle import weights as W

pytestmark = pytest.mark.gpu


def test_device_sgd_equals_torch_sgd_under_a_moving_schedule():
    from margipose_amd.train_helpers import DeviceSGD, make_1cycle
    torch.manual_seed(3)
    shapes = [(192, 128, 3, 3), (128,), (17, 128, 1, 1), (5,), (1,), (64, 3, 7, 7)]
    p_ref = [torch.randn(s, device='cuda').requires_grad_(True) for s in shapes]
    p_dev = [p.detach().clone().requires_grad_(True) for p in p_ref]
    o_ref = torch.optim.SGD(p_ref, lr=0.1, momentum=0.9)
    o_dev = DeviceSGD(p_dev, lr=0.1, momentum=0.9)
    s_ref, s_dev = make_1cycle(o_ref, 10, 1.0, 0.9), make_1cycle(o_dev, 10, 1.0, 0.9)
    for it in range(6):
        grads = [torch.randn(s, device='cuda') for s in shapes]
        for ps, opt, sch in ((p_ref, o_ref, s_ref), (p_dev, o_dev, s_dev)):
            sch.batch_step()
            for p, g in zip(ps, grads):
                p.grad = g.clone()
            opt.step()
        for a, b in zip(p_ref, p_dev):
            err = float((a - b).abs().max() / a.abs().max())
            assert err < 1e-6, (it, err)


def _model(T, seed, x, stem='patch8'):
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64, stem=stem), x.double(), T)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    return m.cuda().train()


@pytest.mark.parametrize('stem', ['patch8', 'inceptionv4'])
def test_graphed_iterations_equal_eager_iterations(stem):
    from margipose_amd.train_helpers import DeviceSGD, GraphedTrainStep, make_1cycle, training_step
    T, seed, B, n_iter = 1, 55, 2, 4
    x, target, mask = W.seeded_inputs(seed, B)
    batches = [W.seeded_inputs(seed + 1 + i, B) for i in range(n_iter)]
    m_e = _model(T, seed, x, stem)
    m_g = copy.deepcopy(m_e)
    # eager: the reference's loop
    opt_e = DeviceSGD(m_e.parameters(), lr=0.05, momentum=0.9)
    sch_e = make_1cycle(opt_e, 20, 0.05, 0.9)
    losses_e = []
    for xb, tb, mb in batches:
        _, loss = training_step(m_e, sch_e, xb.cuda(), tb.cuda(), mb.cuda(), [1] * B)
        losses_e.append(float(loss.detach()))
    # graphed: captured once (the capture's warm-up iterations run on a COPY of the state, which is restored afterwards)
    opt_g = DeviceSGD(m_g.parameters(), lr=0.05, momentum=0.9)
    sch_g = make_1cycle(opt_g, 20, 0.05, 0.9)
    state = copy.deepcopy(m_g.state_dict())
    step = GraphedTrainStep(m_g, opt_g, x.cuda(), target.cuda(), mask.cuda())
    m_g.load_state_dict(state)
    opt_g._bufs.zero_(); opt_g._steps = 0
    losses_g = []
    for xb, tb, mb in batches:
        sch_g.batch_step()
        _, loss = step(xb.cuda(), tb.cuda(), mb.cuda())
        losses_g.append(float(loss))
    assert losses_e == losses_g, (losses_e, losses_g)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize('stem,overlap', [('patch8', True), ('inceptionv4', True), ('patch8', False)])
def test_planned_iterations_equal_eager_iterations(stem, overlap):
    """train_helpers.PlannedTrainStep: the iteration recorded once as a launch plan (csrc/plan.hip) and re-issued from a C loop
    on the eager schedule's two streams gives the same losses and the same weights, bit for bit, as the reference's loop run
    eagerly -- with the weight gradients on the side stream (the default) and on the main stream."""
    from margipose_amd.train_helpers import DeviceSGD, PlannedTrainStep, make_1cycle, training_step
    T, seed, B, n_iter = 1, 55, 2, 4
    x, target, mask = W.seeded_inputs(seed, B)
    batches = [W.seeded_inputs(seed + 1 + i, B) for i in range(n_iter)]
    m_e = _model(T, seed, x, stem)
    m_p = copy.deepcopy(m_e)
    m_e.inner.engine().overlap_wgrad = overlap
    m_p.inner.engine().overlap_wgrad = overlap
    opt_e = DeviceSGD(m_e.parameters(), lr=0.05, momentum=0.9)
    sch_e = make_1cycle(opt_e, 20, 0.05, 0.9)
    losses_e = []
    for xb, tb, mb in batches:
        _, loss = training_step(m_e, sch_e, xb.cuda(), tb.cuda(), mb.cuda(), [1] * B)
        losses_e.append(float(loss.detach()))
    # planned: recorded once (the constructor's iterations run on the state, which is restored afterwards)
    opt_p = DeviceSGD(m_p.parameters(), lr=0.05, momentum=0.9)
    sch_p = make_1cycle(opt_p, 20, 0.05, 0.9)
    state = copy.deepcopy(m_p.state_dict())
    step = PlannedTrainStep(m_p, opt_p, x.cuda(), target.cuda(), mask.cuda())
    assert step.n_launches > 100 and (step.n_waits > 0) == overlap, (step.n_launches, step.n_waits)
    m_p.load_state_dict(state)
    opt_p._bufs.zero_(); opt_p._steps = 0
    # (other users of the allocator between replays must not disturb the plan's buffers: they live in a private pool)
    junk = [torch.randn(1 << 20, device='cuda') for _ in range(8)]
    losses_p = []
    for xb, tb, mb in batches:
        sch_p.batch_step()
        _, loss = step(xb.cuda(), tb.cuda(), mb.cuda())
        losses_p.append(float(loss))
        junk = [torch.randn(1 << 20, device='cuda') for _ in range(8)]
    assert losses_e == losses_p, (losses_e, losses_p)
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_p.state_dict().items()):
        assert torch.equal(a, b), k
    # the parameters' .grad tensors follow the replays (they are views of the recorded iteration's gradient buffer)
    for pe, pp in zip(m_e.parameters(), m_p.parameters()):
        assert torch.equal(pe.grad, pp.grad)


def test_stale_launch_plan_is_refused_and_pending_forwards_survive_a_replay():
    """A launch plan replays raw addresses.  (1) A replayed eval forward (PlannedInference) between an eager training
    forward and its backward overwrites the engine's BatchNorm arenas like any forward does: the pending forward keeps its vectors
    (Engine.before_replay snapshots them) and its gradients equal those of the same forward / backward with no replay in between.
    (2) A replayed TRAINING iteration updates the weights: a pending forward's backward afterwards raises autograd's "modified by an
    inplace operation", like after an eager optimiser step.  (3) A parameter re-bound behind a plan's back (`p.data = ...`,
    load_state_dict(assign=True), model.to()) makes the recorded addresses stale: the next replay raises instead of writing
    through them."""
    from margipose_amd._lib import MposeError
    from margipose_amd.train_helpers import DeviceSGD, PlannedInference, PlannedTrainStep, forward_loss
    T, seed, B = 1, 61, 2
    x, target, mask = W.seeded_inputs(seed, B)
    m = _model(T, seed, x)
    m.eval()
    infer = PlannedInference(m, x.cuda())
    m.train()
    xb, tb, mb = (t.cuda() for t in W.seeded_inputs(seed + 1, B))

    def grads(between):
        out = m(xb)
        loss = forward_loss(m, out, tb, mb, [1] * B)
        if between is not None:
            between()
        for p in m.parameters():
            p.grad = None
        loss.backward()
        return [p.grad.clone() for p in m.parameters()]
    g0, g1 = grads(None), grads(infer)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    opt = DeviceSGD(m.parameters(), lr=0.0, momentum=0.0)
    step = PlannedTrainStep(m, opt, x.cuda(), target.cuda(), mask.cuda())
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        grads(step)
    step()                                                           # the plans themselves are still valid
    infer()
    p0 = next(m.parameters())
    p0.data = p0.data.clone()                                        # a weight re-bound behind the plans' back
    with pytest.raises(MposeError, match='stale'):
        step()
    with pytest.raises(MposeError, match='stale'):
        infer()


def test_training_iteration_launches_only_library_kernels():
    """What a launch plan cannot record must not be in the iteration: every device kernel of one eager training iteration (forward,
    3D loss, backward, DeviceSGD) is a launch of libmargipose_hip.so -- no fills, copies or elementwise kernels of the tensor library."""
    from torch.profiler import ProfilerActivity, profile
    from margipose_amd import dsntnn
    from margipose_amd.train_helpers import DeviceSGD, forward_loss
    T, seed, B = 2, 56, 2
    x, target, mask = [t.cuda() for t in W.seeded_inputs(seed, B)]
    m = _model(T, seed, x.cpu(), 'inceptionv4')
    opt = DeviceSGD(m.parameters(), lr=0.05, momentum=0.9)
    one = torch.ones((), device='cuda')

    def iteration():
        out = m(x)
        loss = forward_loss(m, out, target, mask, [1] * B)
        opt.zero_grad(set_to_none=True)
        loss.backward(one)
        opt.step()

    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        iteration()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert len(names) > 100
    foreign = sorted(set(n for n in names if 'mpose' not in n and 'Memcpy HtoD' not in n))
    assert not foreign, foreign


@pytest.mark.parametrize('bf16,frozen', [(False, False), (True, False), (False, True)])
def test_planned_inference_equals_eager_forward(bf16, frozen):
    """train_helpers.PlannedInference: the eval-mode forward recorded as a launch plan returns, for new inputs and after the weights
    changed, exactly what the eager forward returns (coordinates and every stage's heatmaps), fp32 and bf16 heatmap storage; and the
    eager forward it records launches only library kernels."""
    from torch.profiler import ProfilerActivity, profile
    from margipose_amd.train_helpers import PlannedInference
    T, seed, B = 2, 57, 3
    xs = [W.seeded_inputs(seed + i, B)[0].cuda() for i in range(3)]
    m = _model(T, seed, xs[0].cpu(), 'inceptionv4').eval()
    if bf16:
        m.heatmap_dtype = torch.bfloat16
    with torch.no_grad():
        m(xs[0])
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            m(xs[0])
            torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert len(names) > 50 and not sorted(set(n for n in names if 'mpose' not in n)), sorted(set(n for n in names if 'mpose' not in n))
    pf = PlannedInference(m, xs[0], frozen_weights=frozen)
    assert pf.n_launches > 50
    if frozen:               # (the weight measuring / packing launches are not in the recording, and two calls in a row pack nothing;
        eng = m.inner.engine()           # the eager forwards and the weight update below are noticed and packed once, before the replay)
        pf(xs[0]); e0 = eng._pack_epoch
        pf(xs[1]); assert eng._pack_epoch == e0
        assert pf.n_launches < PlannedInference(m, xs[0]).n_launches
    for it, x in enumerate(xs):
        if it == 2:          # the weights may move between calls: they are read at replay time
            with torch.no_grad():
                for p in m.parameters():
                    p.mul_(1.01)
        got = pf(x).clone()
        got_hm = [h.clone() for h in m.xy_heatmaps + m.zy_heatmaps + m.xz_heatmaps]
        with torch.no_grad():
            want = m(x)
        want_hm = m.xy_heatmaps + m.zy_heatmaps + m.xz_heatmaps
        assert torch.equal(got, want)
        assert len(got_hm) == len(want_hm) and all(torch.equal(a, b) for a, b in zip(got_hm, want_hm))
        assert want_hm[0].dtype == (torch.bfloat16 if bf16 else torch.float32)


def test_batch_stager_pinned_double_buffer():
    """train_helpers.BatchStager (reference bin/train_3d.py:158-161 done with pinned double buffers on a copy stream): values
    arrive intact for float and uint8 frames, slots rotate, and the consumer needs no host synchronisation."""
    from margipose_amd.train_helpers import BatchStager
    st = BatchStager('cuda:0')
    rng = np.random.default_rng(5)
    seen = []
    for it in range(5):
        u8 = it % 2 == 1
        inp = torch.from_numpy(rng.integers(0, 256, (4, 3, 32, 32), dtype=np.uint8)) if u8 else torch.from_numpy(rng.standard_normal((4, 3, 32, 32)))
        batch = {'input': inp, 'target': torch.from_numpy(rng.uniform(-1, 1, (4, 17, 4))), 'joint_mask': torch.ones(4, 17, dtype=torch.float64),
                 'valid_depth': [1, 1, 0, 1]}
        dev = st.stage(batch)
        assert dev['valid_depth'] == [1, 1, 0, 1]
        assert dev['input'].is_cuda and dev['input'].dtype == (torch.uint8 if u8 else torch.float32)
        assert dev['target'].dtype == torch.float32 and dev['joint_mask'].dtype == torch.float32
        seen.append((dev['input'].float().sum() + dev['target'].sum(), float(inp.double().sum() + batch['target'].sum())))
    torch.cuda.synchronize()
    for got, want in seen:
        assert abs(float(got) - want) <= 1e-3 * max(1.0, abs(want))
