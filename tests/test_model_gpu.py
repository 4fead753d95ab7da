"""GPU parity of the whole hot path (margipose_amd.models.MargiPoseModel through the C ABI) against the
oracle (oracle/model_ref.py, fp64 on the host CPU, same seeded weights/inputs) and against the
reference-generated golden vectors (tests/golden/model_T2.npz).

Tolerance: BASELINE.json's north-star gate is 1e-4 relative fp32.  Tensors are compared by relative
L2 / max-norm error per tensor (raw near-zero heatmap tails are meaningless elementwise)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import model_ref as R
from oracle import weights as W

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


_WEIGHTS = {}


def weights(T, seed, x, axis_permutation=True):
    """fp64 state dict with calibrated BN running statistics (see oracle.model_ref.calibrate_running_stats).  The calibration is
    a train-mode fp64 forward on the CPU; a test asks for the same dict two or three times (model, fp64 oracle, fp32 oracle):
    computed once per (T, seed, input), handed out as fresh copies."""
    key = (T, seed, axis_permutation, tuple(x.shape), float(x.double().sum()))
    if key not in _WEIGHTS:
        _WEIGHTS.clear()           # (one entry: the tests of a file walk through their configurations one after the other)
        _WEIGHTS[key] = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64), x.double(), T, axis_permutation)
    return OrderedDict((k, v.clone()) for k, v in _WEIGHTS[key].items())


def build(T, seed, x, axis_permutation=True):
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    m = MargiPoseModel(CanonicalSkeletonDesc, T, axis_permutation, 'patch8', 'jsd')
    sd = weights(T, seed, x, axis_permutation)
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    return m.cuda()


def oracle_step(T, seed, x, target, mask, train=True, axis_permutation=True, valid_depth=None, dtype=torch.float64):
    sd = weights(T, seed, x, axis_permutation)
    sd = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items())
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
    x = x.to(dtype).requires_grad_(True)
    target, mask = target.to(dtype), mask.to(dtype)
    xy, zy, xz = R.inner_forward(sd, x, T, train, axis_permutation)
    out = {'xy': xy, 'zy': zy, 'xz': xz, 'coords': R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1])}
    l3 = R.forward_3d_losses(xy, zy, xz, target)
    out['l3'] = l3
    out['l2'] = R.forward_2d_losses(xy, zy, xz, target)
    if valid_depth is None:
        losses = l3
    else:
        vd = valid_depth.to(dtype)[:, None]
        losses = vd * l3 + (1 - vd) * out['l2']
    loss = R.average_loss(losses, mask)
    if train:
        loss.backward()
        out['grads'] = OrderedDict((k, p.grad) for k, p in params.items())
        out['dx'] = x.grad
    out['loss'] = loss
    out['sd'] = sd
    return out


def report(name, errs, tol=TOL):
    os.makedirs('gpurun_out', exist_ok=True)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:15]
    with open(os.path.join('gpurun_out', 'parity_%s.json' % name), 'w') as f:
        json.dump({'worst': worst, 'n': len(errs), 'max': worst[0][1] if worst else 0.0}, f, indent=1)
    bad = [(k, e) for k, e in worst if not (e < tol)]
    assert not bad, 'parity failures (%s): %s' % (name, bad[:10])


@pytest.mark.parametrize('T,perm', [(1, True), (2, True), (1, False)])
def test_eval_forward(T, perm):
    """Eval-mode forward: conv_igemm_k with three fp16 products, BatchNorm + ReLU + residual sum in its epilogues."""
    seed, B = 400 + T, 2
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    m = build(T, seed, x, perm).eval()
    with torch.no_grad():
        out = m(x.cuda())
        l3 = m.forward_3d_losses(out, target.cuda())
    ref = oracle_step(T, seed, x, target, mask, train=False, axis_permutation=perm)
    errs = {'coords': rel(out.cpu(), ref['coords'].detach()), 'l3': rel(l3.cpu(), ref['l3'].detach())}
    for p in ('xy', 'zy', 'xz'):
        for t in range(T):
            errs['hm_%s%d' % (p, t)] = rel(getattr(m, p + '_heatmaps')[t].cpu(), ref[p][t].detach())
    report('eval_T%d_%d' % (T, perm), errs)


def test_inference_config_batch64_vs_oracle():
    """BASELINE configs[1] at its own size -- batch 64, 3 stages, 256x256, eval mode, bf16 heatmap storage + fp32 soft-argmax --
    against the fp64 oracle with the same rounding points (tests' test_bf16_heatmap_inference_mode explains them).  At B=64 the
    split-K cost model picks the unsplit 128-channel launches with the fused output stage (conv.hip::pick_ks, inference
    constants): the launch plan bench.py's inference leg times, checked here at full size (the small-batch eval tests run other
    plans).  Coordinates 1e-4; stored heatmaps one bf16 ulp where the fp32 values straddle a rounding boundary."""
    T, seed, B = 3, 460, 64
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    m = build(T, seed, x).eval()
    m.heatmap_dtype = torch.bfloat16
    with torch.no_grad():
        out = m(x.cuda())
        assert out.shape == (B, 17, 3) and m.xy_heatmaps[-1].dtype == torch.bfloat16
        sd = weights(T, seed, x, True)
        sd = OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in sd.items())
        un = {}
        # eval mode: every frame is independent of its batch mates, so the fp64 oracle checks every eighth frame of the batch
        # the GPU ran whole (an eighth of the CPU time: the suite's budget; every fourth under MPOSE_LONG_TESTS)
        sub = torch.arange(0, B, 4 if os.environ.get('MPOSE_LONG_TESTS', '0') != '0' else 8)
        xy, zy, xz = R.inner_forward(sd, x[sub].double(), T, False, True, heatmap_dtype=torch.bfloat16, unrounded=un)
        ref = R.heatmaps_to_coords(un['xy'], un['zy'], un['xz'])
    err = rel(out.cpu()[sub], ref)
    worst = 0.0
    for got, want in ((m.xy_heatmaps, xy), (m.zy_heatmaps, zy), (m.xz_heatmaps, xz)):
        for t in range(T):
            g, w = got[t].float().cpu().double()[sub], want[t]
            worst = max(worst, float(((g - w).abs() / w.abs().clamp_min(1e-30)).max()))
            assert float((g != w).double().mean()) < 5e-3
    print('configs[1] B=64 T=3: coords %.2e, worst heatmap element %.2e' % (err, worst))
    assert err < TOL and worst <= 2.0 ** -7, (err, worst)


def grad_noise_gate(name, gpu, ref64, ref32, same_piece=None):
    """Gradient parity gate.  Raw per-tensor gradients of this network are NOT reproducible to 1e-4 by ANY fp32
    implementation: ReLU-mask flips and small-batch BatchNorm make stock PyTorch fp32 (the reference's CPU path)
    deviate from fp64 by ~3e-4 median / ~3e-3 worst (relative L2) at these sizes.  So the gate is
    'as close to the fp64 truth as the reference's own fp32 path', measured in this very run:
        median(err_gpu) <= max(1e-4, 3 x median(err_ref_fp32), p90(err_ref_fp32), p99(err_ref_fp32)),
        p99(err_gpu) <= max(1e-4, 5 x max(err_ref_fp32)).
    (The p90 term: one flipped ReLU / max-pool winner perturbs every gradient UPSTREAM of it by ~1e-3 and nothing
    downstream, so the per-tensor errors of any fp32 path are bimodal and WHERE the split falls differs between two
    correct implementations -- e.g. resnet18, B=2: the fp32 CPU path shows 4e-3 on all stem + first-block tensors and
    1e-5 on the rest, the GPU path 4e-3 / 8e-4 / 1e-5.  The GPU's typical error may therefore reach, but not exceed,
    what the reference's own fp32 path shows on a tenth of the tensors; gpurun_out/gradnoise_*.json keeps every
    per-tensor pair.)
    Round 5: the median term no longer admits p99(err_ref_fp32).  WHERE a flip falls is luck -- one in the last blocks
    of a column perturbs every tensor upstream of it, i.e. most of them, and the median then IS that effect (T=1, B=2 with the H2
    engine's rounding: one flip in block 9 of the xz column, median 1.4e-3 against the fp32 run's p90 1.2e-3) -- so a run that
    misses the free-running median gate is accepted ONLY if `same_piece` (a callable returning the statistics of the comparison
    with both oracles forced onto the GPU's ReLU piece, oracle/piece.py) shows pure arithmetic error inside the strict gates of
    tests/test_grad_parity_gpu.py: every tensor <= max(3e-5, 1.5 x the fp32 oracle's worst), median <= 1.5 x the fp32 oracle's.
    Callers without piece control (ResNet stems, ChatterboxModel) get no escape.
    Parameters whose true gradient is analytically zero (the last shortcut BN's bias: softmax is shift invariant)
    are checked for absolute smallness instead."""
    norms = np.array([float(ref64[k].norm()) for k in ref64])
    typical = float(np.median(norms))
    e_gpu, e_ref, zero_abs = {}, {}, {}
    for k in ref64:
        n = float(ref64[k].norm())
        if n < 1e-9 * typical:
            zero_abs[k] = float(gpu[k].double().norm()) / typical
            continue
        e_gpu[k] = rel_l2(gpu[k], ref64[k])
        e_ref[k] = rel_l2(ref32[k].double(), ref64[k])
    vg, vr = np.array(list(e_gpu.values())), np.array(list(e_ref.values()))
    stats = {'gpu_median': float(np.median(vg)), 'gpu_p99': float(np.quantile(vg, 0.99)), 'gpu_max': float(vg.max()),
             'ref32_median': float(np.median(vr)), 'ref32_p90': float(np.quantile(vr, 0.9)),
             'ref32_p99': float(np.quantile(vr, 0.99)), 'ref32_max': float(vr.max()),
             'zero_grad_abs_max': max(zero_abs.values()) if zero_abs else 0.0,
             'worst_gpu': sorted(e_gpu.items(), key=lambda kv: -kv[1])[:8],
             'per_key': {k: [e_gpu[k], e_ref[k]] for k in e_gpu}}
    os.makedirs('gpurun_out', exist_ok=True)
    with open(os.path.join('gpurun_out', 'gradnoise_%s.json' % name), 'w') as f:
        json.dump(stats, f, indent=1)
    print(name, {k: v for k, v in stats.items() if k not in ('worst_gpu', 'per_key')})
    if stats['gpu_median'] > max(TOL, 3 * stats['ref32_median'], stats['ref32_p90']):
        assert same_piece is not None, stats
        sp = same_piece()
        print(name, 'free-running median outside the gate: a ReLU flip; same piece:', {k: v for k, v in sp.items() if k != 'worst_gpu'})
        assert sp['gpu_max'] <= max(3e-5, 1.5 * sp['ref32_max']) and sp['gpu_median'] <= max(1.5 * sp['ref32_median'], 2e-6), (stats, sp)
    assert stats['gpu_p99'] <= max(TOL, 5 * stats['ref32_max']), stats
    assert stats['zero_grad_abs_max'] < 1e-4, stats


@pytest.mark.parametrize('T,B', [(1, 2)] + ([(2, 2)] if os.environ.get('MPOSE_LONG_TESTS', '0') != '0' else []))
def test_train_step_vs_oracle(T, B):
    """(T = 2 under MPOSE_LONG_TESTS: its forward is test_model_T2_vs_reference_golden's, its gradients
    tests/test_grad_parity_gpu.py::test_grads_on_the_same_relu_piece[2-2-auto]'s.)"""
    seed = 500 + T
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    rng = np.random.default_rng(seed)
    mask = torch.tensor((rng.uniform(0, 1, (B, 17)) > 0.2).astype(np.float32))
    from margipose_amd import dsntnn
    m = build(T, seed, x).train()
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    from oracle import piece
    masks = piece.gpu_relu_masks(m.inner.engine(), m.xy_heatmaps[0].grad_fn.ectx)      # the piece this forward ran on (read before backward)
    l3 = m.forward_3d_losses(out, target.cuda())
    loss = dsntnn.average_loss(l3, mask.cuda())
    loss.backward()
    ref = oracle_step(T, seed, x, target, mask, train=True)
    ref32 = oracle_step(T, seed, x, target, mask, train=True, dtype=torch.float32)
    # forward quantities: strict 1e-4
    errs = {'coords': rel(out.detach().cpu(), ref['coords'].detach()), 'l3': rel(l3.detach().cpu(), ref['l3'].detach()),
            'loss': rel(loss.item(), ref['loss'].item())}
    for p in ('xy', 'zy', 'xz'):
        for t in range(T):
            errs['hm_%s%d' % (p, t)] = rel(getattr(m, p + '_heatmaps')[t].detach().cpu(), ref[p][t].detach())
    sd = m.state_dict()
    for k, v in ref['sd'].items():
        if 'running' in k:
            errs['buf:' + k] = rel(sd[k].cpu(), v)
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == 1
    name = 'train_T%d_B%d' % (T, B)
    report(name, errs)
    # gradients: gated on the reference's own fp32 noise floor
    gpu = OrderedDict((k, p.grad.cpu()) for k, p in m.named_parameters())
    gpu['__dx__'] = xg.grad.cpu()
    r64 = OrderedDict(ref['grads']); r64['__dx__'] = ref['dx']
    r32 = OrderedDict(ref32['grads']); r32['__dx__'] = ref32['dx']

    def same_piece():
        from tests.test_grad_parity_gpu import compare
        sd64 = OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in weights(T, seed, x, True).items())
        g64, _ = piece.oracle_grads(sd64, T, x, target, mask, torch.float64, masks=masks)
        g32, _ = piece.oracle_grads(sd64, T, x, target, mask, torch.float32, masks=masks)
        return compare('fallback_' + name, gpu, g64, g32)

    grad_noise_gate(name, gpu, r64, r32, same_piece)


def test_model_T2_vs_reference_golden(golden_dir):
    """Same run as tools/make_golden.py::gen_model (mixed 2D/3D loss by valid_depth, random mask)."""
    from margipose_amd import dsntnn
    g = np.load(os.path.join(golden_dir, 'model_T2.npz'))
    seed, T, B = int(g['seed']), 2, 2
    x, target, _ = W.seeded_inputs(seed + 1000, B)
    mask = torch.tensor(g['mask'], dtype=torch.float32).cuda()
    m = build(T, seed, x)
    m.eval()
    with torch.no_grad():
        out = m(x.cuda())
        errs = {'coords_eval': rel(out.cpu(), g['coords_eval_f64']),
                'l3_eval': rel(m.forward_3d_losses(out, target.cuda()).cpu(), g['losses3d_eval_f64']),
                'hm_xy_eval': rel(m.xy_heatmaps[-1].cpu().numpy()[:, :, ::4, ::4], g['hm_xy_eval_f64'])}
    m.train()
    out = m(x.cuda())
    l3 = m.forward_3d_losses(out, target.cuda())
    l2 = m.forward_2d_losses(out, target.cuda())
    errs['coords_train'] = rel(out.detach().cpu(), g['coords_train_f64'])
    errs['l3_train'] = rel(l3.detach().cpu(), g['losses3d_train_f64'])
    errs['l2_train'] = rel(l2.detach().cpu(), g['losses2d_train_f64'])
    vd = torch.tensor(g['valid_depth'], dtype=torch.float32).cuda()[:, None]
    loss = dsntnn.average_loss(vd * l3 + (1 - vd) * l2, mask)      # bin/train_3d.py:126-142
    loss.backward()
    errs['loss_mixed'] = rel(loss.item(), g['loss_mixed_f64'])
    keys = [str(k) for k in g['param_keys']]
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.double().norm()) for k in keys])
    typical = float(np.median(g['gnorm_mixed_f64']))
    nz = g['gnorm_mixed_f64'] > 1e-9 * typical            # analytically-zero gradients are excluded (see grad_noise_gate)
    dev_gpu = np.abs(norms - g['gnorm_mixed_f64'])[nz] / g['gnorm_mixed_f64'][nz]
    dev_ref = np.abs(g['gnorm_mixed_f32'] - g['gnorm_mixed_f64'])[nz] / g['gnorm_mixed_f64'][nz]
    print('grad-norm deviation vs reference fp64: ours median %.2e max %.2e | reference fp32 median %.2e max %.2e'
          % (np.median(dev_gpu), dev_gpu.max(), np.median(dev_ref), dev_ref.max()))
    assert np.median(dev_gpu) <= max(TOL, 3 * np.median(dev_ref)) and dev_gpu.max() <= max(TOL, 5 * dev_ref.max())
    assert norms[~nz].max() < 1e-4 * typical
    running = np.concatenate([b.detach().cpu().numpy().flatten() for k, b in m.named_buffers() if 'running' in k])
    errs['running_after'] = rel(running, g['running_after'])
    # reference's own fp32 run vs its fp64 run, for context
    # one SGD step (bin/train_3d.py:186) -> weight norms
    opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
    opt.step()
    wn = np.array([float(p.detach().double().norm()) for p in m.parameters()])
    errs['w_after_sgd'] = float(np.max(np.abs(wn - g['w_after_sgd_norm']) / np.maximum(g['w_after_sgd_norm'], 1e-30)))
    assert errs.pop('w_after_sgd') < 1e-3      # lr * grad noise (see grad_noise_gate), weights themselves O(1)
    report('golden_T2', errs)


def test_second_step_and_grad_accumulation():
    """Two consecutive steps reuse the arenas; .grad accumulates like autograd (no aliasing of the flat buffer)."""
    from margipose_amd import dsntnn
    T, seed, B = 1, 77, 2
    x, target, mask = W.seeded_inputs(seed, B)
    m = build(T, seed, x).train()
    def step():
        out = m(x.cuda())
        loss = dsntnn.average_loss(m.forward_3d_losses(out, target.cuda()), mask.cuda())
        loss.backward()
        return loss.item()
    step()
    g1 = [p.grad.clone() for p in m.parameters()]
    # restore BN running stats influence: grads do not depend on running stats in train mode
    step()
    for a, p in zip(g1, m.parameters()):
        assert torch.allclose(p.grad, 2 * a, rtol=2e-3, atol=1e-7 * float(a.abs().max() + 1e-30) + 1e-12)


def test_full_config_shapes_and_properties():
    """BASELINE cfg2/cfg3 sizes: B=32, T=3, 256x256 -> shapes, normalised heatmaps, finite grads."""
    from margipose_amd import dsntnn
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(12345)
    m = MargiPoseModel(CanonicalSkeletonDesc, 3, True, 'patch8', 'jsd').cuda().train()
    B = 32
    x = torch.randn(B, 3, 256, 256, device='cuda')
    target = torch.rand(B, 17, 3, device='cuda') * 2 - 1
    out = m(x)
    assert out.shape == (B, 17, 3) and m.xy_heatmaps[-1].shape == (B, 17, 32, 32) and len(m.zy_heatmaps) == 3
    for hm in m.xy_heatmaps + m.zy_heatmaps + m.xz_heatmaps:
        assert (hm.flatten(2).sum(-1) - 1).abs().max() < 1e-4
    loss = dsntnn.average_loss(m.forward_3d_losses(out, target), torch.ones(B, 17, device='cuda'))
    loss.backward()
    assert torch.isfinite(loss)
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    # batch-shard consistency (data-parallel invariant): eval-mode outputs of a half batch equal the full batch's half to fp32
    # rounding (a tensor's power-of-two scale follows the largest magnitude in the BATCH, so a sample's roundings depend on its
    # batch mates; this untrained model's peaked heatmaps amplify that to ~2e-5 on a coordinate)
    m.eval()
    with torch.no_grad():
        full = m(x[:8])
        half = m(x[:4])
    assert (full[:4] - half).abs().max() < 1e-4


@pytest.mark.parametrize('size,T', [(384, 1), (512, 1), (128, 2), (768, 1)])
def test_other_input_sizes_vs_oracle(size, T):
    """The model is fully convolutional (384 -> 48x48 heatmaps / mid 24, 512 -> 64x64 / mid 32).  768 -> 96x96 / mid 48
    (round 6): the reference's only size rule is 192 % (S/16) == 0 (models/margipose_model.py:87-97); heatmap rows beyond 4096
    elements run the tail's multi-pass kernels, the stages' wider images the general tile forms of the convolution kernels."""
    from margipose_amd import dsntnn
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    seed, B = 600 + size, 1 if size > 256 else 2
    x, target, mask = W.seeded_inputs(seed, B, size)
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64), x.double(), T)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, 'patch8', 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    m = m.cuda().train()
    out = m(x.cuda())
    F = size // 8
    assert m.xy_heatmaps[-1].shape == (B, 17, F, F)
    l3 = m.forward_3d_losses(out, target.cuda())
    loss = dsntnn.average_loss(l3, mask.cuda())
    loss.backward()
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
    xy, zy, xz = R.inner_forward(sd, x.double(), T, True)
    ref_l3 = R.forward_3d_losses(xy, zy, xz, target.double())
    ref_loss = R.average_loss(ref_l3, mask.double())
    ref_loss.backward()
    errs = {'coords': rel(out.detach().cpu(), R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).detach()),
            'l3': rel(l3.detach().cpu(), ref_l3.detach()), 'hm_zy': rel(m.zy_heatmaps[-1].detach().cpu(), zy[-1].detach())}
    report('size%d' % size, errs)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).cpu())
    gn_ref = float(torch.sqrt(sum((p.grad ** 2).sum() for p in params.values())))
    assert abs(gn - gn_ref) / gn_ref < 2e-3, (gn, gn_ref)      # whole-model gradient norm (see grad_noise_gate for why not 1e-4)


def test_five_stage_model_runs():
    """BASELINE configs[4]: 5-stage model (forward/backward plumbing at 256x256; n_stages is a free ctor arg)."""
    from margipose_amd import dsntnn
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(5)
    m = MargiPoseModel(CanonicalSkeletonDesc, 5, True, 'patch8', 'jsd').cuda().train()
    x = torch.randn(4, 3, 256, 256, device='cuda')
    out = m(x)
    assert len(m.xy_heatmaps) == 5 and out.shape == (4, 17, 3)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, torch.rand(4, 17, 3, device='cuda') * 2 - 1), torch.ones(4, 17, device='cuda'))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_overlap_wgrad_matches_serial():
    """Side-stream weight gradients (the default schedule) give bit-identical gradients to the serial schedule."""
    from margipose_amd import dsntnn
    T, seed, B = 1, 91, 4
    x, target, mask = W.seeded_inputs(seed, B)
    m = build(T, seed, x).train()
    def run(overlap):
        m.inner.engine().overlap_wgrad = overlap
        m.zero_grad(set_to_none=True)
        loss = dsntnn.average_loss(m.forward_3d_losses(m(x.cuda()), target.cuda()), mask.cuda())
        loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in m.parameters()]
    a = run(False)
    b = run(True)
    c = run(True)
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and all(torch.equal(u, v) for u, v in zip(b, c))


@pytest.mark.parametrize('stem,overlap', [('inceptionv4', False), ('patch8', True)])
def test_data_parallel_two_ranks_share_one_gpu(stem, overlap):
    """tools/dp_check.py under torch.distributed.run: 2 ranks on cuda:0 over gloo (RCCL needs distinct devices), 2 stages:
    averaged gradients == mean of the shards' gradients, for this engine's own shard gradients (1e-5) AND for the fp64
    oracle's mean of shards (SURVEY 8e's parity definition); bucket layout; broadcast of parameters and buffers.
    overlap: the side-stream schedule under data parallelism (MPOSE_DP_OVERLAP=1), functionally."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPOSE_DIST_BACKEND='gloo', MPOSE_SINGLE_DEVICE='1')
    if overlap:
        env['MPOSE_DP_OVERLAP'] = '1'
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                          '127.0.0.1', '--master-port', str(29600 + os.getpid() % 300), os.path.join(root, 'tools', 'dp_check.py'), stem],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900).stdout.decode(errors='replace')
    assert 'DP_CHECK_OK' in out, out[-3000:]


def test_data_parallel_rccl_single_rank():
    """tools/dp_nccl_single.py: the `nccl` (RCCL) backend with ONE rank on the box's GPU -- the gradient buckets go through real
    asynchronous ncclAllReduce calls issued from the backward pass with the weight-gradient side stream on (the schedule that
    only ever ran under gloo before); results bit-identical to the plain step.  (MPOSE_DP_GRAPH=1 python tools/dp_nccl_single.py
    additionally captures the step, collectives included, in a HIP graph: informational)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'dp_nccl_single.py')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=600).stdout.decode(errors='replace')
    assert 'DP_NCCL_SINGLE_OK' in out, out[-3000:]
    # (round 5) the same step re-issued from a launch plan: the collectives are host actions between two recorded segments
    assert 'DP_NCCL_PLAN_OK' in out, out[-3000:]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script re-executes itself under
    torch.distributed.run, two ranks share cuda:0 over gloo here (a box has one GPU), and rank 0 prints ONE valid JSON line that
    carries n_gpus, the whole-job images/s and every gradient bucket's all-reduce time -- the line the driver's SCALE run reads.
    `--gpus 1` stays a plain single process (WORLD_SIZE unset, no launcher)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPOSE_DIST_BACKEND='gloo', MPOSE_SINGLE_DEVICE='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    small = ['--steps', '3', '--warmup', '1', '--batch', '4', '--stages', '2', '--stem', 'patch8', '--no-cpu-baseline', '--no-inference']
    lines = {}
    for n in (2, 1):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(n)] + small + (['--no-kernel-timing'] if n == 1 else []),
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        out = r.stdout.decode(errors='replace')
        js = [l for l in out.splitlines() if l.startswith('{')]
        assert r.returncode == 0 and len(js) == 1, (r.returncode, out[-2000:], r.stderr.decode(errors='replace')[-3000:])
        lines[n] = json.loads(js[0])
    two, one = lines[2], lines[1]
    assert two['n_gpus'] == 2 and two['config']['global_batch'] == 8 and two['config']['parallelism'] == 'dp2' and two['scaling'] == 'weak'
    assert abs(two['value'] - 2 * 4 * 3 / (two['ms_per_step'] * 3e-3)) < 1e-6 * two['value']        # whole-job images/s
    bk = two['allreduce_buckets']['buckets']
    assert len(bk) == 3 and all(b['ms'] > 0 and b['bus_GBps'] > 0 for b in bk)                      # stage 1, stage 0, stem
    assert one['n_gpus'] == 1 and 'allreduce_buckets' not in one and one['config']['parallelism'] == 'dp1'


@pytest.mark.skipif(os.environ.get('MPOSE_LONG_TESTS', '0') == '0', reason='eight processes on one GPU: MPOSE_LONG_TESTS=1 (tools/final_check.sh)')
def test_bench_with_eight_ranks_on_one_gpu():
    """Readiness of the first real SCALE run: `python bench.py --gpus 8` as the driver issues it -- eight ranks
    rendezvous on 127.0.0.1, share cuda:0 over gloo (the pool has no multi-GPU node), run the planned data-parallel step with its
    bucketed all-reduces, and rank 0 prints ONE JSON line with n_gpus 8, the whole-job rate and both buckets' timings: no port,
    rendezvous or host-thread issue is left to be found on the 8-GPU node."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPOSE_DIST_BACKEND='gloo', MPOSE_SINGLE_DEVICE='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    # Eight processes on ONE GPU is not what the 8-GPU node runs, and this runtime does not survive it every time: in 10-20 % of the runs one
    # rank dies of HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION inside a torch.zeros() of its set-up, before any kernel of this library has been
    # launched (profiles/r6_eight_rank_flake.txt: stacks taken with HIP_LAUNCH_BLOCKING=1; 0 of 20 runs with four ranks, 0 of 36 with
    # two; eight plain PyTorch processes broadcasting device tensors over gloo die the same way, tools/probe/eight_procs_torch.py).  A run that died of exactly that is repeated; anything else fails the test at once.
    for attempt in range(4):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--batch', '2', '--stages', '1',
                            '--stem', 'patch8', '--no-cpu-baseline', '--no-inference'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
        if r.returncode == 0 or b'HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION' not in r.stderr:
            break
    out = r.stdout.decode(errors='replace')
    js = [l for l in out.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(js) == 1, (r.returncode, out[-2000:], r.stderr.decode(errors='replace')[-3000:])
    d = json.loads(js[0])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 16 and d['config']['parallelism'] == 'dp8' and d['scaling'] == 'weak'
    assert abs(d['value'] - 8 * 2 * 2 / (d['ms_per_step'] * 2e-3)) < 1e-6 * d['value']
    bk = d['allreduce_buckets']['buckets']
    assert len(bk) == 2 and all(b['ms'] > 0 for b in bk)                      # stage 0, stem


def test_unused_stage_gets_zero_grads():
    """A loss on the first stage's heatmaps only: the later stage's parameters receive exactly zero (not the stale
    split-K partial sums of an earlier step), the first stage's gradients match a one-stage run."""
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(7)
    m = MargiPoseModel(CanonicalSkeletonDesc, 2, True, 'patch8', 'jsd').cuda().train()
    x = torch.randn(2, 3, 256, 256, device='cuda')
    tgt = torch.rand(2, 17, 3, device='cuda') * 2 - 1
    out = m(x)                                   # step 1 fills every partial buffer
    m.forward_3d_losses(out, tgt).mean().backward()
    m.zero_grad(set_to_none=True)
    out = m(x)
    (m.xy_heatmaps[0] * torch.linspace(0, 1, 32, device='cuda')).sum().backward()
    for k, p in m.named_parameters():
        if '_hm_cnns.1.' in k:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    assert float(m.inner.xy_hm_cnns[0].down_layers[0].module[0].weight.grad.abs().max()) > 0


@pytest.mark.parametrize('stem,variant', [('patch8', 'default'), ('inceptionv4', 'default'), ('patch8', 'stage_unpack'), ('patch8', 'eval_bn'),
                                          ('patch8', 'frozen')])
def test_every_gradient_element_is_written(stem, variant, monkeypatch):
    """The backward pass does not zero the flat gradient buffer when every stage runs (engine.py, MPOSE_GFLAT_FILL): with the
    buffer poisoned with NaNs first, no parameter's gradient may keep one, and the gradients equal those of a zero-filled run
    bit for bit -- on the default schedule, with the split-K partials unpacked once per stage and the coefficient jobs as their
    own launches ('stage_unpack'), through eval-mode BatchNorm ('eval_bn': other coefficient jobs write the BatchNorm gradients),
    and with a frozen parameter ('frozen': the engine falls back to the zero fill -- ADVICE r5)."""
    import copy
    from margipose_amd import engine as eng_mod
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(11)
    m0 = MargiPoseModel(CanonicalSkeletonDesc, 2, True, stem, 'jsd').cuda().train()
    if variant == 'eval_bn':
        m0.eval()
    if variant == 'frozen':
        next(m0.parameters()).requires_grad_(False)
    m1 = copy.deepcopy(m0)
    x = torch.randn(2, 3, 256, 256, device='cuda')
    tgt = torch.rand(2, 17, 3, device='cuda') * 2 - 1
    grads = []
    for m, mode in ((m0, 1), (m1, 2)):
        monkeypatch.setattr(eng_mod, '_GFLAT_FILL', mode)
        if variant == 'stage_unpack':
            m.inner.engine().inline_unpack = False
            m.inner.engine().fuse_coef = False
        m.forward_3d_losses(m(x), tgt).mean().backward()
        grads.append(dict((k, p.grad.clone()) for k, p in m.named_parameters() if p.grad is not None))
    assert grads[0].keys() == grads[1].keys() and len(grads[1]) > 50
    for k, g in grads[1].items():
        assert bool(torch.isfinite(g).all()), k
        assert torch.equal(g, grads[0][k]), k


@pytest.mark.parametrize('stem', ['patch8', 'inceptionv4'])
def test_uint8_frames_are_normalised_on_device(stem):
    """uint8 RGB frames in -> same result as `ImageSpecs.convert` (to_tensor + (x - mean) / std, reference
    data_specs.py:6-13,38-39) done on the host and fed as float32 (normalisation fused into the first load)."""
    from margipose_amd import dsntnn
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(11)
    m = MargiPoseModel(CanonicalSkeletonDesc, 1, True, stem, 'jsd').cuda().train()
    frames = torch.randint(0, 256, (2, 3, 256, 256), dtype=torch.uint8)
    mean = torch.tensor(m.data_specs.input_specs.mean).view(1, 3, 1, 1)
    std = torch.tensor(m.data_specs.input_specs.stddev).view(1, 3, 1, 1)
    x = ((frames.float() / 255.0 - mean) / std).cuda()
    tgt = torch.rand(2, 17, 3, device='cuda') * 2 - 1
    state = {k: v.clone() for k, v in m.state_dict().items()}
    out_f = m(x)
    loss_f = dsntnn.average_loss(m.forward_3d_losses(out_f, tgt), torch.ones(2, 17, device='cuda'))
    loss_f.backward()
    g_f = m.inner.xy_hm_cnns[0].down_layers[0].module[0].weight.grad.clone()
    m.load_state_dict(state)                     # same BN running statistics for the second pass
    m.zero_grad(set_to_none=True)
    out_u = m(frames.cuda())
    loss_u = dsntnn.average_loss(m.forward_3d_losses(out_u, tgt), torch.ones(2, 17, device='cuda'))
    loss_u.backward()
    g_u = m.inner.xy_hm_cnns[0].down_layers[0].module[0].weight.grad
    assert float((out_u - out_f).abs().max()) < 2e-5, float((out_u - out_f).abs().max())     # (x/255 - mean)/std vs fma form: 1 ulp inputs
    # (inputs that differ in the last bit put a handful of the ~10M ReLU sites of this B=2 step on the other piece: the size of
    #  the resulting gradient change is that of tests/test_grad_parity_gpu.py's free-running comparison, not an arithmetic error)
    assert float((g_u - g_f).norm() / g_f.norm()) < 2e-2


def test_bf16_heatmap_inference_mode():
    """BASELINE configs[1]: eval forward with bf16 heatmap storage + fp32 soft-argmax, against the oracle with the same
    rounding points (heatmaps rounded to bf16 where stored / read by the next combiner, coordinates from the unrounded
    softmax).  Tolerances: coordinates 1e-4 relative (the bar of the fp32 path: rounding happens at identical points, a
    1-ulp disagreement of a single bf16 heatmap value moves the next stage's input by < 1e-6); heatmaps one bf16 ulp
    (2^-8 relative) where the fp32 values straddle a rounding boundary."""
    T, B, seed = 2, 2, 640
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    m = build(T, seed, x, True).eval()
    m.heatmap_dtype = torch.bfloat16
    with torch.no_grad():
        out = m(x.cuda())
    assert out.dtype == torch.float32 and m.xy_heatmaps[0].dtype == torch.bfloat16 and len(m.zy_heatmaps) == T
    sd = weights(T, seed, x, True)
    sd = OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in sd.items())
    un = {}
    xy, zy, xz = R.inner_forward(sd, x.double(), T, False, True, heatmap_dtype=torch.bfloat16, unrounded=un)
    ref = R.heatmaps_to_coords(un['xy'], un['zy'], un['xz'])
    errs = {'coords': rel(out.cpu(), ref)}
    for name, got, want in (('xy', m.xy_heatmaps, xy), ('zy', m.zy_heatmaps, zy), ('xz', m.xz_heatmaps, xz)):
        for t in range(T):
            g, w = got[t].float().cpu().double(), want[t]
            ulp = float(((g - w).abs() / w.abs().clamp_min(1e-30)).max())
            frac_diff = float((g != w).double().mean())
            errs['hm_%s%d_ulp' % (name, t)] = ulp
            assert ulp <= 2.0 ** -7 and frac_diff < 5e-3, (name, t, ulp, frac_diff)
    print(errs)
    assert errs['coords'] < TOL, errs
    # the storage mode is inference-only
    m.train()
    with pytest.raises(Exception, match='inference'):
        m(x.cuda())
    m.heatmap_dtype = torch.float32
    with pytest.raises(Exception):
        m.heatmap_dtype = torch.float16


def _loss_of(m, x, target, mask):
    from margipose_amd import dsntnn
    return dsntnn.average_loss(m.forward_3d_losses(m(x.cuda()), target.cuda()), mask.cuda())


def test_two_forwards_in_flight_then_both_backwards():
    """Plain autograd lets several forwards be alive before their backwards run (an eval pass in between included); the
    engine's BatchNorm arenas belong to one forward at a time, so the pending forward is given a snapshot."""
    T, seed, B = 1, 31, 2
    x1, target, mask = W.seeded_inputs(seed, B)
    x2, _, _ = W.seeded_inputs(seed + 1, B)
    m = build(T, seed, x1).train()

    def alone(x):
        m.zero_grad(set_to_none=True)
        _loss_of(m, x, target, mask).backward()
        return [p.grad.clone() for p in m.parameters()]
    g1, g2 = alone(x1), alone(x2)
    m.zero_grad(set_to_none=True)
    l1 = _loss_of(m, x1, target, mask)
    l2 = _loss_of(m, x2, target, mask)           # overwrites the arenas of the first forward
    with torch.no_grad():
        m.eval(); m(x2.cuda()); m.train()        # ... and so does an inference pass
    l1.backward()
    got1 = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    l2.backward()
    got2 = [p.grad.clone() for p in m.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(g1, got1))
    assert all(torch.equal(a, b) for a, b in zip(g2, got2))


def test_backward_twice_with_retain_graph():
    T, seed, B = 1, 33, 2
    x, target, mask = W.seeded_inputs(seed, B)
    m = build(T, seed, x).train()
    loss = _loss_of(m, x, target, mask)
    loss.backward(retain_graph=True)
    g = [p.grad.clone() for p in m.parameters()]
    loss.backward()
    assert all(torch.equal(p.grad, 2 * a) for a, p in zip(g, m.parameters()))
    with pytest.raises(RuntimeError):
        loss.backward()                          # the graph is gone now, like any autograd graph


def test_parameter_update_between_forward_and_backward_is_an_error():
    T, seed, B = 1, 35, 2
    x, target, mask = W.seeded_inputs(seed, B)
    m = build(T, seed, x).train()
    loss = _loss_of(m, x, target, mask)
    with torch.no_grad():
        m.inner.xy_hm_cnns[0].down_layers[0].module[0].weight.mul_(1.01)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        loss.backward()


@pytest.mark.parametrize('stem', ['patch8', pytest.param('inceptionv4', marks=pytest.mark.skipif(
    os.environ.get('MPOSE_LONG_TESTS', '0') == '0', reason='suite time budget: MPOSE_LONG_TESTS=1 (tools/final_check.sh) runs it'))])
def test_backward_through_eval_mode_batchnorm(stem):
    """model.eval() + loss.backward() (reference bin/eval_3d.py:63 computes losses in eval mode; fine-tuning with frozen
    statistics): running statistics are constants, dx = gamma*invstd*g, and the bias of a convolution in front of a
    BatchNorm gets a real gradient.  patch8: against the fp64 oracle on the GPU's ReLU piece (tests/test_grad_parity_gpu.py);
    inceptionv4 (ReLU / max-pool pieces of the stem are free): against fp64 with the fp32 oracle's own deviation as the bar."""
    from tests import test_grad_parity_gpu as GP
    from margipose_amd import dsntnn
    T, seed, B = 1, 37, 2
    x, target, mask = W.seeded_inputs(seed, B)
    m, sd = GP.build(T, seed, x, stem)
    m.eval()
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    masks = GP.gpu_relu_masks(m.inner.engine(), m.xy_heatmaps[0].grad_fn.ectx)
    loss = dsntnn.average_loss(m.forward_3d_losses(out, target.cuda()), mask.cuda())
    loss.backward()
    gpu = OrderedDict((k, p.grad.detach().cpu()) for k, p in m.named_parameters())
    gpu['__dx__'] = xg.grad.cpu()

    def oracle(dtype, use_masks):
        s = OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())
        params = OrderedDict((k, v.requires_grad_(True)) for k, v in s.items() if v.is_floating_point() and 'running' not in k)
        xr = x.detach().to(dtype).clone().requires_grad_(True)
        R.RELU_MASKS = masks if use_masks else None
        try:
            xy, zy, xz = R.inner_forward(s, xr, T, False)
            l = R.average_loss(R.forward_3d_losses(xy, zy, xz, target.to(dtype)), mask.to(dtype))
            l.backward()
        finally:
            R.RELU_MASKS = None
        g = OrderedDict((k, p.grad) for k, p in params.items())
        g['__dx__'] = xr.grad
        return g, float(l.detach())
    r64, l64 = oracle(torch.float64, True)
    r32, _ = oracle(torch.float32, True)
    assert abs(float(loss.detach()) - l64) <= 1e-4 * abs(l64)
    st = GP.compare('eval_%s' % stem, gpu, r64, r32)
    if stem == 'patch8':
        assert st['gpu_max'] <= 3e-5 and st['gpu_median'] <= 2.0 * st['ref32_median'], st    # (no batch statistics in eval: the fp32 oracle is at 2.6e-6)
    else:
        assert st['gpu_median'] <= max(1e-4, 1.5 * st['ref32_median']) and st['gpu_p99'] <= max(1e-4, 1.5 * st['ref32_p99']), st
        kb = 'inner.in_cnn.7.bias'
        assert rel_l2(gpu[kb], r64[kb]) < 2e-2, rel_l2(gpu[kb], r64[kb])       # the conv bias in front of the last stem BatchNorm
    for k, b in m.named_buffers():               # eval mode must not touch the running statistics
        if 'running' in k:
            assert torch.equal(b.cpu(), sd[k].float()), k


def _build_stem(T, seed, x, stem):
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64, stem=stem), x.double(), T)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    return m.cuda(), sd


def test_default_model_single_frame_vs_oracle():
    """BASELINE configs[0] on the HIP path: the reference's default model (4 stages, InceptionV4 feature extractor,
    models/margipose_model.py:13-22) on ONE 256x256 frame, eval mode (bin/infer_single.py:60-66) -> (1, 17, 3) coordinates."""
    T, seed = 4, 810
    x, target, mask = W.seeded_inputs(seed, 1)
    m, sd = _build_stem(T, seed, x, 'inceptionv4')
    m.eval()
    with torch.no_grad():
        out = m(x.cuda())
        l3 = m.forward_3d_losses(out, target.cuda())
    assert out.shape == (1, 17, 3) and len(m.xy_heatmaps) == 4
    xy, zy, xz = R.inner_forward(sd, x.double(), T, False)
    errs = {'coords': rel(out.cpu(), R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1])),
            'l3': rel(l3.cpu(), R.forward_3d_losses(xy, zy, xz, target.double()))}
    for t in range(T):
        errs['hm_xz%d' % t] = rel(m.xz_heatmaps[t].cpu(), xz[t])
    report('default_T4_inceptionv4_B1', errs)


@pytest.mark.skipif(os.environ.get('MPOSE_LONG_TESTS', '0') == '0', reason='suite time budget (a CPU fp64 pass of five stages at 384 x 384: up to a '
                    'minute on the pool\'s slow hosts): MPOSE_LONG_TESTS=1 (tools/final_check.sh) runs it; the default suite keeps configs[4]\'s '
                    'shape in test_fp16_convolution_mode_vs_oracle[5-384-patch8-1] and 384 x 384 in test_other_input_sizes_vs_oracle')
def test_five_stage_model_at_384_vs_oracle():
    """BASELINE configs[4]'s shape: 5 stages at 384x384 input (48x48 heatmaps, 24x24 mid resolution; the size constraint of
    models/margipose_model.py:87-97), one training step against the fp64 oracle: forward quantities free running, gradients on
    the ReLU piece the GPU took (tests/test_grad_parity_gpu.py's mask control: at B=1 a free-running comparison measures which
    handful of the ~100 M ReLU sites flipped -- 1.5e-4 ... 1e-3 on the median from one build to the next -- not arithmetic)."""
    from margipose_amd import dsntnn
    import tests.test_grad_parity_gpu as G
    T, seed, B, size = 5, 820, 1, 384
    x, target, mask = W.seeded_inputs(seed, B, size)
    m, sd = _build_stem(T, seed, x, 'patch8')
    m.train()
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    assert m.xy_heatmaps[-1].shape == (B, 17, 48, 48) and len(m.zy_heatmaps) == 5
    masks = G.gpu_relu_masks(m.inner.engine(), m.xy_heatmaps[0].grad_fn.ectx)
    l3 = m.forward_3d_losses(out, target.cuda())
    loss = dsntnn.average_loss(l3, mask.cuda())
    loss.backward()
    gpu = OrderedDict((k, p.grad.detach().cpu()) for k, p in m.named_parameters())
    gpu['__dx__'] = xg.grad.cpu()
    with torch.no_grad():                      # forward quantities: the oracle on its own piece
        xy, zy, xz = R.inner_forward(OrderedDict((k, v.double() if v.is_floating_point() else v) for k, v in sd.items()), x.double(), T, True)
        ref_l3 = R.forward_3d_losses(xy, zy, xz, target.double())
    errs = {'coords': rel(out.detach().cpu(), R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).detach()), 'l3': rel(l3.detach().cpu(), ref_l3.detach())}
    for t in range(T):
        errs['hm_zy%d' % t] = rel(m.zy_heatmaps[t].detach().cpu(), zy[t].detach())
    report('T5_384', errs)
    g64, _, g32 = G.oracle_grads_pair(sd, T, x, target, mask, masks=masks)
    st = G.compare('masked_T5_384', gpu, g64, g32)
    assert st['gpu_max'] <= G.MASKED_TOL_CONFIG and st['gpu_median'] <= max(1.5 * st['ref32_median'], 2e-6), st


def _oracle_step(sd, x, target, mask, T, dtype=torch.float64):
    s = OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in s.items() if v.is_floating_point() and 'running' not in k)
    xy, zy, xz = R.inner_forward(s, x.to(dtype), T, True)
    ls = R.forward_3d_losses(xy, zy, xz, target.to(dtype))
    loss = R.average_loss(ls, mask.to(dtype))
    loss.backward()
    return R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).detach(), float(loss.detach()), OrderedDict((k, p.grad) for k, p in params.items())


@pytest.mark.parametrize('T,size,stem,B', [(5, 384, 'patch8', 1), (1, 128, 'inceptionv4', 2), (2, 384, 'inceptionv4', 1),
                                           pytest.param(5, 384, 'inceptionv4', 2, marks=pytest.mark.skipif(
                                               os.environ.get('MPOSE_LONG_TESTS', '0') == '0', reason='suite time budget: the bench leg\'s own '
                                               'combination at full depth runs under MPOSE_LONG_TESTS=1 (tools/final_check.sh); the default suite '
                                               'keeps its two-stage form'))])
def test_fp16_convolution_mode_vs_oracle(T, size, stem, B):
    """model.conv_dtype = torch.float16 -- BASELINE configs[4], "5-stack hourglass at 384x384, fp16 convs with MFMA": every
    convolution of the model (columns AND feature extractor; forward, data- and weight-gradient) multiplies operands rounded to
    fp16 in one MFMA pass with fp32 accumulation (MPOSE_CONV_F16X1); BatchNorm, losses and soft-argmax stay fp32 -- since round 6
    with the regular 128-channel blocks on conv_h2r_k reading the h planes of their producer-written operands (conv_h.hip, X1).  One
    training step at configs[4]'s own shape, a small InceptionV4 case for the feature extractor's convolutions, and (round 6) the
    combination bench.py's configs[4] leg runs -- InceptionV4 at 384 x 384, two stages here, five under MPOSE_LONG_TESTS -- against the fp64
    ORACLE, with the mode's stated tolerance -- not the 1e-4 bar of the fp32 path and not a self-comparison: coordinates 2e-2
    absolute (normalised [-1, 1] units: half a 48x48-heatmap pixel), loss 2 % relative, every gradient tensor of >= 1024
    elements within cosine 0.95 of the oracle's, their median within 0.98, and the whole-model gradient norm within 5 %.  (An
    fp16 operand carries 11 significant bits: 2^-12 relative rounding per element, accumulated over ~60 convolution layers with
    BatchNorm renormalising in between.  Measured: configs[4]'s shape 9e-4 / 5e-6 / worst cosine 0.9956 / norm 0.9992; the
    two-frame InceptionV4 case, whose batch statistics are small-sample, 1.1e-2 / 2.5e-4 / 0.968 / 1.007.)"""
    from margipose_amd import dsntnn
    seed = 840 + T
    x, target, mask = W.seeded_inputs(seed, B, size)
    m, sd = _build_stem(T, seed, x, stem)
    m.train()
    m.conv_dtype = torch.float16
    assert m.conv_dtype == torch.float16
    out = m(x.cuda())
    loss = dsntnn.average_loss(m.forward_3d_losses(out, target.cuda()), mask.cuda())
    loss.backward()
    gpu = OrderedDict((k, p.grad.detach().cpu().double()) for k, p in m.named_parameters())
    coords, ref_loss, g64 = _oracle_step(sd, x, target, mask, T)
    e_c = float((out.detach().cpu().double() - coords).abs().max())
    e_l = abs(float(loss) - ref_loss) / abs(ref_loss)
    big = [k for k in g64 if g64[k].numel() >= 1024 and float(g64[k].norm()) > 0]
    cos = {k: float((gpu[k] * g64[k]).sum() / (gpu[k].norm() * g64[k].norm() + 1e-300)) for k in big}
    worst = min(cos, key=cos.get)
    n_gpu = float(torch.sqrt(sum((v ** 2).sum() for v in gpu.values())))
    n_ref = float(torch.sqrt(sum((v ** 2).sum() for v in g64.values())))
    print('fp16 mode T=%d @%d %s: coords %.2e, loss %.2e, worst cosine %.4f (%s), median cosine %.5f, grad norm ratio %.4f'
          % (T, size, stem, e_c, e_l, cos[worst], worst, float(np.median(list(cos.values()))), n_gpu / n_ref))
    assert e_c < 2e-2 and e_l < 2e-2, (e_c, e_l)
    assert cos[worst] > 0.95 and float(np.median(list(cos.values()))) > 0.98, (worst, cos[worst])
    assert abs(n_gpu / n_ref - 1.0) < 0.05
    assert e_c > 1e-6                        # the mode really is different arithmetic
    m.conv_dtype = torch.float32
    assert m.conv_dtype == torch.float32


@pytest.mark.parametrize('stem', ['patch8', 'inceptionv4'])
def test_step_switches_leave_the_results_bit_identical(stem):
    """The scheduling switches the engine keeps change WHO does a piece of work, not its arithmetic: the split-K
    partials summed right behind each weight-gradient launch or once per stage (Engine.inline_unpack), the last block's residual
    sum + soft-argmax as one launch or two (tail_fuse), the BatchNorm-backward coefficient jobs in the reduction's finishing pass
    or as their own launch (fuse_coef).  Loss, every gradient, and the running statistics must equal the default schedule's bit
    for bit."""
    import copy
    from margipose_amd import dsntnn
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    T, B, seed = 1, 2, 470
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    if stem == 'patch8':
        m0 = build(T, seed, x).train()
    else:
        torch.manual_seed(seed)
        m0 = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd').cuda().train()
    m1 = copy.deepcopy(m0)
    eng = m1.inner.engine()
    eng.inline_unpack = False
    eng.tail_fuse = False
    eng.fuse_coef = False
    res = []
    for m in (m0, m1):
        out = m(x.cuda())
        loss = dsntnn.average_loss(m.forward_3d_losses(out, target.cuda()), mask.cuda())
        loss.backward()
        res.append((out.detach().clone(), float(loss.detach())))
    assert res[0][1] == res[1][1] and torch.equal(res[0][0], res[1][0])
    for (k, a), (_, b) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(a.grad, b.grad), k
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k


def test_rebound_batchnorm_tensors_are_picked_up():
    """The engine bakes the BatchNorm tensors' addresses into device-resident job tables and caches the tensor objects
    (Engine._ensure_arenas).  A buffer or parameter bound to a NEW tensor object after the first forward -- `bn.running_mean = t`,
    load_state_dict(assign=True) -- must rebuild the tables: the eval forward then reads the new statistics (same result as a
    fresh model holding them) and a training forward updates the new buffer, not the orphaned one."""
    import copy
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(5)
    m = MargiPoseModel(CanonicalSkeletonDesc, 1, True, 'patch8', 'jsd').cuda().eval()
    x = torch.randn(2, 3, 256, 256, device='cuda')
    with torch.no_grad():
        out0 = m(x).clone()
    bn = m.inner.xy_hm_cnns[0].down_layers[0].module[1]
    assert isinstance(bn, torch.nn.BatchNorm2d)
    old_mean = bn.running_mean
    bn.running_mean = torch.full_like(old_mean, 0.25)             # a new tensor object at a new address
    bn.weight = torch.nn.Parameter(bn.weight.detach() * 1.5)
    ref = copy.deepcopy(m)                                         # a fresh engine over the same state
    with torch.no_grad():
        out1, out_ref = m(x).clone(), ref(x).clone()
    assert torch.equal(out1, out_ref)
    assert not torch.equal(out1, out0)
    m.train()
    before = bn.running_mean.clone()
    m(x)
    torch.cuda.synchronize()
    assert not torch.equal(bn.running_mean, before)               # the bound buffer is the one updated
    assert float(old_mean.abs().max()) == 0.0                      # the orphan (initial zeros) is left alone
