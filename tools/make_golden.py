#!/usr/bin/env python
"""Generate tests/golden/* by running the REAL reference (/root/reference, imported in place,
read-only) on seeded inputs.  Runs only in the build container; the GPU box never sees the
reference.  Fixtures are data only: seeds + the reference's outputs.

    python tools/make_golden.py            # writes tests/golden/*.npz, *.json

Inputs and weights are regenerated on the test side from the same numpy seeds
(oracle/weights.py), so only outputs are stored.
"""
import hashlib
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _reference_import import import_reference   # noqa: E402
from oracle import weights as W                   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
dsntnn, mm, Skel = import_reference()
torch.set_num_threads(8)


def patch8():
    return nn.Sequential(nn.Conv2d(3, 128, 8, stride=8, bias=False), nn.BatchNorm2d(128),
                         nn.ReLU(inplace=True))


mm.make_image_feature_extractor = lambda name: patch8()


def npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print('wrote', name, '%.1f KB' % (os.path.getsize(path) / 1024))


def t2n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ known answer + state_dict keys
def gen_keys():
    out = {}
    for T in (1, 2, 3, 4):
        m = mm.MargiPoseModel(Skel, T, True, 'patch8', 'jsd')
        sd = m.state_dict()
        items = [[k, list(v.shape)] for k, v in sd.items()]
        n_params = sum(p.numel() for p in m.parameters())
        digest = hashlib.sha256(json.dumps(items).encode()).hexdigest()
        out[str(T)] = {'n_keys': len(items), 'n_params': n_params, 'sha256': digest,
                       'items': items if T <= 2 else None}
        mine = W.schema(T)
        assert [[k, list(s)] for k, s in mine.items()] == items, 'oracle schema mismatch T=%d' % T
    with open(os.path.join(OUT, 'state_dict_keys.json'), 'w') as f:
        json.dump(out, f)
    print('wrote state_dict_keys.json')


# ------------------------------------------------------------------ tail
def tail_case(name, seed, B, F, mask_kind, loss_kind, pixelwise):
    rng = np.random.default_rng(seed)
    logits = [rng.standard_normal((B, 17, F, F)) * 4.0 for _ in range(3)]
    target = rng.uniform(-1, 1, (B, 17, 3))
    mask = np.ones((B, 17)) if mask_kind == 'ones' else (rng.uniform(0, 1, (B, 17)) > 0.3).astype(np.float64)
    res = {}
    for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        lg = [torch.tensor(l, dtype=dt, requires_grad=True) for l in logits]
        hm = [dsntnn.flat_softmax(l) for l in lg]
        model = mm.MargiPoseModel.__new__(mm.MargiPoseModel)
        nn.Module.__init__(model)
        model.pixelwise_loss = 'jsd' if pixelwise else None
        model.xy_heatmaps, model.zy_heatmaps, model.xz_heatmaps = [hm[0]], [hm[1]], [hm[2]]
        coords = mm.MargiPoseModel.heatmaps_to_coords(*hm)
        tgt = torch.tensor(target, dtype=dt)
        if loss_kind == '3d':
            losses = model.forward_3d_losses(coords, tgt)
        else:
            losses = model.forward_2d_losses(coords, tgt)
        loss = dsntnn.average_loss(losses, torch.tensor(mask, dtype=dt))
        loss.backward()
        res['coords_' + tag] = t2n(coords)
        res['losses_' + tag] = t2n(losses)
        res['loss_' + tag] = t2n(loss)
        for p, l in zip(('xy', 'zy', 'xz'), lg):
            g = t2n(l.grad) if l.grad is not None else np.zeros_like(logits[0])
            res['dlogits_%s_%s' % (p, tag)] = g[:, ::4].astype(np.float32) if tag == 'f32' else g[:, ::4]
        for p, h in zip(('xy', 'zy', 'xz'), hm):
            res['hm_%s_%s' % (p, tag)] = t2n(h)[:, :, ::4, ::4]
        if tag == 'f64':
            gauss = dsntnn.make_gauss(tgt[..., :2], (F, F), 1.0)
            res['gauss_xy_f64'] = t2n(gauss)[:, :, ::4, ::4]
            res['js_xy_f64'] = t2n(dsntnn.js_reg_losses(hm[0], tgt[..., :2], 1.0))
    npz('tail_%s.npz' % name, seed=seed, B=B, F=F, mask=mask, loss_kind=loss_kind,
        pixelwise=int(pixelwise), **res)


def gen_tail():
    tail_case('f32x2_3d', 101, 2, 32, 'ones', '3d', True)
    tail_case('f32x2_3d_masked', 102, 2, 32, 'rand', '3d', True)
    tail_case('f32x2_2d', 103, 2, 32, 'rand', '2d', True)
    tail_case('f32x2_3d_nopix', 104, 2, 32, 'ones', '3d', False)
    tail_case('f48x1_3d', 105, 1, 48, 'ones', '3d', True)
    tail_case('f64x1_3d', 106, 1, 64, 'ones', '3d', True)
    # reference's own known-answer test (tests/test_models.py:39-46)
    xy = dsntnn.make_gauss(torch.Tensor([[[-0.5, 0.5]]]), (32, 32), 1, normalize=True)
    zy = dsntnn.make_gauss(torch.Tensor([[[0.1, 0]]]), (32, 32), 1, normalize=True)
    xz = dsntnn.make_gauss(torch.Tensor([[[0, 0.2]]]), (32, 32), 1, normalize=True)
    xyz = mm.MargiPoseModel.heatmaps_to_coords(xy, zy, xz)
    npz('known_answer.npz', xyz=t2n(xyz), expected=np.array([[[-0.5, 0.5, 0.15]]], dtype=np.float32))


# ------------------------------------------------------------------ axis permutation
def gen_perm():
    res = {}
    for S in (16, 24):
        col = mm.HeatmapColumn.__new__(mm.HeatmapColumn)
        nn.Module.__init__(col)
        col.down_layers = nn.Identity()
        col.up_layers = nn.Identity()
        x = torch.arange(192 * S * S, dtype=torch.float64).view(1, 192, S, S)
        for space in ('xy', 'zy', 'xz'):
            col.heatmap_space = space
            res['%s_%d' % (space, S)] = t2n(col(x)).astype(np.int32)
    npz('axis_permutation.npz', **res)


# ------------------------------------------------------------------ one column
def grads_summary(named_params):
    norms, heads = [], []
    for k, p in named_params:
        g = p.grad.detach().double().flatten()
        norms.append(float(g.norm()))
        h = torch.zeros(8, dtype=torch.float64)
        h[:min(8, g.numel())] = g[:8]
        heads.append(h.numpy())
    return np.array(norms), np.stack(heads)


def gen_column():
    for space, seed in (('xy', 201), ('zy', 202), ('xz', 203)):
        res = {}
        rng = np.random.default_rng(seed + 1000)
        x_np = rng.standard_normal((2, 128, 32, 32))
        gy_np = rng.standard_normal((2, 17, 32, 32))
        for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
            col = mm.HeatmapColumn(17, heatmap_space=space)
            sd = W.column_state_dict('c', seed, dt)
            col = col.to(dt)
            col.load_state_dict(OrderedDict((k[2:], v) for k, v in sd.items()), strict=True)
            # eval-mode forward (running stats)
            col.eval()
            x = torch.tensor(x_np, dtype=dt)
            with torch.no_grad():
                res['logits_eval_' + tag] = t2n(col(x)) if tag == 'f64' else t2n(col(x))[:, ::4]
            # train-mode forward + backward
            col.train()
            x = torch.tensor(x_np, dtype=dt, requires_grad=True)
            y = col(x)
            y.backward(torch.tensor(gy_np, dtype=dt))
            res['logits_train_' + tag] = t2n(y) if tag == 'f64' else t2n(y)[:, ::4]
            res['dx_' + tag] = t2n(x.grad)[:, ::16]
            n, h = grads_summary(list(col.named_parameters()))
            res['gnorm_' + tag] = n
            res['ghead_' + tag] = h
            if tag == 'f64':
                res['param_keys'] = np.array([k for k, _ in col.named_parameters()])
                res['running'] = np.concatenate([t2n(b).flatten() for k, b in col.named_buffers()
                                                 if not k.endswith('num_batches_tracked')])
        npz('column_%s.npz' % space, seed=seed, **res)


# ------------------------------------------------------------------ full model, T=2
def gen_model():
    T, seed, B = 2, 301, 2
    res = {}
    x_t, target_t, _ = W.seeded_inputs(seed + 1000, B, dtype=torch.float64)
    rng = np.random.default_rng(seed + 2000)
    mask_np = (rng.uniform(0, 1, (B, 17)) > 0.2).astype(np.float64)
    valid_depth = np.array([1.0, 0.0])
    for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        model = mm.MargiPoseModel(Skel, T, True, 'patch8', 'jsd').to(dt)
        model.load_state_dict(W.make_state_dict(T, seed, dt), strict=True)
        x, target, mask = x_t.to(dt), target_t.to(dt), torch.tensor(mask_np, dtype=dt)
        # calibrate the synthetic running stats (momentum 1, one pass) so eval-mode logits are well conditioned
        bns = [mod for mod in model.modules() if isinstance(mod, nn.BatchNorm2d)]
        for mod in bns:
            mod.momentum = 1.0
        model.train()
        with torch.no_grad():
            model(x)
        for mod in bns:
            mod.momentum = 0.1
            mod.num_batches_tracked.zero_()
        model.eval()
        with torch.no_grad():
            res['coords_eval_' + tag] = t2n(model(x))
            res['hm_xy_eval_' + tag] = t2n(model.xy_heatmaps[-1])[:, :, ::4, ::4]
            res['losses3d_eval_' + tag] = t2n(model.forward_3d_losses(None, target))
        model.train()
        out = model(x)
        l3 = model.forward_3d_losses(out, target)
        l2 = model.forward_2d_losses(out, target)
        res['coords_train_' + tag] = t2n(out)
        res['losses3d_train_' + tag] = t2n(l3)
        res['losses2d_train_' + tag] = t2n(l2)
        for p in ('xy', 'zy', 'xz'):
            for t in range(T):
                res['hm_%s%d_train_%s' % (p, t, tag)] = t2n(getattr(model, p + '_heatmaps')[t])[:, :, ::4, ::4]
        # bin/train_3d.py:126-142 -- mixed 2D/3D selection by valid_depth, then masked mean
        vd = torch.tensor(valid_depth, dtype=dt)[:, None]
        losses = vd * l3 + (1 - vd) * l2
        loss = dsntnn.average_loss(losses, mask)
        model.zero_grad()
        loss.backward()
        res['loss_mixed_' + tag] = t2n(loss)
        n, h = grads_summary(list(model.named_parameters()))
        res['gnorm_mixed_' + tag] = n
        res['ghead_mixed_' + tag] = h
        if tag == 'f64':
            res['param_keys'] = np.array([k for k, _ in model.named_parameters()])
            res['running_after'] = np.concatenate(
                [t2n(b).flatten() for k, b in model.named_buffers() if not k.endswith('num_batches_tracked')])
            res['nbt_after'] = np.array([int(b) for k, b in model.named_buffers()
                                         if k.endswith('num_batches_tracked')])
            # one SGD step (bin/train_3d.py:186, optimiser from :338-340) -> weight checksum
            opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
            opt.step()
            res['w_after_sgd_norm'] = np.array([float(p.detach().norm()) for p in model.parameters()])
    res['mask'] = mask_np
    res['valid_depth'] = valid_depth
    npz('model_T2.npz', seed=seed, **res)


# ------------------------------------------------------------------ 1cycle schedule + a short training curve
def gen_train_curve():
    from margipose.hyperparam_scheduler import make_1cycle
    T, seed, B, n_steps = 1, 701, 2, 5
    x_t, target_t, _ = W.seeded_inputs(seed + 1000, B, dtype=torch.float64)
    res = {}
    for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        model = mm.MargiPoseModel(Skel, T, True, 'patch8', 'jsd').to(dt)
        model.load_state_dict(W.make_state_dict(T, seed, dt), strict=True)
        model.train()
        opt = torch.optim.SGD(model.parameters(), lr=0)
        sched = make_1cycle(opt, 10, lr_max=0.05, momentum=0.9)
        x, target = x_t.to(dt), target_t.to(dt)
        mask = torch.ones(B, 17, dtype=dt)
        losses, lrs, moms = [], [], []
        for it in range(n_steps):
            sched.batch_step()
            lrs.append(opt.param_groups[0]['lr']); moms.append(opt.param_groups[0]['momentum'])
            out = model(x)
            loss = dsntnn.average_loss(model.forward_3d_losses(out, target), mask)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(float(loss))
        res['losses_' + tag] = np.array(losses)
        res['final_coords_' + tag] = t2n(out)
    res['lr'] = np.array(lrs); res['momentum'] = np.array(moms)
    # the schedule over a longer horizon (exact host arithmetic)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0)
    sched = make_1cycle(opt, 1000, lr_max=1.0, momentum=0.9)
    lr_long, mom_long = [], []
    for it in range(1000):
        sched.batch_step(); lr_long.append(opt.param_groups[0]['lr']); mom_long.append(opt.param_groups[0]['momentum'])
    res['lr_1000'] = np.array(lr_long); res['momentum_1000'] = np.array(mom_long)
    npz('train_curve.npz', seed=seed, **res)


# ------------------------------------------------------------------ input normalisation (data_specs.py)
def gen_frames():
    """`ImageSpecs.convert` = normalize_pixels(to_tensor(img), mean, std) (data_specs.py:6-13,38-39).  torchvision is absent
    from this image, so `to_tensor` (published behaviour for uint8 HWC images: CHW float32, divided by 255) is applied here and
    the REFERENCE's own normalize_pixels + the ImageSpecs constants it is called with (margipose_model.py:207) do the rest."""
    from margipose import data_specs as ds
    rng = np.random.default_rng(901)
    frames = rng.integers(0, 256, (2, 3, 16, 16), dtype=np.uint8)
    specs = ds.ImageSpecs(256, mean=ds.ImageSpecs.IMAGENET_MEAN, stddev=ds.ImageSpecs.IMAGENET_STDDEV)
    out32 = np.stack([t2n(ds.normalize_pixels(torch.from_numpy(f).float().div(255), specs.mean, specs.stddev)) for f in frames])
    out64 = np.stack([t2n(ds.normalize_pixels(torch.from_numpy(f).double().div(255), specs.mean, specs.stddev)) for f in frames])
    npz('frames_u8.npz', frames=frames, mean=np.array(specs.mean), stddev=np.array(specs.stddev), expected_f32=out32, expected_f64=out64)


# ------------------------------------------------------------------ make_gauss in 1, 2, 3 dimensions + d js / d mu
def gen_gauss_nd():
    rng = np.random.default_rng(911)
    res = {}
    for tag, size in (('1d', (5,)), ('2d', (6, 10)), ('3d', (4, 6, 8))):
        mu = rng.uniform(-1, 1, (2, 3, len(size)))
        res['mu_' + tag] = mu
        res['size_' + tag] = np.array(size)
        for norm in (True, False):
            res['gauss_%s_%d' % (tag, int(norm))] = t2n(dsntnn.make_gauss(torch.tensor(mu), size, 1.3, normalize=norm))
    mu = torch.tensor(rng.uniform(-1, 1, (2, 3, 2)), requires_grad=True)
    hm = torch.softmax(torch.tensor(rng.standard_normal((2, 3, 64))), -1).view(2, 3, 8, 8)
    js = dsntnn.js_reg_losses(hm, mu, 1.0)
    g, = torch.autograd.grad(js.sum(), mu)
    res.update(js_mu=t2n(mu), js_hm=t2n(hm), js_val=t2n(js), js_dmu=t2n(g))
    npz('gauss_nd.npz', **res)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['keys', 'tail', 'perm', 'column', 'model', 'curve', 'frames', 'gauss']
    for w in which:
        {'keys': gen_keys, 'tail': gen_tail, 'perm': gen_perm, 'column': gen_column, 'model': gen_model, 'curve': gen_train_curve, 'frames': gen_frames, 'gauss': gen_gauss_nd}[w]()
