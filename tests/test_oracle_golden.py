"""Pins the oracle (oracle/*.py) against golden vectors produced by the REAL reference
(tools/make_golden.py, run in the build container) and against the reference's own known-answer
test (reference tests/test_models.py:39-46).  CPU only."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import model_ref as R
from oracle import tail_np as T
from oracle import weights as W

TAIL_CASES = ['f32x2_3d', 'f32x2_3d_masked', 'f32x2_2d', 'f32x2_3d_nopix', 'f48x1_3d', 'f64x1_3d']


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_known_answer(golden_dir):
    g = _load(golden_dir, 'known_answer.npz')
    xy = T.make_gauss(np.array([[[-0.5, 0.5]]], dtype=np.float32), (32, 32), 1)
    zy = T.make_gauss(np.array([[[0.1, 0.0]]], dtype=np.float32), (32, 32), 1)
    xz = T.make_gauss(np.array([[[0.0, 0.2]]], dtype=np.float32), (32, 32), 1)
    xyz = T.heatmaps_to_coords(xy, zy, xz)
    np.testing.assert_allclose(xyz, g['expected'], rtol=1.3e-6, atol=1e-5)   # torch assert_allclose fp32 defaults
    np.testing.assert_allclose(xyz, g['xyz'], rtol=1e-6, atol=1e-7)


def test_state_dict_schema(golden_dir):
    with open(os.path.join(golden_dir, 'state_dict_keys.json')) as f:
        ref = json.load(f)
    import hashlib
    for t in ('1', '2', '3', '4'):
        items = [[k, list(s)] for k, s in W.schema(int(t)).items()]
        assert len(items) == ref[t]['n_keys']
        assert hashlib.sha256(json.dumps(items).encode()).hexdigest() == ref[t]['sha256']
        if ref[t]['items'] is not None:
            assert items == ref[t]['items']


def _tail_inputs(g):
    rng = np.random.default_rng(int(g['seed']))
    B, F = int(g['B']), int(g['F'])
    logits = [rng.standard_normal((B, 17, F, F)) * 4.0 for _ in range(3)]
    target = rng.uniform(-1, 1, (B, 17, 3))
    return logits, target, g['mask']


@pytest.mark.parametrize('case', TAIL_CASES)
def test_tail_numpy_vs_reference(golden_dir, case):
    g = _load(golden_dir, 'tail_%s.npz' % case)
    logits, target, mask = _tail_inputs(g)
    hm = [T.flat_softmax(l) for l in logits]
    pix = bool(int(g['pixelwise']))
    three_d = str(g['loss_kind']) == '3d'
    coords = T.heatmaps_to_coords(*hm)
    np.testing.assert_allclose(coords, g['coords_f64'], rtol=1e-11, atol=1e-13)
    fn = T.forward_3d_losses if three_d else T.forward_2d_losses
    losses = fn([tuple(hm)], target, pixelwise=pix)
    np.testing.assert_allclose(losses, g['losses_f64'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(T.average_loss(losses, mask), g['loss_f64'], rtol=1e-11)
    for p, h in zip(('xy', 'zy', 'xz'), hm):
        np.testing.assert_allclose(h[:, :, ::4, ::4], g['hm_%s_f64' % p], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(T.make_gauss(target[..., :2], hm[0].shape[2:], 1.0)[:, :, ::4, ::4],
                               g['gauss_xy_f64'], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(T.js_reg_losses(hm[0], target[..., :2], 1.0), g['js_xy_f64'], rtol=1e-10)
    # hand-derived backward vs reference autograd
    dloss = mask / max(mask.sum(), 1.0)
    gs = T.stage_loss_grad_wrt_heatmaps(hm[0], hm[1], hm[2], target, dloss, pixelwise=pix, three_d=three_d)
    for p, h, gg in zip(('xy', 'zy', 'xz'), hm, gs):
        d = T.softmax_backward(h, gg)
        ref = g['dlogits_%s_f64' % p]
        np.testing.assert_allclose(d[:, ::4], ref, rtol=1e-8, atol=1e-15 + 1e-9 * np.abs(ref).max())


@pytest.mark.parametrize('case', TAIL_CASES[:3])
def test_tail_torch_restatement_vs_reference(golden_dir, case):
    g = _load(golden_dir, 'tail_%s.npz' % case)
    logits, target, mask = _tail_inputs(g)
    lg = [torch.tensor(l, requires_grad=True) for l in logits]
    hm = [R.flat_softmax(l) for l in lg]
    fn = R.forward_3d_losses if str(g['loss_kind']) == '3d' else R.forward_2d_losses
    losses = fn([hm[0]], [hm[1]], [hm[2]], torch.tensor(target), pixelwise=bool(int(g['pixelwise'])))
    loss = R.average_loss(losses, torch.tensor(mask))
    loss.backward()
    np.testing.assert_allclose(losses.detach().numpy(), g['losses_f64'], rtol=1e-10)
    np.testing.assert_allclose(lg[0].grad.numpy()[:, ::4], g['dlogits_xy_f64'], rtol=1e-8, atol=1e-14)


def test_axis_permutation(golden_dir):
    g = _load(golden_dir, 'axis_permutation.npz')
    for S in (16, 24):
        x = torch.arange(192 * S * S, dtype=torch.float64).view(1, 192, S, S)
        for space in ('xy', 'zy', 'xz'):
            np.testing.assert_array_equal(R.axis_permute(x, space).numpy().astype(np.int32), g['%s_%d' % (space, S)])


@pytest.mark.parametrize('space', ['xy', 'zy', 'xz'])
def test_column_vs_reference(golden_dir, space):
    g = _load(golden_dir, 'column_%s.npz' % space)
    seed = int(g['seed'])
    rng = np.random.default_rng(seed + 1000)
    x_np = rng.standard_normal((2, 128, 32, 32))
    gy_np = rng.standard_normal((2, 17, 32, 32))
    sd = W.column_state_dict('c', seed, torch.float64)
    with torch.no_grad():
        y = R.heatmap_column(sd, 'c', torch.tensor(x_np), space, train=False)
    np.testing.assert_allclose(y.numpy(), g['logits_eval_f64'], rtol=1e-9, atol=1e-10)
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items()
                         if v.is_floating_point() and 'running' not in k)
    x = torch.tensor(x_np, requires_grad=True)
    y = R.heatmap_column(sd, 'c', x, space, train=True)
    y.backward(torch.tensor(gy_np))
    np.testing.assert_allclose(y.detach().numpy(), g['logits_train_f64'], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(x.grad.numpy()[:, ::16], g['dx_f64'], rtol=1e-8, atol=1e-10)
    keys = [str(k) for k in g['param_keys']]
    norms = np.array([float(params['c.' + k].grad.norm()) for k in keys])
    np.testing.assert_allclose(norms, g['gnorm_f64'], rtol=1e-8)
    running = np.concatenate([v.numpy().flatten() for k, v in sd.items() if 'running' in k])
    np.testing.assert_allclose(running, g['running'], rtol=1e-10)
    # how far the reference's OWN fp32 run is from its fp64 run (context for the 1e-4 gate)
    ref32 = np.abs(g['gnorm_f32'] - g['gnorm_f64']) / g['gnorm_f64']
    assert ref32.max() < 1e-3


def test_model_T2_vs_reference(golden_dir):
    g = _load(golden_dir, 'model_T2.npz')
    seed, T_, B = int(g['seed']), 2, 2
    sd = W.make_state_dict(T_, seed, torch.float64)
    x, target, _ = W.seeded_inputs(seed + 1000, B, dtype=torch.float64)
    mask = torch.tensor(g['mask'])
    R.calibrate_running_stats(sd, x, T_)
    with torch.no_grad():
        xy, zy, xz = R.inner_forward(sd, x, T_, train=False)
        np.testing.assert_allclose(R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).numpy(), g['coords_eval_f64'],
                                   rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(R.forward_3d_losses(xy, zy, xz, target).numpy(), g['losses3d_eval_f64'], rtol=1e-8)
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items()
                         if v.is_floating_point() and 'running' not in k)
    xy, zy, xz = R.inner_forward(sd, x, T_, train=True)
    np.testing.assert_allclose(R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).detach().numpy(),
                               g['coords_train_f64'], rtol=1e-8, atol=1e-10)
    l3 = R.forward_3d_losses(xy, zy, xz, target)
    l2 = R.forward_2d_losses(xy, zy, xz, target)
    np.testing.assert_allclose(l3.detach().numpy(), g['losses3d_train_f64'], rtol=1e-8)
    np.testing.assert_allclose(l2.detach().numpy(), g['losses2d_train_f64'], rtol=1e-8)
    vd = torch.tensor(g['valid_depth'])[:, None]
    loss = R.average_loss(vd * l3 + (1 - vd) * l2, mask)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss_mixed_f64'], rtol=1e-9)
    keys = [str(k) for k in g['param_keys']]
    norms = np.array([float(params[k].grad.norm()) for k in keys])
    np.testing.assert_allclose(norms, g['gnorm_mixed_f64'], rtol=1e-7, atol=1e-14)
    heads = np.stack([np.pad(params[k].grad.flatten()[:8].numpy(), (0, max(0, 8 - params[k].numel()))) for k in keys])
    np.testing.assert_allclose(heads, g['ghead_mixed_f64'], rtol=1e-6, atol=1e-12)
    running = np.concatenate([v.numpy().flatten() for k, v in sd.items() if 'running' in k])
    np.testing.assert_allclose(running, g['running_after'], rtol=1e-10)


def test_frames_normalisation_fixture(golden_dir):
    """tests/golden/frames_u8.npz (reference data_specs.normalize_pixels on to_tensor'd uint8 frames) == (x/255 - mean)/std,
    the formula margipose_amd's uint8 entry implements on the device (mpose_frames_u8, mpose_im2col_k3s2)."""
    g = _load(golden_dir, 'frames_u8.npz')
    x = g['frames'].astype(np.float64) / 255.0
    want = (x - g['mean'].reshape(1, 3, 1, 1)) / g['stddev'].reshape(1, 3, 1, 1)
    assert np.abs(want - g['expected_f64']).max() < 1e-12
    assert np.abs(want - g['expected_f32']).max() < 5e-7


# ---------------------------------------------------------------------------------------------------------------
# ChatterboxModel's two dilated heads (reference models/chatterbox_model.py:87-221): oracle/chatterbox_ref.py against
# the imported reference class (tools/make_golden_chatterbox.py)
# ---------------------------------------------------------------------------------------------------------------
def test_chatterbox_schema(golden_dir):
    with open(os.path.join(golden_dir, 'chatterbox_keys.json')) as f:
        ref = json.load(f)
    for tag, sw in (('w', True), ('h', False)):
        assert [[k, list(s)] for k, s in W.chatterbox_cnn_entries('', sw)] == ref[tag]
    full = W.chatterbox_schema()
    for head, tag in (('zy_hm_cnn.', 'w'), ('xz_hm_cnn.', 'h')):      # (:241-242: zy shrinks the width, xz the height)
        assert [[k[len(head):], list(s)] for k, s in full.items() if k.startswith(head)] == ref[tag]


@pytest.mark.parametrize('tag', ['w', 'h'])
def test_chatterbox_cnn_vs_reference(golden_dir, tag):
    from oracle import chatterbox_ref as C
    g = _load(golden_dir, 'chatterbox_cnn.npz')
    with open(os.path.join(golden_dir, 'chatterbox_keys.json')) as f:
        pnames = json.load(f)['params_' + tag]
    sw = tag == 'w'
    seed_w, seed_x = (int(v) for v in g['seeds'])
    rng = np.random.default_rng(seed_x)
    x = torch.from_numpy(rng.standard_normal((1, 128, 32, 32))).float().requires_grad_(True)
    probe = torch.from_numpy(rng.standard_normal((1, 17, 32, 32))).float()
    sd = W.fill_like(OrderedDict(W.chatterbox_cnn_entries('', sw)), seed_w)
    with torch.no_grad():
        y_eval = C.chatterbox_cnn(sd, '', x, sw, False)
    np.testing.assert_allclose(y_eval.numpy(), g['eval_out_' + tag], rtol=1e-4, atol=1e-4)
    for n in pnames:
        sd[n].requires_grad_(True)
    y = C.chatterbox_cnn(sd, '', x, sw, True)
    (y * probe).sum().backward()
    # same ATen kernels on the same machine; the tolerance leaves room for a different summation order in another build
    np.testing.assert_allclose(y.detach().numpy(), g['train_out_' + tag], rtol=1e-4, atol=1e-4)
    ref_dx = g['train_dx_' + tag]
    assert np.abs(x.grad.numpy() - ref_dx).max() <= 1e-4 * np.abs(ref_dx).max()
    gn = np.array([float(sd[n].grad.double().norm()) for n in pnames])
    np.testing.assert_allclose(gn, g['train_gnorm_' + tag], rtol=1e-3)
    np.testing.assert_allclose(sd['down_convs.5.running_mean'].numpy(), g['running_mean_k8_' + tag], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd['down_convs.5.running_var'].numpy(), g['running_var_k8_' + tag], rtol=1e-4, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# Image feature extractors (reference models/margipose_model.py:103-139): the reference's REAL
# make_image_feature_extractor run over stand-in third-party constructors (tools/make_golden_stems.py).  Pins the slice
# features[0:7], the Conv2d(384,128,1)+BN+ReLU head, the padding rewrite, the ResNet slice and its 1x1-head rule, and the
# state-dict schema; the third-party layer definitions themselves stay unpinned (their source is not in the image).
# ---------------------------------------------------------------------------------------------------------------
STEMS = ['inceptionv4', 'resnet18', 'resnet34', 'resnet50']


@pytest.mark.parametrize('stem', STEMS)
def test_stem_schema_vs_reference_assembled_model(golden_dir, stem):
    import hashlib
    with open(os.path.join(golden_dir, 'stem_keys.json')) as f:
        ref = json.load(f)
    items = [[k, list(s)] for k, s in W.schema(1, stem=stem).items()]
    assert len(items) == ref[stem]['n_keys']
    assert hashlib.sha256(json.dumps(items).encode()).hexdigest() == ref[stem]['sha256']
    assert [it for it in items if it[0].startswith('inner.in_cnn.')] == ref[stem]['in_cnn_items']
    # the product model carries the same keys and shapes (a reference checkpoint loads strictly) and the reference's error text
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    m = MargiPoseModel(CanonicalSkeletonDesc, 1, True, stem, 'jsd')
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == items
    assert sum(p.numel() for p in m.parameters()) == ref[stem]['n_params']
    with pytest.raises(Exception) as e:
        MargiPoseModel(CanonicalSkeletonDesc, 1, True, 'vgg16', 'jsd')
    assert str(e.value) == ref['unknown_name_message']
    # the padding rewrite (:111-117) touches Conv2d and MaxPool2d only: every one ends at kernel // 2, the AvgPool2d keeps its own
    pads = dict((k, tuple(v)) for k, v in ref[stem]['paddings_after_rewrite'])
    if stem == 'inceptionv4':
        assert pads['0.conv'] == (1, 1) and pads['3.maxpool'] == (1, 1) and pads['4.branch1.1.conv'] == (0, 3)
        assert pads['4.branch1.2.conv'] == (3, 0) and pads['4.branch0.0.conv'] == (0, 0) and pads['6.branch3.0'] == (1, 1) and pads['7'] == (0, 0)
    else:
        assert pads['0'] == (3, 3) and pads['3'] == (1, 1)


@pytest.mark.parametrize('stem', STEMS)
def test_stem_model_vs_reference(golden_dir, stem):
    """oracle/model_ref.py (functional restatement) against the reference's MargiPoseModel with the reference-assembled stem:
    the stem's output alone (eval and train mode), then the T=1 model's coordinates, losses, gradients and running statistics."""
    g = _load(golden_dir, 'stem_%s.npz' % stem)
    seed, T_, B = int(g['seed']), 1, 2
    sd = W.make_state_dict(T_, seed, torch.float64, stem=stem)
    x, target, _ = W.seeded_inputs(seed + 1000, B, dtype=torch.float64)
    mask = torch.tensor(g['mask'])
    R.calibrate_running_stats(sd, x, T_)
    with torch.no_grad():
        feat = R.stem_forward(sd, x, False)
        np.testing.assert_allclose(feat.numpy()[:, ::4], g['feat_eval_f64'], rtol=2e-6, atol=2e-7)       # (stored as fp32)
        xy, zy, xz = R.inner_forward(sd, x, T_, train=False)
        np.testing.assert_allclose(R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).numpy(), g['coords_eval_f64'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(xy[-1].numpy()[:, :, ::4, ::4], g['hm_xy_eval_f64'], rtol=1e-7, atol=1e-14)
        np.testing.assert_allclose(R.forward_3d_losses(xy, zy, xz, target).numpy(), g['losses3d_eval_f64'], rtol=1e-8)
        sd_tr = OrderedDict((k, v.clone()) for k, v in sd.items())
        np.testing.assert_allclose(R.stem_forward(sd_tr, x, True).numpy()[:, ::4], g['feat_train_f64'], rtol=2e-6, atol=2e-7)
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
    xr = x.clone().requires_grad_(True)
    xy, zy, xz = R.inner_forward(sd, xr, T_, train=True)
    np.testing.assert_allclose(R.heatmaps_to_coords(xy[-1], zy[-1], xz[-1]).detach().numpy(), g['coords_train_f64'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(xz[-1].detach().numpy()[:, :, ::4, ::4], g['hm_xz_train_f64'], rtol=1e-7, atol=1e-14)
    l3 = R.forward_3d_losses(xy, zy, xz, target)
    np.testing.assert_allclose(l3.detach().numpy(), g['losses3d_train_f64'], rtol=1e-8)
    loss = R.average_loss(l3, mask)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g['loss_f64'], rtol=1e-9)
    np.testing.assert_allclose(xr.grad.numpy()[:, :, ::8, ::8], g['dx_f64'], rtol=1e-6, atol=1e-13)
    keys = [str(k) for k in g['param_keys']]
    norms = np.array([float(params[k].grad.norm()) for k in keys])
    np.testing.assert_allclose(norms, g['gnorm_f64'], rtol=1e-7, atol=1e-13)
    heads = np.stack([np.pad(params[k].grad.flatten()[:8].numpy(), (0, max(0, 8 - params[k].numel()))) for k in keys])
    np.testing.assert_allclose(heads, g['ghead_f64'], rtol=1e-6, atol=1e-12)
    running = np.concatenate([v.numpy().flatten() for k, v in sd.items() if 'running' in k])
    np.testing.assert_allclose(running, g['running_after'], rtol=1e-10)
