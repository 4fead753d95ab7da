"""Reference-generated fixtures consumed DIRECTLY by the HIP path (tools/make_golden.py ran the imported reference):

  * tests/golden/column_{xy,zy,xz}.npz -- one HeatmapColumn (reference models/margipose_model.py:43-100) on a given feature
    tensor, eval and train mode: the engine is fed the fixture's features (Engine.forward(features=...)) and its heatmaps are
    compared with flat_softmax of the reference's logits;
  * tests/golden/axis_permutation.npz -- the column's axis permutation (:91-99) through mpose_axis_permute;
  * tests/golden/frames_u8.npz -- `ImageSpecs.convert` normalisation (data_specs.py:6-13,38-39) of uint8 frames through
    mpose_frames_u8 and through the InceptionV4 stem's fused first-layer gather (mpose_im2col_k3s2);
  * tests/golden/chatterbox_cnn.npz (tools/make_golden_chatterbox.py) -- the reference's _ChatterboxCnn
    (models/chatterbox_model.py:87-221), both orientations, on non-negative features: ChatterboxModel's zy / xz heads are fed the
    fixture's features (Engine.graph_forward(features=...)) and their logits compared with the reference's."""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
PLANES = ('xy', 'zy', 'xz')


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize('space', PLANES)
def test_column_fixture_through_the_engine(golden_dir, space):
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    g = np.load(os.path.join(golden_dir, 'column_%s.npz' % space), allow_pickle=True)
    seed = int(g['seed'])
    c = PLANES.index(space)
    x = torch.from_numpy(np.random.default_rng(seed + 1000).standard_normal((2, 128, 32, 32))).float().cuda()
    m = MargiPoseModel(CanonicalSkeletonDesc, 1, True, 'patch8', 'jsd')
    col_sd = W.column_state_dict('c', seed, torch.float32)
    sd = m.state_dict()
    for k, v in col_sd.items():                      # the fixture's weights go into the column of its own space
        sd['inner.%s_hm_cnns.0.%s' % (space, k[2:])] = v
    m.load_state_dict(sd)
    m = m.cuda()
    eng = m.inner.engine()
    for train, key, tol in ((False, 'logits_eval_f64', 1e-4), (True, 'logits_train_f64', 1e-4)):
        with torch.no_grad():
            hms, _, _ = eng.forward(None, train, False, features=x)
        want = torch.softmax(torch.from_numpy(g[key]).flatten(2), -1).view(2, 17, 32, 32)
        got = hms[c][0].cpu().double()
        assert rel(got, want) < tol, (space, train, rel(got, want))
    # train mode updated the running statistics exactly once, with the reference's values
    running = np.concatenate([b.detach().cpu().numpy().flatten() for k, b in getattr(m.inner, space + '_hm_cnns')[0].named_buffers()
                              if 'running' in k])
    assert rel(running, g['running']) < 1e-5


def test_axis_permutation_fixture(golden_dir):
    from margipose_amd import _lib
    g = np.load(os.path.join(golden_dir, 'axis_permutation.npz'))
    L = _lib.lib()
    for S in (16, 24):
        x = torch.arange(192 * S * S, dtype=torch.float32).view(1, 192, S, S).permute(0, 2, 3, 1).contiguous().cuda()     # NHWC
        ins = [x, x.clone(), x.clone()]
        outs = [torch.empty_like(x) for _ in range(3)]
        spaces = (ctypes.c_int * 3)(0, 1, 2)
        _lib.check(L.mpose_axis_permute(_lib.ptr_array(ins), _lib.ptr_array(outs), spaces, 3, 1, S, 192, _lib.stream_ptr()), 'permute')
        torch.cuda.synchronize()
        for c, space in enumerate(PLANES):
            got = (x if space == 'xy' else outs[c]).permute(0, 3, 1, 2).cpu().numpy().astype(np.int64)
            assert np.array_equal(got, g['%s_%d' % (space, S)].astype(np.int64)), (space, S)


def test_uint8_frames_against_the_reference_normalisation(golden_dir):
    from margipose_amd import _lib
    g = np.load(os.path.join(golden_dir, 'frames_u8.npz'))
    L = _lib.lib()
    frames = torch.from_numpy(g['frames']).cuda()
    B, _, H, Wd = frames.shape
    mean3 = (ctypes.c_float * 3)(*[float(v) for v in g['mean']])
    std3 = (ctypes.c_float * 3)(*[float(v) for v in g['stddev']])
    out = torch.empty(B, 3, H, Wd, device='cuda')
    _lib.check(L.mpose_frames_u8(ctypes.c_void_p(frames.data_ptr()), mean3, std3, _lib.ptr(out), B, H, Wd, 0, _lib.stream_ptr()), 'frames_u8')
    torch.cuda.synchronize()
    assert float((out.cpu().double() - torch.from_numpy(g['expected_f64'])).abs().max()) < 5e-7        # fp32 rounding of O(1) values
    assert float((out.cpu() - torch.from_numpy(g['expected_f32'])).abs().max()) < 5e-7
    # the InceptionV4 stem's first-layer gather: patches of the uint8 frames == patches of the reference-normalised floats
    ref = torch.from_numpy(g['expected_f32']).cuda()
    pu = torch.empty(B, H // 2, Wd // 2, 32, device='cuda')
    pf = torch.empty_like(pu)
    _lib.check(L.mpose_im2col_k3s2(ctypes.c_void_p(frames.data_ptr()), 1, mean3, std3, _lib.ptr(pu), B, H, Wd, _lib.stream_ptr()), 'im2col u8')
    _lib.check(L.mpose_im2col_k3s2(ctypes.c_void_p(ref.data_ptr()), 0, None, None, _lib.ptr(pf), B, H, Wd, _lib.stream_ptr()), 'im2col f32')
    torch.cuda.synchronize()
    assert float((pu - pf).abs().max()) < 5e-7


@pytest.mark.parametrize('tag', ['w', 'h'])
def test_chatterbox_head_fixture_through_the_graph(golden_dir, tag):
    from margipose_amd.models import CanonicalSkeletonDesc, ChatterboxModel
    g = np.load(os.path.join(golden_dir, 'chatterbox_cnn.npz'))
    seed_w, seed_x = (int(v) for v in g['seeds'])
    head, idx = ('zy_hm_cnn.', 1) if tag == 'w' else ('xz_hm_cnn.', 2)        # (:241-242: zy shrinks the width, xz the height)
    x = torch.from_numpy(np.random.default_rng(seed_x).standard_normal((1, 128, 32, 32))).float().abs().cuda()
    m = ChatterboxModel(CanonicalSkeletonDesc, 'jsd')
    sd = m.state_dict()
    for k, v in W.fill_like(OrderedDict(W.chatterbox_cnn_entries('', tag == 'w')), seed_w).items():
        sd[head + k] = v
    m.load_state_dict(sd)
    eng = m.cuda().engine()
    for train, key in ((False, 'eval_out_pos_'), (True, 'train_out_pos_')):
        with torch.no_grad():
            outs, _ = eng.graph_forward(None, train, False, features=x)
        got = outs[idx][..., :17].permute(0, 3, 1, 2).cpu()
        want = g[key + tag]
        assert rel(got, want) < 1e-4, (tag, train, rel(got, want))


@pytest.mark.parametrize('stem', ['inceptionv4', 'resnet18', 'resnet34', 'resnet50'])
def test_stem_fixture_through_the_model(golden_dir, stem):
    """tests/golden/stem_<name>.npz (tools/make_golden_stems.py: the reference's REAL make_image_feature_extractor,
    models/margipose_model.py:103-139, over stand-in third-party constructors, inside the reference's MargiPoseModel) consumed
    by the HIP path through the drop-in surface only: MargiPoseModel(skel, 1, True, stem, 'jsd'), strict load_state_dict of the
    reference-schema weights, eval forward, train forward + loss + backward.  Forward quantities at the north star's 1e-4;
    gradient norms against the reference's fp64 run, gated on the deviation of the reference's OWN fp32 run (stored beside it)."""
    from margipose_amd import dsntnn
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    from oracle import model_ref as R
    g = np.load(os.path.join(golden_dir, 'stem_%s.npz' % stem), allow_pickle=False)
    seed, T, B = int(g['seed']), 1, 2
    x, target, _ = W.seeded_inputs(seed + 1000, B)
    mask = torch.tensor(g['mask'], dtype=torch.float32).cuda()
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64, stem=stem), x.double(), T)      # (what the generator did)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(x.cuda())
        errs = {'coords_eval': rel(out.cpu(), g['coords_eval_f64']),
                'l3_eval': rel(m.forward_3d_losses(out, target.cuda()).cpu(), g['losses3d_eval_f64']),
                'hm_xy_eval': rel(m.xy_heatmaps[-1].cpu().numpy()[:, :, ::4, ::4], g['hm_xy_eval_f64'])}
    m.train()
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    l3 = m.forward_3d_losses(out, target.cuda())
    errs['coords_train'] = rel(out.detach().cpu(), g['coords_train_f64'])
    errs['l3_train'] = rel(l3.detach().cpu(), g['losses3d_train_f64'])
    errs['hm_xz_train'] = rel(m.xz_heatmaps[-1].detach().cpu().numpy()[:, :, ::4, ::4], g['hm_xz_train_f64'])
    loss = dsntnn.average_loss(l3, mask)
    loss.backward()
    errs['loss'] = rel(loss.item(), g['loss_f64'])
    running = np.concatenate([b.detach().cpu().numpy().flatten() for k, b in m.named_buffers() if 'running' in k])
    errs['running_after'] = rel(running, g['running_after'])
    print('stem fixture', stem, errs)
    assert max(errs.values()) < 1e-4, errs
    keys = [str(k) for k in g['param_keys']]
    params = dict(m.named_parameters())
    norms = np.array([float(params[k].grad.double().norm()) for k in keys])
    typical = float(np.median(g['gnorm_f64']))
    nz = g['gnorm_f64'] > 1e-9 * typical            # (analytically-zero gradients: the last shortcut BatchNorm's bias)
    dev_gpu = np.abs(norms - g['gnorm_f64'])[nz] / g['gnorm_f64'][nz]
    dev_ref = np.abs(g['gnorm_f32'] - g['gnorm_f64'])[nz] / g['gnorm_f64'][nz]
    dxe = rel(xg.grad.cpu().numpy()[:, :, ::8, ::8], g['dx_f64'])
    dxr = rel(g['dx_f32'], g['dx_f64'])
    print('grad-norm deviation vs the reference fp64: ours median %.2e p90 %.2e max %.2e | reference fp32 median %.2e p90 %.2e max %.2e | dx %.2e (fp32 %.2e)'
          % (np.median(dev_gpu), np.quantile(dev_gpu, 0.9), dev_gpu.max(), np.median(dev_ref), np.quantile(dev_ref, 0.9), dev_ref.max(), dxe, dxr))
    assert np.median(dev_gpu) <= max(1e-4, 3 * np.median(dev_ref), np.quantile(dev_ref, 0.9)) and dev_gpu.max() <= max(1e-4, 5 * dev_ref.max())
    assert norms[~nz].max() < 1e-4 * typical if (~nz).any() else True
    assert dxe <= max(1e-4, 5 * dxr)
