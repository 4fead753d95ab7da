"""GPU parity of the soft-argmax tail kernels (csrc/tail.hip, via the C ABI) against the oracle and the
reference-generated golden vectors.  Tolerances: fp32 kernels vs fp64 oracle, rtol 1e-4 is the
north-star gate; observed errors are ~1e-6 and the tests assert 2e-5 where stated."""
import os

import numpy as np
import pytest
import torch

from oracle import tail_np as T

pytestmark = pytest.mark.gpu

TAIL_CASES = ['f32x2_3d', 'f32x2_3d_masked', 'f32x2_2d', 'f32x2_3d_nopix', 'f48x1_3d', 'f64x1_3d']


@pytest.fixture(scope='module')
def D():
    from margipose_amd import dsntnn
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return dsntnn


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device='cuda')


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_known_answer_gpu(D, golden_dir):
    """reference tests/test_models.py:39-46."""
    from margipose_amd.models.margipose_model import MargiPoseModel
    xy = D.make_gauss(dev([[[-0.5, 0.5]]]), (32, 32), 1, normalize=True)
    zy = D.make_gauss(dev([[[0.1, 0]]]), (32, 32), 1, normalize=True)
    xz = D.make_gauss(dev([[[0, 0.2]]]), (32, 32), 1, normalize=True)
    xyz = MargiPoseModel.heatmaps_to_coords(xy, zy, xz)
    torch.testing.assert_close(xyz.cpu(), torch.tensor([[[-0.5, 0.5, 0.15]]]), rtol=1.3e-6, atol=1e-5)


def _inputs(g):
    rng = np.random.default_rng(int(g['seed']))
    B, F = int(g['B']), int(g['F'])
    logits = [rng.standard_normal((B, 17, F, F)) * 4.0 for _ in range(3)]
    target = rng.uniform(-1, 1, (B, 17, 3))
    return logits, target, g['mask']


@pytest.mark.parametrize('case', TAIL_CASES)
def test_tail_vs_golden(D, golden_dir, case):
    from margipose_amd.models.margipose_model import MargiPoseModel
    g = np.load(os.path.join(golden_dir, 'tail_%s.npz' % case))
    logits, target, mask = _inputs(g)
    pix = bool(int(g['pixelwise'])); three_d = str(g['loss_kind']) == '3d'
    lg = [dev(l).requires_grad_(True) for l in logits]
    hm = [D.flat_softmax(l) for l in lg]
    for p, h in zip(('xy', 'zy', 'xz'), hm):
        np.testing.assert_allclose(h.detach().cpu().numpy()[:, :, ::4, ::4], g['hm_%s_f64' % p], rtol=2e-5, atol=1e-30)
        assert abs(float(h.sum()) - h.shape[0] * 17) < 1e-3
    coords = MargiPoseModel.heatmaps_to_coords(*hm)
    np.testing.assert_allclose(coords.detach().cpu().numpy(), g['coords_f64'], rtol=1e-5, atol=2e-6)
    losses = D.stage_losses(hm[0], hm[1], hm[2], dev(target), 1.0, pix, three_d)
    np.testing.assert_allclose(losses.detach().cpu().numpy(), g['losses_f64'], rtol=2e-5, atol=1e-6)
    loss = D.average_loss(losses, dev(mask))
    np.testing.assert_allclose(loss.item(), g['loss_f64'], rtol=2e-5)
    loss.backward()
    for p, l in zip(('xy', 'zy', 'xz'), lg):
        ref = g['dlogits_%s_f64' % p]
        got = l.grad.cpu().numpy()[:, ::4] if l.grad is not None else np.zeros_like(ref)
        assert rel_err(got, ref) < 2e-5, (p, rel_err(got, ref))
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-5 * np.abs(ref).max() + 1e-12)


@pytest.mark.parametrize('F', [32, 48, 64, 96, 128])      # (96, 128: rows beyond 4096 elements -- the multi-pass kernels of round 6)
def test_individual_ops_vs_oracle(D, F):
    rng = np.random.default_rng(7 + F)
    B = 3
    logits = rng.standard_normal((B, 17, F, F)) * 3.0
    mu = rng.uniform(-1, 1, (B, 17, 2))
    p64 = T.flat_softmax(logits)
    x = dev(logits).requires_grad_(True)
    p = D.flat_softmax(x)
    assert rel_err(p.detach().cpu().numpy(), p64) < 1e-5
    c = D.dsnt(p)
    np.testing.assert_allclose(c.detach().cpu().numpy(), T.dsnt(p64), rtol=1e-5, atol=2e-6)
    js = D.js_reg_losses(p, dev(mu), 1.0)
    np.testing.assert_allclose(js.detach().cpu().numpy(), T.js_reg_losses(p64, mu, 1.0), rtol=2e-5, atol=1e-6)
    # backward of (sum js + sum coords*w) through softmax
    wc = rng.standard_normal((B, 17, 2))
    (js.sum() + (c * dev(wc)).sum()).backward()
    xs = T.normalized_linspace(F)[None, None, None, :]; ys = T.normalized_linspace(F)[None, None, :, None]
    g64 = T.js_grad_wrt_p(p64, T.make_gauss(mu, (F, F), 1.0)) + wc[..., 0][..., None, None] * xs + wc[..., 1][..., None, None] * ys
    d64 = T.softmax_backward(p64, g64)
    assert rel_err(x.grad.cpu().numpy(), d64) < 2e-5


def test_full_size_properties(D):
    """BASELINE config sizes (B=64, 17 joints, 32x32): size-independent properties."""
    torch.manual_seed(12345)
    B = 64
    lg = [torch.randn(B, 17, 32, 32, device='cuda') * 4 for _ in range(3)]
    hm = [D.flat_softmax(l) for l in lg]
    for h in hm:
        s = h.flatten(2).sum(-1)
        assert (s - 1).abs().max() < 1e-5 and h.min() >= 0
    from margipose_amd.models.margipose_model import MargiPoseModel
    xyz = MargiPoseModel.heatmaps_to_coords(*hm)
    assert xyz.shape == (B, 17, 3) and xyz.abs().max() < 1
    # softmax shift invariance + dsnt linearity
    h2 = D.flat_softmax(lg[0] + 3.0)
    assert (h2 - hm[0]).abs().max() < 1e-6
    mix = 0.25 * hm[0] + 0.75 * hm[1]
    torch.testing.assert_close(D.dsnt(mix), 0.25 * D.dsnt(hm[0]) + 0.75 * D.dsnt(hm[1]), rtol=1e-4, atol=1e-6)
    # JS of a heatmap with itself-as-target is ~0 and JS <= ln 2
    tgt = torch.rand(B, 17, 3, device='cuda') * 1.6 - 0.8
    gauss = D.make_gauss(tgt[..., :2].contiguous(), (32, 32), 1.0)
    js0 = D.js_reg_losses(gauss.contiguous(), tgt[..., :2].contiguous(), 1.0)
    assert js0.abs().max() < 1e-5
    js = D.js_reg_losses(hm[0], tgt[..., :2].contiguous(), 1.0)
    assert js.min() >= 0 and js.max() <= np.log(2) + 1e-5


def test_empty_and_errors(D):
    e = torch.empty(0, 17, 32, 32, device='cuda')
    assert D.flat_softmax(e).shape == (0, 17, 32, 32)
    assert D.dsnt(e).shape == (0, 17, 2)
    from margipose_amd import _lib
    with pytest.raises(_lib.MposeError):
        D.flat_softmax(torch.zeros(1, 17, 32, 32))          # CPU tensor: no fallback
    with pytest.raises(_lib.MposeError):
        D.flat_softmax(torch.zeros(1, 17, 30, 30, device='cuda'))   # W % 4 != 0
    with pytest.raises(AssertionError):
        D.average_loss(torch.zeros(2, 17, device='cuda'), torch.zeros(2, 16, device='cuda'))


def test_bf16_heatmap_io(D):
    """cfg2 of BASELINE.json: bf16 heatmaps + fp32 soft-argmax."""
    from margipose_amd import _lib
    rng = np.random.default_rng(3)
    B = 4
    lg32 = [torch.tensor(rng.standard_normal((B, 17, 32, 32)) * 4, dtype=torch.float32) for _ in range(3)]
    lgb = [l.to(torch.bfloat16).cuda() for l in lg32]
    hm = [torch.empty_like(l) for l in lgb]
    xyz = torch.empty(B, 17, 3, device='cuda')
    _lib.check(_lib.lib().mpose_softmax_dsnt_fwd(_lib.ptr_array(lgb), _lib.ptr_array(hm), None, _lib.ptr(xyz), 3, B * 17, 32,
                                                  32, 1, _lib.stream_ptr()), 'softmax bf16')
    ref_hm = [T.flat_softmax(l.float().cpu().numpy().astype(np.float64)) for l in lgb]
    ref_xyz = T.heatmaps_to_coords(*ref_hm)
    np.testing.assert_allclose(xyz.cpu().numpy(), ref_xyz, rtol=1e-5, atol=2e-6)    # fp32 soft-argmax on bf16 logits
    for h, r in zip(hm, ref_hm):
        np.testing.assert_allclose(h.float().cpu().numpy(), r, rtol=8e-3, atol=1e-30)  # bf16 rounding of outputs


@pytest.mark.parametrize('F,bf16,B', [(32, False, 5), (32, True, 5), (48, False, 5), (16, False, 5), (64, False, 5),
                                      # inputs beyond 128 MB: the launcher takes the all-joints form (every channel line read once)
                                      (32, False, 180), (16, True, 700)])
def test_fused_residual_sum_softmax_is_bit_identical(F, bf16, B):
    """mpose_bn_add_softmax_fwd (the last ResidualBlock's sum + flat_softmax + dsnt, logits kept in LDS) against the two launches
    it replaces (mpose_bn_add_fwd layout 1, then mpose_softmax_dsnt_fwd): same heatmaps and coordinates, bit for bit -- in the
    four-joints-per-workgroup form and in the all-joints form the launcher picks when the inputs exceed the Infinity Cache."""
    from margipose_amd import _lib
    from margipose_amd._lib import BnAddOperands
    L = _lib.lib()
    gen = torch.Generator(device='cuda').manual_seed(11 + F + B)
    C, J = 32, 17
    g = lambda *s: torch.randn(*s, generator=gen, device='cuda', dtype=torch.float32)
    a = [g(B, F, F, C) * 3 for _ in range(3)]
    b = [g(B, F, F, C) * 2 for _ in range(3)]
    co = [[g(C) for _ in range(4)] for _ in range(3)]
    logits = [torch.empty(B, J, F, F, device='cuda') for _ in range(3)]
    dt = torch.bfloat16 if bf16 else torch.float32
    heat0 = [torch.empty(B, J, F, F, device='cuda', dtype=dt) for _ in range(3)]
    heat1 = [torch.full((B, J, F, F), 7.0, device='cuda', dtype=dt) for _ in range(3)]
    pc0 = torch.empty(3, B * J, 2, device='cuda'); pc1 = torch.full((3, B * J, 2), 7.0, device='cuda')
    xyz0 = torch.empty(B, J, 3, device='cuda'); xyz1 = torch.empty(B, J, 3, device='cuda')
    ops = []
    for c in range(3):
        ao = BnAddOperands()
        ao.a, ao.a_scale, ao.a_shift = a[c].data_ptr(), co[c][0].data_ptr(), co[c][1].data_ptr()
        ao.b, ao.b_scale, ao.b_shift = b[c].data_ptr(), co[c][2].data_ptr(), co[c][3].data_ptr()
        ao.out = logits[c].data_ptr()
        ops.append(ao)
    st = _lib.stream_ptr()
    _lib.check(L.mpose_bn_add_fwd((BnAddOperands * 3)(*ops), 3, F * F, B, C, 1, J, st), 'bn_add')
    _lib.check(L.mpose_softmax_dsnt_fwd(_lib.ptr_array(logits), _lib.ptr_array(heat0), _lib.ptr(pc0), _lib.ptr(xyz0), 3, B * J, F, F,
                                        2 if bf16 else 0, st), 'softmax')
    _lib.check(L.mpose_bn_add_softmax_fwd((BnAddOperands * 3)(*ops), _lib.ptr_array(heat1), _lib.ptr(pc1), 3, B, F, F, C, J,
                                          2 if bf16 else 0, st), 'fused')
    _lib.check(L.mpose_coords_merge(_lib.ptr(pc1), _lib.ptr(xyz1), B * J, st), 'merge')
    torch.cuda.synchronize()
    for h0, h1 in zip(heat0, heat1):
        assert torch.equal(h0, h1)
    assert torch.equal(pc0, pc1) and torch.equal(xyz0, xyz1)
    # and against the oracle (fp64): the fused path is still the reference's tail
    lg = [torch.relu(a[c][..., :J] * co[c][0][:J] + co[c][1][:J]) + (b[c][..., :J] * co[c][2][:J] + co[c][3][:J]) for c in range(3)]
    ref = [T.flat_softmax(l.permute(0, 3, 1, 2).double().cpu().numpy()) for l in lg]
    if not bf16:
        np.testing.assert_allclose(xyz1.cpu().numpy(), T.heatmaps_to_coords(*ref), rtol=1e-4, atol=5e-6)


def test_fused_residual_sum_softmax_rejects():
    from margipose_amd import _lib
    from margipose_amd._lib import BnAddOperands
    L = _lib.lib()
    t = torch.zeros(1, 64, 64, 32, device='cuda'); v = torch.zeros(32, device='cuda'); h = torch.zeros(1, 17, 64, 64, device='cuda')
    ao = BnAddOperands()
    ao.a = ao.b = t.data_ptr(); ao.a_scale = ao.a_shift = ao.b_scale = ao.b_shift = v.data_ptr()
    ops = (BnAddOperands * 3)(ao)
    hp = _lib.ptr_array([h])
    assert L.mpose_bn_add_softmax_fwd(ops, hp, None, 1, 1, 68, 68, 32, 17, 0, _lib.stream_ptr()) == -22   # H*W > 4096
    assert L.mpose_bn_add_softmax_fwd(ops, hp, None, 1, 1, 30, 30, 32, 17, 0, _lib.stream_ptr()) == -22   # W % 4
    assert L.mpose_bn_add_softmax_fwd(ops, hp, None, 1, 1, 32, 32, 32, 33, 0, _lib.stream_ptr()) == -22   # J > C
    assert L.mpose_bn_add_softmax_fwd(ops, hp, None, 1, 0, 32, 32, 32, 17, 0, _lib.stream_ptr()) == 0                   # empty batch
