"""GPU parity of the three-product fp16 form of the convolutions (MPOSE_CONV_F16X3, include/margipose_hip.h) through the C ABI:
mpose_absmax / mpose_weights_absmax -> mpose_pack_weights (layout 2) -> mpose_conv_fwd / mpose_conv_wgrad.

Claim under test: with per-tensor power-of-two scales and a two-way fp16 split, three fp16 MFMA products are fp32 arithmetic in
everything but the instruction -- the error against float64 is held to the same gate as the six-product bf16 form
(tests/test_conv_gpu.py: at most 2x the error of torch's own fp32 convolution of the same data) -- including on data an
unscaled fp16 could not carry: gradients of magnitude 1e-7, activations of magnitude 1e+4, heavy-tailed tensors whose
largest element is 1e5 x the typical one, and a sum of two inputs nine orders of magnitude apart.

Reference layers: src/margipose/models/margipose_model.py:33,36,67-68,73-74 and their autograd gradients."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
F16X3 = 32


SLOT = 16 * 64          # floats per activation amax slot (MPOSE_AMAX_SUBSLOTS x MPOSE_AMAX_STRIDE)


def _amax(L, _lib, tensors, C, scale=None, shift=None, relu=False):
    """mpose_absmax over NHWC device tensors -> one amax slot each (rows of the returned tensor)."""
    from margipose_amd._lib import AbsmaxOperands
    slots = torch.zeros(len(tensors), SLOT, dtype=torch.float32, device='cuda')
    ops = []
    for i, t in enumerate(tensors):
        ao = AbsmaxOperands()
        ao.src, ao.dst = t.data_ptr(), slots[i].data_ptr()
        if scale is not None:
            ao.scale, ao.shift = scale.data_ptr(), shift.data_ptr()
        ops.append(ao)
    npix = tensors[0].numel() // C
    _lib.check(L.mpose_absmax((AbsmaxOperands * len(ops))(*ops), len(ops), ctypes.c_int64(npix), C, int(relu), _lib.stream_ptr()), 'absmax')
    return slots


def _pack(L, _lib, eng, w, cout, cin, T, transposed_layout=False):
    """Weights -> (packed fp16 planes, device amax slot, npad)."""
    npad = (cout + 63) // 64 * 64
    kpad = (cin + 31) // 32 * 32
    packed = torch.zeros(T * kpad * npad * 3 // 2, dtype=torch.float32, device='cuda')
    amax = torch.zeros(1, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'], j['amax'] = w.data_ptr(), packed.data_ptr(), amax.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = cout, cin, T, npad, kpad, 2
    if transposed_layout:
        j['sn'], j['sk'], j['st'] = T, cout * T, 1
    else:
        j['sn'], j['sk'], j['st'] = cin * T, T, 1
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_weights_absmax(_lib.ptr(dev), 1, _lib.stream_ptr()), 'weights_absmax')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * kpad * npad, _lib.stream_ptr()), 'pack')
    return packed, amax, npad


def _errs(got_nhwc, ref, f32):
    scale = ref.abs().max()
    got = got_nhwc.cpu().double().permute(0, 3, 1, 2)
    return float((got - ref).abs().max() / scale), float((f32.double() - ref).abs().max() / scale)


def _check(e_gpu, e_f32):
    assert e_gpu <= 2.0 * e_f32 + 2e-7, (e_gpu, e_f32)


def _data(rng, shape, kind):
    if kind == 'normal':
        return rng.standard_normal(shape)
    if kind == 'tiny':                       # gradients late in training
        return rng.standard_normal(shape) * 1e-7
    if kind == 'huge':
        return rng.standard_normal(shape) * 1e4
    if kind == 'heavy':                      # log-normal magnitudes: max ~ 1e5 x median
        return rng.standard_normal(shape) * np.exp(3.0 * rng.standard_normal(shape))
    if kind == 'relu':
        return np.maximum(rng.standard_normal(shape), 0.0)
    raise ValueError(kind)


def test_absmax_and_weight_scale():
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for C, npix in ((128, 2048), (192, 777), (32, 64), (8, 5)):
        x = torch.from_numpy(_data(rng, (npix, C), 'heavy')).float().cuda()
        sc = torch.from_numpy(rng.uniform(-1.5, 1.5, C)).float().cuda()
        sh = torch.from_numpy(rng.standard_normal(C)).float().cuda()
        got = _amax(L, _lib, [x, -x, 2 * x], C).cpu()
        assert int((got != 0).sum(1).max()) <= 16 and float(got.view(3, 16, 64)[:, :, 1:].abs().max()) == 0.0      # sub-slots only
        assert got.max(1).values.tolist() == [float(x.abs().max()), float(x.abs().max()), float((2 * x).abs().max())]
        got = _amax(L, _lib, [x], C, sc, sh, relu=True).cpu()
        assert float(got.max()) == float(torch.relu(torch.addcmul(sh, x, sc)).max())       # (addcmul: the same fused multiply-add)
    # layout 2: h + l rebuilds w * 2^k to 2^-22 relative, largest magnitude in [2^14, 2^15)
    cout, cin = 64, 32
    w = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * np.exp(rng.uniform(-6, 2, (cout, cin, 1, 1)))).float().cuda()
    packed, amax, npad = _pack(L, _lib, eng, w, cout, cin, 1)
    torch.cuda.synchronize()
    assert float(amax) == float(w.abs().max())
    planes = packed.view(torch.float16)[:(cin // 16) * 2 * npad * 16].view(cin // 16, 2, npad, 2, 8).double()
    rebuilt = planes.sum(1).permute(1, 0, 2, 3).reshape(npad, cin)[:cout]
    k = 14 - int(np.floor(np.log2(float(amax))))
    ref = w.view(cout, cin).double() * 2.0 ** k
    assert 2.0 ** 14 <= float(ref.abs().max()) < 2.0 ** 15
    # 22 significant bits down to fp16's subnormal spacing (2^-24 in scaled units = 2^-38 of the largest weight)
    assert bool(((rebuilt.cpu() - ref.cpu()).abs() <= 2.0 ** -21 * ref.cpu().abs() + 2.0 ** -24).all())


@pytest.mark.parametrize('B,H,cin,cout,kind', [(2, 32, 128, 128, 'normal'), (8, 16, 192, 192, 'relu'), (1, 12, 64, 96, 'heavy'),
                                              (32, 32, 128, 128, 'tiny'), (3, 8, 32, 32, 'huge'), (4, 32, 128, 32, 'heavy')])
def test_conv3x3_fp32_equivalent(B, H, cin, cout, kind):
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(_data(rng, (B, cin, H, H), kind)).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, w_amax, npad = _pack(L, _lib, eng, w.cuda(), cout, cin, 9)
    x_amax = _amax(L, _lib, [xg], cin)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.in_amax, op.w0_amax = xg.data_ptr(), packed.data_ptr(), out.data_ptr(), x_amax.data_ptr(), w_amax.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    _check(*_errs(out, F.conv2d(x.double(), w.double(), padding=1), F.conv2d(x, w, padding=1)))


def test_f16x3_needs_its_amax_operands():
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    x = torch.zeros(1, 8, 8, 32, device='cuda'); out = torch.zeros(1, 8, 8, 32, device='cuda')
    packed, w_amax, npad = _pack(L, _lib, eng, torch.zeros(32, 32, 3, 3, device='cuda'), 32, 32, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(1, 8, 32, 8, 32, 0, 8, 1, 1, [(0, 0, t9)], npad)
    op = ConvOperands()
    op.in_, op.w0, op.out0 = x.data_ptr(), packed.data_ptr(), out.data_ptr()
    assert L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()) == -22
    # an all-zero tensor (amax 0) is legal and yields zeros
    x_amax = _amax(L, _lib, [x], 32)
    op.in_amax, op.w0_amax = x_amax.data_ptr(), w_amax.data_ptr()
    out.fill_(float('nan'))
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()), 'conv')
    assert float(out.abs().max()) == 0.0


def test_conv_fused_shortcut_and_stride2():
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    B, H, cin, cout = 4, 32, 128, 192
    rng = np.random.default_rng(7)
    x = torch.from_numpy(_data(rng, (B, cin, H, H), 'relu')).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * 0.03).float()
    ws = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * 9.0).float()          # (300x the main weights: own scale)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, wa, npad = _pack(L, _lib, eng, w.cuda(), cout, cin, 9)
    packed1, wa1, _ = _pack(L, _lib, eng, ws.cuda(), cout, cin, 1)
    xa = _amax(L, _lib, [xg], cin)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H // 2, cout, cout, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    out = torch.full((B, H // 2, H // 2, cout), float('nan'), device='cuda')
    out1 = torch.full((B, H // 2, H // 2, cout), float('nan'), device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.w1, op.out1 = xg.data_ptr(), packed.data_ptr(), out.data_ptr(), packed1.data_ptr(), out1.data_ptr()
    op.in_amax, op.w0_amax, op.w1_amax = xa.data_ptr(), wa.data_ptr(), wa1.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    _check(*_errs(out, F.conv2d(x.double(), w.double(), stride=2, padding=1), F.conv2d(x, w, stride=2, padding=1)))
    _check(*_errs(out1, F.conv2d(x.double(), ws.double(), stride=2), F.conv2d(x, ws, stride=2)))


def test_conv_transposed_stride2_classes():
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    B, H, cin, cout = 4, 16, 192, 128
    rng = np.random.default_rng(11)
    x = torch.from_numpy(_data(rng, (B, cin, H, H), 'tiny')).float()
    w = torch.from_numpy(rng.standard_normal((cin, cout, 3, 3)) * 0.03).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, wa, npad = _pack(L, _lib, eng, w.cuda(), cout, cin, 9, transposed_layout=True)
    xa = _amax(L, _lib, [xg], cin)
    g = eng._geom(B, H, cin, 2 * H, cout, 0, H, 1, 2, eng._up_classes(False), npad)
    out = torch.full((B, 2 * H, 2 * H, cout), float('nan'), device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.in_amax, op.w0_amax = xg.data_ptr(), packed.data_ptr(), out.data_ptr(), xa.data_ptr(), wa.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    fn = lambda a, b: F.conv_transpose2d(a, b, stride=2, padding=1, output_padding=1)
    _check(*_errs(out, fn(x.double(), w.double()), fn(x, w)))


@pytest.mark.parametrize('B,H,C', [(2, 32, 128), (8, 16, 192), (2, 8, 32), (1, 12, 64)])
def test_conv_prologue_and_bn_statistics(B, H, C):
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(100 + H)
    x = torch.from_numpy(rng.standard_normal((B, C, H, H)) * 37.0).float()
    w = torch.from_numpy(rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C) / 37.0).float()
    sh = torch.from_numpy(rng.standard_normal(C) * 0.3).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, wa, npad = _pack(L, _lib, eng, w.cuda(), C, C, 9)
    scg, shg = sc.cuda(), sh.cuda()
    xa = _amax(L, _lib, [xg], C, scg, shg, relu=True)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, C), float('nan'), device='cuda')
    stats = torch.zeros(C, 2, dtype=torch.float64, device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0 = xg.data_ptr(), packed.data_ptr(), out.data_ptr()
    op.in_scale, op.in_shift, op.stats0 = scg.data_ptr(), shg.data_ptr(), stats.data_ptr()
    op.in_amax, op.w0_amax = xa.data_ptr(), wa.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    a64 = torch.relu(x.double() * sc.double().view(1, C, 1, 1) + sh.double().view(1, C, 1, 1))
    a32 = torch.relu(torch.addcmul(sh.view(1, C, 1, 1), x, sc.view(1, C, 1, 1)))          # the kernel's fp32 prologue, exactly
    # against the convolution of the SAME fp32 activations: the fp32-equivalence gate
    _check(*_errs(out, F.conv2d(a32.double(), w.double(), padding=1), F.conv2d(a32, w, padding=1)))
    ref = F.conv2d(a64, w.double(), padding=1)
    got = out.cpu().double().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    s_ref = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1)
    err = (stats.cpu() - s_ref).abs() / (s_ref.abs() + ref.abs().max() * (B * H * H) ** 0.5)
    assert float(err.max()) < 1e-4, float(err.max())


@pytest.mark.parametrize('B,H,cin,cout,ratio', [(2, 32, 128, 128, 1.0), (8, 16, 192, 192, 1e-9), (1, 12, 64, 96, 1e6), (4, 32, 32, 128, 1e-3),
                                               (2, 16, 64, 64, 0.0), (2, 16, 64, 64, 1e-30)])
def test_conv_sum_of_two_inputs(B, H, cin, cout, ratio):
    """MPOSE_CONV_SUM_INPUTS with the two inputs at very different magnitudes: each has its own scale, the first pass is
    re-expressed in the second pass's units (a power of two) before the second accumulates on top.  ratio 0 / 1e-30: a second input
    that is all zeros (a shortcut BatchNorm frozen at gamma = 0) or negligible must not overflow that re-expression."""
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(B + H + cin)
    x0 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, H)) * ratio).float()
    w0 = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    w1 = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).float()
    p0, wa0, npad = _pack(L, _lib, eng, w0.cuda(), cout, cin, 9)
    p1, wa1, _ = _pack(L, _lib, eng, w1.cuda(), cout, cin, 1)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, cout, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    x0g, x1g = (t.permute(0, 2, 3, 1).contiguous().cuda() for t in (x0, x1))
    xa = _amax(L, _lib, [x0g, x1g], cin)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    op = ConvOperands()
    op.in_, op.in1, op.w0, op.w1, op.out0 = x0g.data_ptr(), x1g.data_ptr(), p0.data_ptr(), p1.data_ptr(), out.data_ptr()
    op.in_amax, op.in1_amax, op.w0_amax, op.w1_amax = xa[0].data_ptr(), xa[1].data_ptr(), wa0.data_ptr(), wa1.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, 2 | F16X3, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    fn = lambda a0, a1, v0, v1: F.conv2d(a0, v0, padding=1) + F.conv2d(a1, v1)
    _check(*_errs(out, fn(x0.double(), x1.double(), w0.double(), w1.double()), fn(x0, x1, w0, w1)))


def _wgrad(L, _lib, eng, g, x_nhwc, gout_nhwc, cout, cin, T, npad, n_split, amaxes, scale=None, shift=None, gout1=None, cout1=0):
    from margipose_amd._lib import WgradOperands
    kpad = (cin + 31) // 32 * 32
    part = torch.full((n_split * T * kpad * npad,), float('nan'), device='cuda')
    wo = WgradOperands()
    wo.in_, wo.gout0, wo.dw0 = x_nhwc.data_ptr(), gout_nhwc.data_ptr(), part.data_ptr()
    wo.in_amax, wo.gout0_amax = amaxes[0].data_ptr(), amaxes[1].data_ptr()
    if scale is not None:
        wo.in_scale, wo.in_shift = scale.data_ptr(), shift.data_ptr()
    part1 = None
    if gout1 is not None:
        part1 = torch.full((n_split * kpad * npad,), float('nan'), device='cuda')
        wo.gout1, wo.dw1, wo.gout1_amax = gout1.data_ptr(), part1.data_ptr(), amaxes[2].data_ptr()
    _lib.check(L.mpose_conv_wgrad(ctypes.byref(g), (WgradOperands * 1)(wo), 1, n_split, _lib.stream_ptr()), 'wgrad')
    outs = []
    for p, co, t in ((part, cout, T), (part1, cout1, 1)):
        if p is None:
            continue
        dw = torch.full((co, cin, t), float('nan'), device='cuda')
        jobs = np.zeros(1, dtype=eng.UNPACK_DT)
        j = jobs[0]
        j['src'], j['dst'] = p.data_ptr(), dw.data_ptr()
        j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['n_split'] = co, cin, t, npad, kpad, n_split
        j['sn'], j['sk'], j['st'], j['accumulate'] = cin * t, t, 1, 0
        dev = eng._jobs_to_device(jobs, 'cuda')
        _lib.check(L.mpose_unpack_wgrads(_lib.ptr(dev), 1, co * cin * t, _lib.stream_ptr()), 'unpack')
        outs.append(dw)
    torch.cuda.synchronize()
    return outs


def _wgrad_errs(dw, x, go, fn, shape):
    def grad(dtype):
        w = torch.zeros(shape, dtype=dtype, requires_grad=True)
        fn(x.to(dtype), w).backward(go.to(dtype))
        return w.grad.double()
    ref, f32 = grad(torch.float64), grad(torch.float32)
    scale = ref.abs().max()
    return float((dw.cpu().double().reshape(shape) - ref).abs().max() / scale), float((f32 - ref).abs().max() / scale)


@pytest.mark.parametrize('B,H,cin,cout,n_split,pro,gkind', [(2, 32, 128, 128, 3, False, 'normal'), (4, 16, 192, 192, 2, True, 'tiny'),
                                                         (1, 16, 64, 96, 1, False, 'heavy'), (3, 8, 32, 32, 4, True, 'normal'),
                                                         (2, 24, 128, 64, 5, False, 'tiny'), (2, 16, 96, 128, 2, True, 'heavy'),
                                                         # (the one- and two-wave workgroups of the 32-channel tiles; 24: an odd number of octets per row)
                                                         (2, 16, 32, 64, 3, False, 'heavy'), (2, 24, 32, 32, 5, False, 'tiny'),
                                                         (1, 24, 32, 64, 2, True, 'normal')])
def test_weight_gradient_fp32_equivalent(B, H, cin, cout, n_split, pro, gkind):
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    rng = np.random.default_rng(B * 100 + H + cin)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    go = torch.from_numpy(_data(rng, (B, cout, H, H), gkind)).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cin)).float()
    sh = torch.from_numpy(rng.standard_normal(cin) * 0.3).float()
    npad = (cout + 63) // 64 * 64
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    xg, gg = x.permute(0, 2, 3, 1).contiguous().cuda(), go.permute(0, 2, 3, 1).contiguous().cuda()
    scg, shg = (sc.cuda(), sh.cuda()) if pro else (None, None)
    amaxes = torch.cat([_amax(L, _lib, [xg], cin, scg, shg, relu=pro), _amax(L, _lib, [gg], cout)])
    dw, = _wgrad(L, _lib, eng, g, xg, gg, cout, cin, 9, npad, n_split, amaxes, scg, shg)

    def fn(a, w):
        if pro:
            a = F.relu(a * sc.to(a.dtype).view(1, -1, 1, 1) + sh.to(a.dtype).view(1, -1, 1, 1))
        return F.conv2d(a, w, padding=1)
    _check(*_wgrad_errs(dw, x, go, fn, (cout, cin, 3, 3)))


def test_weight_gradient_stride2_with_fused_shortcut():
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    B, H, cin, cout = 2, 32, 128, 192
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    go = torch.from_numpy(rng.standard_normal((B, cout, H // 2, H // 2)) * 1e-6).float()
    go1 = torch.from_numpy(rng.standard_normal((B, cout, H // 2, H // 2)) * 3.0).float()
    npad = (cout + 63) // 64 * 64
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H // 2, cout, cout, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xg, gg, gg1 = to(x), to(go), to(go1)
    amaxes = torch.cat([_amax(L, _lib, [xg], cin), _amax(L, _lib, [gg, gg1], cout)])
    dw, dw1 = _wgrad(L, _lib, eng, g, xg, gg, cout, cin, 9, npad, 2, amaxes, gout1=gg1, cout1=cout)
    _check(*_wgrad_errs(dw, x, go, lambda a, w: F.conv2d(a, w, stride=2, padding=1), (cout, cin, 3, 3)))
    _check(*_wgrad_errs(dw1, x, go1, lambda a, w: F.conv2d(a, w, stride=2), (cout, cin, 1, 1)))


@pytest.mark.parametrize('n_split', [1, 3])
def test_weight_gradient_transposed_stride2_with_fused_shortcut(n_split):
    """ConvTranspose2d(192, 128, 3, stride 2, padding 1, output_padding 1) + the 1x1 transposed shortcut of the columns' up block
    (reference models/margipose_model.py:67-82): the weight gradient of the four output-phase classes -- on the row-of-taps kernel the
    GRADIENT is the strided side (every second row and pixel, one view per class; csrc/wgrad.hip build_units)."""
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    B, H, cin, cout = 2, 16, 192, 128
    rng = np.random.default_rng(78 + n_split)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    go = torch.from_numpy(rng.standard_normal((B, cout, 2 * H, 2 * H)) * 1e-3).float()
    go1 = torch.from_numpy(rng.standard_normal((B, cout, 2 * H, 2 * H)) * 2.0).float()
    npad = (cout + 63) // 64 * 64
    g = eng._geom(B, H, cin, 2 * H, cout, cout, H, 1, 2, eng._up_classes(True), npad, npad)
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    xg, gg, gg1 = to(x), to(go), to(go1)
    amaxes = torch.cat([_amax(L, _lib, [xg], cin), _amax(L, _lib, [gg, gg1], cout)])
    dw, dw1 = _wgrad(L, _lib, eng, g, xg, gg, cout, cin, 9, npad, n_split, amaxes, gout1=gg1, cout1=cout)
    # (the partials are [widx][k = input channel][n = gradient channel]: the helper unpacks (n, k, tap); ConvTranspose2d's weight is (k, n, ..))
    _check(*_wgrad_errs(dw.permute(1, 0, 2).contiguous(), x, go,
                        lambda a, w: F.conv_transpose2d(a, w, stride=2, padding=1, output_padding=1), (cin, cout, 3, 3)))
    _check(*_wgrad_errs(dw1.permute(1, 0, 2).contiguous(), x, go1,
                        lambda a, w: F.conv_transpose2d(a, w, stride=2, output_padding=1), (cin, cout, 1, 1)))


@pytest.mark.parametrize('B,H,cin,cout,add,f16', [(2, 32, 128, 128, True, True), (4, 16, 192, 192, False, True), (3, 8, 32, 32, True, True),
                                                 (1, 12, 64, 96, True, False)])
def test_fused_output_stage(B, H, cin, cout, add, f16):
    """Inference epilogue of conv_igemm_k: y = relu(es * conv + et) [+ as * add_src + at] stored as fp32, max |y| accumulated
    into the amax slot the next convolution reads (reference models/margipose_model.py:31-40 with running-statistics BatchNorm)."""
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(B * 31 + H)
    x = torch.from_numpy(_data(rng, (B, cin, H, H), 'relu')).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    es, et = torch.from_numpy(rng.uniform(0.5, 1.5, cout)).float(), torch.from_numpy(rng.standard_normal(cout) * 0.3).float()
    a_s, a_t = torch.from_numpy(rng.uniform(-1.5, 1.5, cout)).float(), torch.from_numpy(rng.standard_normal(cout) * 0.3).float()
    src = torch.from_numpy(rng.standard_normal((B, cout, H, H))).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    srcg = src.permute(0, 2, 3, 1).contiguous().cuda()
    if f16:
        packed, w_amax, npad = _pack(L, _lib, eng, w.cuda(), cout, cin, 9)
    else:
        import tests.test_conv_gpu as T6
        packed, npad, _ = T6._pack(L, _lib, eng, w.cuda(), cout, cin, 9)
    x_amax = _amax(L, _lib, [xg], cin)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    slot = torch.zeros(SLOT, device='cuda')
    dev = [t.cuda() for t in (es, et, a_s, a_t)]
    op = ConvOperands()
    op.in_, op.w0, op.out0 = xg.data_ptr(), packed.data_ptr(), out.data_ptr()
    if f16:
        op.in_amax, op.w0_amax = x_amax.data_ptr(), w_amax.data_ptr()
    op.epi_scale0, op.epi_shift0, op.out0_amax = dev[0].data_ptr(), dev[1].data_ptr(), slot.data_ptr()
    if add:
        op.add_src, op.add_scale, op.add_shift = srcg.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, (F16X3 if f16 else 0) | 16, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()

    def fn(a, v, dt):
        y = torch.relu(F.conv2d(a, v, padding=1) * es.to(dt).view(1, -1, 1, 1) + et.to(dt).view(1, -1, 1, 1))
        return y + (src.to(dt) * a_s.to(dt).view(1, -1, 1, 1) + a_t.to(dt).view(1, -1, 1, 1)) if add else y
    _check(*_errs(out, fn(x.double(), w.double(), torch.float64), fn(x, w, torch.float32)))
    assert float(slot.max()) == float(out.abs().max())
    # stats / mask / accumulate do not combine with the output stage
    op.stats0 = slot.data_ptr()
    assert L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, (F16X3 if f16 else 0) | 16, _lib.stream_ptr()) == -22


@pytest.mark.parametrize('B,H,cin,cout,f16', [(2, 32, 128, 128, True), (4, 16, 192, 192, True), (3, 8, 32, 64, False)])
def test_consumer_bn_sums_in_epilogue(B, H, cin, cout, f16):
    """mpose_conv_operands.red_*: while the two-input data-gradient stores g = conv3x3(in, w0) + conv1x1(in1, w1), its epilogue
    accumulates the BatchNorm-backward sums of g's consumer -- (sum g*m, sum g*m*a, sum g, sum g*b) per channel with
    m = [scale*a + shift > 0] -- what mpose_bn_bwd_reduce computes in a pass of its own (reference: autograd through
    models/margipose_model.py:34-40)."""
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(B * 7 + H)
    x0 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, H)) * 0.1).float()
    w0 = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    w1 = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).float()
    a = torch.from_numpy(rng.standard_normal((B, cout, H, H))).float()
    b = torch.from_numpy(rng.standard_normal((B, cout, H, H))).float()
    sc = torch.from_numpy(rng.uniform(-1.5, 1.5, cout)).float()
    sh = torch.from_numpy(rng.standard_normal(cout) * 0.3).float()
    if f16:
        p0, wa0, npad = _pack(L, _lib, eng, w0.cuda(), cout, cin, 9)
        p1, wa1, _ = _pack(L, _lib, eng, w1.cuda(), cout, cin, 1)
    else:
        import tests.test_conv_gpu as T6
        p0, npad, _ = T6._pack(L, _lib, eng, w0.cuda(), cout, cin, 9)
        p1, _, _ = T6._pack(L, _lib, eng, w1.cuda(), cout, cin, 1)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, cout, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    x0g, x1g, ag, bg = to(x0), to(x1), to(a), to(b)
    xa = _amax(L, _lib, [x0g, x1g], cin)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    sums = torch.zeros(cout, 4, dtype=torch.float64, device='cuda')
    scg, shg = sc.cuda(), sh.cuda()
    op = ConvOperands()
    op.in_, op.in1, op.w0, op.w1, op.out0 = x0g.data_ptr(), x1g.data_ptr(), p0.data_ptr(), p1.data_ptr(), out.data_ptr()
    if f16:
        op.in_amax, op.in1_amax, op.w0_amax, op.w1_amax = xa[0].data_ptr(), xa[1].data_ptr(), wa0.data_ptr(), wa1.data_ptr()
    op.red_a, op.red_b, op.red_scale, op.red_shift, op.red_sums = ag.data_ptr(), bg.data_ptr(), scg.data_ptr(), shg.data_ptr(), sums.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, 2 | (F16X3 if f16 else 0), _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    gout = out.cpu().double().permute(0, 3, 1, 2)                      # the sums are defined on the values the kernel stored
    m = (torch.addcmul(sh.view(1, -1, 1, 1), a, sc.view(1, -1, 1, 1)) > 0).double()
    ref = torch.stack([(gout * m).sum((0, 2, 3)), (gout * m * a.double()).sum((0, 2, 3)), gout.sum((0, 2, 3)),
                       (gout * b.double()).sum((0, 2, 3))], 1)
    scale = torch.stack([gout.abs().sum((0, 2, 3)), (gout * a.double()).abs().sum((0, 2, 3)), gout.abs().sum((0, 2, 3)),
                         (gout * b.double()).abs().sum((0, 2, 3))], 1)
    assert float(((sums.cpu() - ref).abs() / scale).max()) < 2e-6      # fp32 partial sums over <= 256 rows, fp64 across workgroups
    op.stats0 = sums.data_ptr()                                       # forward statistics and consumer sums exclude each other
    assert L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, 2 | (F16X3 if f16 else 0), _lib.stream_ptr()) == -22


@pytest.mark.parametrize('B,hw,cin,cout,dil,n_split,pro', [(2, (16, 32), 128, 128, (2, 2), 4, False), (2, (16, 32), 64, 192, (4, 4), 4, True),
                                                        (3, (8, 32), 128, 64, (1, 4), 8, False), (2, (32, 16), 128, 128, (4, 1), 3, True),
                                                        (2, (16, 32), 64, 64, (2, 2), 3, False)])
def test_weight_gradient_of_dilated_kernels(B, hw, cin, cout, dil, n_split, pro):
    """Dilated 3x3 (reference models/chatterbox_model.py:62-72, 143-150) in the three-product form: rows 2 / 4 pixels apart go
    straight to the row-of-taps kernel; a kernel dilated by d ALONG x is computed as d launches over the residues of x mod d
    (mpose_conv_wgrad_phases), each into n_split / d of the partials -- or, when n_split is not a multiple of d (last case), by
    conv_wgrad_k.  Same gate as every weight gradient."""
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    rng = np.random.default_rng(B * 100 + hw[0] + cin + dil[1])
    x = torch.from_numpy(rng.standard_normal((B, cin) + hw)).float()
    go = torch.from_numpy(rng.standard_normal((B, cout) + hw)).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cin)).float()
    sh = torch.from_numpy(rng.standard_normal(cin) * 0.3).float()
    npad = (cout + 63) // 64 * 64
    g = eng.conv_geom('f', False, B, hw, cin, hw, cout, (3, 3), (1, 1), dil, dil, npad)
    d = int(L.mpose_conv_wgrad_phases(ctypes.byref(g)))
    assert d == (dil[1] if dil[1] > 1 else 1)
    xg, gg = x.permute(0, 2, 3, 1).contiguous().cuda(), go.permute(0, 2, 3, 1).contiguous().cuda()
    scg, shg = (sc.cuda(), sh.cuda()) if pro else (None, None)
    amaxes = torch.cat([_amax(L, _lib, [xg], cin, scg, shg, relu=pro), _amax(L, _lib, [gg], cout)])
    dw, = _wgrad(L, _lib, eng, g, xg, gg, cout, cin, 9, npad, n_split, amaxes, scg, shg)

    def fn(a, w):
        if pro:
            a = F.relu(a * sc.to(a.dtype).view(1, -1, 1, 1) + sh.to(a.dtype).view(1, -1, 1, 1))
        return F.conv2d(a, w, padding=dil, dilation=dil)
    _check(*_wgrad_errs(dw, x, go, fn, (cout, cin, 3, 3)))


@pytest.mark.parametrize('B,hw,cin,cout,dil,pro', [(2, (16, 32), 128, 128, (2, 2), False), (2, (32, 16), 64, 128, (4, 1), True),
                                                (3, (8, 32), 128, 64, (1, 4), True), (2, (16, 32), 64, 256, (4, 4), False)])
def test_dilated_convolution_forward_and_data_gradient(B, hw, cin, cout, dil, pro):
    """Dilated 3x3 (reference models/chatterbox_model.py:62-72, 143-150) through mpose_conv_fwd in the three-product form: kernel
    rows 2 / 4 pixels apart run the row-group loop as they are, a kernel dilated ALONG x one residue of x per launch; forward
    (with the BatchNorm + ReLU prologue and the statistics epilogue) and data-gradient (accumulating), same gate as every
    convolution."""
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(B * 10 + hw[0] + cin + dil[0] * 3 + dil[1])
    x = torch.from_numpy(rng.standard_normal((B, cin) + hw)).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cin)).float()
    sh = torch.from_numpy(rng.standard_normal(cin) * 0.3).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    scg, shg = (sc.cuda(), sh.cuda()) if pro else (None, None)
    packed, w_amax, npad = _pack(L, _lib, eng, w.cuda(), cout, cin, 9)
    x_amax = _amax(L, _lib, [xg], cin, scg, shg, relu=pro)
    g = eng.conv_geom('f', False, B, hw, cin, hw, cout, (3, 3), (1, 1), dil, dil, npad)
    out = torch.full((B,) + hw + (cout,), float('nan'), device='cuda')
    stats = torch.zeros(cout, 2, dtype=torch.float64, device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.in_amax, op.w0_amax = xg.data_ptr(), packed.data_ptr(), out.data_ptr(), x_amax.data_ptr(), w_amax.data_ptr()
    op.stats0 = stats.data_ptr()
    if pro:
        op.in_scale, op.in_shift = scg.data_ptr(), shg.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()

    def fn(a, b):
        if pro:
            a = F.relu(a * sc.to(a.dtype).view(1, -1, 1, 1) + sh.to(a.dtype).view(1, -1, 1, 1))
        return F.conv2d(a, b, padding=dil, dilation=dil)
    ref = fn(x.double(), w.double())
    _check(*_errs(out, ref, fn(x, w)))
    want = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1)
    assert float((stats.cpu() - want).abs().max() / want.abs().max()) < 1e-5
    # data gradient: accumulates into a tensor that already holds something
    go = torch.from_numpy(rng.standard_normal((B, cout) + hw)).float()
    gg = go.permute(0, 2, 3, 1).contiguous().cuda()
    npad_d = (cin + 63) // 64 * 64
    pd = torch.zeros(9 * cout * npad_d * 3 // 2, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    wg = w.cuda()
    j['src'], j['dst'], j['amax'], j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = wg.data_ptr(), pd.data_ptr(), w_amax.data_ptr(), cin, cout, 9, npad_d, cout, 2
    j['sn'], j['sk'], j['st'] = 9, cin * 9, 1
    _lib.check(L.mpose_pack_weights(_lib.ptr(eng._jobs_to_device(jobs, 'cuda')), 1, 9 * cout * npad_d, _lib.stream_ptr()), 'pack')
    gd = eng.conv_geom('d', False, B, hw, cin, hw, cout, (3, 3), (1, 1), dil, dil, npad_d)
    base = torch.from_numpy(rng.standard_normal((B,) + hw + (cin,))).float().cuda()
    dx = base.clone()
    g_amax = _amax(L, _lib, [gg], cout)
    od = ConvOperands()
    od.in_, od.w0, od.out0, od.in_amax, od.w0_amax = gg.data_ptr(), pd.data_ptr(), dx.data_ptr(), g_amax.data_ptr(), w_amax.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(gd), (ConvOperands * 1)(od), 1, F16X3 | 1, _lib.stream_ptr()), 'dgrad')
    torch.cuda.synchronize()

    def dgrad(dtype):
        a = x.to(dtype).requires_grad_(True)
        F.conv2d(a, w.to(dtype), padding=dil, dilation=dil).backward(go.to(dtype))
        return a.grad
    r64 = dgrad(torch.float64)
    _check(*_errs(dx - base, r64, dgrad(torch.float32)))


@pytest.mark.parametrize('layout', [2, 3])
@pytest.mark.parametrize('cout,cin,T,transposed', [(128, 128, 9, False), (192, 128, 9, True), (17, 128, 9, False), (128, 51, 1, False),
                                                   (96, 64, 7, False), (64, 160, 1, True), (32, 27, 1, False), (64, 3, 49, False)])
def test_packed_weight_layouts_bit_for_bit(layout, cout, cin, T, transposed):
    """mpose_pack_weights, layouts 2 and 3 of include/margipose_hip.h, against a numpy statement of them: [T][Kpad/16][h, l][Npad][2][8]
    (layout 3: [T][Kpad/16][h, l][2][Npad][8]) fp16 of w * 2^k, h = rn16(w 2^k), l = rn16(w 2^k - h), k from the tensor's largest
    magnitude -- for the forward view of Conv2d / ConvTranspose2d weights (round 5's tiled packer: source-order reads, an LDS turn,
    16-byte fragment stores) and a 49-tap case the element-wise packer keeps."""
    import numpy as np
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    rng = np.random.default_rng(cout * 1000 + cin * 10 + T)
    shape = (cin, cout, T) if transposed else (cout, cin, T)
    w_np = (rng.standard_normal(shape) * rng.choice([1e-3, 1.0, 30.0])).astype(np.float32)
    w = torch.from_numpy(w_np).cuda()
    npad = (cout + 63) // 64 * 64
    kpad = (cin + 31) // 32 * 32
    packed = torch.full((T * kpad * npad,), float('nan'), dtype=torch.float32, device='cuda')      # 2 planes x 2 bytes = 4 bytes per element
    amax = torch.zeros(1, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'], j['amax'] = w.data_ptr(), packed.data_ptr(), amax.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = cout, cin, T, npad, kpad, layout
    j['sn'], j['sk'], j['st'] = (T, cout * T, 1) if transposed else (cin * T, T, 1)
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_weights_absmax(_lib.ptr(dev), 1, _lib.stream_ptr()), 'weights_absmax')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * kpad * npad, _lib.stream_ptr()), 'pack')
    torch.cuda.synchronize()
    assert float(amax) == float(np.abs(w_np).max())
    e = (np.float32(np.abs(w_np).max()).view(np.uint32) >> 23) & 0xff
    k = 141 - int(min(max(e, 27), 254))
    wnk = np.zeros((npad, kpad, T), np.float32)                       # [n][k][t], zero padded
    wnk[:cout, :cin] = w_np.transpose(1, 0, 2) if transposed else w_np
    vs = wnk * np.float32(2.0 ** k)
    h = vs.astype(np.float16)
    l = (vs - h.astype(np.float32)).astype(np.float16)
    pl = np.stack([h, l], 0).reshape(2, npad, kpad // 16, 2, 8, T)     # [plane][n][k16][half][8][t]
    if layout == 2:
        want = pl.transpose(5, 2, 0, 1, 3, 4)                           # [t][k16][plane][n][half][8]
    else:
        want = pl.transpose(5, 2, 0, 3, 1, 4)                           # [t][k16][plane][half][n][8]
    got = packed.cpu().numpy().view(np.float16).reshape(want.shape)
    assert np.array_equal(got.view(np.uint16), np.ascontiguousarray(want).view(np.uint16))
