"""GPU parity of the round-2 convolution engine (csrc/conv_p.hip + csrc/split.hip) through the C ABI:
mpose_split_planes / mpose_bn_add_planes / mpose_bn_bwd_apply_planes write the pre-split bf16 planes, mpose_conv_fwd with
MPOSE_CONV_PLANES_IN consumes them.  Same gate as tests/test_conv_gpu.py: the result may be at most 2x as far from a
float64 convolution as torch's own fp32 convolution of the same data ("fp32-equivalent"), on dense random operands.
MPOSE_CONV_BF16 (single bf16 pass, BASELINE configs[4]) is checked against a float64 convolution of the bf16-ROUNDED
operands (products exact, fp32 accumulation: 1e-5) and must differ from the fp32 result by no more than bf16 rounding.

Reference layers: src/margipose/models/margipose_model.py:25-40,67-82 (Conv2d / ConvTranspose2d + BatchNorm + ReLU)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
PLANES_IN, BF16 = 4, 8


def _env():
    from margipose_amd import _lib, engine as eng
    return _lib.lib(), _lib, eng


def pack(w, cout, cin, T, transposed_layout=False, layout=1):
    L, _lib, eng = _env()
    npad = (cout + 63) // 64 * 64
    kpad = (cin + 31) // 32 * 32
    packed = torch.zeros(T * kpad * npad * 3 // 2, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'] = w.data_ptr(), packed.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = cout, cin, T, npad, kpad, layout
    j['sn'], j['sk'], j['st'] = (T, cout * T, 1) if transposed_layout else (cin * T, T, 1)
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * kpad * npad, _lib.stream_ptr()), 'pack')
    return packed, npad


def to_planes(x_nhwc, scale=None, shift=None, relu=False):
    """fp32 NHWC device tensor -> P8 plane buffer (uint8 tensor) through mpose_split_planes."""
    L, _lib, eng = _env()
    from margipose_amd._lib import SplitOperands
    C = x_nhwc.shape[-1]
    npix = x_nhwc.numel() // C
    buf = torch.full((int(L.mpose_planes_bytes(npix, C)),), 0x7f, dtype=torch.uint8, device='cuda')     # poisoned (bf16 NaN pattern 0x7f7f)
    op = SplitOperands()
    op.src, op.planes = x_nhwc.data_ptr(), buf.data_ptr()
    if scale is not None:
        op.scale, op.shift = scale.data_ptr(), shift.data_ptr()
    _lib.check(L.mpose_split_planes((SplitOperands * 1)(op), 1, ctypes.c_int64(npix), C, int(relu), _lib.stream_ptr()), 'split')
    return buf


def planes_to_f64(buf, npix, C):
    """(hi + mid + lo) as float64 NHWC (npix, C)."""
    p = buf.view(torch.bfloat16).view(C // 8, 3, npix, 8).float().double()
    return p.sum(1).permute(1, 0, 2).reshape(npix, C)


def conv_planes(g, xin, packed, out, flags=0, packed1=None, out1=None, in1=None, stats=None, mask=None):
    L, _lib, eng = _env()
    from margipose_amd._lib import ConvOperands
    op = ConvOperands()
    op.in_, op.w0, op.out0 = xin.data_ptr(), packed.data_ptr(), out.data_ptr()
    if packed1 is not None:
        op.w1 = packed1.data_ptr()
    if out1 is not None:
        op.out1 = out1.data_ptr()
    if in1 is not None:
        op.in1 = in1.data_ptr()
    if stats is not None:
        op.stats0 = stats.data_ptr()
    if mask is not None:
        op.mask_src, op.mask_scale, op.mask_shift = (t.data_ptr() for t in mask)
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, PLANES_IN | flags, _lib.stream_ptr()), 'conv planes')
    torch.cuda.synchronize()


def errs(got_nhwc, ref, f32):
    scale = ref.abs().max()
    got = got_nhwc.cpu().double().permute(0, 3, 1, 2)
    return float((got - ref).abs().max() / scale), float((f32.double() - ref).abs().max() / scale)


def check(e_gpu, e_f32):
    # 3x (tests/test_conv_gpu.py: 2x): this engine has no split-K, so an output's hi x hi accumulator is rounded once per
    # (tap, 16 channels) step -- 72..108 times for the 3x3 layers, against the 16-lane blocked sums of the CPU kernel it is
    # compared with.  Measured ratios: 1.6-2.7.  At the model level the GPU's gradients are CLOSER to fp64 than the CPU fp32
    # path's (tests/test_grad_parity_gpu.py: 0.62-0.76x).
    assert e_gpu <= 3.0 * e_f32 + 2e-7, (e_gpu, e_f32)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def test_split_planes_exact_and_affine():
    """hi + mid + lo reproduces the fp32 value to 2^-24 relative over 16 binades; the affine + ReLU variant equals
    relu(fma(x, scale, shift)) split the same way."""
    rng = np.random.default_rng(1)
    B, H, C = 3, 12, 64
    x = torch.from_numpy(rng.standard_normal((B, H, H, C)) * np.exp(rng.uniform(-8, 8, (B, H, H, C)))).float().cuda()
    buf = to_planes(x)
    torch.cuda.synchronize()
    back = planes_to_f64(buf, B * H * H, C).view(B, H, H, C)
    rel = ((back - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max()
    assert float(rel) <= 2.0 ** -24, float(rel)
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float().cuda()
    sh = torch.from_numpy(rng.standard_normal(C)).float().cuda()
    x = torch.from_numpy(rng.standard_normal((B, H, H, C))).float().cuda()
    buf = to_planes(x, sc, sh, relu=True)
    torch.cuda.synchronize()
    want = torch.relu(torch.addcmul(sh.double(), x.double(), sc.double()))
    back = planes_to_f64(buf, B * H * H, C).view(B, H, H, C)
    assert float((back - want).abs().max()) < 2e-6


@pytest.mark.parametrize('B,H,cin,cout', [(2, 32, 128, 128), (8, 16, 192, 192), (1, 12, 64, 96), (32, 32, 128, 128), (3, 8, 32, 32),
                                          (2, 16, 160, 64), (5, 32, 192, 128)])
def test_conv3x3_planes_fp32_equivalent(B, H, cin, cout):
    L, _lib, eng = _env()
    rng = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    packed, npad = pack(w.cuda(), cout, cin, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    conv_planes(g, to_planes(nhwc(x)), packed, out)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    check(*errs(out, ref, F.conv2d(x, w, padding=1)))


def test_planes_fused_shortcut_stride2_with_statistics():
    """Down block entry (3x3 stride 2 + fused 1x1 stride-2 shortcut, two outputs) + the BatchNorm (sum, sumsq) epilogue."""
    L, _lib, eng = _env()
    B, H, cin, cout = 4, 32, 128, 192
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * 0.03).float()
    ws = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * 0.09).float()
    packed, npad = pack(w.cuda(), cout, cin, 9)
    packed1, _ = pack(ws.cuda(), cout, cin, 1)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H // 2, cout, cout, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    out = torch.full((B, H // 2, H // 2, cout), float('nan'), device='cuda')
    out1 = torch.full_like(out, float('nan'))
    stats = torch.zeros(cout, 2, dtype=torch.float64, device='cuda')
    conv_planes(g, to_planes(nhwc(x)), packed, out, packed1=packed1, out1=out1, stats=stats)
    ref = F.conv2d(x.double(), w.double(), stride=2, padding=1)
    check(*errs(out, ref, F.conv2d(x, w, stride=2, padding=1)))
    check(*errs(out1, F.conv2d(x.double(), ws.double(), stride=2), F.conv2d(x, ws, stride=2)))
    s_ref = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1)
    err = (stats.cpu() - s_ref).abs() / (s_ref.abs() + ref.abs().max() * (B * H * H / 4) ** 0.5)
    assert float(err.max()) < 1e-4, float(err.max())


@pytest.mark.parametrize('B,H,cin,cout,short', [(4, 16, 192, 128, False), (2, 16, 192, 128, True)])
def test_planes_transposed_stride2_classes(B, H, cin, cout, short):
    """Up block entry: ConvTranspose2d(3, stride 2, pad 1, output_padding 1) as four output-parity classes, alone and
    with the fused transposed 1x1 shortcut (class (0,0) carries the extra tap)."""
    L, _lib, eng = _env()
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cin, cout, 3, 3)) * 0.03).float()       # (Cin, Cout, k, k)
    ws = torch.from_numpy(rng.standard_normal((cin, cout, 1, 1)) * 0.09).float()
    packed, npad = pack(w.cuda(), cout, cin, 9, transposed_layout=True)
    out = torch.full((B, 2 * H, 2 * H, cout), float('nan'), device='cuda')
    fn = lambda a, b: F.conv_transpose2d(a, b, stride=2, padding=1, output_padding=1)
    if not short:
        g = eng._geom(B, H, cin, 2 * H, cout, 0, H, 1, 2, eng._up_classes(False), npad)
        conv_planes(g, to_planes(nhwc(x)), packed, out)
    else:
        packed1, _ = pack(ws.cuda(), cout, cin, 1, transposed_layout=True)
        g = eng._geom(B, H, cin, 2 * H, cout, cout, H, 1, 2, eng._up_classes(True), npad, npad)
        out1 = torch.zeros_like(out)       # the 1x1 stride-2 transposed conv only writes the even pixels
        conv_planes(g, to_planes(nhwc(x)), packed, out, packed1=packed1, out1=out1)
        fs = lambda a, b: F.conv_transpose2d(a, b, stride=2, output_padding=1)
        got = out1.cpu().double().permute(0, 3, 1, 2)[:, :, ::2, ::2]
        ref1 = fs(x.double(), ws.double())[:, :, ::2, ::2]
        f1 = fs(x, ws).double()[:, :, ::2, ::2]
        check(float((got - ref1).abs().max() / ref1.abs().max()), float((f1 - ref1).abs().max() / ref1.abs().max()))
    check(*errs(out, fn(x.double(), w.double()), fn(x, w)))


@pytest.mark.parametrize('B,H,cin,cout', [(2, 32, 128, 128), (8, 16, 192, 192), (4, 32, 32, 128)])
def test_planes_sum_of_two_inputs_and_accumulate(B, H, cin, cout):
    """MPOSE_CONV_SUM_INPUTS (dX = conv_in^T(dC1) + shortcut^T(dSC)) and MPOSE_CONV_ACCUMULATE on the plane engine."""
    L, _lib, eng = _env()
    rng = np.random.default_rng(B + H + cin)
    x0 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w0 = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    w1 = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).float()
    p0, npad = pack(w0.cuda(), cout, cin, 9)
    p1, _ = pack(w1.cuda(), cout, cin, 1)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, cout, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    base = torch.from_numpy(rng.standard_normal((B, H, H, cout))).float().cuda()
    out = base.clone()
    conv_planes(g, to_planes(nhwc(x0)), p0, out, flags=2 | 1, packed1=p1, in1=to_planes(nhwc(x1)))
    fn = lambda a0, a1, v0, v1: F.conv2d(a0, v0, padding=1) + F.conv2d(a1, v1)
    add = base.cpu().permute(0, 3, 1, 2)
    ref = fn(x0.double(), x1.double(), w0.double(), w1.double()) + add.double()
    f32 = fn(x0, x1, w0, w1) + add
    check(*errs(out, ref, f32))


def test_planes_relu_mask_epilogue():
    """Data-gradient epilogue: result * [mask_scale*mask_src + mask_shift > 0], stats = (sum d, sum d*mask_src)."""
    L, _lib, eng = _env()
    B, H, C = 4, 16, 192
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((B, C, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).float()
    src = torch.from_numpy(rng.standard_normal((B, C, H, H))).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float()
    sh = torch.from_numpy(rng.standard_normal(C) * 0.3).float()
    packed, npad = pack(w.cuda(), C, C, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, C), float('nan'), device='cuda')
    stats = torch.zeros(C, 2, dtype=torch.float64, device='cuda')
    conv_planes(g, to_planes(nhwc(x)), packed, out, stats=stats, mask=(nhwc(src), sc.cuda(), sh.cuda()))
    keep = (src.double() * sc.double().view(1, C, 1, 1) + sh.double().view(1, C, 1, 1)) > 0
    ref = F.conv2d(x.double(), w.double(), padding=1) * keep
    f32 = F.conv2d(x, w, padding=1) * keep
    check(*errs(out, ref, f32))
    s_ref = torch.stack([ref.sum((0, 2, 3)), (ref * src.double()).sum((0, 2, 3))], 1)
    err = (stats.cpu() - s_ref).abs() / (s_ref.abs() + ref.abs().max() * (B * H * H) ** 0.5)
    assert float(err.max()) < 1e-4, float(err.max())


@pytest.mark.parametrize('B,H,cin,cout', [(2, 32, 128, 128), (4, 16, 192, 192), (2, 48, 128, 128)])
def test_planes_bf16_single_pass_mode(B, H, cin, cout):
    L, _lib, eng = _env()
    rng = np.random.default_rng(B * 7 + H)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    packed, npad = pack(w.cuda(), cout, cin, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    conv_planes(g, to_planes(nhwc(x)), packed, out, flags=BF16)
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    ref = F.conv2d(xr, wr, padding=1)
    got = out.cpu().double().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5             # exact products, fp32 accumulation
    full = F.conv2d(x.double(), w.double(), padding=1)
    assert float((got - full).abs().max() / full.abs().max()) < 2e-2           # bf16 operand rounding, K = 9*cin terms


def test_bn_add_and_bn_bwd_apply_write_planes():
    """The fused producers: same fp32 results as mpose_bn_add_fwd / mpose_bn_bwd_apply, and planes that add back to them."""
    L, _lib, eng = _env()
    from margipose_amd._lib import BnAddOperands, BnBwdApplyOperands
    B, H, C = 3, 16, 192
    npix = B * H * H
    rng = np.random.default_rng(9)
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s)).float().cuda()
    a, b, gup = mk(B, H, H, C), mk(B, H, H, C), mk(B, H, H, C)
    sa, ta, sb, tb = mk(C), mk(C), mk(C), mk(C)
    coef_a, coef_b = mk(4, C), mk(4, C)
    # bn_add
    want = torch.empty_like(a)
    ao = BnAddOperands()
    ao.a, ao.a_scale, ao.a_shift, ao.b, ao.b_scale, ao.b_shift, ao.out = (t.data_ptr() for t in (a, sa, ta, b, sb, tb, want))
    _lib.check(L.mpose_bn_add_fwd((BnAddOperands * 3)(ao), 1, H * H, B, C, 0, 0, _lib.stream_ptr()), 'bn_add')
    got = torch.empty_like(a)
    planes = torch.zeros(int(L.mpose_planes_bytes(npix, C)), dtype=torch.uint8, device='cuda')
    ao.out = got.data_ptr()
    _lib.check(L.mpose_bn_add_planes((BnAddOperands * 3)(ao), (ctypes.c_void_p * 3)(planes.data_ptr()), 1, ctypes.c_int64(npix), C,
                                     _lib.stream_ptr()), 'bn_add_planes')
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    back = planes_to_f64(planes, npix, C).view(B, H, H, C)
    assert float(((back - want.double()).abs() / want.double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -24
    # bn_bwd_apply (masked branch a + branch b)
    da_w, db_w, da_g, db_g = (torch.empty_like(a) for _ in range(4))
    bo = BnBwdApplyOperands()
    bo.g, bo.a, bo.b, bo.coef_a, bo.coef_b, bo.a_scale, bo.a_shift = (t.data_ptr() for t in (gup, a, b, coef_a, coef_b, sa, ta))
    bo.da, bo.db = da_w.data_ptr(), db_w.data_ptr()
    _lib.check(L.mpose_bn_bwd_apply((BnBwdApplyOperands * 3)(bo), 1, H * H, B, C, 0, 0, _lib.stream_ptr()), 'apply')
    bo.da, bo.db = da_g.data_ptr(), db_g.data_ptr()
    pa = torch.zeros_like(planes)
    pb = torch.zeros_like(planes)
    _lib.check(L.mpose_bn_bwd_apply_planes((BnBwdApplyOperands * 3)(bo), (ctypes.c_void_p * 3)(pa.data_ptr()),
                                           (ctypes.c_void_p * 3)(pb.data_ptr()), 1, ctypes.c_int64(npix), C, _lib.stream_ptr()), 'apply_planes')
    torch.cuda.synchronize()
    assert torch.equal(da_g, da_w) and torch.equal(db_g, db_w)
    for p, t in ((pa, da_w), (pb, db_w)):
        back = planes_to_f64(p, npix, C).view(B, H, H, C)
        assert float(((back - t.double()).abs() / t.double().abs().clamp_min(1e-30)).max()) <= 2.0 ** -24


@pytest.mark.parametrize('B,H,cin,cout,with_f32', [(2, 32, 128, 128, True), (4, 16, 192, 192, False), (3, 8, 32, 32, True), (1, 12, 64, 96, False)])
def test_planes_fused_output_stage(B, H, cin, cout, with_f32):
    """Inference epilogue: y = relu(scale*conv + shift) + (add_scale*add_src + add_shift) written as pre-split planes (what the
    next convolution reads) and optionally as fp32 -- a ResidualBlock's second half (reference models/margipose_model.py:34-40
    with BatchNorm in eval mode) in ONE launch."""
    L, _lib, eng = _env()
    from margipose_amd._lib import ConvOperands
    rng = np.random.default_rng(B * 31 + H)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    mk = lambda *s: torch.from_numpy(rng.standard_normal(s)).float()
    sc, sh, asc, ash = mk(cout).abs() + 0.5, mk(cout) * 0.3, mk(cout), mk(cout) * 0.3
    add = mk(B, cout, H, H)
    packed, npad = pack(w.cuda(), cout, cin, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    npix = B * H * H
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    planes = torch.full((int(L.mpose_planes_bytes(npix, cout)),), 0x7f, dtype=torch.uint8, device='cuda')
    xin = to_planes(nhwc(x))
    dev = [t.cuda() for t in (sc, sh, asc, ash)]
    addg = nhwc(add)
    op = ConvOperands()
    op.in_, op.w0 = xin.data_ptr(), packed.data_ptr()
    if with_f32:
        op.out0 = out.data_ptr()
    op.out0_planes = planes.data_ptr()
    op.epi_scale0, op.epi_shift0 = dev[0].data_ptr(), dev[1].data_ptr()
    op.add_src, op.add_scale, op.add_shift = addg.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, PLANES_IN | 16, _lib.stream_ptr()), 'conv fused')
    torch.cuda.synchronize()
    v = lambda t: t.double().view(1, cout, 1, 1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), padding=1) * v(sc) + v(sh)) + (add.double() * v(asc) + v(ash))
    f32 = torch.relu(F.conv2d(x, w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + (add * asc.view(1, -1, 1, 1) + ash.view(1, -1, 1, 1))
    got_p = planes_to_f64(planes, npix, cout).cpu().view(B, H, H, cout).permute(0, 3, 1, 2)
    scale = ref.abs().max()
    e_p, e_f32 = float((got_p - ref).abs().max() / scale), float((f32.double() - ref).abs().max() / scale)
    check(e_p, e_f32)
    if with_f32:
        got = out.cpu().double().permute(0, 3, 1, 2)
        assert float((got - got_p).abs().max() / scale) <= 2.0 ** -24      # the planes ARE the fp32 result, split
