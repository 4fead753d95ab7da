"""GPU parity of the implicit-GEMM convolution kernel called directly through the C ABI
(mpose_pack_weights + mpose_conv_fwd), against a float64 convolution of the same inputs.

The kernel computes fp32 convolutions on the bf16 matrix cores with 3-way split operands ("bf16x6",
margipose_amd/csrc/conv.hip).  The claim tested here is that this is fp32 arithmetic in everything but the
instruction used: its error against float64 must be no larger than the error of an ordinary fp32 convolution
(torch CPU, fp32) of the same data -- tolerance written below -- on dense random data, which is the worst case
for the dropped 2^-26 cross terms.  Covers stride 1 / 2, the transposed stride-2 parity classes, the fused
shortcut pass, rows that do not fill a 64-pixel tile, and every split-K factor.

Reference layers: src/margipose/models/margipose_model.py:33,36,67-68,73-74 (Conv2d / ConvTranspose2d, no bias).
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pack(L, _lib, eng, w, cout, cin, T, transposed_layout=False):
    """Pack a torch-layout (Cout, Cin, kh, kw) weight for the forward kernel; returns the packed arena."""
    npad = (cout + 63) // 64 * 64
    kpad = (cin + 31) // 32 * 32
    packed = torch.zeros(T * kpad * npad * 3 // 2, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'] = w.data_ptr(), packed.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'] = cout, cin, T, npad, kpad
    if transposed_layout:            # ConvTranspose2d weight is (Cin, Cout, kh, kw)
        j['sn'], j['sk'], j['st'] = T, cout * T, 1
    else:
        j['sn'], j['sk'], j['st'] = cin * T, T, 1
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * kpad * npad, _lib.stream_ptr()), 'pack')
    return packed, npad, kpad


def _run(L, _lib, g, x, packed, out, packed1=None, out1=None):
    from margipose_amd._lib import ConvOperands
    op = ConvOperands()
    op.in_, op.w0, op.out0 = x.data_ptr(), packed.data_ptr(), out.data_ptr()
    if packed1 is not None:
        op.w1, op.out1 = packed1.data_ptr(), out1.data_ptr()
    arr = (ConvOperands * 1)(op)
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), arr, 1, 0, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()


def _errs(got_nhwc, x_nchw, w, fn):
    """(kernel error, torch-fp32 error), both max-abs against float64 and scaled by max |ref|."""
    ref = fn(x_nchw.double(), w.double())
    f32 = fn(x_nchw.float(), w.float()).double()
    scale = ref.abs().max()
    got = got_nhwc.cpu().double().permute(0, 3, 1, 2)
    return float((got - ref).abs().max() / scale), float((f32 - ref).abs().max() / scale)


# tolerance: the kernel may be at most 2x as far from float64 as torch's own fp32 convolution (plus 2e-7 slack for
# tiny cases where both are ~1 ulp); in practice it is closer (fewer, wider accumulation steps).
def _check(e_gpu, e_f32):
    assert e_gpu <= 2.0 * e_f32 + 2e-7, (e_gpu, e_f32)


@pytest.mark.parametrize('B,H,cin,cout', [(2, 32, 128, 128), (8, 16, 192, 192), (1, 12, 64, 96), (32, 32, 128, 128), (3, 8, 32, 32)])
def test_conv3x3_fp32_equivalent(B, H, cin, cout):
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    rng = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    wg = w.cuda()
    packed, npad, kpad = _pack(L, _lib, eng, wg, cout, cin, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    _run(L, _lib, g, xg, packed, out)
    e_gpu, e_f32 = _errs(out, x, w, lambda a, b: torch.nn.functional.conv2d(a, b, padding=1))
    _check(e_gpu, e_f32)


def test_conv_fused_shortcut_and_stride2():
    """Down block entry: 3x3 stride 2 + fused 1x1 stride-2 shortcut from the same input (two outputs, one launch)."""
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    B, H, cin, cout = 4, 32, 128, 192
    rng = np.random.default_rng(7)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * 0.03).float()
    ws = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * 0.09).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, npad, _ = _pack(L, _lib, eng, w.cuda(), cout, cin, 9)
    packed1, _, _ = _pack(L, _lib, eng, ws.cuda(), cout, cin, 1)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H // 2, cout, cout, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    out = torch.full((B, H // 2, H // 2, cout), float('nan'), device='cuda')
    out1 = torch.full((B, H // 2, H // 2, cout), float('nan'), device='cuda')
    _run(L, _lib, g, xg, packed, out, packed1, out1)
    _check(*_errs(out, x, w, lambda a, b: torch.nn.functional.conv2d(a, b, stride=2, padding=1)))
    _check(*_errs(out1, x, ws, lambda a, b: torch.nn.functional.conv2d(a, b, stride=2)))


def test_conv_transposed_stride2_classes():
    """Up block entry: ConvTranspose2d(3, stride 2, pad 1, output_padding 1) as four output-parity classes."""
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    B, H, cin, cout = 4, 16, 192, 128
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((cin, cout, 3, 3)) * 0.03).float()       # (Cin, Cout, k, k)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, npad, _ = _pack(L, _lib, eng, w.cuda(), cout, cin, 9, transposed_layout=True)
    g = eng._geom(B, H, cin, 2 * H, cout, 0, H, 1, 2, eng._up_classes(False), npad)
    out = torch.full((B, 2 * H, 2 * H, cout), float('nan'), device='cuda')
    _run(L, _lib, g, xg, packed, out)
    _check(*_errs(out, x, w, lambda a, b: torch.nn.functional.conv_transpose2d(a, b, stride=2, padding=1, output_padding=1)))


def test_split_planes_are_exact():
    """pack_weights_k's three bf16 planes add back to the fp32 weight to within 2^-24 relative (x = hi + mid + lo)."""
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    cout, cin = 64, 32
    rng = np.random.default_rng(3)
    w = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * np.exp(rng.uniform(-8, 8, (cout, cin, 1, 1)))).float().cuda()
    packed, npad, kpad = _pack(L, _lib, eng, w, cout, cin, 1)
    torch.cuda.synchronize()
    planes = packed.view(torch.bfloat16).view(1, kpad // 16, 3, npad, 2, 8).float()        # [T][K16][plane][n][half][8]
    total = planes.sum(2).double()                                                           # exact in float64
    rebuilt = total.permute(0, 2, 1, 3, 4).reshape(npad, kpad)[:cout, :cin]                  # [n][k]
    ref = w.view(cout, cin).double()
    assert float(((rebuilt - ref).abs() / ref.abs()).max()) <= 2.0 ** -24


@pytest.mark.parametrize('B,H,C', [(2, 32, 128), (8, 16, 192), (2, 8, 32), (1, 12, 64)])
def test_conv_prologue_and_bn_statistics(B, H, C):
    """conv2 of a ResidualBlock: BN+ReLU of the producer applied while staging (in_scale/in_shift), and the
    per-channel (sum, sum of squares) of the output accumulated by the epilogue (fp64 atomics) for the next BatchNorm."""
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(100 + H)
    x = torch.from_numpy(rng.standard_normal((B, C, H, H))).float()
    w = torch.from_numpy(rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float()
    sh = torch.from_numpy(rng.standard_normal(C) * 0.3).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, npad, _ = _pack(L, _lib, eng, w.cuda(), C, C, 9)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, t9)], npad)
    out = torch.full((B, H, H, C), float('nan'), device='cuda')
    stats = torch.zeros(C, 2, dtype=torch.float64, device='cuda')
    scg, shg = sc.cuda(), sh.cuda()
    op = ConvOperands()
    op.in_, op.w0, op.out0 = xg.data_ptr(), packed.data_ptr(), out.data_ptr()
    op.in_scale, op.in_shift, op.stats0 = scg.data_ptr(), shg.data_ptr(), stats.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, 0, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    a = torch.relu(x.double() * sc.double().view(1, C, 1, 1) + sh.double().view(1, C, 1, 1))
    ref = torch.nn.functional.conv2d(a, w.double(), padding=1)
    got = out.cpu().double().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5          # fp32 prologue + fp32-equivalent conv
    s_ref = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1)
    err = (stats.cpu() - s_ref).abs() / (s_ref.abs() + ref.abs().max() * (B * H * H) ** 0.5)
    assert float(err.max()) < 1e-4, float(err.max())


@pytest.mark.parametrize('B,H,cin,cout', [(2, 32, 128, 128), (8, 16, 192, 192), (1, 12, 64, 96), (4, 32, 32, 128)])
def test_conv_sum_of_two_inputs(B, H, cin, cout):
    """MPOSE_CONV_SUM_INPUTS: out = conv3x3(in, w0) + conv1x1(in1, w1) in one launch (the data-gradient of a block input)."""
    from margipose_amd import _lib, engine as eng
    from margipose_amd._lib import ConvOperands
    L = _lib.lib()
    rng = np.random.default_rng(B + H + cin)
    x0 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    w0 = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    w1 = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).float()
    p0, npad, _ = _pack(L, _lib, eng, w0.cuda(), cout, cin, 9)
    p1, _, _ = _pack(L, _lib, eng, w1.cuda(), cout, cin, 1)
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, cout, H, 1, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    x0g, x1g = (t.permute(0, 2, 3, 1).contiguous().cuda() for t in (x0, x1))
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    op = ConvOperands()
    op.in_, op.in1, op.w0, op.w1, op.out0 = x0g.data_ptr(), x1g.data_ptr(), p0.data_ptr(), p1.data_ptr(), out.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, 2, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    fn = lambda a0, a1, v0, v1: torch.nn.functional.conv2d(a0, v0, padding=1) + torch.nn.functional.conv2d(a1, v1)
    ref = fn(x0.double(), x1.double(), w0.double(), w1.double())
    f32 = fn(x0, x1, w0, w1).double()
    got = out.cpu().double().permute(0, 3, 1, 2)
    _check(float((got - ref).abs().max() / ref.abs().max()), float((f32 - ref).abs().max() / ref.abs().max()))


def _wgrad(L, _lib, eng, g, x_nhwc, gout_nhwc, cout, cin, T, npad, n_split, scale=None, shift=None, gout1=None, cout1=0):
    """mpose_conv_wgrad into n_split partial buffers + mpose_unpack_wgrads -> torch-layout (Cout, Cin, kh*kw) gradient(s)."""
    from margipose_amd._lib import WgradOperands
    kpad = (cin + 31) // 32 * 32
    part = torch.full((n_split * T * kpad * npad,), float('nan'), device='cuda')
    wo = WgradOperands()
    wo.in_, wo.gout0, wo.dw0 = x_nhwc.data_ptr(), gout_nhwc.data_ptr(), part.data_ptr()
    if scale is not None:
        wo.in_scale, wo.in_shift = scale.data_ptr(), shift.data_ptr()
    part1 = None
    if gout1 is not None:
        part1 = torch.full((n_split * kpad * npad,), float('nan'), device='cuda')
        wo.gout1, wo.dw1 = gout1.data_ptr(), part1.data_ptr()
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
    """(kernel error, torch-fp32 error) of a weight gradient against float64 autograd, max-abs scaled by max |ref|."""
    def grad(dtype):
        w = torch.zeros(shape, dtype=dtype, requires_grad=True)
        fn(x.to(dtype), w).backward(go.to(dtype))
        return w.grad.double()
    ref, f32 = grad(torch.float64), grad(torch.float32)
    scale = ref.abs().max()
    return float((dw.cpu().double().reshape(shape) - ref).abs().max() / scale), float((f32 - ref).abs().max() / scale)


@pytest.mark.parametrize('B,H,cin,cout,n_split,pro', [(2, 32, 128, 128, 3, False), (4, 16, 192, 192, 2, True), (1, 16, 64, 96, 1, False),
                                                   (3, 8, 32, 32, 4, True), (2, 24, 128, 64, 5, False)])
def test_weight_gradient_fp32_equivalent(B, H, cin, cout, n_split, pro):
    """conv_wgrad_k through mpose_conv_wgrad + mpose_unpack_wgrads: the weight gradient of a 3x3 convolution (optionally
    of relu(scale*x+shift), the BN+ReLU prologue) must be as close to float64 autograd as torch's fp32 gradient is (same
    2x gate as the forward kernel), for every tile shape, rows that do not fill a split, and several split-K factors."""
    import torch.nn.functional as F
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    rng = np.random.default_rng(B * 100 + H + cin)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    go = torch.from_numpy(rng.standard_normal((B, cout, H, H))).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cin)).float()
    sh = torch.from_numpy(rng.standard_normal(cin) * 0.3).float()
    npad = (cout + 63) // 64 * 64
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, t9)], npad)
    xg, gg = x.permute(0, 2, 3, 1).contiguous().cuda(), go.permute(0, 2, 3, 1).contiguous().cuda()
    dw, = _wgrad(L, _lib, eng, g, xg, gg, cout, cin, 9, npad, n_split, sc.cuda() if pro else None, sh.cuda() if pro else None)

    def fn(a, w):
        if pro:
            a = F.relu(a * sc.to(a.dtype).view(1, -1, 1, 1) + sh.to(a.dtype).view(1, -1, 1, 1))
        return F.conv2d(a, w, padding=1)
    e_gpu, e_f32 = _wgrad_errs(dw, x, go, fn, (cout, cin, 3, 3))
    _check(e_gpu, e_f32)


def test_weight_gradient_stride2_with_fused_shortcut():
    """The down block's geometry: 3x3 stride 2 + 1x1 stride-2 shortcut in one launch (acc-1 tap -> second gradient)."""
    import torch.nn.functional as F
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    B, H, cin, cout = 2, 32, 128, 192
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    go = torch.from_numpy(rng.standard_normal((B, cout, H // 2, H // 2))).float()
    go1 = torch.from_numpy(rng.standard_normal((B, cout, H // 2, H // 2))).float()
    npad = (cout + 63) // 64 * 64
    t9 = [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
    g = eng._geom(B, H, cin, H // 2, cout, cout, H // 2, 2, 1, [(0, 0, t9 + [(0, 0, 0, 1)])], npad, npad)
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    dw, dw1 = _wgrad(L, _lib, eng, g, to(x), to(go), cout, cin, 9, npad, 2, gout1=to(go1), cout1=cout)
    _check(*_wgrad_errs(dw, x, go, lambda a, w: F.conv2d(a, w, stride=2, padding=1), (cout, cin, 3, 3)))
    _check(*_wgrad_errs(dw1, x, go1, lambda a, w: F.conv2d(a, w, stride=2), (cout, cin, 1, 1)))


# ---------------------------------------------------------------------------------------------------------------
# Per-axis kernel / stride / dilation / padding (mpose_conv_geom.in_mul_x / out_mul_x): the layer shapes of the reference's
# ChatterboxModel (models/chatterbox_model.py:96-126, 143-150, 189-198), each one forward, data-gradient and weight-gradient
# ---------------------------------------------------------------------------------------------------------------
CHATTERBOX_LAYERS = [
    # (transposed, kernel, stride, dilation, padding, output_padding, src (H, W), flat)
    (False, (3, 3), (1, 2), (1, 1), (1, 1), (0, 0), (16, 16), False),      # _DownBlock conv1, width halved
    (False, (3, 3), (2, 1), (1, 2), (1, 2), (0, 0), (16, 16), False),      # ... height halved, width dilated
    (False, (3, 3), (1, 1), (4, 1), (4, 1), (0, 0), (16, 8), False),       # conv2, dilation (4, 1)
    (False, (1, 1), (1, 2), (1, 1), (0, 0), (0, 0), (16, 16), False),      # resample
    (True, (3, 3), (1, 2), (4, 1), (4, 1), (0, 1), (16, 8), False),        # _UpBlock conv1
    (True, (3, 3), (1, 1), (1, 4), (1, 4), (0, 0), (8, 16), False),        # ... stride 1
    (True, (1, 1), (2, 1), (1, 1), (0, 0), (1, 0), (8, 16), False),        # _UpBlock resample (every other row is zero)
    (False, (1, 8), (1, 8), (1, 1), (0, 0), (0, 0), (16, 8), True),        # Conv2d(512, 1024, (1, 8)) on the 8-wide map
    (True, (1, 8), (1, 8), (1, 1), (0, 0), (0, 0), (16, 1), True),         # ConvTranspose2d(1024, 512, (1, 8))
    (False, (8, 1), (8, 1), (1, 1), (0, 0), (0, 0), (8, 16), False),       # the same pair along the height
    (True, (8, 1), (8, 1), (1, 1), (0, 0), (0, 0), (1, 16), False),
]


@pytest.mark.parametrize('case', range(len(CHATTERBOX_LAYERS)))
def test_per_axis_convolution_geometries(case):
    import torch.nn.functional as F
    from margipose_amd import _lib, engine as eng
    L = _lib.lib()
    tr, k, stride, dil, pad, opad, src_hw, flat = CHATTERBOX_LAYERS[case]
    B, cin, cout = 3, 64, 96
    T = k[0] * k[1]
    rng = np.random.default_rng(300 + case)
    x = torch.from_numpy(rng.standard_normal((B, cin) + src_hw)).float()
    wshape = (cin, cout) + k if tr else (cout, cin) + k
    w = torch.from_numpy(rng.standard_normal(wshape) * (2.0 / (T * cin)) ** 0.5).float()
    # (a full-extent kernel is described with stride = kernel: the same single output position)
    real_stride = (1, 1) if k in ((1, 8), (8, 1)) else stride

    def fn(a, b):
        if tr:
            return F.conv_transpose2d(a, b, None, stride=real_stride, padding=pad, output_padding=opad, dilation=dil)
        return F.conv2d(a, b, None, stride=real_stride, padding=pad, dilation=dil)
    ref = fn(x.double(), w.double())
    dst_hw = tuple(ref.shape[2:])
    go = torch.from_numpy(rng.standard_normal((B, cout) + dst_hw)).float()
    s_hw, d_hw = ((1, src_hw[0] * src_hw[1]), (1, dst_hw[0] * dst_hw[1])) if flat else (src_hw, dst_hw)
    to = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    wg = w.cuda()
    # forward
    packed, npad, _ = _pack(L, _lib, eng, wg, cout, cin, T, transposed_layout=tr)
    gf = eng.conv_geom('f', tr, B, s_hw, cin, d_hw, cout, k, stride, dil, pad, npad)
    out = torch.full((B,) + dst_hw + (cout,), float('nan'), device='cuda')
    _run(L, _lib, gf, to(x), packed, out)
    _check(*_errs(out, x, w, fn))
    # data gradient: K = cout, N = cin, the same taps seen from the other side
    npad_d = (cin + 63) // 64 * 64
    pd = torch.zeros(T * cout * npad_d * 3 // 2, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'], j['N'], j['K'], j['T'], j['Npad'], j['Kpad'] = wg.data_ptr(), pd.data_ptr(), cin, cout, T, npad_d, cout
    j['sn'], j['sk'], j['st'] = (cout * T, T, 1) if tr else (T, cin * T, 1)
    _lib.check(L.mpose_pack_weights(_lib.ptr(eng._jobs_to_device(jobs, 'cuda')), 1, T * cout * npad_d, _lib.stream_ptr()), 'pack')
    gd = eng.conv_geom('d', tr, B, s_hw, cin, d_hw, cout, k, stride, dil, pad, npad_d)
    dx = torch.full((B,) + src_hw + (cin,), float('nan'), device='cuda')
    _run(L, _lib, gd, to(go), pd, dx)

    def dgrad(dtype):
        a = x.to(dtype).requires_grad_(True)
        fn(a, w.to(dtype)).backward(go.to(dtype))
        return a.grad.double()
    r64, r32 = dgrad(torch.float64), dgrad(torch.float32)
    sc = r64.abs().max()
    _check(float((dx.cpu().double().permute(0, 3, 1, 2) - r64).abs().max() / sc), float((r32 - r64).abs().max() / sc))
    # weight gradient (unpacked as (Cout, Cin, taps); a ConvTranspose2d stores (Cin, Cout, ...))
    dw, = _wgrad(L, _lib, eng, gf, to(x), to(go), cout, cin, T, npad, 2)

    def wgradient(dtype):
        b = w.to(dtype).requires_grad_(True)
        fn(x.to(dtype), b).backward(go.to(dtype))
        g_ = b.grad.double().reshape(wshape[0], wshape[1], T)
        return g_.permute(1, 0, 2) if tr else g_
    r64, r32 = wgradient(torch.float64), wgradient(torch.float32)
    sc = r64.abs().max()
    _check(float((dw.cpu().double() - r64).abs().max() / sc), float((r32 - r64).abs().max() / sc))
