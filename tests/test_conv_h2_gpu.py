"""GPU parity of the H2 convolution engine (csrc/conv_h.hip, MPOSE_CONV_H2_IN) through the C ABI:
mpose_absmax -> mpose_split_h2 (activations as two fp16 planes of x * 2^k, layout H8[C/8][2][npix][8]) + mpose_pack_weights
(layout 3) -> mpose_conv_fwd.

Claims under test: (1) the engine is fp32 arithmetic in everything but the instruction, held to the gate of the other engines
(tests/test_conv_gpu.py: at most 2x the error of torch's own fp32 convolution against float64), on normal, ReLU, 1e-7-sized,
1e+4-sized and heavy-tailed data, at image borders, batch ends and ragged last tiles; (2) a LOOSE amax slot (the bound a producer
works with, 2^8 above the true maximum) costs no measurable precision; (3) its epilogues -- BatchNorm statistics, channel
extremes, ReLU mask + BatchNorm-backward sums, consumer sums, output amax -- equal float64 sums of what it stored, both as fp64
atomics and as per-workgroup partial rows (MPOSE_CONV_STATS_PART); (4) the fused shortcut (second output) and the two-input sum.

Reference layers: src/margipose/models/margipose_model.py:33,36,67-68 (regular ResidualBlock convolutions) and their autograd
data-gradients."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
F16X3, H2, PART = 32, 128, 256
SLOT = 16 * 64


def _lib_eng():
    from margipose_amd import _lib, engine as eng
    return _lib.lib(), _lib, eng


def _amax(tensors, C, scale=None, shift=None, relu=False, loosen=1.0):
    L, _lib, _ = _lib_eng()
    from margipose_amd._lib import AbsmaxOperands
    slots = torch.zeros(len(tensors), SLOT, dtype=torch.float32, device='cuda')
    ops = []
    for i, t in enumerate(tensors):
        ao = AbsmaxOperands()
        ao.src, ao.dst = t.data_ptr(), slots[i].data_ptr()
        if scale is not None:
            ao.scale, ao.shift = scale.data_ptr(), shift.data_ptr()
        ops.append(ao)
    _lib.check(L.mpose_absmax((AbsmaxOperands * len(ops))(*ops), len(ops), ctypes.c_int64(tensors[0].numel() // C), C, int(relu),
                              _lib.stream_ptr()), 'absmax')
    return slots * loosen           # (a bound above the true maximum: what a producer that cannot measure first works with)


def _split(tensors, slots, C, scale=None, shift=None, relu=False):
    L, _lib, _ = _lib_eng()
    from margipose_amd._lib import SplitH2Operands
    npix = tensors[0].numel() // C
    assert int(L.mpose_h2_bytes(ctypes.c_int64(npix), C)) == npix * C * 4
    planes = [torch.empty(npix * C, dtype=torch.float32, device='cuda') for _ in tensors]
    ops = []
    for t, pl, sl in zip(tensors, planes, slots):
        so = SplitH2Operands()
        so.src, so.planes, so.amax = t.data_ptr(), pl.data_ptr(), sl.data_ptr()
        if scale is not None:
            so.scale, so.shift = scale.data_ptr(), shift.data_ptr()
        ops.append(so)
    _lib.check(L.mpose_split_h2((SplitH2Operands * len(ops))(*ops), len(ops), ctypes.c_int64(npix), C, int(relu), _lib.stream_ptr()), 'split_h2')
    return planes


def _pack(w, cout, cin, T, layout=3):
    L, _lib, eng = _lib_eng()
    npad = (cout + 63) // 64 * 64
    packed = torch.zeros(T * cin * npad * 3 // 2, dtype=torch.float32, device='cuda')
    amax = torch.zeros(1, dtype=torch.float32, device='cuda')
    jobs = np.zeros(1, dtype=eng.PACK_DT)
    j = jobs[0]
    j['src'], j['dst'], j['amax'] = w.data_ptr(), packed.data_ptr(), amax.data_ptr()
    j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['layout'] = cout, cin, T, npad, cin, layout
    j['sn'], j['sk'], j['st'] = cin * T, T, 1
    dev = eng._jobs_to_device(jobs, 'cuda')
    _lib.check(L.mpose_weights_absmax(_lib.ptr(dev), 1, _lib.stream_ptr()), 'weights_absmax')
    _lib.check(L.mpose_pack_weights(_lib.ptr(dev), 1, T * cin * npad, _lib.stream_ptr()), 'pack')
    return packed, amax, npad


def _data(rng, shape, kind):
    if kind == 'normal':
        return rng.standard_normal(shape)
    if kind == 'tiny':
        return rng.standard_normal(shape) * 1e-7
    if kind == 'huge':
        return rng.standard_normal(shape) * 1e4
    if kind == 'heavy':
        return rng.standard_normal(shape) * np.exp(3.0 * rng.standard_normal(shape))
    if kind == 'relu':
        return np.maximum(rng.standard_normal(shape), 0.0)
    raise ValueError(kind)


def _errs(got_nhwc, ref, f32):
    scale = ref.abs().max()
    got = got_nhwc.cpu().double().permute(0, 3, 1, 2)
    return float((got - ref).abs().max() / scale), float((f32.double() - ref).abs().max() / scale)


def _check(e_gpu, e_f32):
    assert e_gpu <= 2.0 * e_f32 + 2e-7, (e_gpu, e_f32)


T9 = lambda eng: [(ky - 1, kx - 1, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]
T9D = lambda eng: [(1 - ky, 1 - kx, ky * 3 + kx, 0) for ky, kx in eng.TAPS3]


def test_split_h2_planes_rebuild_the_tensor():
    """h + l == x * 2^k to 2^-22 relative (down to fp16's subnormal spacing), largest magnitude below 2^15, layout H8."""
    rng = np.random.default_rng(3)
    for npix, C, loosen in ((300, 64, 1.0), (1024, 128, 1.0), (77, 32, 256.0)):
        x = torch.from_numpy(_data(rng, (npix, C), 'heavy')).float().cuda()
        slot = _amax([x], C, loosen=loosen)
        pl, = _split([x], slot, C)
        torch.cuda.synchronize()
        amax = float(slot.max())
        k = 14 - int(np.floor(np.log2(amax)))
        h8 = pl.view(torch.float16).view(C // 8, 2, npix, 8).double().cpu()
        rebuilt = h8.sum(1).permute(1, 0, 2).reshape(npix, C)
        ref = x.double().cpu() * 2.0 ** k
        assert float(h8[:, 0].abs().max()) < 2.0 ** 15
        assert bool(((rebuilt - ref).abs() <= 2.0 ** -21 * ref.abs() + 2.0 ** -24).all())


@pytest.mark.parametrize('B,H,cin,cout,kind,loosen', [(2, 32, 128, 128, 'normal', 1.0), (8, 16, 192, 192, 'relu', 1.0), (3, 10, 64, 64, 'heavy', 1.0),
                                                     (32, 32, 128, 128, 'tiny', 1.0), (3, 8, 32, 32, 'huge', 1.0), (1, 48, 128, 64, 'normal', 1.0),
                                                     (2, 32, 128, 128, 'normal', 256.0), (5, 12, 96, 32, 'relu', 64.0)])
def test_conv3x3_fp32_equivalent(B, H, cin, cout, kind, loosen):
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import ConvOperands
    rng = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(_data(rng, (B, cin, H, H), kind)).float()
    w = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    packed, w_amax, npad = _pack(w.cuda(), cout, cin, 9)
    x_amax = _amax([xg], cin, loosen=loosen)
    xh, = _split([xg], x_amax, cin)
    g = eng._geom(B, H, cin, H, cout, 0, H, 1, 1, [(0, 0, T9(eng))], npad)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.in_amax, op.w0_amax = xh.data_ptr(), packed.data_ptr(), out.data_ptr(), x_amax.data_ptr(), w_amax.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3 | H2, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    _check(*_errs(out, F.conv2d(x.double(), w.double(), padding=1), F.conv2d(x, w, padding=1)))


def test_h2_rejects_what_it_does_not_do():
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import ConvOperands
    x = torch.zeros(1, 8, 8, 32, device='cuda'); out = torch.zeros(1, 8, 8, 32, device='cuda')
    packed, w_amax, npad = _pack(torch.zeros(32, 32, 3, 3, device='cuda'), 32, 32, 9)
    slot = _amax([x], 32)
    g = eng._geom(1, 8, 32, 8, 32, 0, 8, 1, 1, [(0, 0, T9(eng))], npad)
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.in_amax, op.w0_amax = x.data_ptr(), packed.data_ptr(), out.data_ptr(), slot.data_ptr(), w_amax.data_ptr()
    assert L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, H2, _lib.stream_ptr()) == -22             # needs MPOSE_CONV_F16X3
    assert L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3 | H2 | 1, _lib.stream_ptr()) == -22  # no accumulate
    op.in_scale, op.in_shift = slot.data_ptr(), slot.data_ptr()
    assert L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3 | H2, _lib.stream_ptr()) == -22      # no prologue
    op.in_scale = op.in_shift = None
    # an all-zero tensor (amax 0) is legal and yields zeros
    out.fill_(float('nan'))
    xh, = _split([x], slot, 32)
    op.in_ = xh.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3 | H2, _lib.stream_ptr()), 'conv')
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize('B,H,C,part', [(2, 32, 128, False), (2, 32, 128, True), (8, 16, 192, True), (3, 10, 64, False), (3, 10, 64, True)])
def test_forward_block_launches(B, H, C, part):
    """The two forward launches of a regular ResidualBlock: 3x3 + fused 1x1 shortcut on one input (two outputs, two sets of
    statistics, channel extremes of the first), then the 3x3 over relu(bn1(.)) written as planes by mpose_split_h2."""
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import ConvOperands
    rng = np.random.default_rng(H + C)
    x = torch.from_numpy(rng.standard_normal((B, C, H, H)) * 3.0).float()
    w = torch.from_numpy(rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).float()
    ws = torch.from_numpy(rng.standard_normal((C, C, 1, 1)) * 5.0).float()                 # (own scale: 100x the main weights)
    xg = x.permute(0, 2, 3, 1).contiguous().cuda()
    p3, a3, npad = _pack(w.cuda(), C, C, 9)
    p1, a1, _ = _pack(ws.cuda(), C, C, 1)
    xa = _amax([xg], C)
    xh, = _split([xg], xa, C)
    g = eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, T9(eng) + [(0, 0, 0, 1)])], npad, npad)
    out0 = torch.full((B, H, H, C), float('nan'), device='cuda'); out1 = out0.clone()
    rows = -(-(B * H * H) // 64)
    if part:
        st0, st1, mm = (torch.full((4 + rows * C * 2,), float('nan'), device='cuda') for _ in range(3))
    else:
        st0 = torch.zeros(C, 2, dtype=torch.float64, device='cuda'); st1 = st0.clone()
        mm = torch.zeros(C, 2, dtype=torch.int32, device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.w1, op.out0, op.out1 = xh.data_ptr(), p3.data_ptr(), p1.data_ptr(), out0.data_ptr(), out1.data_ptr()
    op.in_amax, op.w0_amax, op.w1_amax = xa.data_ptr(), a3.data_ptr(), a1.data_ptr()
    op.stats0, op.stats1, op.mm0 = st0.data_ptr(), st1.data_ptr(), mm.data_ptr()
    flags = F16X3 | H2 | (PART if part else 0)
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, flags, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    _check(*_errs(out0, F.conv2d(x.double(), w.double(), padding=1), F.conv2d(x, w, padding=1)))
    _check(*_errs(out1, F.conv2d(x.double(), ws.double()), F.conv2d(x, ws)))

    def sums(buf, k):
        if not part:
            return buf.cpu().double()
        n = int(buf[:1].view(torch.int32))
        assert n == int(L.mpose_conv_stat_rows(ctypes.byref(g), (ConvOperands * 1)(op), 1, flags)) and 0 < n <= rows
        return buf[4:4 + n * C * k].view(n, C, k).cpu().double()
    for buf, o in ((st0, out0), (st1, out1)):
        o64 = o.cpu().double().view(-1, C)
        s = sums(buf, 2)
        s = s.sum(0) if part else s
        ref = torch.stack([o64.sum(0), (o64 * o64).sum(0)], 1)
        assert float(((s - ref).abs() / (ref.abs() + o64.abs().max() * (B * H * H) ** 0.5)).max()) < 1e-5
    o64 = out0.cpu().double().view(-1, C)
    if part:
        m = sums(mm, 2).max(0).values
        got_max, got_neg = m[:, 0], m[:, 1]
    else:
        key = mm.cpu().view(torch.int32).long() & 0xffffffff
        f = torch.where(key >= 2 ** 31, key - 2 ** 31, (~key) & 0xffffffff).to(torch.int32).view(torch.float32).double()
        got_max, got_neg = f[:, 0], f[:, 1]
    assert torch.equal(got_max, o64.max(0).values) and torch.equal(got_neg, (-o64).max(0).values)

    # second launch: relu(bn1(out0)) as planes -> 3x3
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float().cuda(); sh = torch.from_numpy(rng.standard_normal(C) * 0.3).float().cuda()
    aa = _amax([out0], C, sc, sh, relu=True)
    ah, = _split([out0], aa, C, sc, sh, relu=True)
    g2 = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, T9(eng))], npad)
    out2 = torch.full((B, H, H, C), float('nan'), device='cuda')
    op2 = ConvOperands()
    op2.in_, op2.w0, op2.out0, op2.in_amax, op2.w0_amax = ah.data_ptr(), p3.data_ptr(), out2.data_ptr(), aa.data_ptr(), a3.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g2), (ConvOperands * 1)(op2), 1, F16X3 | H2, _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    a32 = torch.relu(torch.addcmul(sh.cpu().view(1, C, 1, 1), out0.cpu().permute(0, 3, 1, 2), sc.cpu().view(1, C, 1, 1)))
    _check(*_errs(out2, F.conv2d(a32.double(), w.double(), padding=1), F.conv2d(a32, w, padding=1)))


@pytest.mark.parametrize('B,H,C,part', [(2, 32, 128, True), (4, 16, 192, False), (3, 10, 64, True)])
def test_data_gradient_with_relu_mask_and_bn_sums(B, H, C, part):
    """The data-gradient of a block's second 3x3: flipped taps, the ReLU mask of the first BatchNorm applied to what is stored,
    and stats0 = (sum d, sum d * mask_src) per channel -- the BatchNorm-backward sums of that BatchNorm."""
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import ConvOperands
    rng = np.random.default_rng(7 * H + C)
    go = torch.from_numpy(_data(rng, (B, C, H, H), 'tiny')).float()
    w = torch.from_numpy(rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).float()       # (out, in, ky, kx) of the forward conv
    c1 = torch.from_numpy(rng.standard_normal((B, C, H, H))).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, C)).float(); sh = torch.from_numpy(rng.standard_normal(C) * 0.3).float()
    gg = go.permute(0, 2, 3, 1).contiguous().cuda(); c1g = c1.permute(0, 2, 3, 1).contiguous().cuda()
    # dgrad weights: N = the forward's input channels, K = its output channels
    wd = w.permute(1, 0, 2, 3).contiguous()
    p3, a3, npad = _pack(wd.cuda(), C, C, 9)
    ga = _amax([gg], C)
    gh, = _split([gg], ga, C)
    g = eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, T9D(eng))], npad)
    out = torch.full((B, H, H, C), float('nan'), device='cuda')
    rows = -(-(B * H * H) // 64)
    st = torch.full((4 + rows * C * 2,), float('nan'), device='cuda') if part else torch.zeros(C, 2, dtype=torch.float64, device='cuda')
    op = ConvOperands()
    op.in_, op.w0, op.out0, op.in_amax, op.w0_amax = gh.data_ptr(), p3.data_ptr(), out.data_ptr(), ga.data_ptr(), a3.data_ptr()
    scg, shg = sc.cuda(), sh.cuda()
    op.mask_src, op.mask_scale, op.mask_shift, op.stats0 = c1g.data_ptr(), scg.data_ptr(), shg.data_ptr(), st.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, F16X3 | H2 | (PART if part else 0), _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    mask = (torch.addcmul(sh.view(1, C, 1, 1), c1, sc.view(1, C, 1, 1)) > 0)
    fn = lambda a, b: F.conv_transpose2d(a, b, padding=1) * mask.to(a.dtype)
    _check(*_errs(out, fn(go.double(), w.double()), fn(go, w)))
    o64 = out.cpu().double().view(-1, C); x64 = c1g.cpu().double().view(-1, C)
    ref = torch.stack([o64.sum(0), (o64 * x64).sum(0)], 1)
    if part:
        n = int(st[:1].view(torch.int32))
        got = st[4:4 + n * C * 2].view(n, C, 2).cpu().double().sum(0)
    else:
        got = st.cpu()
    assert float(((got - ref).abs() / (ref.abs() + o64.abs().max() * x64.abs().max() * (B * H * H) ** 0.5)).max()) < 1e-5


@pytest.mark.parametrize('B,H,cin,cout,ratio,part', [(2, 32, 128, 128, 1.0, False), (2, 32, 128, 128, 0.3, True), (4, 16, 192, 192, 1e-9, False),
                                                      (3, 10, 64, 64, 1e6, False), (3, 10, 64, 64, 2.0, True), (2, 16, 64, 64, 0.0, False),
                                                      (5, 8, 32, 32, 1.0, True)])
def test_sum_of_two_inputs_with_consumer_sums_and_amax(B, H, cin, cout, ratio, part):
    """MPOSE_CONV_SUM_INPUTS (dX = conv_in^T(dC1) + shortcut^T(dSC)) with the inputs at very different magnitudes (0: an all-zero
    second input is dropped, not overflowed), red_* = the consumer's four BatchNorm-backward sums of what is stored, out0_amax.
    Round 6: the stride-1 shapes run on conv_h2r_k<., 2> (two K loops, the epilogue through LDS; ragged last tiles at H = 10 and 8,
    the 32-channel tile), red_sums as fp64 atomics and as per-workgroup partial rows."""
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import ConvOperands
    rng = np.random.default_rng(B + H + cin)
    x0 = torch.from_numpy(rng.standard_normal((B, cin, H, H))).float()
    x1 = torch.from_numpy(rng.standard_normal((B, cin, H, H)) * ratio).float()
    w0 = torch.from_numpy(rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).float()
    w1 = torch.from_numpy(rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).float()
    ra = torch.from_numpy(rng.standard_normal((B, cout, H, H))).float(); rb = torch.from_numpy(rng.standard_normal((B, cout, H, H))).float()
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout)).float(); sh = torch.from_numpy(rng.standard_normal(cout) * 0.3).float()
    p0, wa0, npad = _pack(w0.cuda(), cout, cin, 9)
    p1, wa1, _ = _pack(w1.cuda(), cout, cin, 1)
    g = eng._geom(B, H, cin, H, cout, cout, H, 1, 1, [(0, 0, T9(eng) + [(0, 0, 0, 1)])], npad, npad)
    x0g, x1g, rag, rbg = (t.permute(0, 2, 3, 1).contiguous().cuda() for t in (x0, x1, ra, rb))
    xa = _amax([x0g, x1g], cin)
    x0h, x1h = _split([x0g, x1g], xa, cin)
    out = torch.full((B, H, H, cout), float('nan'), device='cuda')
    rows = -(-(B * H * H) // 64)
    red = torch.full((4 + rows * cout * 4,), float('nan'), device='cuda') if part else torch.zeros(cout, 4, dtype=torch.float64, device='cuda')
    oam = torch.zeros(SLOT, device='cuda')
    scg, shg = sc.cuda(), sh.cuda()
    op = ConvOperands()
    op.in_, op.in1, op.w0, op.w1, op.out0 = x0h.data_ptr(), x1h.data_ptr(), p0.data_ptr(), p1.data_ptr(), out.data_ptr()
    op.in_amax, op.in1_amax, op.w0_amax, op.w1_amax = xa[0].data_ptr(), xa[1].data_ptr(), wa0.data_ptr(), wa1.data_ptr()
    op.red_a, op.red_b, op.red_scale, op.red_shift, op.red_sums = rag.data_ptr(), rbg.data_ptr(), scg.data_ptr(), shg.data_ptr(), red.data_ptr()
    op.out0_amax = oam.data_ptr()
    _lib.check(L.mpose_conv_fwd(ctypes.byref(g), (ConvOperands * 1)(op), 1, 2 | F16X3 | H2 | (PART if part else 0), _lib.stream_ptr()), 'conv')
    torch.cuda.synchronize()
    if part:
        n = int(red[:1].view(torch.int32))
        assert 0 < n <= rows
        red = red[4:4 + n * cout * 4].view(n, cout, 4).double().sum(0)
    fn = lambda a0, a1, v0, v1: F.conv2d(a0, v0, padding=1) + F.conv2d(a1, v1)
    _check(*_errs(out, fn(x0.double(), x1.double(), w0.double(), w1.double()), fn(x0, x1, w0, w1)))
    o64 = out.cpu().double().view(-1, cout); a64 = rag.cpu().double().view(-1, cout); b64 = rbg.cpu().double().view(-1, cout)
    m = (torch.addcmul(shg.cpu().view(1, -1), rag.cpu().view(-1, cout), scg.cpu().view(1, -1)) > 0).double()
    ref = torch.stack([(o64 * m).sum(0), (o64 * m * a64).sum(0), o64.sum(0), (o64 * b64).sum(0)], 1)
    tol = o64.abs().max() * max(1.0, float(a64.abs().max())) * (B * H * H) ** 0.5
    assert float(((red.cpu() - ref).abs() / (ref.abs() + tol)).max()) < 1e-5
    assert float(oam.max()) == float(out.abs().max())


@pytest.mark.parametrize('B,H,n_split,single,loosen', [(2, 32, 3, False, 1.0), (3, 16, 2, False, 64.0), (1, 32, 1, True, 1.0)])
def test_weight_gradient_from_plane_operands(B, H, n_split, single, loosen):
    """mpose_wgrad_operands.planes_in (round 6): the row-of-taps weight gradient of a regular 128-channel block reading BOTH operands
    as the H8 planes their producers wrote -- the 3x3 + fused 1x1 shortcut launch (two gradients, ten taps) and the plain 3x3 --
    against torch's float64 / float32 gradients (the gate of tests/test_conv_f16x3_gpu.py), with a loose bound in the slots,
    several pixel splits, an odd number of images, and the single-product (fp16-rounded) arithmetic; and bit for bit against the
    same launch on the fp32 tensors whose planes those are (same scale exponents, same two pieces: same products)."""
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import WgradOperands
    C = 128
    rng = np.random.default_rng(B * 10 + H)
    x = torch.from_numpy(_data(rng, (B, C, H, H), 'relu')).float()
    g0 = torch.from_numpy(_data(rng, (B, C, H, H), 'heavy') * 1e-3).float()
    g1 = torch.from_numpy(rng.standard_normal((B, C, H, H)) * 1e-2).float()
    xg, g0g, g1g = (t.permute(0, 2, 3, 1).contiguous().cuda() for t in (x, g0, g1))
    slots = _amax([xg, g0g, g1g], C, loosen=loosen)
    xh, g0h, g1h = _split([xg, g0g, g1g], slots, C)
    npad = 128
    geom = eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, T9(eng) + [(0, 0, 0, 1)])], npad, npad)

    def run(planes):
        p0 = torch.full((n_split * 9 * C * npad,), float('nan'), device='cuda'); p1 = torch.full((n_split * C * npad,), float('nan'), device='cuda')
        wo = WgradOperands()
        src = (xh, g0h, g1h) if planes else (xg, g0g, g1g)
        wo.in_, wo.gout0, wo.gout1, wo.dw0, wo.dw1 = src[0].data_ptr(), src[1].data_ptr(), src[2].data_ptr(), p0.data_ptr(), p1.data_ptr()
        wo.in_amax, wo.gout0_amax, wo.gout1_amax = slots[0].data_ptr(), slots[1].data_ptr(), slots[2].data_ptr()
        wo.single_product, wo.planes_in = int(single), int(planes)
        _lib.check(L.mpose_conv_wgrad(ctypes.byref(geom), (WgradOperands * 1)(wo), 1, n_split, _lib.stream_ptr()), 'wgrad')
        outs = []
        for p, t in ((p0, 9), (p1, 1)):
            dw = torch.full((C, C, t), float('nan'), device='cuda')
            jobs = np.zeros(1, dtype=eng.UNPACK_DT)
            j = jobs[0]
            j['src'], j['dst'] = p.data_ptr(), dw.data_ptr()
            j['N'], j['K'], j['T'], j['Npad'], j['Kpad'], j['n_split'] = C, C, t, npad, C, n_split
            j['sn'], j['sk'], j['st'], j['accumulate'] = C * t, t, 1, 0
            dev = eng._jobs_to_device(jobs, 'cuda')
            _lib.check(L.mpose_unpack_wgrads(_lib.ptr(dev), 1, C * C * t, _lib.stream_ptr()), 'unpack')
            outs.append(dw)
        torch.cuda.synchronize()
        return outs
    dw3, dw1 = run(True)
    f3, f1 = run(False)
    assert torch.equal(dw3, f3) and torch.equal(dw1, f1)

    def errs(dw, go, fn, shape):
        def grad(dtype):
            w = torch.zeros(shape, dtype=dtype, requires_grad=True)
            fn(x.to(dtype), w).backward(go.to(dtype))
            return w.grad.double()
        ref, f32 = grad(torch.float64), grad(torch.float32)
        scale = ref.abs().max()
        return float((dw.cpu().double().reshape(shape) - ref).abs().max() / scale), float((f32 - ref).abs().max() / scale)
    if not single:
        _check(*errs(dw3, g0, lambda a, w: F.conv2d(a, w, padding=1), (C, C, 3, 3)))
        _check(*errs(dw1, g1, lambda a, w: F.conv2d(a, w), (C, C, 1, 1)))
    else:          # fp16-rounded operands: 2^-11 per factor
        e3, _ = errs(dw3, g0, lambda a, w: F.conv2d(a, w, padding=1), (C, C, 3, 3))
        assert e3 < 2e-3, e3
    # what the row form does not take as planes is refused, not computed on misread bytes
    wo = WgradOperands()
    g_bad = eng._geom(B, H, 192, H, 192, 0, H, 1, 1, [(0, 0, T9(eng))], 192)
    big = torch.zeros(B * H * H * 192, device='cuda')
    wo.in_, wo.gout0, wo.dw0 = big.data_ptr(), big.data_ptr(), torch.zeros(9 * 192 * 192, device='cuda').data_ptr()
    wo.in_amax, wo.gout0_amax, wo.planes_in = slots[0].data_ptr(), slots[1].data_ptr(), 1
    assert L.mpose_conv_wgrad(ctypes.byref(g_bad), (WgradOperands * 1)(wo), 1, 1, _lib.stream_ptr()) != 0


@pytest.mark.parametrize('B,H', [(2, 32), (3, 16), (2, 24)])
def test_single_product_mode_reads_the_h_planes(B, H):
    """MPOSE_CONV_F16X1 | MPOSE_CONV_H2_IN (round 6: conv_h2r_k<., ., X1>): the fp16-rounded mode on producer-split planes -- the kernel
    reads the h planes alone (half the DMA and fragment reads, one product) -- against conv_igemm_k's MPOSE_CONV_F16X1 on the fp32
    tensors those planes were split from: the same fp16-rounded operands (same slots, same scale exponents), exact products, fp32
    accumulation in another order.  All three launch kinds of a regular block: the 3x3, the 3x3 + fused 1x1 shortcut (two outputs),
    the two-input sum; image widths whose halo tile has an odd and an even number of 64-row groups."""
    L, _lib, eng = _lib_eng()
    from margipose_amd._lib import ConvOperands
    C, X1 = 128, 64
    rng = np.random.default_rng(H)
    x0 = torch.from_numpy(_data(rng, (B, H, H, C), 'relu')).float().cuda()
    x1 = torch.from_numpy(rng.standard_normal((B, H, H, C)) * 0.1).float().cuda()
    w3 = torch.from_numpy(rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5).float().cuda()
    w1 = torch.from_numpy(rng.standard_normal((C, C, 1, 1)) * (2.0 / C) ** 0.5).float().cuda()
    slots = _amax([x0, x1], C)
    x0h, x1h = _split([x0, x1], slots, C)
    packs = {lay: (_pack(w3, C, C, 9, lay), _pack(w1, C, C, 1, lay)) for lay in (2, 3)}
    geoms = {0: eng._geom(B, H, C, H, C, 0, H, 1, 1, [(0, 0, T9(eng))], 128),
             1: eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, T9(eng) + [(0, 0, 0, 1)])], 128, 128),
             2: eng._geom(B, H, C, H, C, C, H, 1, 1, [(0, 0, T9D(eng) + [(0, 0, 0, 1)])], 128, 128)}

    def run(mode, h2):
        (p3, a3, _), (p1, a1, _) = packs[3 if h2 else 2]
        out0 = torch.full((B, H, H, C), float('nan'), device='cuda'); out1 = out0.clone()
        op = ConvOperands()
        op.in_, op.in_amax = (x0h if h2 else x0).data_ptr(), slots[0].data_ptr()
        op.w0, op.w0_amax, op.out0 = p3.data_ptr(), a3.data_ptr(), out0.data_ptr()
        if mode:
            op.w1, op.w1_amax = p1.data_ptr(), a1.data_ptr()
        if mode == 1:
            op.out1 = out1.data_ptr()
        if mode == 2:
            op.in1, op.in1_amax = (x1h if h2 else x1).data_ptr(), slots[1].data_ptr()
        flags = F16X3 | X1 | (H2 if h2 else 0) | (2 if mode == 2 else 0)
        _lib.check(L.mpose_conv_fwd(ctypes.byref(geoms[mode]), (ConvOperands * 1)(op), 1, flags, _lib.stream_ptr()), 'conv')
        torch.cuda.synchronize()
        return out0, out1
    for mode in (0, 1, 2):
        a0, a1_ = run(mode, True)
        b0, b1_ = run(mode, False)
        assert bool(torch.isfinite(a0).all())
        assert float((a0 - b0).abs().max() / b0.abs().max()) < 2e-6, mode
        if mode == 1:
            assert float((a1_ - b1_).abs().max() / b1_.abs().max()) < 2e-6
    # and it IS the reduced-precision arithmetic: 2^-11-sized roundings against the fp32-equivalent form
    ref = F.conv2d(x0.permute(0, 3, 1, 2).double().cpu(), w3.double().cpu(), padding=1).permute(0, 2, 3, 1)
    e = float((run(0, True)[0].cpu().double() - ref).abs().max() / ref.abs().max())
    assert 1e-6 < e < 3e-3, e
