"""MI355X implementation of the reference's DSNT operator module.

Same names, argument meaning and error behaviour as reference src/margipose/dsntnn.py, backed by the
gfx950 kernels in csrc/tail.hip through the C ABI (include/margipose_hip.h).  All tensors must be
contiguous float32 ROCm-device tensors; there is no CPU path.
"""
import torch

from . import _lib
from ._lib import c_float, c_int, check, dev_f32, lib, ptr, ptr_array, stream_ptr


def _rows_hw(t):
    if t.dim() != 4:
        raise _lib.MposeError('expected a (B, J, H, W) tensor, got shape %s' % (tuple(t.shape),))
    b, j, h, w = t.shape
    if w % 4 != 0 or h * w > (1 << 24):      # (rows beyond 4096 elements run csrc/tail.hip's multi-pass kernels)
        raise _lib.MposeError('heatmap size %dx%d unsupported (need W %% 4 == 0)' % (h, w))
    return b * j, h, w


def _normalized_linspace(length, dtype=None, device=None):
    """reference dsntnn.py:12-36."""
    if isinstance(length, torch.Tensor):
        length = length.to(device, dtype)
    first = -(length - 1.0) / length
    return torch.arange(length, dtype=dtype, device=device) * (2.0 / length) + first


# ---------------------------------------------------------------------------------------------
class _FlatSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp):
        x = dev_f32(inp.contiguous(), 'inp')
        rows, h, w = _rows_hw(x)
        out = torch.empty_like(x)
        check(lib().mpose_softmax_dsnt_fwd(ptr_array([x]), ptr_array([out]), None, None, 1, rows, h, w, 0,
                                           stream_ptr()), 'mpose_softmax_dsnt_fwd')
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad):
        (p,) = ctx.saved_tensors
        g = dev_f32(grad.contiguous(), 'grad')
        rows, h, w = _rows_hw(p)
        d = torch.empty_like(p)
        check(lib().mpose_softmax_bwd(ptr_array([p]), ptr_array([g]), None, ptr_array([d]), 1, rows, h * w,
                                      stream_ptr()), 'mpose_softmax_bwd')
        return d


def flat_softmax(inp):
    """Softmax with all but the first two dimensions combined (reference dsntnn.py:124-130)."""
    return _FlatSoftmax.apply(inp)


class _Dsnt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heatmaps):
        hm = dev_f32(heatmaps.contiguous(), 'heatmaps')
        rows, h, w = _rows_hw(hm)
        mu = torch.empty(hm.shape[0], hm.shape[1], 2, dtype=torch.float32, device=hm.device)
        check(lib().mpose_dsnt_fwd(ptr_array([hm]), ptr(mu), None, 1, rows, h, w, stream_ptr()), 'mpose_dsnt_fwd')
        ctx.shape = hm.shape
        return mu

    @staticmethod
    def backward(ctx, grad):
        g = dev_f32(grad.contiguous(), 'grad')
        b, j, h, w = ctx.shape
        d = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().mpose_dsnt_bwd(ptr(g), ptr_array([d]), 1, b * j, h, w, 0, stream_ptr()), 'mpose_dsnt_bwd')
        return d


def dsnt(heatmaps):
    """Differentiable spatial to numerical transform (reference dsntnn.py:84-96): (B,J,H,W) -> (B,J,2)."""
    return _Dsnt.apply(heatmaps)


class _HeatmapsToCoords(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xy_hm, zy_hm, xz_hm):
        hms = [dev_f32(t.contiguous(), 'heatmap') for t in (xy_hm, zy_hm, xz_hm)]
        rows, h, w = _rows_hw(hms[0])
        for t in hms[1:]:
            if t.shape != hms[0].shape:
                raise _lib.MposeError('heatmap shapes differ')
        xyz = torch.empty(hms[0].shape[0], hms[0].shape[1], 3, dtype=torch.float32, device=hms[0].device)
        check(lib().mpose_dsnt_fwd(ptr_array(hms), None, ptr(xyz), 3, rows, h, w, stream_ptr()), 'mpose_dsnt_fwd')
        ctx.shape = hms[0].shape
        return xyz

    @staticmethod
    def backward(ctx, grad):
        g = dev_f32(grad.contiguous(), 'grad')
        b, j, h, w = ctx.shape
        zero = torch.zeros_like(g[..., 0])
        # models/margipose_model.py:259: z = 0.5 * (zy.x + xz.y)
        dpc = torch.stack([torch.stack([g[..., 0], g[..., 1]], -1),
                           torch.stack([0.5 * g[..., 2], zero], -1),
                           torch.stack([zero, 0.5 * g[..., 2]], -1)], 0).contiguous()
        outs = [torch.empty(ctx.shape, dtype=torch.float32, device=g.device) for _ in range(3)]
        check(lib().mpose_dsnt_bwd(ptr(dpc), ptr_array(outs), 3, b * j, h, w, 0, stream_ptr()), 'mpose_dsnt_bwd')
        return tuple(outs)


def heatmaps_to_coords(xy_hm, zy_hm, xz_hm):
    """reference models/margipose_model.py:254-261 as one launch."""
    return _HeatmapsToCoords.apply(xy_hm, zy_hm, xz_hm)


def average_loss(losses, mask=None):
    """Average of per-location losses (reference dsntnn.py:99-121)."""
    if mask is not None:
        assert mask.size() == losses.size(), 'mask must be the same size as losses'
    return _AverageLoss.apply(losses, mask)


class _AverageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, losses, mask):
        l = dev_f32(losses.contiguous(), 'losses')
        m = dev_f32(mask.contiguous(), 'mask') if mask is not None else None
        out2 = torch.empty(2, dtype=torch.float32, device=l.device)
        check(lib().mpose_average_loss_fwd(ptr(l), ptr(m), ptr(out2), l.numel(), stream_ptr()),
              'mpose_average_loss_fwd')
        ctx.save_for_backward(out2, m)
        ctx.shape = l.shape
        return out2[0]

    @staticmethod
    def backward(ctx, grad):
        out2, m = ctx.saved_tensors
        g = dev_f32(grad.contiguous(), 'grad')
        d = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)      # mask * (grad / denominator), one launch
        check(lib().mpose_average_loss_bwd(ptr(g), ptr(out2), ptr(m), ptr(d), _lib.c_int64(d.numel()), stream_ptr()),
              'mpose_average_loss_bwd')
        return d, None


class _AddLosses(torch.autograd.Function):
    """a + b of two per-location loss tensors (a is None: 0 + b) -- the reference's `losses = 0; losses += ...` over stages
    (models/margipose_model.py:238-252) as launches of the library, so that a launch plan records them."""

    @staticmethod
    def forward(ctx, a, b):
        bb = dev_f32(b.contiguous(), 'losses')
        aa = dev_f32(a.contiguous(), 'losses') if a is not None else None
        if aa is not None and aa.shape != bb.shape:
            raise _lib.MposeError('loss tensors of different shapes')
        out = torch.empty_like(bb)
        check(lib().mpose_add_f32(ptr(aa), ptr(bb), ptr(out), _lib.c_int64(bb.numel()), stream_ptr()), 'mpose_add_f32')
        ctx.has_a = aa is not None
        return out

    @staticmethod
    def backward(ctx, grad):
        return (grad if ctx.has_a else None), grad


def add_losses(a, b):
    return _AddLosses.apply(a, b)


def euclidean_losses(actual, target):
    """Per-point Euclidean distance (reference dsntnn.py:133-151).  (B,L,D)-sized glue, a handful of
    elements per sample; the training path uses the fused stage-loss kernel instead."""
    assert actual.size() == target.size(), 'input tensors must have the same size'
    diff = actual - target
    return diff.pow(2).sum(-1, keepdim=False).sqrt()


def make_gauss(means, size, sigma, normalize=True):
    """Separable Gaussians of standard deviation `sigma` PIXELS centred at `means` (normalised coordinates, ordered x, y, z, ...
    while `size` is ordered [..., depth, height, width]) on a grid of `size` cells -- reference dsntnn.py:154-195, any number of
    dimensions, differentiable w.r.t. `means`.  API-surface helper built from device tensor ops: the loss kernels regenerate
    their targets in registers (csrc/tail.hip) and never call this."""
    n = len(size)
    if means.size(-1) != n:
        raise _lib.MposeError('make_gauss: %d-dimensional means for a %d-dimensional grid' % (means.size(-1), n))
    out = None
    for axis in range(n):                       # axis 0 = x = the LAST grid dimension
        length = size[n - 1 - axis]
        grid = _normalized_linspace(length, dtype=means.dtype, device=means.device)
        inv_std = length / (2.0 * sigma)         # 1 / (sigma pixels in normalised units)
        factor = torch.exp(-0.5 * ((grid - means[..., axis:axis + 1]) * inv_std) ** 2)      # (..., length)
        shape = list(factor.shape[:-1]) + [1] * n
        shape[len(shape) - 1 - axis] = length
        factor = factor.reshape(shape)
        out = factor if out is None else out * factor
    if not normalize:
        return out
    total = out.sum(dim=tuple(range(-n, 0)), keepdim=True)
    return out / (total + 1e-24)


def _js_from_tensors(p, q, ndims):
    """reference dsntnn.py:198-207 with device tensor ops (differentiable in both arguments)."""
    eps = 1e-24
    m = 0.5 * (p + q)
    dims = tuple(range(-ndims, 0))
    kl_pm = (p * ((p + eps).log() - (m + eps).log())).sum(dims)
    kl_qm = (q * ((q + eps).log() - (m + eps).log())).sum(dims)
    return 0.5 * kl_pm + 0.5 * kl_qm


class _JsRegLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heatmaps, mu_t, sigma_t):
        hm = dev_f32(heatmaps.contiguous(), 'heatmaps')
        mu = dev_f32(mu_t.contiguous(), 'mu_t')
        rows, h, w = _rows_hw(hm)
        js = torch.empty(hm.shape[:2], dtype=torch.float32, device=hm.device)
        check(lib().mpose_js_fwd(ptr(hm), ptr(mu), ptr(js), rows, h, w, c_float(sigma_t), stream_ptr()), 'mpose_js_fwd')
        ctx.save_for_backward(hm, mu)
        ctx.sigma = float(sigma_t)
        return js

    @staticmethod
    def backward(ctx, grad):
        hm, mu = ctx.saved_tensors
        g = dev_f32(grad.contiguous(), 'grad')
        rows, h, w = _rows_hw(hm)
        d = torch.empty_like(hm)
        check(lib().mpose_js_bwd(ptr(hm), ptr(mu), ptr(g), ptr(d), rows, h, w, c_float(ctx.sigma), stream_ptr()),
              'mpose_js_bwd')
        return d, None, None


def js_reg_losses(heatmaps, mu_t, sigma_t):
    """Jensen-Shannon divergence between heatmaps and target Gaussians (reference dsntnn.py:220-232).
    2D heatmaps with constant targets -- every reference call site -- run the fused kernel (Gaussians regenerated in registers).
    The rest of the reference's API (a gradient w.r.t. `mu_t`, 1D / 3D heatmaps) is off the hot path and is composed from
    make_gauss + device tensor ops, so autograd reaches the means as it does in the reference."""
    ndims = mu_t.size(-1)
    assert heatmaps.dim() == ndims + 2, 'expected heatmaps to be a {}D tensor'.format(ndims + 2)
    assert heatmaps.size()[:-ndims] == mu_t.size()[:-1]
    if ndims != 2 or (mu_t.requires_grad and torch.is_grad_enabled()):
        if not heatmaps.is_cuda:
            raise _lib.MposeError('heatmaps must live on a ROCm device: margipose_amd has no CPU path')
        return _js_from_tensors(heatmaps, make_gauss(mu_t, heatmaps.size()[2:], sigma_t), ndims)
    return _JsRegLosses.apply(heatmaps, mu_t, sigma_t)


class _StageLoss(torch.autograd.Function):
    """One stage of forward_3d_losses / forward_2d_losses (models/margipose_model.py:223-252) fused:
    3 x JS + DSNT + z-merge + Euclidean in one launch, analytic backward in one launch."""

    @staticmethod
    def forward(ctx, xy_hm, zy_hm, xz_hm, target, sigma, pixelwise, three_d):
        hms = [dev_f32(t.contiguous(), 'heatmap') for t in (xy_hm, zy_hm, xz_hm)]
        rows, h, w = _rows_hw(hms[0])
        tgt = dev_f32(target.contiguous(), 'target')
        if tgt.shape != (hms[0].shape[0], hms[0].shape[1], 3):
            raise _lib.MposeError('target must be (B, J, 3)')
        losses = torch.empty(hms[0].shape[:2], dtype=torch.float32, device=hms[0].device)
        xyz = torch.empty(hms[0].shape[0], hms[0].shape[1], 3, dtype=torch.float32, device=hms[0].device)
        check(lib().mpose_stage_loss_fwd(ptr_array(hms), ptr(tgt), ptr(losses), ptr(xyz), rows, h, w, c_float(sigma),
                                         int(pixelwise), int(three_d), 0, stream_ptr()), 'mpose_stage_loss_fwd')
        ctx.save_for_backward(hms[0], hms[1], hms[2], tgt, xyz)
        ctx.cfg = (float(sigma), int(pixelwise), int(three_d))
        return losses

    @staticmethod
    def backward(ctx, grad):
        xy, zy, xz, tgt, xyz = ctx.saved_tensors
        sigma, pixelwise, three_d = ctx.cfg
        g = dev_f32(grad.contiguous(), 'grad')
        rows, h, w = _rows_hw(xy)
        outs = [torch.empty_like(xy) for _ in range(3)]
        check(lib().mpose_stage_loss_bwd(ptr_array([xy, zy, xz]), ptr(tgt), ptr(xyz), ptr(g), ptr_array(outs), rows, h, w,
                                         c_float(sigma), pixelwise, three_d, 0, stream_ptr()), 'mpose_stage_loss_bwd')
        return outs[0], outs[1], outs[2], None, None, None, None


def stage_losses(xy_hm, zy_hm, xz_hm, target_xyz, sigma=1.0, pixelwise=True, three_d=True):
    return _StageLoss.apply(xy_hm, zy_hm, xz_hm, target_xyz, sigma, pixelwise, three_d)
