"""Piece control for the gradient checks (test infrastructure, like everything under oracle/): read the ReLU masks and
max-pool window choices the HIP path actually used, and differentiate the oracle on the SAME piece of the piecewise-linear network.

The network is piecewise linear in its ReLUs.  Two correct fp32 implementations whose pre-activations differ in the last bit pick
different pieces wherever a pre-activation sits at ~0, and the gradients of neighbouring pieces differ by O(1e-3) of a tensor's
norm.  These helpers separate "different piece" from "arithmetic error".  Used by tests/test_grad_parity_gpu.py,
tests/test_model_gpu.py and __graft_entry__.smoke() (the shipped entry point must not import the tests package).
Reference: the sites are the nn.ReLU / MaxPool2d modules of src/margipose/models/margipose_model.py:25-40,104-118."""
from collections import OrderedDict

import numpy as np
import torch

from oracle import model_ref as R
from oracle import weights as W


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def build(T, seed, x, stem='patch8'):
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64, stem=stem), x.double(), T)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, stem, 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    return m.cuda().train(), sd


def gpu_step(m, x, target, mask):
    """One forward + loss + backward of the HIP path.  Returns (gradients by parameter name + '__dx__', masks of the piece it
    ran on, loss, coordinates): ONE forward, so a training-mode model's running statistics are updated once."""
    from margipose_amd import dsntnn
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    ectx = m.xy_heatmaps[0].grad_fn.ectx               # the engine's saved activations of THIS forward
    masks = gpu_relu_masks(m.inner.engine(), ectx)      # (before backward: the BatchNorm vectors are this forward's)
    if m.inner.engine().stem is not None:
        masks.update(gpu_stem_masks(m, m.inner.engine(), ectx))
    loss = dsntnn.average_loss(m.forward_3d_losses(out, target.cuda()), mask.cuda())
    loss.backward()
    grads = OrderedDict((k, p.grad.detach().cpu()) for k, p in m.named_parameters())
    grads['__dx__'] = xg.grad.cpu()
    return grads, masks, float(loss.detach()), out.detach().cpu()


def gpu_relu_masks(eng, ectx):
    """site -> (B, C, H, W) bool tensor: the sign test the kernels evaluate, fmaf(x, scale, shift) > 0, reproduced
    exactly in fp64 (x*scale is exact there and one rounding cannot change the sign of a non-zero sum)."""
    masks = {}

    def site_mask(raw, n):
        C, Cs = n.C, n.Cs
        sc = eng.bnf[n.f_off:n.f_off + C].double()
        sh = eng.bnf[n.f_off + Cs:n.f_off + Cs + C].double()
        return ((raw[..., :C].double() * sc + sh) > 0).permute(0, 3, 1, 2).contiguous().cpu()

    for t, saved in enumerate(ectx['blocks']):
        for i, sv in enumerate(saved):
            for c, plane in enumerate(R.PLANES):
                b = eng.stage_blocks[t][i][c]
                pre = 'inner.%s_hm_cnns.%d.%s.%d' % (plane, t, 'down_layers' if i < 5 else 'up_layers', i % 5)
                masks[pre + '.relu1'] = site_mask(sv['c1'][c], b.bn1)
                masks[pre + '.relu2'] = site_mask(sv['c2'][c], b.bn2)
    if eng.stem is None:
        masks['inner.in_cnn.relu'] = site_mask(ectx['stem_raw'], eng.stem_bn)
    return masks


def gpu_stem_masks(m, eng, ectx):
    """The InceptionV4 feature extractor's pieces (round 4): every BasicConv2d ReLU as the sign of fmaf(raw, scale, shift) of the
    node's saved pre-activation, and the window choice of the two max-pools taken on the activations the pooling kernel sees
    (the fp64 product-sum rounded once to fp32 = its fmaf; torch's return_indices convention)."""
    import torch.nn.functional as F
    from margipose_amd import stem as S
    st = eng.stem
    assert isinstance(st, S.InceptionV4Stem)
    names = {id(mod): name for name, mod in m.named_modules()}
    raw = ectx['stem_ctx']['raw']
    masks = {}

    def pre(n):         # (B, H, W, C) fp64: scale * raw + shift of the whole node (pooled channel ranges carry scale 1, shift 0)
        sc = st.f_arena[n.f_off:n.f_off + n.C].double()
        sh = st.f_arena[n.f_off + n.C:n.f_off + 2 * n.C].double()
        return raw[n.name].double() * sc + sh

    for n in st.nodes:
        if n.is_image or not any(p[2] is not None for p in n.parts):
            continue
        v = pre(n)
        for (a, b, bn, eps, bias) in n.parts:
            if bn is not None:
                masks[names[id(bn)] + '.relu'] = (v[..., a:b] > 0).permute(0, 3, 1, 2).contiguous().cpu()
    pools = [op for op in st.ops if isinstance(op, S._PoolOp) and op.kind == 0]
    assert len(pools) == 2
    for op, site in zip(pools, ('inner.in_cnn.3.maxpool', 'inner.in_cnn.5.maxpool')):
        act = pre(op.src).float().clamp_min(0).permute(0, 3, 1, 2).contiguous()
        masks[site] = F.max_pool2d(act, 3, stride=2, padding=1, return_indices=True)[1].cpu()
    return masks


def oracle_grads(sd, T, x, target, mask, dtype, masks=None, record=None):
    sd = OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
    xr = x.detach().to(dtype).clone().requires_grad_(True)
    R.RELU_MASKS, R.RELU_RECORD = masks, record
    try:
        xy, zy, xz = R.inner_forward(sd, xr, T, True)
        loss = R.average_loss(R.forward_3d_losses(xy, zy, xz, target.to(dtype)), mask.to(dtype))
        loss.backward()
    finally:
        R.RELU_MASKS, R.RELU_RECORD = None, None
    g = OrderedDict((k, p.grad) for k, p in params.items())
    g['__dx__'] = xr.grad
    return g, float(loss)


def oracle_grads_pair(sd, T, x, target, mask, masks=None):
    """The fp64 and the fp32 oracle pass of one case side by side (two host threads: the passes are independent, ATen releases
    the GIL, and one CPU backward pass at B=32 leaves most of the box's cores idle).  Returns (g64, loss64, g32)."""
    import threading
    out = {}

    def run(dtype):
        s_ = OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())
        params = OrderedDict((k, v.requires_grad_(True)) for k, v in s_.items() if v.is_floating_point() and 'running' not in k)
        xr = x.detach().to(dtype).clone().requires_grad_(True)
        xy, zy, xz = R.inner_forward(s_, xr, T, True)
        loss = R.average_loss(R.forward_3d_losses(xy, zy, xz, target.to(dtype)), mask.to(dtype))
        loss.backward()
        g = OrderedDict((k, p.grad) for k, p in params.items())
        g['__dx__'] = xr.grad
        out[dtype] = (g, float(loss))

    R.RELU_MASKS, R.RELU_RECORD = masks, None          # (read-only for both threads)
    try:
        ths = [threading.Thread(target=run, args=(dt,)) for dt in (torch.float64, torch.float32)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    finally:
        R.RELU_MASKS, R.RELU_RECORD = None, None
    if len(out) != 2:
        raise RuntimeError('an oracle pass failed')
    return out[torch.float64][0], out[torch.float64][1], out[torch.float32][0]
