"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's
soft-argmax / loss tail, forward AND hand-derived backward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (margipose_amd/) never imports anything under oracle/.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
  * the reference's only known-answer test for this path (tests/test_models.py:39-46), and
  * golden vectors generated in the build container by importing the reference itself
    (tools/make_golden.py -> tests/golden/tail_*.npz).

Each function cites the reference lines (relative to /root/reference/src/margipose) it restates.
"""
import numpy as np

EPS_KL = 1e-24      # dsntnn.py:199
EPS_GAUSS = 1e-24   # dsntnn.py:194


def normalized_linspace(length, dtype=np.float64):
    """dsntnn.py:12-36 -- cell-centre coordinates in (-1, 1): i*(2/L) - (L-1)/L."""
    length_f = dtype(length)
    first = -(length_f - dtype(1.0)) / length_f
    return (np.arange(length, dtype=dtype) * (dtype(2.0) / length_f) + first).astype(dtype)


def flat_softmax(logits):
    """dsntnn.py:124-130 -- softmax over all dims after the first two (max-subtracted, as ATen does)."""
    b, j = logits.shape[:2]
    flat = logits.reshape(b, j, -1)
    flat = flat - flat.max(axis=-1, keepdims=True)
    e = np.exp(flat)
    return (e / e.sum(axis=-1, keepdims=True)).reshape(logits.shape)


def dsnt(hm):
    """dsntnn.py:39-62,84-96 -- expectation of the coordinate grid; last dim order (x=width, y=height)."""
    h, w = hm.shape[-2:]
    xs = normalized_linspace(w, hm.dtype.type)
    ys = normalized_linspace(h, hm.dtype.type)
    mu_x = (hm.sum(axis=-2) * xs).sum(axis=-1)   # sum over H first, then weight W (dsntnn.py:55-60)
    mu_y = (hm.sum(axis=-1) * ys).sum(axis=-1)
    return np.stack([mu_x, mu_y], axis=-1)


def heatmaps_to_coords(xy_hm, zy_hm, xz_hm):
    """models/margipose_model.py:254-261 -- x,y from xy; z = mean of zy's width axis and xz's height axis."""
    xy = dsnt(xy_hm)
    zy = dsnt(zy_hm)
    xz = dsnt(xz_hm)
    z = 0.5 * (zy[..., 0:1] + xz[..., 1:2])
    return np.concatenate([xy, z], axis=-1)


def make_gauss(means, size, sigma, normalize=True):
    """dsntnn.py:154-195 -- separable Gaussian at `means` (x, y order), sigma in pixels."""
    h, w = size
    dt = means.dtype.type
    xs = normalized_linspace(w, dt)
    ys = normalized_linspace(h, dt)
    kx = dt(-0.5) * (dt(1.0) / (dt(2.0) * dt(sigma) / dt(w))) ** 2     # dsntnn.py:179-180
    ky = dt(-0.5) * (dt(1.0) / (dt(2.0) * dt(sigma) / dt(h))) ** 2
    ex = np.exp((xs - means[..., 0:1]) ** 2 * kx)       # (..., W)
    ey = np.exp((ys - means[..., 1:2]) ** 2 * ky)       # (..., H)
    g = ey[..., :, None] * ex[..., None, :]
    if not normalize:
        return g
    return g / (g.sum(axis=(-1, -2), keepdims=True) + dt(EPS_GAUSS))


def _kl(p, q):
    """dsntnn.py:198-202."""
    eps = p.dtype.type(EPS_KL)
    return (p * (np.log(p + eps) - np.log(q + eps))).sum(axis=(-1, -2))


def js(p, q):
    """dsntnn.py:205-207."""
    m = 0.5 * (p + q)
    return 0.5 * _kl(p, m) + 0.5 * _kl(q, m)


def js_reg_losses(hm, mu_t, sigma_t):
    """dsntnn.py:210-232."""
    g = make_gauss(mu_t, hm.shape[2:], sigma_t)
    return js(hm, g)


def euclidean_losses(actual, target):
    """dsntnn.py:133-151."""
    d = actual - target
    return np.sqrt((d * d).sum(axis=-1))


def average_loss(losses, mask=None):
    """dsntnn.py:99-121."""
    if mask is None:
        return losses.sum() / max(losses.size, 1)
    return (losses * mask).sum() / max(mask.sum(), 1.0)


def forward_3d_losses(stages, target_xyz, pixelwise=True, sigma=1.0):
    """models/margipose_model.py:236-252.  `stages` = list of (xy_hm, zy_hm, xz_hm)."""
    t_xy = target_xyz[..., [0, 1]]
    t_zy = target_xyz[..., [2, 1]]
    t_xz = target_xyz[..., [0, 2]]
    losses = 0
    for xy, zy, xz in stages:
        if pixelwise:
            losses = losses + js_reg_losses(xy, t_xy, sigma)
            losses = losses + js_reg_losses(zy, t_zy, sigma)
            losses = losses + js_reg_losses(xz, t_xz, sigma)
        losses = losses + euclidean_losses(heatmaps_to_coords(xy, zy, xz), target_xyz)
    return losses


def forward_2d_losses(stages, target, pixelwise=True, sigma=1.0):
    """models/margipose_model.py:223-234."""
    t_xy = target[..., :2]
    losses = 0
    for xy, zy, xz in stages:
        if pixelwise:
            losses = losses + js_reg_losses(xy, t_xy, sigma)
        losses = losses + euclidean_losses(heatmaps_to_coords(xy, zy, xz)[..., :2], t_xy)
    return losses


# ----------------------------------------------------------------------------------------------
# Hand-derived backward (SURVEY.md §8 row a-T).  The reference has no explicit backward (autograd);
# these formulas are pinned against reference autograd through tests/golden/tail_*.npz.
# ----------------------------------------------------------------------------------------------

def js_grad_wrt_p(p, g):
    """d JS(p, g) / d p, elementwise, g treated as constant (targets carry no grad)."""
    eps = p.dtype.type(EPS_KL)
    m = 0.5 * (p + g)
    return 0.5 * (np.log(p + eps) - np.log(m + eps) + p / (p + eps) - 0.5 * (p + g) / (m + eps))


def stage_loss_grad_wrt_heatmaps(xy, zy, xz, target_xyz, dloss, pixelwise=True, sigma=1.0, three_d=True):
    """Gradient of  sum_{b,j} dloss[b,j] * stage_loss[b,j]  w.r.t. the three heatmaps of ONE stage."""
    dt = xy.dtype.type
    h, w = xy.shape[-2:]
    xs = normalized_linspace(w, dt)[None, None, None, :]
    ys = normalized_linspace(h, dt)[None, None, :, None]
    mu = heatmaps_to_coords(xy, zy, xz)
    wgt = dloss[..., None, None]
    if three_d:
        diff = mu - target_xyz
        e = diff / np.sqrt((diff * diff).sum(-1, keepdims=True))
        ex, ey, ez = (e[..., i][..., None, None] for i in range(3))
        g_xy = wgt * (ex * xs + ey * ys)
        g_zy = wgt * (0.5 * ez * xs)          # zy plane: width axis is z
        g_xz = wgt * (0.5 * ez * ys)          # xz plane: height axis is z
        if pixelwise:
            g_xy = g_xy + wgt * js_grad_wrt_p(xy, make_gauss(target_xyz[..., [0, 1]], (h, w), sigma))
            g_zy = g_zy + wgt * js_grad_wrt_p(zy, make_gauss(target_xyz[..., [2, 1]], (h, w), sigma))
            g_xz = g_xz + wgt * js_grad_wrt_p(xz, make_gauss(target_xyz[..., [0, 2]], (h, w), sigma))
    else:
        diff = mu[..., :2] - target_xyz[..., :2]
        e = diff / np.sqrt((diff * diff).sum(-1, keepdims=True))
        ex, ey = (e[..., i][..., None, None] for i in range(2))
        g_xy = wgt * (ex * xs + ey * ys)
        if pixelwise:
            g_xy = g_xy + wgt * js_grad_wrt_p(xy, make_gauss(target_xyz[..., [0, 1]], (h, w), sigma))
        g_zy = np.zeros_like(zy)
        g_xz = np.zeros_like(xz)
    return g_xy, g_zy, g_xz


def softmax_backward(p, g):
    """d/d logits of flat_softmax given upstream g = dL/dp:  p * (g - sum_k p_k g_k)."""
    s = (p * g).sum(axis=(-1, -2), keepdims=True)
    return p * (g - s)
