"""ORACLE (test infrastructure, NOT product code) -- functional PyTorch-CPU restatement of the
reference MargiPose backbone + loss graph ("the reference PyTorch CPU path" of BASELINE.json).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (margipose_amd/) never imports anything under oracle/.

Why torch and not numpy here: the reference's arithmetic for this part IS stock ATen CPU
(conv2d / conv_transpose2d / batch_norm / softmax + autograd); restating it over a flat
`state_dict` with torch.nn.functional keeps the identical kernels (MKL-DNN) and gives the backward
for free.  The tail also exists as an independent numpy restatement with a hand-derived backward
(oracle/tail_np.py).

Parity status: PINNED against the imported reference (tools/make_golden.py ->
tests/golden/{column,model}_*.npz, checked by tests/test_oracle_golden.py).
The stem is the in-repo deterministic `patch8` stem (one 8x8/stride-8 conv + BN + ReLU), NOT the
reference's InceptionV4 stem, whose arithmetic lives in the absent third-party package
pretrainedmodels==0.6.0 (requirements.txt:8): stem parity is UNPINNED and says so everywhere.

Citations are relative to /root/reference/src/margipose.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
PLANES = ('xy', 'zy', 'xz')

# Mask control for gradient-parity experiments (tests/test_grad_parity_gpu.py).  ReLU makes the network piecewise
# linear; two correct fp32 implementations that disagree in the last bit of a pre-activation sitting at ~0 choose
# different pieces, and the gradients of the two pieces differ by O(1e-3), not O(1e-7).  With RELU_MASKS set to
# {site: 0/1 tensor} the oracle evaluates relu(x) as x * mask, i.e. it is forced onto the piece another
# implementation chose, which separates "different piece" from "arithmetic error".  RELU_RECORD (a dict) receives
# the oracle's own masks.  Sites are named '<block prefix>.relu1' / '.relu2', 'inner.in_cnn.relu' (patch8 stem), and for the
# InceptionV4 stem '<BatchNorm module name>.relu' plus the two max-pools 'inner.in_cnn.3.maxpool' / 'inner.in_cnn.5.maxpool'
# (window choices, see _maxpool3s2).
RELU_MASKS = None
RELU_RECORD = None


def _relu(x, site):
    if RELU_RECORD is not None:
        RELU_RECORD[site] = (x > 0).detach()
    if RELU_MASKS is not None and site in RELU_MASKS:
        return x * RELU_MASKS[site].to(x.dtype)
    return F.relu(x)


def _maxpool3s2(x, site):
    """max_pool2d(3, stride 2, padding 1) with the same piece control as _relu: RELU_MASKS[site], when present, holds the flat
    H*W index (torch's return_indices convention) of the element every window selects; RELU_RECORD receives the oracle's own."""
    if RELU_RECORD is not None:
        RELU_RECORD[site] = F.max_pool2d(x.detach(), 3, stride=2, padding=1, return_indices=True)[1]
    if RELU_MASKS is not None and site in RELU_MASKS:
        idx = RELU_MASKS[site].to(x.device)
        return x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    return F.max_pool2d(x, 3, stride=2, padding=1)


def _bn(sd, key, x, train):
    """nn.BatchNorm2d defaults (models/margipose_model.py:31,34,37): batch stats + running update in
    train mode, running stats in eval mode."""
    return F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'],
                        sd[key + '.weight'], sd[key + '.bias'], training=train,
                        momentum=BN_MOMENTUM, eps=BN_EPS)


def _conv_in(sd, key, x, kind, k):
    w = sd[key + '.weight']
    pad = k // 2
    if kind == 'regular':
        return F.conv2d(x, w, None, stride=1, padding=pad)
    if kind == 'down':
        return F.conv2d(x, w, None, stride=2, padding=pad)
    if kind == 'up':   # ConvTranspose2d(k, padding=k//2, stride=2, output_padding=1): :79-82
        return F.conv_transpose2d(x, w, None, stride=2, padding=pad, output_padding=1)
    raise ValueError(kind)


def residual_block(sd, prefix, x, kind, train):
    """models/margipose_model.py:25-40 -- [conv_in,BN,ReLU,conv3x3,BN,ReLU](x) + [conv_sc,BN](x)."""
    m = _conv_in(sd, prefix + '.module.0', x, kind, 3)
    m = _relu(_bn(sd, prefix + '.module.1', m, train), prefix + '.relu1')
    m = F.conv2d(m, sd[prefix + '.module.3.weight'], None, stride=1, padding=1)
    m = _relu(_bn(sd, prefix + '.module.4', m, train), prefix + '.relu2')
    s = _conv_in(sd, prefix + '.shortcut.0', x, kind, 1)
    s = _bn(sd, prefix + '.shortcut.1', s, train)
    return m + s


DOWN_KINDS = ('regular', 'regular', 'down', 'regular', 'regular')   # :47-53
UP_KINDS = ('regular', 'regular', 'up', 'regular', 'regular')       # :54-60


def axis_permute(mid, space):
    """models/margipose_model.py:91-99 -- swap (chunk-channel <-> W) for zy, (chunk-channel <-> H) for xz."""
    if space == 'xy':
        return mid
    b, c, h, w = mid.shape
    assert h == w and c % h == 0
    v = mid.reshape(b, c // h, h, h, w)          # (b, chunk, i, h, w)
    if space == 'zy':
        v = v.permute(0, 1, 4, 3, 2)             # out[b,k,i,h,j] = in[b,k,j,h,i]
    else:
        v = v.permute(0, 1, 3, 2, 4)             # out[b,k,i,j,w] = in[b,k,j,i,w]
    return v.reshape(b, c, h, w)


def heatmap_column(sd, prefix, x, space, train):
    """models/margipose_model.py:84-100."""
    for i, kind in enumerate(DOWN_KINDS):
        x = residual_block(sd, '%s.down_layers.%d' % (prefix, i), x, kind, train)
    x = axis_permute(x, space)
    for i, kind in enumerate(UP_KINDS):
        x = residual_block(sd, '%s.up_layers.%d' % (prefix, i), x, kind, train)
    return x


def patch8_stem(sd, x, train):
    """In-repo deterministic stem (NOT the reference's InceptionV4 stem; parity unpinned)."""
    f = F.conv2d(x, sd['inner.in_cnn.0.weight'], None, stride=8)
    return _relu(_bn(sd, 'inner.in_cnn.1', f, train), 'inner.in_cnn.relu')


def _basic_conv(sd, key, x, train, stride=1):
    """pretrainedmodels' BasicConv2d after the reference's padding rewrite (models/margipose_model.py:111-117):
    Conv2d(bias=False, padding=k//2) -> BatchNorm2d(eps=1e-3) -> ReLU."""
    w = sd[key + '.conv.weight']
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[2] // 2, w.shape[3] // 2))
    y = F.batch_norm(y, sd[key + '.bn.running_mean'], sd[key + '.bn.running_var'], sd[key + '.bn.weight'], sd[key + '.bn.bias'],
                     training=train, momentum=BN_MOMENTUM, eps=1e-3)
    return _relu(y, key + '.bn.relu')


def inceptionv4_stem(sd, x, train):
    """models/margipose_model.py:104-118 with inceptionv4().features[0:7] restated from SURVEY.md Appendix B
    (the third-party source, pretrainedmodels==0.6.0, is not in the reference tree: parity UNPINNED)."""
    p = 'inner.in_cnn.'
    x = _basic_conv(sd, p + '0', x, train, 2)
    x = _basic_conv(sd, p + '1', x, train)
    x = _basic_conv(sd, p + '2', x, train)
    x = torch.cat([_maxpool3s2(x, p + '3.maxpool'), _basic_conv(sd, p + '3.conv', x, train, 2)], 1)                     # Mixed_3a
    b0 = _basic_conv(sd, p + '4.branch0.1', _basic_conv(sd, p + '4.branch0.0', x, train), train)
    b1 = x
    for i in range(4):
        b1 = _basic_conv(sd, p + '4.branch1.%d' % i, b1, train)
    x = torch.cat([b0, b1], 1)                                                                                          # Mixed_4a
    x = torch.cat([_basic_conv(sd, p + '5.conv', x, train, 2), _maxpool3s2(x, p + '5.maxpool')], 1)                      # Mixed_5a
    b0 = _basic_conv(sd, p + '6.branch0', x, train)
    b1 = _basic_conv(sd, p + '6.branch1.1', _basic_conv(sd, p + '6.branch1.0', x, train), train)
    b2 = x
    for i in range(3):
        b2 = _basic_conv(sd, p + '6.branch2.%d' % i, b2, train)
    b3 = _basic_conv(sd, p + '6.branch3.1', F.avg_pool2d(x, 3, stride=1, padding=1, count_include_pad=False), train)
    x = torch.cat([b0, b1, b2, b3], 1)                                                                                  # Inception_A
    y = F.conv2d(x, sd[p + '7.weight'], sd[p + '7.bias'])
    return _relu(_bn(sd, p + '8', y, train), p + '8.relu')


def resnet_stem(sd, x, train):
    """models/margipose_model.py:119-137: torchvision resnet{18,34,50} conv1, bn1, relu, maxpool, layer1, layer2 (+ the 1x1
    head when layer2 has more than 128 channels).  torchvision==0.3.0 is not in the reference tree: the blocks are restated
    from the published architecture (BasicBlock: 3x3-3x3; Bottleneck: 1x1-3x3(stride)-1x1, expansion 4) -- parity UNPINNED."""
    p = 'inner.in_cnn.'
    x = F.relu(_bn(sd, p + '1', F.conv2d(x, sd[p + '0.weight'], None, stride=2, padding=3), train))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for idx, stride in ((4, 1), (5, 2)):
        b = 0
        while '%s%d.%d.conv1.weight' % (p, idx, b) in sd:
            q = '%s%d.%d.' % (p, idx, b)
            st = stride if b == 0 else 1
            if q + 'conv3.weight' in sd:          # Bottleneck
                y = F.relu(_bn(sd, q + 'bn1', F.conv2d(x, sd[q + 'conv1.weight']), train))
                y = F.relu(_bn(sd, q + 'bn2', F.conv2d(y, sd[q + 'conv2.weight'], None, stride=st, padding=1), train))
                y = _bn(sd, q + 'bn3', F.conv2d(y, sd[q + 'conv3.weight']), train)
            else:                                  # BasicBlock
                y = F.relu(_bn(sd, q + 'bn1', F.conv2d(x, sd[q + 'conv1.weight'], None, stride=st, padding=1), train))
                y = _bn(sd, q + 'bn2', F.conv2d(y, sd[q + 'conv2.weight'], None, padding=1), train)
            if q + 'downsample.0.weight' in sd:
                x = _bn(sd, q + 'downsample.1', F.conv2d(x, sd[q + 'downsample.0.weight'], None, stride=st), train)
            x = F.relu(y + x)
            b += 1
    if p + '6.weight' in sd:
        x = F.relu(_bn(sd, p + '7', F.conv2d(x, sd[p + '6.weight'], sd[p + '6.bias']), train))
    return x


def stem_forward(sd, x, train):
    if 'inner.in_cnn.0.conv.weight' in sd:
        return inceptionv4_stem(sd, x, train)
    if 'inner.in_cnn.4.0.conv1.weight' in sd:
        return resnet_stem(sd, x, train)
    return patch8_stem(sd, x, train)


def flat_softmax(x):
    """dsntnn.py:124-130."""
    return F.softmax(x.flatten(2), dim=-1).view_as(x)


def inner_forward(sd, x, n_stages, train, axis_permutation=True, heatmap_dtype=None, unrounded=None):
    """models/margipose_model.py:179-200 -- returns three lists (xy, zy, xz) of per-stage heatmaps.
    heatmap_dtype=torch.bfloat16 restates BASELINE configs[1] ("bf16 heatmaps + fp32 soft-argmax"): every stage's heatmaps
    are rounded to bf16 where they are stored (what the next combiner reads); `unrounded` (a dict) receives the last
    stage's softmax before rounding, from which the fp32 soft-argmax coordinates are taken."""
    inp = stem_forward(sd, x, train)
    spaces = PLANES if axis_permutation else ('xy', 'xy', 'xy')
    outs = {p: [] for p in PLANES}
    for t in range(n_stages):
        if t > 0:
            cat = torch.cat([outs[p][t - 1] for p in PLANES], dim=1)
            inp = inp + F.conv2d(cat, sd['inner.hm_combiners.%d.conv.weight' % (t - 1)])   # :145-150,:195
        for p, space in zip(PLANES, spaces):
            logits = heatmap_column(sd, 'inner.%s_hm_cnns.%d' % (p, t), inp, space, train)
            hm = flat_softmax(logits)
            if unrounded is not None:
                unrounded[p] = hm
            outs[p].append(hm if heatmap_dtype is None else hm.to(heatmap_dtype).to(hm.dtype))
    return outs['xy'], outs['zy'], outs['xz']


# ---- tail, torch flavour (autograd-capable) ---------------------------------------------------

def _linspace(n, like):
    n_f = float(n)
    return torch.arange(n, dtype=like.dtype) * (2.0 / n_f) + (-(n_f - 1.0) / n_f)     # dsntnn.py:35-36


def dsnt(hm):
    """dsntnn.py:39-62,84-96."""
    xs = _linspace(hm.shape[-1], hm)
    ys = _linspace(hm.shape[-2], hm)
    return torch.stack([(hm.sum(-2) * xs).sum(-1), (hm.sum(-1) * ys).sum(-1)], -1)


def heatmaps_to_coords(xy, zy, xz):
    """models/margipose_model.py:254-261."""
    a, b, c = dsnt(xy), dsnt(zy), dsnt(xz)
    return torch.cat([a, 0.5 * (b[..., 0:1] + c[..., 1:2])], -1)


def make_gauss(mu, size, sigma):
    """dsntnn.py:154-195."""
    h, w = size
    kx = -0.5 * (1.0 / (2.0 * sigma / w)) ** 2
    ky = -0.5 * (1.0 / (2.0 * sigma / h)) ** 2
    ex = ((_linspace(w, mu) - mu[..., 0:1]) ** 2 * kx).exp()
    ey = ((_linspace(h, mu) - mu[..., 1:2]) ** 2 * ky).exp()
    g = ey.unsqueeze(-1) * ex.unsqueeze(-2)
    return g / (g.sum((-1, -2), keepdim=True) + 1e-24)


def _kl(p, q):
    return (p * ((p + 1e-24).log() - (q + 1e-24).log())).sum((-1, -2))     # dsntnn.py:198-202


def js_reg_losses(hm, mu_t, sigma=1.0):
    g = make_gauss(mu_t, hm.shape[-2:], sigma)
    m = 0.5 * (hm + g)
    return 0.5 * _kl(hm, m) + 0.5 * _kl(g, m)                               # dsntnn.py:205-207


def euclidean_losses(actual, target):
    return (actual - target).pow(2).sum(-1).sqrt()                         # dsntnn.py:133-151


def forward_3d_losses(xy_list, zy_list, xz_list, target, pixelwise=True):
    """models/margipose_model.py:236-252."""
    t = target[..., :3]
    t_xy, t_zy, t_xz = t[..., [0, 1]], t[..., [2, 1]], t[..., [0, 2]]
    losses = 0
    for xy, zy, xz in zip(xy_list, zy_list, xz_list):
        if pixelwise:
            losses = losses + js_reg_losses(xy, t_xy) + js_reg_losses(zy, t_zy) + js_reg_losses(xz, t_xz)
        losses = losses + euclidean_losses(heatmaps_to_coords(xy, zy, xz), t)
    return losses


def forward_2d_losses(xy_list, zy_list, xz_list, target, pixelwise=True):
    """models/margipose_model.py:223-234."""
    t_xy = target[..., :2]
    losses = 0
    for xy, zy, xz in zip(xy_list, zy_list, xz_list):
        if pixelwise:
            losses = losses + js_reg_losses(xy, t_xy)
        losses = losses + euclidean_losses(heatmaps_to_coords(xy, zy, xz)[..., :2], t_xy)
    return losses


def average_loss(losses, mask):
    """dsntnn.py:99-121."""
    return (losses * mask).sum() / mask.sum().clamp(1)


def calibrate_running_stats(sd, x, n_stages, axis_permutation=True):
    """Make the synthetic BN running statistics consistent with the activations (one train-mode pass with
    momentum 1): without this, eval-mode logits of a randomly initialised net reach +-3000 and every
    comparison through the softmax is ill-conditioned.  Mirrors tools/make_golden.py::gen_model."""
    global BN_MOMENTUM
    old, BN_MOMENTUM = BN_MOMENTUM, 1.0
    try:
        with torch.no_grad():
            inner_forward(sd, x, n_stages, True, axis_permutation)
    finally:
        BN_MOMENTUM = old
    return sd


def train_step_reference(sd, x, target, mask, n_stages, axis_permutation=True):
    """One fwd + 3D loss + backward on CPU.  `sd` float tensors that require grad get .grad filled.
    Returns (coords, heatmap lists, per-(b,j) losses, scalar loss)."""
    xy, zy, xz = inner_forward(sd, x, n_stages, True, axis_permutation)
    coords = heatmaps_to_coords(xy[-1], zy[-1], xz[-1])
    losses = forward_3d_losses(xy, zy, xz, target)
    loss = average_loss(losses, mask)
    loss.backward()
    return coords, (xy, zy, xz), losses, loss
