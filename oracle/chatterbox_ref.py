"""ORACLE (test infrastructure, NOT product code) -- functional PyTorch-CPU restatement of the reference ChatterboxModel
(/root/reference/src/margipose/models/chatterbox_model.py) over a flat state_dict with the reference's key names.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; margipose_amd/ never does.

Parity status:
  * `chatterbox_cnn` (_ChatterboxCnn, :87-221) is PINNED: tools/make_golden_chatterbox.py runs the imported reference class on
    seeded inputs, tests/golden/chatterbox_cnn.npz holds its outputs and input gradients, tests/test_oracle_golden.py checks
    this restatement against them (both orientations, train and eval mode);
  * `resnet_features` / `xy_cnn` (:37-84) restate torchvision 0.3.0's resnet34 (requirements.txt:7), which is NOT in the
    reference tree (BasicBlock: 3x3 - BN - ReLU - 3x3 - BN, + identity or 1x1/BN downsample, ReLU; layers of 3, 4, 6, 3 blocks
    with 64, 128, 256, 512 planes) -- parity UNPINNED for these two, as for the ResNet stems of oracle/model_ref.py.  The
    reference's own modification of layer3 / layer4 (:62-72: strides set to 1, 3x3 convolutions that were NOT strided get
    dilation 2 / 4 and matching padding) is restated from the reference.

The tail (flat_softmax, dsnt, losses) is oracle/model_ref.py's, which is pinned.
"""
import torch
import torch.nn.functional as F

from . import model_ref as R

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _bn(sd, key, x, train):
    return F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'], sd[key + '.weight'], sd[key + '.bias'],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _basic_block(sd, q, x, train, stride, dil1, dil2):
    """torchvision BasicBlock with the (stride, dilation) the caller decided on."""
    y = F.relu(_bn(sd, q + 'bn1', F.conv2d(x, sd[q + 'conv1.weight'], None, stride=stride, padding=dil1, dilation=dil1), train))
    y = _bn(sd, q + 'bn2', F.conv2d(y, sd[q + 'conv2.weight'], None, padding=dil2, dilation=dil2), train)
    if q + 'downsample.0.weight' in sd:
        x = _bn(sd, q + 'downsample.1', F.conv2d(x, sd[q + 'downsample.0.weight'], None, stride=stride), train)
    return F.relu(y + x)


def _blocks(sd, p):
    b = 0
    while '%s%d.conv1.weight' % (p, b) in sd:
        yield b, '%s%d.' % (p, b)
        b += 1


def resnet_features(sd, x, train, p='in_cnn.'):
    """ResNetFeatureExtractor.forward (:45-54): conv1, bn1, relu, max_pool2d(3, 2, 1), layer1, layer2."""
    x = F.relu(_bn(sd, p + 'bn1', F.conv2d(x, sd[p + 'conv1.weight'], None, stride=2, padding=3), train))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for layer, stride in (('layer1.', 1), ('layer2.', 2)):
        for b, q in _blocks(sd, p + layer):
            x = _basic_block(sd, q, x, train, stride if b == 0 else 1, 1, 1)
    return x


def xy_cnn(sd, t, train, p='xy_hm_cnn.'):
    """_XYCnn (:57-84).  layer1 / layer2 are resnet34's layer3 / layer4; `module.stride == (2, 2)` convolutions (block 0's
    conv1 and its 1x1 downsample) become stride 1 and -- the `elif` -- keep dilation 1; every other 3x3 gets dilation
    2 ** (i + 1) and padding (dil * 2 + 1) // 2 = dil."""
    for i, layer in enumerate(('layer1.', 'layer2.')):
        dil = 2 ** (i + 1)
        for b, q in _blocks(sd, p + layer):
            t = _basic_block(sd, q, t, train, 1, 1 if b == 0 else dil, dil)
    return F.conv2d(t, sd[p + 'hm_conv.weight'])


def _down_block(sd, q, x, train, stride, dilation, dilation_in):
    """_ChatterboxCnn._DownBlock (:132-171)."""
    out = F.conv2d(x, sd[q + 'conv1.weight'], None, stride=stride, padding=dilation_in, dilation=dilation_in)
    out = F.relu(_bn(sd, q + 'bn1', out, train))
    out = _bn(sd, q + 'bn2', F.conv2d(out, sd[q + 'conv2.weight'], None, padding=dilation, dilation=dilation), train)
    if q + 'resample.0.weight' in sd:
        x = _bn(sd, q + 'resample.1', F.conv2d(x, sd[q + 'resample.0.weight'], None, stride=stride), train)
    return F.relu(out + x)


def _up_block(sd, q, x, train, stride, dilation, dilation_in, output_padding):
    """_ChatterboxCnn._UpBlock (:173-214)."""
    out = F.conv_transpose2d(x, sd[q + 'conv1.weight'], None, stride=stride, padding=dilation_in, output_padding=output_padding,
                             dilation=dilation_in)
    out = F.relu(_bn(sd, q + 'bn1', out, train))
    out = _bn(sd, q + 'bn2', F.conv2d(out, sd[q + 'conv2.weight'], None, padding=dilation, dilation=dilation), train)
    if q + 'resample.0.weight' in sd:
        x = _bn(sd, q + 'resample.1', F.conv_transpose2d(x, sd[q + 'resample.0.weight'], None, stride=stride,
                                                         output_padding=output_padding), train)
    return F.relu(out + x)


def chatterbox_cnn(sd, p, t, shrink_width, train):
    """_ChatterboxCnn.forward (:216-221) with the layer table of :96-126.  f(a, b) orders (height, width) arguments:
    shrink_width halves the WIDTH (128x32x32 -> 256x32x16 -> 512x32x8 -> 1024x32x1 and back)."""
    def f(a, b):
        return (a, b) if shrink_width else (b, a)
    d, u = p + 'down_convs.', p + 'up_convs.'
    t = _down_block(sd, d + '0.', t, train, f(1, 2), f(2, 1), f(1, 1))
    t = _down_block(sd, d + '1.', t, train, (1, 1), f(2, 1), f(2, 1))
    t = _down_block(sd, d + '2.', t, train, f(1, 2), f(4, 1), f(2, 1))
    t = _down_block(sd, d + '3.', t, train, (1, 1), f(4, 1), f(4, 1))
    t = F.relu(_bn(sd, d + '5', F.conv2d(t, sd[d + '4.weight']), train))                   # kernel f(1, 8): 1024 x 32 x 1
    t = F.relu(_bn(sd, u + '1', F.conv_transpose2d(t, sd[u + '0.weight']), train))         # kernel f(1, 8): 512 x 32 x 8
    t = _up_block(sd, u + '3.', t, train, (1, 1), f(4, 1), f(4, 1), (0, 0))
    t = _up_block(sd, u + '4.', t, train, f(1, 2), f(2, 1), f(4, 1), f(0, 1))
    t = _up_block(sd, u + '5.', t, train, (1, 1), f(2, 1), f(2, 1), (0, 0))
    t = _up_block(sd, u + '6.', t, train, f(1, 2), f(1, 1), f(2, 1), f(0, 1))
    return F.conv2d(t, sd[u + '7.weight'])


def chatterbox_forward(sd, x, train):
    """ChatterboxModel.forward (:273-289): returns (xyz coordinates, (xy, zy, xz) heatmaps)."""
    t = resnet_features(sd, x, train)
    xy = R.flat_softmax(xy_cnn(sd, t, train))
    zy = R.flat_softmax(chatterbox_cnn(sd, 'zy_hm_cnn.', t, True, train))
    xz = R.flat_softmax(chatterbox_cnn(sd, 'xz_hm_cnn.', t, False, train))
    return R.heatmaps_to_coords(xy, zy, xz), (xy, zy, xz)


def chatterbox_losses(hms, target, three_d=True, pixelwise=True):
    """forward_3d_losses (:255-271) / forward_2d_losses (:246-253): the single-stage case of oracle/model_ref.py's loops."""
    xy, zy, xz = hms
    fn = R.forward_3d_losses if three_d else R.forward_2d_losses
    return fn([xy], [zy], [xz], target, pixelwise)
