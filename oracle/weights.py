"""ORACLE helper (test infrastructure) -- the reference's state_dict key schema and a
machine-independent deterministic weight generator shared by golden generation (build container)
and the parity tests (GPU box).  Values come from numpy's PCG64 stream, so no tensor file has to be
committed: the goldens store only inputs' seeds and the reference's outputs.

Schema source: models/margipose_model.py:25-100 (ResidualBlock / HeatmapColumn), :142-177
(HeatmapCombiner / MargiPoseModelInner); pinned by tests/golden/state_dict_keys.json, which was
dumped from the imported reference.
"""
from collections import OrderedDict

import numpy as np
import torch

PLANES = ('xy', 'zy', 'xz')


def _bn_entries(prefix, c):
    return [(prefix + '.weight', (c,)), (prefix + '.bias', (c,)), (prefix + '.running_mean', (c,)),
            (prefix + '.running_var', (c,)), (prefix + '.num_batches_tracked', ())]


def _block_entries(prefix, cin, cout, kind):
    if kind == 'up':      # ConvTranspose2d weight is (Cin, Cout, k, k)
        w_in, w_sc = (cin, cout, 3, 3), (cin, cout, 1, 1)
    else:
        w_in, w_sc = (cout, cin, 3, 3), (cout, cin, 1, 1)
    e = [(prefix + '.module.0.weight', w_in)]
    e += _bn_entries(prefix + '.module.1', cout)
    e += [(prefix + '.module.3.weight', (cout, cout, 3, 3))]
    e += _bn_entries(prefix + '.module.4', cout)
    e += [(prefix + '.shortcut.0.weight', w_sc)]
    e += _bn_entries(prefix + '.shortcut.1', cout)
    return e


def column_entries(prefix, n_joints=17):
    down = [(128, 128, 'regular'), (128, 128, 'regular'), (128, 192, 'down'), (192, 192, 'regular'),
            (192, 192, 'regular')]
    up = [(192, 192, 'regular'), (192, 192, 'regular'), (192, 128, 'up'), (128, 128, 'regular'),
          (128, n_joints, 'regular')]
    e = []
    for i, (a, b, k) in enumerate(down):
        e += _block_entries('%s.down_layers.%d' % (prefix, i), a, b, k)
    for i, (a, b, k) in enumerate(up):
        e += _block_entries('%s.up_layers.%d' % (prefix, i), a, b, k)
    return e


def _basic_entries(prefix, cin, cout, k):
    k = k if isinstance(k, tuple) else (k, k)
    return [(prefix + '.conv.weight', (cout, cin) + k)] + _bn_entries(prefix + '.bn', cout)


def inceptionv4_stem_entries(p='inner.in_cnn.'):
    """pretrainedmodels' InceptionV4 features[0:7] + the reference's 1x1 head (unpinned)."""
    e = _basic_entries(p + '0', 3, 32, 3) + _basic_entries(p + '1', 32, 32, 3) + _basic_entries(p + '2', 32, 64, 3)
    e += _basic_entries(p + '3.conv', 64, 96, 3)
    e += _basic_entries(p + '4.branch0.0', 160, 64, 1) + _basic_entries(p + '4.branch0.1', 64, 96, 3)
    e += _basic_entries(p + '4.branch1.0', 160, 64, 1) + _basic_entries(p + '4.branch1.1', 64, 64, (1, 7))
    e += _basic_entries(p + '4.branch1.2', 64, 64, (7, 1)) + _basic_entries(p + '4.branch1.3', 64, 96, 3)
    e += _basic_entries(p + '5.conv', 192, 192, 3)
    e += _basic_entries(p + '6.branch0', 384, 96, 1)
    e += _basic_entries(p + '6.branch1.0', 384, 64, 1) + _basic_entries(p + '6.branch1.1', 64, 96, 3)
    e += _basic_entries(p + '6.branch2.0', 384, 64, 1) + _basic_entries(p + '6.branch2.1', 64, 96, 3) + _basic_entries(p + '6.branch2.2', 96, 96, 3)
    e += _basic_entries(p + '6.branch3.1', 384, 96, 1)
    e += [(p + '7.weight', (128, 384, 1, 1)), (p + '7.bias', (128,))] + _bn_entries(p + '8', 128)
    return e


RESNET_LAYERS = {'resnet18': ('basic', 2, 2), 'resnet34': ('basic', 3, 4), 'resnet50': ('bottleneck', 3, 4)}


def resnet_stem_entries(name, p='inner.in_cnn.'):
    """torchvision ResNet conv1/bn1/layer1/layer2 as nn.Sequential children 0,1,4,5 (models/margipose_model.py:128-136)
    + the reference's 1x1 head for resnet50 (:122-127).  torchvision is not in the reference tree: unpinned."""
    kind, n1, n2 = RESNET_LAYERS[name]
    ex = 1 if kind == 'basic' else 4
    e = [(p + '0.weight', (64, 3, 7, 7))] + _bn_entries(p + '1', 64)
    cin = 64
    for li, (idx, planes, n, stride) in enumerate(((4, 64, n1, 1), (5, 128, n2, 2))):
        for b in range(n):
            q = '%s%d.%d.' % (p, idx, b)
            st = stride if b == 0 else 1
            if kind == 'basic':
                e += [(q + 'conv1.weight', (planes, cin, 3, 3))] + _bn_entries(q + 'bn1', planes)
                e += [(q + 'conv2.weight', (planes, planes, 3, 3))] + _bn_entries(q + 'bn2', planes)
            else:
                e += [(q + 'conv1.weight', (planes, cin, 1, 1))] + _bn_entries(q + 'bn1', planes)
                e += [(q + 'conv2.weight', (planes, planes, 3, 3))] + _bn_entries(q + 'bn2', planes)
                e += [(q + 'conv3.weight', (planes * 4, planes, 1, 1))] + _bn_entries(q + 'bn3', planes * 4)
            if st != 1 or cin != planes * ex:
                e += [(q + 'downsample.0.weight', (planes * ex, cin, 1, 1))] + _bn_entries(q + 'downsample.1', planes * ex)
            cin = planes * ex
    if cin != 128:
        e += [(p + '6.weight', (128, cin, 1, 1)), (p + '6.bias', (128,))] + _bn_entries(p + '7', 128)
    return e


def chatterbox_cnn_entries(p, shrink_width, n_joints=17):
    """_ChatterboxCnn (models/chatterbox_model.py:87-214) in state_dict order: a block registers resample.{0,1} (when it has
    one) BEFORE conv1, bn1, conv2, bn2.  Pinned by tests/golden/chatterbox_keys.json (dumped from the imported reference)."""
    def f(a, b):
        return (a, b) if shrink_width else (b, a)

    def block(q, cin, cout, resample, up):
        w1 = (cin, cout, 3, 3) if up else (cout, cin, 3, 3)
        e = []
        if resample:
            e += [(q + 'resample.0.weight', (cin, cout, 1, 1) if up else (cout, cin, 1, 1))] + _bn_entries(q + 'resample.1', cout)
        e += [(q + 'conv1.weight', w1)] + _bn_entries(q + 'bn1', cout)
        e += [(q + 'conv2.weight', (cout, cout, 3, 3))] + _bn_entries(q + 'bn2', cout)
        return e
    d, u = p + 'down_convs.', p + 'up_convs.'
    e = block(d + '0.', 128, 256, True, False) + block(d + '1.', 256, 256, False, False)
    e += block(d + '2.', 256, 512, True, False) + block(d + '3.', 512, 512, False, False)
    e += [(d + '4.weight', (1024, 512) + f(1, 8))] + _bn_entries(d + '5', 1024)
    e += [(u + '0.weight', (1024, 512) + f(1, 8))] + _bn_entries(u + '1', 512)
    e += block(u + '3.', 512, 512, False, True) + block(u + '4.', 512, 256, True, True)
    e += block(u + '5.', 256, 256, False, True) + block(u + '6.', 256, 128, True, True)
    e += [(u + '7.weight', (n_joints, 128, 1, 1))]
    return e


def _basic_block_entries(q, cin, planes, downsample):
    e = [(q + 'conv1.weight', (planes, cin, 3, 3))] + _bn_entries(q + 'bn1', planes)
    e += [(q + 'conv2.weight', (planes, planes, 3, 3))] + _bn_entries(q + 'bn2', planes)
    if downsample:
        e += [(q + 'downsample.0.weight', (planes, cin, 1, 1))] + _bn_entries(q + 'downsample.1', planes)
    return e


def chatterbox_schema(n_joints=17):
    """ChatterboxModel (models/chatterbox_model.py:223-244): in_cnn (resnet34 conv1, bn1, layer1, layer2), xy_hm_cnn (resnet34
    layer3 / layer4 as layer1 / layer2, hm_conv), zy_hm_cnn, xz_hm_cnn.  The resnet34 part restates torchvision (unpinned)."""
    e = [('in_cnn.conv1.weight', (64, 3, 7, 7))] + _bn_entries('in_cnn.bn1', 64)
    for name, cin, planes, n in (('in_cnn.layer1.', 64, 64, 3), ('in_cnn.layer2.', 64, 128, 4),
                                 ('xy_hm_cnn.layer1.', 128, 256, 6), ('xy_hm_cnn.layer2.', 256, 512, 3)):
        for b in range(n):
            e += _basic_block_entries('%s%d.' % (name, b), cin if b == 0 else planes, planes, b == 0 and cin != planes)
    e += [('xy_hm_cnn.hm_conv.weight', (n_joints, 512, 1, 1))]
    e += chatterbox_cnn_entries('zy_hm_cnn.', True, n_joints) + chatterbox_cnn_entries('xz_hm_cnn.', False, n_joints)
    return OrderedDict(e)


def schema(n_stages, n_joints=17, stem='patch8'):
    """Ordered key -> shape map of MargiPoseModel(n_stages) with the given stem."""
    if stem == 'inceptionv4':
        e = inceptionv4_stem_entries()
    elif stem in RESNET_LAYERS:
        e = resnet_stem_entries(stem)
    else:
        e = [('inner.in_cnn.0.weight', (128, 3, 8, 8))] + _bn_entries('inner.in_cnn.1', 128)
    # nn.ModuleList registration order (models/margipose_model.py:158-162): all xy columns, then zy,
    # then xz, then the combiners.
    for p in PLANES:
        for t in range(n_stages):
            e += column_entries('inner.%s_hm_cnns.%d' % (p, t), n_joints)
    for t in range(n_stages - 1):
        e += [('inner.hm_combiners.%d.conv.weight' % t, (128, 3 * n_joints, 1, 1))]
    return OrderedDict(e)


def fill_like(shapes, seed, dtype=torch.float32):
    """Deterministic values for an ordered key->shape map (conv weights Kaiming-scaled as
    nn_helpers.py:7-21 would; BN affine/running stats deliberately NON-trivial so tests exercise them)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for key, shape in shapes.items():
        shape = tuple(shape)
        if key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
            continue
        if key.endswith('running_var'):
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith('running_mean'):
            v = rng.standard_normal(shape) * 0.1
        elif len(shape) == 1 and key.endswith('.weight'):
            v = rng.uniform(0.5, 1.5, shape)
        elif len(shape) == 1:
            v = rng.standard_normal(shape) * 0.1
        else:
            fan_out = shape[0] * shape[2] * shape[3]          # kaiming_normal_(mode='fan_out'): the LEADING dimension
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_out)
        sd[key] = torch.from_numpy(np.asarray(v)).to(dtype)
    return sd


def make_state_dict(n_stages, seed, dtype=torch.float32, stem='patch8'):
    return fill_like(schema(n_stages, stem=stem), seed, dtype)


def column_state_dict(prefix, seed, dtype=torch.float32):
    return fill_like(OrderedDict(column_entries(prefix)), seed, dtype)


def seeded_inputs(seed, batch, size=256, dtype=torch.float32):
    """Synthetic frames / targets / mask as SURVEY.md §8(d) prescribes (numpy stream, portable)."""
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((batch, 3, size, size))).to(dtype)
    target = torch.from_numpy(rng.uniform(-1.0, 1.0, (batch, 17, 3))).to(dtype)
    mask = torch.ones(batch, 17, dtype=dtype)
    return x, target, mask
