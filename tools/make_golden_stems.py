#!/usr/bin/env python
"""Pin what CAN be pinned of the image feature extractors (missing #1): the part that is reference code.

The reference builds its stems in `make_image_feature_extractor` (/root/reference/src/margipose/models/margipose_model.py:103-139)
out of two third-party packages that are not in this image (pretrainedmodels==0.6.0, torchvision==0.3.0):

    :104-110   which `inceptionv4().features[i]` are taken (0..6) and the Conv2d(384, 128, 1) + BatchNorm2d + ReLU head
    :111-117   the padding rewrite -- every Conv2d / MaxPool2d (and nothing else: not the AvgPool2d) gets padding = kernel // 2
    :119-137   the ResNet slice conv1, bn1, relu, maxpool, layer1, layer2 and the `resnet_out_chans != 128` rule for the 1x1 head

This script runs THAT function -- the real one, imported in place -- over stand-in third-party constructors: live `nn.Module`s
written here from the published architectures (the same recollection SURVEY.md Appendix B records), with the third-party packages'
OWN paddings (so that the reference's rewrite loop is what produces the final geometry) and their own attribute names (so that the
state-dict keys are the ones a real checkpoint carries).  It then drives the reference's MargiPoseModel with that stem on seeded
inputs and stores outputs only.  What the fixtures pin: the slice, the head, the padding rewrite, the state-dict schema, and the
functional restatement oracle/model_ref.py::{inceptionv4_stem, resnet_stem} (independent code: nn.functional calls with explicit
paddings) against the reference-assembled module graph.  What they cannot pin: the stand-ins themselves against the third-party
originals -- that part stays "unpinned" (DESIGN.md section 1).

    python tools/make_golden_stems.py      # writes tests/golden/stem_<name>.npz and stem_keys.json (build container only)
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from _reference_import import import_reference   # noqa: E402
from oracle import weights as W                   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
dsntnn, mm, Skel = import_reference()
torch.set_num_threads(8)


# ---------------------------------------------------------------------------------------------------------------------
# stand-in for pretrainedmodels.inceptionv4 (features[0:7] only; the third-party paddings, NOT the rewritten ones)
# ---------------------------------------------------------------------------------------------------------------------
class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001, momentum=0.1, affine=True)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


class Mixed_3a(nn.Module):
    def __init__(self):
        super().__init__()
        self.maxpool = nn.MaxPool2d(3, stride=2)
        self.conv = BasicConv2d(64, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat((self.maxpool(x), self.conv(x)), 1)


class Mixed_4a(nn.Module):
    def __init__(self):
        super().__init__()
        self.branch0 = nn.Sequential(BasicConv2d(160, 64, kernel_size=1, stride=1), BasicConv2d(64, 96, kernel_size=3, stride=1))
        self.branch1 = nn.Sequential(BasicConv2d(160, 64, kernel_size=1, stride=1),
                                     BasicConv2d(64, 64, kernel_size=(1, 7), stride=1, padding=(0, 3)),
                                     BasicConv2d(64, 64, kernel_size=(7, 1), stride=1, padding=(3, 0)),
                                     BasicConv2d(64, 96, kernel_size=(3, 3), stride=1))

    def forward(self, x):
        return torch.cat((self.branch0(x), self.branch1(x)), 1)


class Mixed_5a(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = BasicConv2d(192, 192, kernel_size=3, stride=2)
        self.maxpool = nn.MaxPool2d(3, stride=2)

    def forward(self, x):
        return torch.cat((self.conv(x), self.maxpool(x)), 1)


class Inception_A(nn.Module):
    def __init__(self):
        super().__init__()
        self.branch0 = BasicConv2d(384, 96, kernel_size=1, stride=1)
        self.branch1 = nn.Sequential(BasicConv2d(384, 64, kernel_size=1, stride=1), BasicConv2d(64, 96, kernel_size=3, stride=1, padding=1))
        self.branch2 = nn.Sequential(BasicConv2d(384, 64, kernel_size=1, stride=1), BasicConv2d(64, 96, kernel_size=3, stride=1, padding=1),
                                     BasicConv2d(96, 96, kernel_size=3, stride=1, padding=1))
        self.branch3 = nn.Sequential(nn.AvgPool2d(3, stride=1, padding=1, count_include_pad=False),
                                     BasicConv2d(384, 96, kernel_size=1, stride=1))

    def forward(self, x):
        return torch.cat((self.branch0(x), self.branch1(x), self.branch2(x), self.branch3(x)), 1)


class _InceptionV4(nn.Module):
    def __init__(self):
        super().__init__()
        # (the original continues with three more Inception_A, Reduction_A, ...: the reference takes indices 0..6 only)
        self.features = nn.Sequential(BasicConv2d(3, 32, kernel_size=3, stride=2), BasicConv2d(32, 32, kernel_size=3, stride=1),
                                      BasicConv2d(32, 64, kernel_size=3, stride=1, padding=1), Mixed_3a(), Mixed_4a(), Mixed_5a(),
                                      Inception_A(), Inception_A())


def inceptionv4(*a, **k):
    return _InceptionV4()


# ---------------------------------------------------------------------------------------------------------------------
# stand-in for torchvision.models.resnet{18,34,50} (v1.5 bottleneck: the stride sits on the 3x3)
# ---------------------------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        return self.relu(self.bn3(self.conv3(out)) + idt)


class _ResNet(nn.Module):
    def __init__(self, block, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)      # (:121 reads layer3[0].conv1.in_channels)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)

    def _make_layer(self, block, planes, n, stride=1):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False), nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)


_RESNETS = {'resnet18': (BasicBlock, [2, 2, 2, 2]), 'resnet34': (BasicBlock, [3, 4, 6, 3]), 'resnet50': (Bottleneck, [3, 4, 6, 3])}

# the names the reference's module looks up at call time (`from pretrainedmodels import inceptionv4`, `import torchvision.models`)
mm.inceptionv4 = inceptionv4
mm.torchvision = types.SimpleNamespace(models=types.SimpleNamespace(
    **{name: (lambda pretrained=False, _n=name: _ResNet(*_RESNETS[_n])) for name in _RESNETS}))


def t2n(t):
    return t.detach().cpu().numpy()


def grads_summary(named_params):
    norms, heads = [], []
    for k, p in named_params:
        g = p.grad.detach().double().flatten()
        norms.append(float(g.norm()))
        h = torch.zeros(8, dtype=torch.float64)
        h[:min(8, g.numel())] = g[:8]
        heads.append(h.numpy())
    return np.array(norms), np.stack(heads)


def gen(stem, seed, keys_out):
    T, B = 1, 2
    res = {}
    x_t, target_t, _ = W.seeded_inputs(seed + 1000, B, dtype=torch.float64)
    rng = np.random.default_rng(seed + 2000)
    mask_np = (rng.uniform(0, 1, (B, 17)) > 0.2).astype(np.float64)
    for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        model = mm.MargiPoseModel(Skel, T, True, stem, 'jsd')       # the REAL make_image_feature_extractor runs in here
        if tag == 'f64':
            items = [[k, list(v.shape)] for k, v in model.state_dict().items()]
            mine = [[k, list(s)] for k, s in W.schema(T, stem=stem).items()]
            assert mine == items, 'oracle schema differs from the reference-assembled model (%s)' % stem
            pads = [[k, list(m_.padding) if isinstance(m_.padding, tuple) else [m_.padding, m_.padding]]
                    for k, m_ in model.inner.in_cnn.named_modules() if isinstance(m_, (nn.Conv2d, nn.MaxPool2d, nn.AvgPool2d))]
            keys_out[stem] = {'n_keys': len(items), 'n_params': sum(p.numel() for p in model.parameters()),
                              'sha256': hashlib.sha256(json.dumps(items).encode()).hexdigest(),
                              'in_cnn_items': [it for it in items if it[0].startswith('inner.in_cnn.')],
                              'paddings_after_rewrite': pads}
        model = model.to(dt)
        model.load_state_dict(W.make_state_dict(T, seed, dt, stem=stem), strict=True)
        x, target, mask = x_t.to(dt), target_t.to(dt), torch.tensor(mask_np, dtype=dt)
        feats = []
        hook = model.inner.in_cnn.register_forward_hook(lambda mod, inp, out: feats.append(out.detach()))
        bns = [mod for mod in model.modules() if isinstance(mod, nn.BatchNorm2d)]
        for mod in bns:                  # calibrate the synthetic running statistics (oracle.model_ref.calibrate_running_stats)
            mod.momentum = 1.0
        model.train()
        with torch.no_grad():
            model(x)
        for mod in bns:
            mod.momentum = 0.1
            mod.num_batches_tracked.zero_()
        model.eval()
        feats.clear()
        with torch.no_grad():
            res['coords_eval_' + tag] = t2n(model(x))
            if tag == 'f64':
                res['feat_eval_f64'] = t2n(feats[-1])[:, ::4].astype(np.float32)    # the stem's output (every fourth channel, stored as fp32)
            res['hm_xy_eval_' + tag] = t2n(model.xy_heatmaps[-1])[:, :, ::4, ::4]
            res['losses3d_eval_' + tag] = t2n(model.forward_3d_losses(None, target))
        model.train()
        xg = x.clone().requires_grad_(True)
        out = model(xg)
        if tag == 'f64':
            res['feat_train_f64'] = t2n(feats[-1])[:, ::4].astype(np.float32)
        l3 = model.forward_3d_losses(out, target)
        res['coords_train_' + tag] = t2n(out)
        res['losses3d_train_' + tag] = t2n(l3)
        res['hm_xz_train_' + tag] = t2n(model.xz_heatmaps[-1])[:, :, ::4, ::4]
        loss = dsntnn.average_loss(l3, mask)
        model.zero_grad()
        loss.backward()
        res['loss_' + tag] = t2n(loss)
        res['dx_' + tag] = t2n(xg.grad)[:, :, ::8, ::8]
        n, h = grads_summary(list(model.named_parameters()))
        res['gnorm_' + tag] = n
        res['ghead_' + tag] = h
        hook.remove()
        if tag == 'f64':
            res['param_keys'] = np.array([k for k, _ in model.named_parameters()])
            res['running_after'] = np.concatenate(
                [t2n(b).flatten() for k, b in model.named_buffers() if not k.endswith('num_batches_tracked')])
    res['mask'] = mask_np
    path = os.path.join(OUT, 'stem_%s.npz' % stem)
    np.savez_compressed(path, seed=seed, **{k: np.asarray(v) for k, v in res.items()})
    print('wrote stem_%s.npz %.1f KB' % (stem, os.path.getsize(path) / 1024))


if __name__ == '__main__':
    keys = {}
    for i, stem in enumerate(('inceptionv4', 'resnet18', 'resnet34', 'resnet50')):
        gen(stem, 1401 + i, keys)
    try:
        mm.make_image_feature_extractor('vgg16')
        raise AssertionError('expected the reference to reject an unknown feature extractor')
    except Exception as e:      # (:139)
        keys['unknown_name_message'] = str(e)
    with open(os.path.join(OUT, 'stem_keys.json'), 'w') as f:
        json.dump(keys, f)
    print('wrote stem_keys.json')
