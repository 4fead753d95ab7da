"""ChatterboxModel on MI355X (reference src/margipose/models/chatterbox_model.py:223-289, factory :292-303): same
constructor, attributes (`xy_heatmaps` / `zy_heatmaps` / `xz_heatmaps`, `data_specs`, `pixelwise_loss`), methods and
state_dict keys (`in_cnn.*`, `xy_hm_cnn.{layer1,layer2,hm_conv}.*`, `{zy,xz}_hm_cnn.{down_convs,up_convs}.<i>.*`); every
convolution, BatchNorm, residual sum, softmax, soft-argmax and loss runs on the gfx950 kernels behind
include/margipose_hip.h (stem.py::ChatterboxGraph executes the network, dsntnn.py the tail).

The nn.Module tree only owns parameters and buffers.  torchvision (requirements.txt:7) is not in the reference tree: the
ResNet-34 layers are restated from the published architecture (as in stem.py), so `in_cnn` / `xy_hm_cnn` parity is checked
against this repo's oracle only; _ChatterboxCnn is reference code proper and is pinned by tests/golden/chatterbox_cnn.pt.
ImageNet weights (resnet34(pretrained=True), :235) cannot be downloaded: torchvision's initialisation is used.
"""
import torch
from torch import nn

from .. import _lib, dsntnn
from ..engine import Engine
from ..nn_helpers import init_parameters
from ..stem import BasicBlock, make_chatterbox_cnn_modules
from .margipose_model import (CanonicalSkeletonDesc, DataSpecs, IMAGENET_MEAN, IMAGENET_STDDEV, ImageSpecs, JointsSpecs)

Default_Chatterbox_Desc = {
    'type': 'chatterbox',
    'version': '1.3.0',
    'settings': {
        'pixelwise_loss': 'jsd',
    },
}


def _resnet_layer(cin, planes, n, stride):
    return nn.Sequential(*[BasicBlock(cin if i == 0 else planes, planes, stride if i == 0 else 1) for i in range(n)])


class ResNetFeatureExtractor(nn.Module):
    """conv1, bn1, layer1, layer2 of ResNet-34 (:37-54)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _resnet_layer(64, 64, 3, 1)
        self.layer2 = _resnet_layer(64, 128, 4, 2)


class _XYCnn(nn.Module):
    """ResNet-34 layer3 / layer4 (6 and 3 BasicBlocks; run with stride 1 and dilation 2 / 4, :57-84) + the 1x1 to heatmaps."""

    def __init__(self, n_joints):
        super().__init__()
        self.layer1 = _resnet_layer(128, 256, 6, 2)      # (built with the strided block so that its downsample.{0,1} exist)
        self.layer2 = _resnet_layer(256, 512, 3, 2)
        self.hm_conv = nn.Conv2d(512, n_joints, kernel_size=1, bias=False)


class _ChatterboxCnn(nn.Module):
    def __init__(self, n_joints, shrink_width=True):
        super().__init__()
        self.down_convs, self.up_convs = make_chatterbox_cnn_modules(n_joints, shrink_width)
        init_parameters(self)


class _GraphFn(torch.autograd.Function):
    """The three heads' heatmap logits from the image as ONE autograd node."""

    @staticmethod
    def forward(ctx, engine, train, J, x, *params):
        outs, ectx = engine.graph_forward(x, train, save=True)
        ctx.engine, ctx.ectx, ctx.J = engine, ectx, J
        return tuple(o[..., :J].permute(0, 3, 1, 2).contiguous() for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        engine = ctx.engine
        gs = []
        for g in grads:
            if g is None:
                gs.append(None)
                continue
            B, J, H, W = g.shape
            t = torch.zeros(B, H, W, 32, dtype=torch.float32, device=g.device)
            t[..., :J] = g.permute(0, 2, 3, 1)
            gs.append(t)
        gflat, dx = engine.graph_backward(ctx.ectx, gs, ctx.needs_input_grad[3])
        flat = gflat.clone()
        if engine.dp is not None:
            flat.div_(engine.dp[1])
        return (None, None, None, dx) + tuple(engine.grads_from_flat(flat))


class ChatterboxModel(nn.Module):
    n_stages = 0                     # what engine.Engine reads from its owner: a graph-only model
    feature_extractor_name = 'chatterbox'
    spaces = (0, 1, 2)

    def __init__(self, skel_desc, pixelwise_loss):
        super().__init__()
        self.data_specs = DataSpecs(ImageSpecs(256, mean=IMAGENET_MEAN, stddev=IMAGENET_STDDEV), JointsSpecs(skel_desc, n_dims=3))
        self.pixelwise_loss = pixelwise_loss
        self.n_joints = skel_desc.n_joints
        self.in_cnn = ResNetFeatureExtractor()
        self.xy_hm_cnn = _XYCnn(skel_desc.n_joints)
        for m in list(self.in_cnn.modules()) + list(self.xy_hm_cnn.modules()):       # torchvision's ResNet initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        init_parameters(self.xy_hm_cnn.hm_conv)
        self.zy_hm_cnn = _ChatterboxCnn(skel_desc.n_joints, shrink_width=True)
        self.xz_hm_cnn = _ChatterboxCnn(skel_desc.n_joints, shrink_width=False)
        self.xy_hm_cnns = self.zy_hm_cnns = self.xz_hm_cnns = self.hm_combiners = ()
        self.xy_heatmaps = self.zy_heatmaps = self.xz_heatmaps = None
        self._engine = None
        self._last_out = None

    def engine(self):
        if self._engine is None:
            object.__setattr__(self, '_engine', Engine(self))
            self._engine.input_norm = (list(IMAGENET_MEAN), list(IMAGENET_STDDEV))
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self._engine is not None:
            self._engine.invalidate()
        return out

    def _pixelwise_flag(self):
        if self.pixelwise_loss == 'jsd':
            return True
        if self.pixelwise_loss is None:
            return False
        raise Exception('unrecognised pixelwise loss: {}'.format(self.pixelwise_loss))

    def _losses(self, out_var, target_var, three_d):
        if out_var is self._last_out:
            # the model's own output: JS of the plane(s) + DSNT + z-merge + Euclidean in one launch, the same sum as :246-271
            target = target_var.narrow(-1, 0, 3).contiguous() if target_var.size(-1) >= 3 else \
                torch.cat([target_var.narrow(-1, 0, 2), torch.zeros_like(target_var.narrow(-1, 0, 1))], -1).contiguous()
            return dsntnn.stage_losses(self.xy_heatmaps[-1], self.zy_heatmaps[-1], self.xz_heatmaps[-1], target, 1.0,
                                       self._pixelwise_flag(), three_d)
        # any other `out_var` (detached, post-processed, another model's): the Euclidean term is taken on IT, as the reference
        # does (:247-248, :256-257, :266); the pixelwise terms on this model's stored heatmaps
        n = 3 if three_d else 2
        tgt = target_var.narrow(-1, 0, n)
        losses = dsntnn.euclidean_losses(out_var.narrow(-1, 0, n).contiguous(), tgt.contiguous())
        if self._pixelwise_flag():
            losses = losses + dsntnn.js_reg_losses(self.xy_heatmaps[-1], tgt.narrow(-1, 0, 2).contiguous(), 1.0)
            if three_d:
                zy = torch.cat([tgt.narrow(-1, 2, 1), tgt.narrow(-1, 1, 1)], -1).contiguous()
                xz = torch.cat([tgt.narrow(-1, 0, 1), tgt.narrow(-1, 2, 1)], -1).contiguous()
                losses = losses + dsntnn.js_reg_losses(self.zy_heatmaps[-1], zy, 1.0) + dsntnn.js_reg_losses(self.xz_heatmaps[-1], xz, 1.0)
        return losses

    def forward_2d_losses(self, out_var, target_var):
        """euclidean_losses(out_var xy) + JS(xy) (:246-253)."""
        return self._losses(out_var, target_var, False)

    def forward_3d_losses(self, out_var, target_var):
        """euclidean_losses(out_var xyz) + JS(xy) + JS(zy) + JS(xz) (:255-271)."""
        return self._losses(out_var, target_var, True)

    def forward(self, *inputs):
        x = inputs[0]
        eng = self.engine()
        params = eng.param_list()
        J = self.n_joints
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            logits = _GraphFn.apply(eng, self.training, J, x, *params)
        else:
            outs, _ = eng.graph_forward(x, self.training, save=False)
            logits = [o[..., :J].permute(0, 3, 1, 2).contiguous() for o in outs]
        self.xy_heatmaps = [dsntnn.flat_softmax(logits[0])]
        self.zy_heatmaps = [dsntnn.flat_softmax(logits[1])]
        self.xz_heatmaps = [dsntnn.flat_softmax(logits[2])]
        # x, y from the xy map; z = the mean of the two maps that carry it (:283-289)
        out = dsntnn.heatmaps_to_coords(self.xy_heatmaps[-1], self.zy_heatmaps[-1], self.xz_heatmaps[-1])
        object.__setattr__(self, '_last_out', out)
        return out


def create_chatterbox_model(model_desc):
    """ChatterboxModelFactory (:292-303): type 'chatterbox', version ^1.3.0."""
    s = model_desc['settings']
    return ChatterboxModel(skel_desc=CanonicalSkeletonDesc, pixelwise_loss=s.get('pixelwise_loss', 'jsd'))
