"""MargiPoseModel on MI355X: same constructor, attributes, methods, state_dict keys and error
behaviour as the reference model (reference src/margipose/models/margipose_model.py:203-284), with
every FLOP of forward and backward executed by the gfx950 kernels behind include/margipose_hip.h.

The nn.Module tree below exists ONLY to own parameters/buffers under the reference's key schema
(`inner.in_cnn.*`, `inner.{xy,zy,xz}_hm_cnns.<t>.{down,up}_layers.<i>.{module,shortcut}.<k>.*`,
`inner.hm_combiners.<t>.conv.weight`); none of the torch.nn layers is ever called.  The compute is
one autograd Function around margipose_amd.engine.Engine.

Stems: the reference's 'inceptionv4' (default) and 'resnet18/34/50' feature extractors (stem.py; their layer
definitions come from pretrainedmodels / torchvision, which are not in the reference tree: SURVEY.md §8c, parity
unpinned; ImageNet weights cannot be downloaded, so they start from those packages' initialisation), plus
`feature_extractor='patch8'`, an in-repo deterministic stem (8x8/stride-8 conv + BN + ReLU) used by fixtures.
Any other name raises the reference's exception text.
"""
from collections import namedtuple

import torch
from torch import nn

from .. import _lib, dsntnn
from ..engine import Engine
from ..nn_helpers import init_parameters

Default_MargiPose_Desc = {
    'type': 'margipose',
    'version': '6.0.1',
    'settings': {
        'n_stages': 4,
        'axis_permutation': True,
        'feature_extractor': 'inceptionv4',
        'pixelwise_loss': 'jsd',
    },
}

# data_specs stand-ins with the attributes the reference drivers read (data_specs.py:26-64)
ImageSpecs = namedtuple('ImageSpecs', ['size', 'mean', 'stddev'])
JointsSpecs = namedtuple('JointsSpecs', ['skeleton', 'n_dims'])
DataSpecs = namedtuple('DataSpecs', ['input_specs', 'output_specs'])
IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STDDEV = [0.229, 0.224, 0.225]
SkeletonDesc = namedtuple('SkeletonDesc', ['joint_names', 'n_joints'])
# data/skeleton.py:51-74 -- only n_joints == 17 touches the hot path
CanonicalSkeletonDesc = SkeletonDesc(
    joint_names=['head_top', 'neck', 'right_shoulder', 'right_elbow', 'right_wrist', 'left_shoulder', 'left_elbow',
                 'left_wrist', 'right_hip', 'right_knee', 'right_ankle', 'left_hip', 'left_knee', 'left_ankle', 'pelvis',
                 'spine', 'head'],
    n_joints=17)

_BLOCK_TABLE = (   # (sequential name, index, kind, cin, cout) -- models/margipose_model.py:47-60
    ('down_layers', 0, 'regular', 128, 128), ('down_layers', 1, 'regular', 128, 128), ('down_layers', 2, 'down', 128, 192),
    ('down_layers', 3, 'regular', 192, 192), ('down_layers', 4, 'regular', 192, 192),
    ('up_layers', 0, 'regular', 192, 192), ('up_layers', 1, 'regular', 192, 192), ('up_layers', 2, 'up', 192, 128),
    ('up_layers', 3, 'regular', 128, 128), ('up_layers', 4, 'regular', 128, None))


def _conv_holder(kind, cin, cout, k):
    if kind == 'up':
        return nn.ConvTranspose2d(cin, cout, kernel_size=k, padding=k // 2, stride=2, output_padding=1, bias=False)
    return nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, stride=2 if kind == 'down' else 1, bias=False)


class _ParamBlock(nn.Module):
    """Parameter holder with ResidualBlock's key layout: module.{0,1,3,4}, shortcut.{0,1}."""

    def __init__(self, kind, cin, cout):
        super().__init__()
        self.module = nn.Sequential(_conv_holder(kind, cin, cout, 3), nn.BatchNorm2d(cout), nn.Identity(),
                                    nn.Conv2d(cout, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout),
                                    nn.Identity())
        self.shortcut = nn.Sequential(_conv_holder(kind, cin, cout, 1), nn.BatchNorm2d(cout))


class HeatmapColumn(nn.Module):
    """Parameter holder of one column (10 residual blocks, 4,739,599 parameters for 17 joints)."""

    def __init__(self, n_joints, heatmap_space):
        super().__init__()
        if heatmap_space not in ('xy', 'zy', 'xz'):
            raise Exception()
        self.n_joints = n_joints
        self.heatmap_space = heatmap_space
        seqs = {'down_layers': [], 'up_layers': []}
        for seq, _, kind, cin, cout in _BLOCK_TABLE:
            seqs[seq].append(_ParamBlock(kind, cin, n_joints if cout is None else cout))
        self.down_layers = nn.Sequential(*seqs['down_layers'])
        self.up_layers = nn.Sequential(*seqs['up_layers'])
        init_parameters(self)


class HeatmapCombiner(nn.Module):
    def __init__(self, n_joints):
        super().__init__()
        self.conv = nn.Conv2d(n_joints * 3, 128, kernel_size=1, bias=False)
        init_parameters(self)


def make_image_feature_extractor(model_name):
    if model_name == 'patch8':
        return nn.Sequential(nn.Conv2d(3, 128, kernel_size=8, stride=8, bias=False), nn.BatchNorm2d(128), nn.Identity())
    if model_name == 'inceptionv4':
        # parameter holders with pretrainedmodels' key layout; PyTorch default initialisation (the reference loads
        # ImageNet weights, which cannot be downloaded here) -- see stem.py for the "unpinned" caveat
        from ..stem import make_inceptionv4_stem_modules
        return make_inceptionv4_stem_modules()
    if model_name in {'resnet18', 'resnet34', 'resnet50'}:      # reference :119-138 (torchvision layers, see stem.py)
        from ..stem import make_resnet_stem_modules
        return make_resnet_stem_modules(model_name)
    raise Exception('unsupported image feature extractor model name: ' + model_name)


class _BackboneFn(torch.autograd.Function):
    """Whole backbone (stem, stages of 3 columns, combiners, softmax) as ONE autograd node."""

    @staticmethod
    def forward(ctx, engine, train, x, *params):
        hms, xyz, ectx = engine.forward(x, train, save=True)
        flat = tuple(hms[0]) + tuple(hms[1]) + tuple(hms[2])
        ctx.engine, ctx.ectx = engine, ectx
        ctx.save_for_backward(*flat)
        return flat

    @staticmethod
    def backward(ctx, *grads):
        engine, T = ctx.engine, ctx.engine.T
        saved = ctx.saved_tensors
        hms = [list(saved[p * T:(p + 1) * T]) for p in range(3)]
        g_hms = [list(grads[p * T:(p + 1) * T]) for p in range(3)]
        # (ctx.ectx stays: with retain_graph=True the node may run again; the saved activations go when the graph does)
        gflat, dx = engine.backward(ctx.ectx, hms, g_hms, ctx.needs_input_grad[2])
        if engine.dp is not None:       # the per-stage buckets were summed over replicas during the backward pass: the mean
            flat = torch.empty_like(gflat)
            _lib.check(_lib.lib().mpose_copy_div_f32(_lib.ptr(gflat), _lib.ptr(flat), _lib.c_float(float(engine.dp[1])), _lib.c_int64(gflat.numel()),
                                                     _lib.stream_ptr()), 'mpose_copy_div_f32')
        elif engine.grad_views:         # (PlannedTrainStep: the iteration's owner reads .grad before the next backward runs)
            flat = gflat
        else:
            flat = _lib.copy_into(torch.empty_like(gflat), gflat)      # (a launch of the library: a launch plan records it)
        out = engine.grads_from_flat(flat)
        return (None, None, dx) + tuple(out)


class MargiPoseModelInner(nn.Module):
    def __init__(self, n_joints, n_stages, axis_permutation, feature_extractor):
        super().__init__()
        self.n_stages = n_stages
        self.n_joints = n_joints
        self.in_cnn = make_image_feature_extractor(feature_extractor)
        self.feature_extractor_name = feature_extractor
        self.xy_hm_cnns = nn.ModuleList()
        self.zy_hm_cnns = nn.ModuleList()
        self.xz_hm_cnns = nn.ModuleList()
        self.hm_combiners = nn.ModuleList()
        names = ('xy', 'zy', 'xz') if axis_permutation else ('xy', 'xy', 'xy')
        self.spaces = tuple({'xy': 0, 'zy': 1, 'xz': 2}[n] for n in names)
        for t in range(n_stages):
            if t > 0:
                self.hm_combiners.append(HeatmapCombiner(n_joints))
            self.xy_hm_cnns.append(HeatmapColumn(n_joints, heatmap_space=names[0]))
            self.zy_hm_cnns.append(HeatmapColumn(n_joints, heatmap_space=names[1]))
            self.xz_hm_cnns.append(HeatmapColumn(n_joints, heatmap_space=names[2]))
        self._engine = None

    def engine(self):
        if self._engine is None:
            object.__setattr__(self, '_engine', Engine(self))
            if getattr(self, '_input_norm', None) is not None:
                self._engine.input_norm = self._input_norm
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self._engine is not None:
            self._engine.invalidate()
        return out

    def forward(self, *inputs):
        x = inputs[0]
        eng = self.engine()
        params = eng.param_list()
        T = self.n_stages
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            if getattr(self, 'heatmap_dtype', torch.float32) != torch.float32:
                raise _lib.MposeError('heatmap_dtype=bfloat16 is an inference storage mode: use model.eval() under torch.no_grad()')
            flat = _BackboneFn.apply(eng, self.training, x, *params)
            return list(flat[0:T]), list(flat[T:2 * T]), list(flat[2 * T:3 * T])
        bf16 = getattr(self, 'heatmap_dtype', torch.float32) == torch.bfloat16
        hms, xyz, _ = eng.forward(x, self.training, save=False, hm_bf16=bf16)
        object.__setattr__(self, '_last_xyz', xyz)
        return hms[0], hms[1], hms[2]


class MargiPoseModel(nn.Module):
    def __init__(self, skel_desc, n_stages, axis_permutation, feature_extractor, pixelwise_loss):
        super().__init__()
        self.data_specs = DataSpecs(ImageSpecs(256, mean=IMAGENET_MEAN, stddev=IMAGENET_STDDEV),
                                    JointsSpecs(skel_desc, n_dims=3))
        self.pixelwise_loss = pixelwise_loss
        self.inner = MargiPoseModelInner(skel_desc.n_joints, n_stages, axis_permutation, feature_extractor)
        # uint8 (B,3,H,W) frames are accepted too: `ImageSpecs.convert` (to_tensor + normalisation) then happens on
        # the device, fused into the stem's first load
        object.__setattr__(self.inner, '_input_norm', (list(IMAGENET_MEAN), list(IMAGENET_STDDEV)))
        self.xy_heatmaps = self.zy_heatmaps = self.xz_heatmaps = None

    def _pixelwise_flag(self):
        if self.pixelwise_loss == 'jsd':
            return True
        if self.pixelwise_loss is None:
            return False
        raise Exception('unrecognised pixelwise loss: {}'.format(self.pixelwise_loss))

    def _calculate_pixelwise_loss(self, hm, target_coords):
        sigma = 1.0
        if self._pixelwise_flag():
            return dsntnn.js_reg_losses(hm, target_coords.contiguous(), sigma)
        return 0

    def _stage_loop(self, target_var, three_d):
        pix = self._pixelwise_flag()
        target = target_var.narrow(-1, 0, 3).contiguous() if target_var.size(-1) >= 3 else \
            torch.cat([target_var.narrow(-1, 0, 2), torch.zeros_like(target_var.narrow(-1, 0, 1))], -1).contiguous()
        losses = None
        for xy_hm, zy_hm, xz_hm in zip(self.xy_heatmaps, self.zy_heatmaps, self.xz_heatmaps):
            # one fused launch per stage: JS of the plane(s) + DSNT + z-merge + Euclidean; `losses = 0; losses += ...` (reference
            # :238-252) with the additions as launches of the library
            losses = dsntnn.add_losses(losses, dsntnn.stage_losses(xy_hm, zy_hm, xz_hm, target, 1.0, pix, three_d))
        return losses

    def forward_2d_losses(self, out_var, target_var):
        """JS(xy) + Euclid(xy) summed over stages (reference :223-234)."""
        return self._stage_loop(target_var, three_d=False)

    def forward_3d_losses(self, out_var, target_var):
        """JS(xy) + JS(zy) + JS(xz) + Euclid(xyz) summed over stages (reference :236-252)."""
        return self._stage_loop(target_var, three_d=True)

    @staticmethod
    def heatmaps_to_coords(xy_hm, zy_hm, xz_hm):
        return dsntnn.heatmaps_to_coords(xy_hm, zy_hm, xz_hm)

    @property
    def heatmap_dtype(self):
        """torch.float32 (the reference's) or torch.bfloat16: inference-only storage mode of BASELINE configs[1] -- heatmaps are
        written (and read by the next stage's combiner) as bf16, the soft-argmax runs in fp32 on the unrounded softmax."""
        return getattr(self.inner, 'heatmap_dtype', torch.float32)

    @heatmap_dtype.setter
    def heatmap_dtype(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise _lib.MposeError('heatmap_dtype must be torch.float32 or torch.bfloat16')
        object.__setattr__(self.inner, 'heatmap_dtype', dtype)

    @property
    def conv_dtype(self):
        """torch.float32 (default: fp32-equivalent convolutions -- three exact fp16 products of two-way split, per-tensor-scaled
        operands) or the reduced-precision mode of BASELINE configs[4] ("fp16 convs with MFMA"):
          torch.float16   EVERY convolution of the model -- feature extractor and columns, forward, data- and weight-gradient --
                          multiplies operands ROUNDED to fp16 after the same per-tensor power-of-two scale (so fp16's 5-bit exponent
                          never overflows) in a single MFMA pass with fp32 accumulation (MPOSE_CONV_F16X1).
        BatchNorm, the losses and the soft-argmax stay fp32.  NOT within the 1e-4 parity bar of the fp32 path: the stated
        tolerances are in tests/test_model_gpu.py (against the fp64 oracle).  (Round 2's bf16 variant on the plane engine left the
        library in round 6: one convolution engine per precision mode.)"""
        return torch.float16 if self.inner.engine().conv_f16x1 else torch.float32

    @conv_dtype.setter
    def conv_dtype(self, dtype):
        if dtype not in (torch.float32, torch.float16):
            raise _lib.MposeError('conv_dtype must be torch.float32 or torch.float16')
        self.inner.engine().conv_f16x1 = dtype == torch.float16

    def forward(self, *inputs):
        self.xy_heatmaps, self.zy_heatmaps, self.xz_heatmaps = self.inner(*inputs)
        if self.xy_heatmaps[-1].dtype == torch.bfloat16:        # coordinates of the fp32 soft-argmax (same kernel, before rounding)
            return self.inner._last_xyz
        return self.heatmaps_to_coords(self.xy_heatmaps[-1], self.zy_heatmaps[-1], self.xz_heatmaps[-1])


def _caret_match(version, base):
    """semantic_version's '^a.b.c' for a >= 1: same major, not older than the base (model_factory.py:10-14)."""
    try:
        v = tuple(int(t) for t in version.split('-')[0].split('.')[:3])
    except ValueError:
        return False
    return len(v) == 3 and v[0] == base[0] and v >= base


def create_model(model_desc):
    """Registry entry point of reference models/__init__.py:16-27: types 'margipose' (^6.0.0) and 'chatterbox' (^1.3.0)."""
    type_name, version = model_desc['type'], str(model_desc['version'])
    if type_name == 'chatterbox' and _caret_match(version, (1, 3, 0)):          # ChatterboxModelFactory ('chatterbox', '^1.3.0')
        from .chatterbox_model import create_chatterbox_model
        return create_chatterbox_model(model_desc)
    if type_name != 'margipose' or version.split('.')[0] != '6':
        raise Exception('unrecognised model {} v{}'.format(type_name, version))
    s = model_desc['settings']
    return MargiPoseModel(skel_desc=CanonicalSkeletonDesc, n_stages=s.get('n_stages', 4),
                          axis_permutation=s.get('axis_permutation', True),
                          feature_extractor=s.get('feature_extractor', 'inceptionv4'),
                          pixelwise_loss=s.get('pixelwise_loss', 'jsd'))


def load_model(model_file):
    """reference models/__init__.py:30-34: {'model_desc', 'state_dict'} checkpoints."""
    details = torch.load(model_file, map_location='cpu')
    model = create_model(details['model_desc'])
    model.load_state_dict(details['state_dict'])
    return model
