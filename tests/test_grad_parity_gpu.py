"""Gradient parity of the HIP path (row a15 of SURVEY.md §8), with the ReLU pieces under control.

The network is piecewise linear in its ReLUs.  Two correct fp32 implementations whose pre-activations differ in the
last bit pick different pieces wherever a pre-activation sits at ~0, and the gradients of neighbouring pieces differ by
O(1e-3) of a tensor's norm -- that is why stock PyTorch fp32 deviates from its own fp64 run by 1e-4..1e-3 on this
model.  To separate "different piece" from "arithmetic error" these tests read the masks the GPU actually used
(pre-activation * scale + shift > 0, from the tensors the engine saved for backward) and force the fp64 / fp32 oracles
onto the same piece (oracle.model_ref.RELU_MASKS).  What is left is pure arithmetic error, and THAT is gated:

  * every parameter gradient and dL/dx within MASKED_TOL relative L2 of the fp64 oracle on the same piece;
  * the GPU's median error <= 1.5 x the fp32 oracle's (stock ATen CPU kernels) on the same piece.

`test_config_size_*` repeats the free-running comparison (no mask control) at BASELINE.json's training configuration
(batch 32, three stages), where batch statistics are no longer small-sample, against the fp64 and fp32 oracles."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import model_ref as R
from oracle import weights as W

pytestmark = pytest.mark.gpu
MASKED_TOL = 3e-5          # relative L2 per tensor against fp64 on the same ReLU piece (fp32 arithmetic through ~60 layers;
                           # the worst tensors are bias gradients = sums over pixels with heavy cancellation)
MASKED_TOL_CONFIG = 1e-4   # the same at the configuration size (B=32: sums over 32768 pixels) = BASELINE.json configs[2]'s own
                           # "grads within 1e-4".  Measured worst tensor: 4.6e-5 with the default three-product fp16 convolutions
                           # (the shortcut BatchNorm bias gradients of each stage's first block: the 2^-22 representation
                           # residual of a WEIGHT is the same for every pixel, so it does not average out of a sum over pixels
                           # the way activation roundings do), 1e-5 on the plane engine (six bf16 products, two accumulators).
P99_RATIO = 4.0            # same-piece p99 against the fp32 oracle's p99 at the configuration's size
FREE_RATIO = 3.0           # free running, the GPU may deviate from fp64 by this multiple of what the fp32 oracle does.  Which
                           # ReLU sites flip is luck, how many scales with the forward error: 6.6e-6 on a heatmap for the default
                           # form, 2.6e-6 for the plane engine (gated at 1.5 below), 1.4e-5 for round 1's six-product conv_igemm_k


from oracle.piece import (build, gpu_relu_masks, gpu_step, gpu_stem_masks, oracle_grads, oracle_grads_pair,  # noqa: F401
                          rel_l2)


def compare(name, gpu, ref64, ref32, extra=None):
    typical = float(np.median([float(v.norm()) for v in ref64.values()]))
    e_gpu, e_ref, zero = {}, {}, {}
    for k, r in ref64.items():
        if float(r.norm()) < 1e-9 * typical:       # analytically zero (last shortcut BN bias: softmax is shift invariant)
            zero[k] = float(gpu[k].double().norm()) / typical
            continue
        e_gpu[k] = rel_l2(gpu[k], r)
        e_ref[k] = rel_l2(ref32[k].double(), r)
    vg, vr = np.array(list(e_gpu.values())), np.array(list(e_ref.values()))
    stats = {'n': len(vg), 'gpu_median': float(np.median(vg)), 'gpu_p99': float(np.quantile(vg, 0.99)), 'gpu_max': float(vg.max()),
             'ref32_median': float(np.median(vr)), 'ref32_p99': float(np.quantile(vr, 0.99)), 'ref32_max': float(vr.max()),
             'zero_grad_abs_max': max(zero.values()) if zero else 0.0,
             'worst_gpu': sorted(e_gpu.items(), key=lambda kv: -kv[1])[:8]}
    if extra:
        stats.update(extra)
    os.makedirs('gpurun_out', exist_ok=True)
    with open(os.path.join('gpurun_out', 'gradparity_%s.json' % name), 'w') as f:
        json.dump(dict(stats, per_key={k: [e_gpu[k], e_ref[k]] for k in e_gpu}), f, indent=1)
    print(name, {k: v for k, v in stats.items() if k != 'worst_gpu'})
    return stats


def mask_flips(a, b):
    n = sum(int(a[k].numel()) for k in a)
    return sum(int((a[k] != b[k]).sum()) for k in a), n


@pytest.mark.parametrize('T,B,engine', [(1, 3, 'auto'), (2, 2, 'auto'), (1, 2, 'inceptionv4'),
                                        pytest.param(1, 8, 'auto', marks=pytest.mark.skipif(os.environ.get('MPOSE_LONG_TESTS', '0') == '0',
                                                                                            reason='suite time budget: MPOSE_LONG_TESTS=1'))])
def test_grads_on_the_same_relu_piece(T, B, engine):
    """What training runs (one convolution engine per launch kind since round 6): three fp16 products everywhere; the regular
    128-channel blocks on conv_h2r_k with planes end to end (forward, both data gradients, both weight gradients read
    producer-written fp16 planes under a-priori bounds), conv_igemm_k / the row-of-taps weight gradient elsewhere.  B = 3: an odd
    batch, ragged last tiles.  'inceptionv4' = the same behind the reference's default feature extractor, its ReLU / max-pool pieces
    controlled too.  (Three stages on a common piece: the configuration-size test below.  The free-running fp64 pass -- how many
    ReLU sites sit on another piece, and what that alone does to the gradients -- runs for ONE case: it is a CPU fp64 backward
    pass per case and the suite has a time budget.)"""
    seed = 700 + 10 * T + B
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    rng = np.random.default_rng(seed)
    mask = torch.tensor((rng.uniform(0, 1, (B, 17)) > 0.2).astype(np.float32))
    m, sd = build(T, seed, x, 'inceptionv4' if engine == 'inceptionv4' else 'patch8')
    tag = 'T%d_B%d' % (T, B)
    if engine == 'inceptionv4':
        tag += '_inceptionv4'
    gpu, masks, loss_gpu, _ = gpu_step(m, x, target, mask)
    ref64, loss64 = oracle_grads(sd, T, x, target, mask, torch.float64, masks=masks)
    ref32, _ = oracle_grads(sd, T, x, target, mask, torch.float32, masks=masks)
    extra = {}
    if (T, B, engine) == (2, 2, 'auto'):
        own = {}
        free64, _ = oracle_grads(sd, T, x, target, mask, torch.float64, record=own)      # the oracle on ITS piece (for the record)
        flips, total = mask_flips(masks, own)
        free = compare('free_' + tag, gpu, free64, ref32)
        extra = {'relu_sites_flipped_vs_fp64': flips, 'relu_sites': total, 'free_running_gpu_median': free['gpu_median'],
                 'free_running_gpu_max': free['gpu_max']}
    st = compare('masked_' + tag, gpu, ref64, ref32, extra)
    assert abs(loss_gpu - loss64) <= 1e-5 * abs(loss64)
    # (the InceptionV4 case at B=2 is a worse-conditioned problem for EVERY fp32 implementation: the fp32 oracle itself sits at
    #  median 2.0e-5 / max 4.8e-5 of fp64 there, the GPU at 2.3e-5 / 4.4e-5 -- its absolute gate is the fp32 oracle's own level)
    assert st['gpu_max'] <= max(MASKED_TOL, 1.5 * st['ref32_max']), st
    assert st['gpu_max'] <= 1e-4, st
    assert st['gpu_median'] <= max(1.5 * st['ref32_median'], 2e-6), st
    assert st['zero_grad_abs_max'] < 1e-5, st


_LONG = os.environ.get('MPOSE_LONG_TESTS', '0') != '0'


@pytest.mark.parametrize('stem', [pytest.param('patch8', marks=pytest.mark.skipif(
    not _LONG, reason='two more CPU backward passes at B=32 (2-4 minutes of oracle time: the suite has a 30-minute limit on the '
                      "pool's slowest hosts); MPOSE_LONG_TESTS=1 (tools/final_check.sh) runs it and the free-running comparisons -- "
                      'profiles/r5_gradient_parity.json holds the numbers.  The inceptionv4 case below keeps a configuration-size same-piece '
                      'gate on the SAME column launches (H2 + igemm) in the default suite')),
    'inceptionv4'])
def test_config_size_train_step_gradients(stem):
    """BASELINE.json configs[2]: batch 32, three stages, JS + Euclidean loss -- every gradient against the fp64 and the fp32
    oracle on a COMMON piece (the oracle forced onto the ReLU masks and max-pool window choices the GPU used): pure arithmetic
    error at the configuration's size, for the in-repo patch8 stem and (round 4) for the reference's default InceptionV4 feature
    extractor, whose ReLU and max-pool sites are controlled too (gpu_stem_masks).  Free-running comparisons (every implementation
    on its own piece) run at small sizes: test_grads_on_the_same_relu_piece[2-2-auto], tests/test_model_gpu.py's gradient-noise
    gates, and profiles/r2_gradient_parity.json holds the B=32 ones of round 2."""
    T, B, seed = 3, 32, 900
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    m, sd = build(T, seed, x, stem)
    gpu, masks, loss_gpu, _ = gpu_step(m, x, target, mask)
    m64, loss64, m32 = oracle_grads_pair(sd, T, x, target, mask, masks=masks)
    sm = compare('config_%s_T3_B32_masked' % stem, gpu, m64, m32)
    assert abs(loss_gpu - loss64) <= 1e-5 * abs(loss64)
    assert sm['gpu_max'] <= MASKED_TOL_CONFIG, sm
    assert sm['gpu_median'] <= 1.5 * sm['ref32_median'], sm
    # the tail, not only the median: the worst percentile sits above the fp32 oracle's (the weight-residual bias of the
    # first blocks' shortcut BatchNorm bias gradients, see MASKED_TOL_CONFIG) -- gated so that it cannot grow unnoticed
    assert sm['gpu_p99'] <= P99_RATIO * sm['ref32_p99'], sm
    if _LONG:
        # REPORTED, not gated: every implementation on its OWN piece at the configuration's size -- the GPU and the
        # fp32 oracle against the free-running fp64 oracle.  Which ReLU sites flip is luck; profiles/r5_gradient_parity.json keeps it.
        f64, _, f32 = oracle_grads_pair(sd, T, x, target, mask, masks=None)
        compare('config_%s_T3_B32_free' % stem, gpu, f64, f32)
