"""Training-step throughput of ChatterboxModel (reference models/chatterbox_model.py) on one MI355X, in the shape of
bench.py's JSON line (which stays on the MargiPose workload BASELINE.json names): whole step = forward + JS/Euclidean loss +
backward + SGD on synthetic 256x256 frames resident in HBM; `roofline` = the convolution launch that takes the most time,
measured with HIP events inside the timed region (one eager step in forty); `cpu_baseline` = oracle/chatterbox_ref.py on the
host cores.

    python tools/bench_chatterbox.py [--batch 32] [--steps 20] [--warmup 3] > profiles/r3_bench_chatterbox.json
"""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
PEAK_16BIT_MFMA_TFLOPS = 2500.0


def cpu_baseline(cpu_batch):
    from oracle import chatterbox_ref as C
    from oracle import model_ref as R
    from oracle import weights as W
    sd = W.fill_like(W.chatterbox_schema(), 12345)
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
    x, target, mask = W.seeded_inputs(12345, cpu_batch)

    def step():
        for p in params.values():
            p.grad = None
        coords, hms = C.chatterbox_forward(sd, x, True)
        R.average_loss(C.chatterbox_losses(hms, target), mask).backward()
    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        dt = time.perf_counter() - t0
        if n >= 3 and (dt > 12.0 or n >= 6):
            break
    return {'value': cpu_batch * n / dt, 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d timed fwd+loss+bwd steps of batch %d (256x256, fp32) with oracle/chatterbox_ref.py on torch CPU' % (n, cpu_batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--cpu-batch', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the iteration from one HIP graph (train_helpers.GraphedTrainStep)')
    args = ap.parse_args()
    from margipose_amd import dsntnn
    from margipose_amd.engine import KernelTimer
    from margipose_amd.models import CanonicalSkeletonDesc, ChatterboxModel
    from margipose_amd.train_helpers import DeviceSGD, GraphedTrainStep
    device = torch.device('cuda', 0)
    torch.manual_seed(12345)
    model = ChatterboxModel(CanonicalSkeletonDesc, 'jsd').to(device).train()
    opt = DeviceSGD(model.parameters(), lr=0.01, momentum=0.9)
    g = torch.Generator(device='cpu').manual_seed(12345)
    B = args.batch
    x = torch.randn(B, 3, 256, 256, generator=g).to(device)
    target = (torch.rand(B, 17, 3, generator=g) * 2 - 1).to(device)
    mask = torch.ones(B, 17, device=device)

    def step():
        out = model(x)
        loss = dsntnn.average_loss(model.forward_3d_losses(out, target), mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    eager_step = step
    if args.graph:
        graphed = GraphedTrainStep(model, opt, x, target, mask, warmup=2)
        step = lambda: graphed()[1]
    for _ in range(args.warmup):
        loss = step()
    timer = KernelTimer()
    timer.calibrate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % 40 == 0:
            model.engine().timer = timer
            loss = eager_step()
            model.engine().timer = None
        else:
            loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {'metric': 'images/sec fwd+bwd at 256x256, 17 joints (training step: forward + JS/Euclidean loss + backward + SGD)',
           'value': B * args.steps / dt, 'unit': 'images/sec', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 (3xfp16 split operands, fp32 accumulate)', 'data': 'synthetic',
           'config': {'workload': 'ChatterboxModel (reference models/chatterbox_model.py; NOT the workload BASELINE.json names): training '
                                  'step, batch %d, ResNet-34 conv1..layer2 + dilated layer3/4 xy head + two one-axis chatterbox heads, '
                                  '256x256 input, 17 joints, 32x32 heatmaps, JS + Euclidean loss, SGD(momentum 0.9); torchvision layers '
                                  'restated (unpinned), random init' % B,
                      'global_batch': B, 'final_loss': float(loss.detach()),
                      'step_dispatch': 'hip graph replay (train_helpers.GraphedTrainStep)' if args.graph else 'eager launches'}}
    summ = timer.summary()
    convs = {k: v for k, v in summ.items() if k.startswith('conv:') or k.startswith('wgrad:')}
    top = max(convs.items(), key=lambda kv: kv[1]['total_ms'])
    tf = top[1]['work_per_launch'] / (top[1]['avg_us'] * 1e-6) / 1e12
    all_flops = sum(v['work'] for v in convs.values())
    all_ms = sum(v['total_ms'] for v in convs.values())
    peak = PEAK_16BIT_MFMA_TFLOPS / 3.0
    res['roofline'] = {'bound': 'mfma', 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak, 'traffic': None,
                       'kernel': top[0], 'avg_launch_us': top[1]['avg_us'], 'launches': top[1]['n'],
                       'flops_per_launch': top[1]['work_per_launch'],
                       'note': 'achieved = algorithmic fp32 FLOPs / launch duration (HIP events, one eager step); three 16-bit MFMA '
                               'products per multiply-add, peak = 2500 / 3',
                       'all_convs': {'tflops': all_flops / (all_ms * 1e-3) / 1e12, 'frac': all_flops / (all_ms * 1e-3) / 1e12 / peak,
                                     'ms_per_step': all_ms, 'gflop_per_step': all_flops / 1e9},
                       'kernels_ms_per_step': sum(v['total_ms'] for v in summ.values()),
                       'top5': [{'kernel': k, 'avg_us': v['avg_us'], 'n': v['n'], 'total_ms': v['total_ms'],
                                 'tflops': (v['work_per_launch'] / (v['avg_us'] * 1e-6) / 1e12) if v['work_per_launch'] else None}
                                for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['total_ms'])[:5]]}
    if not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(args.cpu_batch)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
