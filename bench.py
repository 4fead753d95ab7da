#!/usr/bin/env python
"""Benchmark of the MargiPose hot path on MI355X (driver contract: see the task statement).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (launches its own N ranks under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic frames: forward (stem + n_stages x 3
columns + soft-argmax) + 3D loss (JS + Euclidean) + backward + (N > 1: ONE all-reduce of the flat gradient
buffer over RCCL) + SGD update.  Workload = BASELINE.json configs[2] ("1xMI355X training step, batch=32,
JS-reg + pixelwise loss on"): the config the metric "images/sec fwd+bwd at 256x256, 17 joints" is quoted on,
with the 3-stage model of configs[1].  fp32 arithmetic throughout (the reference's precision).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (an implicit-GEMM convolution or its
weight gradient: fp32 arithmetic carried out as THREE fp16 MFMA products per multiply-add of two-way split,
per-tensor-scaled operands, so the bound is the dense 16-bit MFMA peak / 3; algorithmic fp32 FLOPs / HIP-event
launch duration, nothing subtracted: from a fully bracketed serial step run between the warm-up and the timed steps,
with the same kernel's duration inside the timed region -- where the side stream's launches share the GPU with it --
reported beside it as `in_timed_region`);
`tail_roofline` is the soft-argmax kernel the metric also names (algorithmic bytes / launch duration, HBM
bound); `cpu_baseline` times the oracle (the stock-PyTorch CPU restatement of the reference, oracle/model_ref.py)
on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL / cross-process tensor sharing fail with the legacy IPC mode (set before HIP starts)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0    # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
PEAK_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0     # fp32-equivalent: 6 bf16 products per fp32 multiply-add
PEAK_FP32_MFMA_TFLOPS = 157.3     # for comparison: v_mfma_f32_32x32x2_f32 (what a plain fp32 MFMA kernel is bound by)
PEAK_HBM_GBPS = 8000.0


def pmc_traffic(label):
    """HBM-side bytes per launch of the kernel `label` from the committed rocprofv3 PMC pass (profiles/*_pmc_traffic.json,
    written by tools/summarize_profile.py from FETCH_SIZE / WRITE_SIZE collected in their own passes), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if label in d:
            return d[label]
    return None


def traffic_fields(label):
    t = pmc_traffic(label)
    if t is None:
        return None, None
    return t.get('hbm_bytes_per_launch'), t


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (weak scaling)')
    ap.add_argument('--stages', type=int, default=3)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--stem', default='inceptionv4', choices=['patch8', 'inceptionv4', 'resnet18', 'resnet34', 'resnet50'],
                    help="'inceptionv4' = the reference's default feature extractor; 'resnet*' = its other options; 'patch8' = the light in-repo stem")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-inference', action='store_true', help='skip the configs[1] inference micro-benchmark (profiling runs)')
    ap.add_argument('--no-configs4', action='store_true', help='skip the short BASELINE configs[4] leg (5 stages, 384x384, fp16 convolutions)')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--no-overlap-wgrad', action='store_true', help='keep the weight-gradient GEMMs on the main stream (the default '
                    'runs them on a side stream, +2.3 %% step rate; the steps whose kernels are bracketed by HIP events for the '
                    'roofline block always run serially, so per-kernel durations are clean)')
    ap.add_argument('--graph', action='store_true', help='replay the captured HIP graph of the iteration (train_helpers.GraphedTrainStep) instead '
                    'of enqueueing ~1000 launches per step from Python.  Bit-identical results; the step is GPU-bound, and a replay '
                    'measured ~1 %% SLOWER than eager launches (37.2 vs 36.8 ms, one box), so the benchmark default is eager')
    ap.add_argument('--eager', action='store_true', help='issue every launch from Python (the default replays the iteration from a launch '
                    'plan: train_helpers.PlannedTrainStep -- the same schedule without its host cost)')
    ap.add_argument('--no-plan', action='store_true', help='same as --eager')
    ap.add_argument('--conv-dtype', default='f32', choices=['f32', 'f16'], help="BASELINE configs[4]'s reduced-precision "
                    "convolutions -- 'f16': every convolution on fp16-rounded operands, one MFMA product (model.conv_dtype = "
                    "torch.float16; with --stages 5 --size 384 that is configs[4]'s workload).  A DIFFERENT workload, reported with its own "
                    'dtype, never the headline')
    return ap.parse_args()


def cpu_baseline(stages, size, cpu_batch, stem='inceptionv4'):
    """The oracle's training step on the host CPU (bounded sample)."""
    from collections import OrderedDict
    from oracle import model_ref as R
    from oracle import weights as W
    threads = torch.get_num_threads()
    sd = W.make_state_dict(stages, 12345, stem=stem)
    params = OrderedDict((k, v.requires_grad_(True)) for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
    x, target, mask = W.seeded_inputs(12345, cpu_batch, size)

    def step():
        for p in params.values():
            p.grad = None
        R.train_step_reference(sd, x, target, mask, stages)

    step()                                  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:                             # at least 3 timed steps, then until ~12 s of CPU work or 6 steps
        step()
        n += 1
        dt = time.perf_counter() - t0
        if n >= 3 and (dt > 12.0 or n >= 6):
            break
    return {'value': cpu_batch * n / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': '%d timed fwd+loss+bwd steps of batch %d (T=%d, %dx%d, fp32) with oracle/model_ref.py on torch CPU, '
                      '%d threads, %s stem' % (n, cpu_batch, stages, size, size, threads, stem)}


def tail_microbench(device, B, bf16_out=False, launches=200):
    """The soft-argmax kernel (flat_softmax + dsnt + heatmaps_to_coords, 3 planes per launch) on its own: `launches`
    back-to-back launches between two HIP events on the launch stream, no calibration term subtracted -- the quotient is what
    rocprofv3 reports as the kernel's average duration plus the ~1.5 us dependent-launch boundary (profiles/ holds the trace
    of tools/prof_tail.py, the same loop).  Algorithmic bytes: logits read once + heatmaps written once + coords."""
    from margipose_amd import _lib
    F = 32
    lg = [torch.randn(B, 17, F, F, device=device) * 4 for _ in range(3)]
    hm = [torch.empty(B, 17, F, F, device=device, dtype=torch.bfloat16 if bf16_out else torch.float32) for _ in range(3)]
    xyz = torch.empty(B, 17, 3, device=device)
    L = _lib.lib()

    def run():
        _lib.check(L.mpose_softmax_dsnt_fwd(_lib.ptr_array(lg), _lib.ptr_array(hm), None, _lib.ptr(xyz), 3, B * 17, F, F,
                                            2 if bf16_out else 0, _lib.stream_ptr()), 'softmax')
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(launches):
        run()
    e.record()
    torch.cuda.synchronize()
    sec = s.elapsed_time(e) * 1e-3 / launches
    nbytes = 3 * B * 17 * F * F * (6 if bf16_out else 8) + B * 17 * 12
    return {'bound': 'hbm', 'achieved': nbytes / sec / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': nbytes / sec / 1e9 / PEAK_HBM_GBPS,
            'traffic': None, 'us_per_launch': sec * 1e6, 'bytes_per_launch': nbytes, 'launches': launches,
            'kernel': 'softmax_dsnt_fwd_k (3 planes, B=%d, %s heatmaps)' % (B, 'bf16' if bf16_out else 'fp32')}


def fused_tail_microbench(device, B, bf16_out=False, launches=200):
    """What a training / inference step launches for the soft-argmax since round 4: the last ResidualBlock's residual sum, flat_softmax
    and dsnt as one kernel (mpose_bn_add_softmax_fwd, tail.hip bn_add_softmax_k).  Algorithmic bytes: the 17 joint channels of the
    two NHWC inputs read once, the heatmaps written once, the plane coordinates."""
    from margipose_amd import _lib
    from margipose_amd._lib import BnAddOperands
    F, C, J = 32, 32, 17
    a = [torch.randn(B, F, F, C, device=device) for _ in range(3)]
    b = [torch.randn(B, F, F, C, device=device) for _ in range(3)]
    v = [torch.randn(C, device=device) for _ in range(4)]
    hm = [torch.empty(B, J, F, F, device=device, dtype=torch.bfloat16 if bf16_out else torch.float32) for _ in range(3)]
    pc = torch.empty(3, B * J, 2, device=device)
    ops = []
    for c in range(3):
        ao = BnAddOperands()
        ao.a, ao.a_scale, ao.a_shift = a[c].data_ptr(), v[0].data_ptr(), v[1].data_ptr()
        ao.b, ao.b_scale, ao.b_shift = b[c].data_ptr(), v[2].data_ptr(), v[3].data_ptr()
        ops.append(ao)
    ops = (BnAddOperands * 3)(*ops)
    L = _lib.lib()

    def run():
        _lib.check(L.mpose_bn_add_softmax_fwd(ops, _lib.ptr_array(hm), _lib.ptr(pc), 3, B, F, F, C, J, 2 if bf16_out else 0,
                                              _lib.stream_ptr()), 'bn_add_softmax')
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(launches):
        run()
    e.record()
    torch.cuda.synchronize()
    sec = s.elapsed_time(e) * 1e-3 / launches
    nbytes = 3 * B * J * F * F * (10 if bf16_out else 12) + 3 * B * J * 8
    return {'bound': 'hbm', 'achieved': nbytes / sec / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': nbytes / sec / 1e9 / PEAK_HBM_GBPS,
            'traffic': None, 'us_per_launch': sec * 1e6, 'bytes_per_launch': nbytes, 'launches': launches,
            'kernel': 'bn_add_softmax_k (residual sum + flat_softmax + dsnt, 3 columns, B=%d, %s heatmaps)' % (B, 'bf16' if bf16_out else 'fp32')}


def train_tail_microbench(device, B, launches=200):
    """The training tail's other three launches per stage: stage_loss_fwd_k (3 x JS + DSNT + z-merge + Euclidean: the
    heatmaps read once, 4 E bytes, E = 3 B 17 F^2), stage_loss_bwd_k (heatmaps read, their gradient written: 8 E) and softmax_bwd_k
    (heatmaps + the loss gradient + the next stage's combiner gradient read, the logits' gradient written: 16 E; 12 E at the last
    stage).  Same method as tail_microbench: back-to-back launches between two HIP events, nothing subtracted."""
    from margipose_amd import _lib
    F, J = 32, 17
    L = _lib.lib()
    hm = [torch.softmax(torch.randn(B, J, F * F, device=device) * 4, -1).view(B, J, F, F).contiguous() for _ in range(3)]
    tgt = (torch.rand(B, J, 3, device=device) * 2 - 1).contiguous()
    losses = torch.empty(B, J, device=device)
    xyz = torch.empty(B, J, 3, device=device)
    g = torch.full((B, J), 1.0 / (B * J), device=device)
    d_hm = [torch.empty_like(h) for h in hm]
    g_comb = [torch.randn_like(h) * 1e-3 for h in hm]
    d_log = [torch.empty_like(h) for h in hm]
    E = 3 * B * J * F * F
    cf = _lib.c_float

    def fwd():
        _lib.check(L.mpose_stage_loss_fwd(_lib.ptr_array(hm), _lib.ptr(tgt), _lib.ptr(losses), _lib.ptr(xyz), B * J, F, F, cf(1.0), 1, 1, 0,
                                          _lib.stream_ptr()), 'stage_loss_fwd')

    def bwd():
        _lib.check(L.mpose_stage_loss_bwd(_lib.ptr_array(hm), _lib.ptr(tgt), _lib.ptr(xyz), _lib.ptr(g), _lib.ptr_array(d_hm), B * J, F, F, cf(1.0),
                                          1, 1, 0, _lib.stream_ptr()), 'stage_loss_bwd')

    def sbwd():
        _lib.check(L.mpose_softmax_bwd(_lib.ptr_array(hm), _lib.ptr_array(d_hm), _lib.ptr_array(g_comb), _lib.ptr_array(d_log), 3, B * J, F * F,
                                       _lib.stream_ptr()), 'softmax_bwd')

    out = {}
    for name, fn, nbytes in (('stage_loss_fwd_k', fwd, 4 * E), ('stage_loss_bwd_k', bwd, 8 * E), ('softmax_bwd_k', sbwd, 16 * E)):
        for _ in range(5):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(launches):
            fn()
        e.record()
        torch.cuda.synchronize()
        sec = s.elapsed_time(e) * 1e-3 / launches
        out[name] = {'us_per_launch': sec * 1e6, 'bytes_per_launch': nbytes, 'achieved_GBps': nbytes / sec / 1e9, 'frac': nbytes / sec / 1e9 / PEAK_HBM_GBPS}
    return out


def configs4_leg(device, steps=8, warmup=3, batch=32):
    """BASELINE configs[4] on one GPU, short: 5-stage MargiPose at 384 x 384 (48 x 48 heatmaps), convolution operands rounded to
    fp16 (one MFMA product per multiply-add, fp32 accumulate), same loss and optimiser as the headline step.  One extra step runs
    with per-kernel events for its own roofline block (peak = the full dense 16-bit MFMA peak)."""
    from margipose_amd import dsntnn
    from margipose_amd.engine import KernelTimer
    from margipose_amd.train_helpers import DeviceSGD
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    torch.manual_seed(12345)
    model = MargiPoseModel(CanonicalSkeletonDesc, 5, True, 'inceptionv4', 'jsd').to(device).train()
    model.conv_dtype = torch.float16
    opt = DeviceSGD(model.parameters(), lr=0.01, momentum=0.9)
    g = torch.Generator(device='cpu').manual_seed(12345)
    x = torch.randn(batch, 3, 384, 384, generator=g).to(device)
    target = (torch.rand(batch, 17, 3, generator=g) * 2 - 1).to(device)
    mask = torch.ones(batch, 17, device=device)

    def step():
        out = model(x)
        loss = dsntnn.average_loss(model.forward_3d_losses(out, target), mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    # (the same dispatch as the headline step: the iteration from a launch plan, so that the figure does not depend on the host)
    planned, dispatch = None, 'eager launches (Python / ctypes)'
    if True:
        try:
            from margipose_amd.train_helpers import PlannedTrainStep
            planned = PlannedTrainStep(model, opt, x, target, mask, warmup=1)
            dispatch = 'launch plan replay (%d launches)' % planned.n_launches
        except Exception as e:
            sys.stderr.write('bench.py: configs[4] launch-plan recording failed (%s: %s); running eagerly\n' % (type(e).__name__, e))
            planned = None
    run = (lambda: planned()[1]) if planned is not None else step
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del planned
    res = {'workload': 'BASELINE configs[4] on 1 GPU: training step, batch %d, 5-stage MargiPose, 384x384 input, 48x48 heatmaps, '
                       'convolutions on fp16-rounded operands (fp32 accumulate; BatchNorm / loss / soft-argmax fp32)' % batch,
           'images_per_sec': batch / dt, 'ms_per_step': 1e3 * dt, 'steps': steps, 'warmup': warmup, 'dtype': 'f16', 'step_dispatch': dispatch,
           'final_loss': float(loss.detach())}
    timer = KernelTimer()
    timer.calibrate()
    eng = model.inner.engine()
    eng.timer = timer
    step()
    eng.timer = None
    torch.cuda.synchronize()
    summ = timer.summary()
    convs = {k: v for k, v in summ.items() if k.startswith('conv:') or k.startswith('wgrad:')}
    if convs:
        top = max(convs.items(), key=lambda kv: kv[1]['total_ms'])
        tf = top[1]['work_per_launch'] / (top[1]['avg_us'] * 1e-6) / 1e12
        all_flops = sum(v['work'] for v in convs.values())
        all_ms = sum(v['total_ms'] for v in convs.values())
        res['roofline'] = {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_BF16_MFMA_TFLOPS,
                           'traffic': None, 'kernel': top[0], 'avg_launch_us': top[1]['avg_us'], 'launches': top[1]['n'],
                           'all_conv_kernels_tflops': all_flops / (all_ms * 1e-3) / 1e12,
                           'all_conv_kernels_frac': all_flops / (all_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                           'note': 'one MFMA product per multiply-add: priced against the full dense 16-bit MFMA peak; HIP events '
                                   'around every convolution launch of one serial step'}
    del model, opt, x
    torch.cuda.empty_cache()
    return res


def inference_microbench(model, device, size):
    """BASELINE configs[1]: batch-64 eval-mode forward (fp32), reported next to the training metric."""
    model.eval()
    x = torch.randn(64, 3, size, size, device=device)
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the same forward from a launch plan (train_helpers.PlannedInference): no Python between its ~250 launches
        planned_ips, plan_note, frozen_ips = None, None, None
        try:
            from margipose_amd.train_helpers import PlannedInference
            pf = PlannedInference(model, x)
            for _ in range(2):
                pf()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                pf()
            torch.cuda.synchronize()
            planned_ips = 64 * n / (time.perf_counter() - t0)
            plan_note = '%d launches re-issued from one C loop' % pf.n_launches
            del pf
            pf = PlannedInference(model, x, frozen_weights=True)     # (the weights are packed once, not per forward: serving)
            for _ in range(2):
                pf()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                pf()
            torch.cuda.synchronize()
            frozen_ips = 64 * n / (time.perf_counter() - t0)
            del pf
        except Exception as e:
            plan_note = 'recording failed: %s: %s' % (type(e).__name__, e)
        model.heatmap_dtype = torch.bfloat16       # configs[1] as worded: bf16 heatmap storage + fp32 soft-argmax
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
        dt16 = time.perf_counter() - t0
        model.heatmap_dtype = torch.float32
    model.train()
    return {'images_per_sec': 64 * n / dt, 'batch': 64, 'ms_per_forward': 1e3 * dt / n, 'dtype': 'f32',
            'images_per_sec_bf16_heatmaps': 64 * n / dt16,
            'images_per_sec_launch_plan': planned_ips, 'launch_plan': plan_note,
            'images_per_sec_launch_plan_frozen_weights': frozen_ips,
            'note': 'eval-mode forward (running-stat BatchNorm), heatmaps + coordinates for all stages; the bf16 figure stores '
                    'the heatmaps as bf16 (fp32 convolutions and soft-argmax): they are <1% of the bytes, so it is the same rate'}


def self_launch(args):
    """`python bench.py --gpus N` on its own (no torchrun around it): re-execute this command under torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1 at a free port -- the driver's own N > 1 form, which keeps working unchanged because
    it sets WORLD_SIZE.  Rank 0's JSON line reaches this process's stdout through the launcher."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus))))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    from margipose_amd import dsntnn, parallel
    from margipose_amd.engine import KernelTimer
    from margipose_amd.train_helpers import DeviceSGD, GraphedTrainStep, PlannedTrainStep
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit('--gpus %d does not match WORLD_SIZE %d (launch with torch.distributed.run)' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the hot path has no CPU fallback')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    torch.manual_seed(12345)                # the seed the reference's eval/infer use (bin/eval_3d.py:123)
    model = MargiPoseModel(CanonicalSkeletonDesc, args.stages, True, args.stem, 'jsd').to(device).train()
    parallel.broadcast_parameters(model)
    parallel.attach(model)
    model.inner.engine().overlap_wgrad = not args.no_overlap_wgrad
    if args.conv_dtype == 'f16':
        model.conv_dtype = torch.float16
    # the reference's optimiser, SGD(lr, momentum) (bin/train_3d.py:339), as one launch with device-resident hyper-parameters
    opt = DeviceSGD(model.parameters(), lr=0.01, momentum=0.9)
    g = torch.Generator(device='cpu').manual_seed(12345 + rank)
    B = args.batch
    x = torch.randn(B, 3, args.size, args.size, generator=g).to(device)
    target = (torch.rand(B, 17, 3, generator=g) * 2 - 1).to(device)
    mask = torch.ones(B, 17, device=device)

    def eager_step():
        out = model(x)
        loss = dsntnn.average_loss(model.forward_3d_losses(out, target), mask)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The iteration is captured once as a HIP graph and replayed (every buffer of a step has a fixed address); eager mode costs
    # ~1000 launches of host time per step.  Under data parallelism the graph would have to contain the RCCL all-reduces: opt-in.
    use_graph = args.graph and not args.eager and (world == 1 or os.environ.get('MPOSE_DP_GRAPH') == '1')
    graphed = None
    if use_graph:
        try:
            graphed = GraphedTrainStep(model, opt, x, target, mask, warmup=2)
        except Exception as e:          # a failed capture must not cost the measurement: fall back to eager launches
            sys.stderr.write('bench.py: HIP graph capture failed (%s: %s); running eagerly\n' % (type(e).__name__, e))
            torch.cuda.synchronize()
            use_graph = False

    # Default dispatch on one GPU: the iteration recorded once as a LAUNCH PLAN (csrc/plan.hip) and re-issued from a C loop -- the
    # eager two-stream schedule, kernel for kernel, without ~700 Python-issued launches per step (15 ms of host time on the pool's
    # fast hosts, 28 ms on its slow ones, where the eager step is host-bound).  --eager issues every launch from Python.
    planned = None
    if graphed is None and not args.eager and not args.no_plan:
        try:
            planned = PlannedTrainStep(model, opt, x, target, mask, warmup=2)
        except Exception as e:          # a failed recording must not cost the measurement: fall back to eager launches
            sys.stderr.write('bench.py: launch-plan recording failed (%s: %s); running eagerly\n' % (type(e).__name__, e))
            torch.cuda.synchronize()
            planned = None

    def step():
        if graphed is not None:
            return graphed()[1]
        if planned is not None:
            return planned()[1]
        return eager_step()

    for _ in range(args.warmup):
        loss = step()
    # Per-kernel HIP events (for the roofline block).  (1) A SURVEY step, untimed, right after the warm-up: every convolution / tail
    # launch bracketed, serial stream schedule, eager -- which launch dominates, and the duration of every launch on its own.  Such a
    # step takes ~45 ms instead of ~24 (an event is a barrier packet between two kernels); inside the timed region it cost +1 ms per
    # step at the driver's 20 steps, which is why it left it.  (2) In the TIMED region every launch of the dominant kernel -- and only
    # those -- is bracketed, in every step, with the step's normal schedule: the duration the roofline block reports is measured over
    # the timed region, next to whatever the side stream runs at that moment (the survey's serial duration is reported beside it).
    survey = None
    timer = None
    top_label = None
    if not args.no_kernel_timing and rank != 0:
        eager_step()                      # (every rank runs the survey step -- it contains the gradient all-reduces -- rank 0 brackets it)
    if rank == 0 and not args.no_kernel_timing:
        survey = KernelTimer()
        survey.calibrate()
        model.inner.engine().timer = survey
        eager_step()
        model.inner.engine().timer = None
        torch.cuda.synchronize()
        ssum = survey.summary()
        if os.environ.get('MPOSE_SURVEY_DUMP'):       # every launch label of the survey step (tools/: where a step goes, label by label)
            with open(os.environ['MPOSE_SURVEY_DUMP'], 'w') as f:
                json.dump(ssum, f, indent=1)
        sconv ={k: v for k, v in ssum.items() if k.startswith('conv:') or k.startswith('wgrad:')}
        if sconv:
            top_label = max(sconv.items(), key=lambda kv: kv[1]['total_ms'])[0]
            if graphed is None and planned is None:
                timer = KernelTimer(only=[top_label])
                model.inner.engine().timer = timer
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    model.inner.engine().timer = None
    loss_value = float(loss.detach())
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # Data parallel: the collective on its own -- each gradient bucket's all-reduce (the slices Engine._finish_bucket sends while
    # the backward pass runs), timed back to back after the step loop: ms and bus bandwidth 2(N-1)/N * bytes / time, to be read
    # against the 7 x ~153 GB/s of xGMI links per GPU.  Never measured before the driver's first multi-GPU run.
    allreduce = None
    if world > 1:
        eng = model.inner.engine()
        allreduce = []
        for bi, (lo, hi) in enumerate(eng._buckets):
            sl = eng.gflat[lo:hi]
            for _ in range(2):
                torch.distributed.all_reduce(sl)
            barrier()
            t1 = time.perf_counter()
            n_rep = 5
            for _ in range(n_rep):
                torch.distributed.all_reduce(sl)
            torch.cuda.synchronize()
            sec = (time.perf_counter() - t1) / n_rep
            nbytes = (hi - lo) * 4
            allreduce.append({'bucket': bi, 'what': 'stem' if bi == len(eng._buckets) - 1 else 'stage %d' % (args.stages - 1 - bi),
                              'bytes': nbytes, 'ms': 1e3 * sec, 'bus_GBps': 2.0 * (world - 1) / world * nbytes / sec / 1e9})
    if rank != 0:
        if world > 1:
            torch.distributed.barrier()
        return
    images = world * B * args.steps
    res = {
        'metric': 'images/sec fwd+bwd at 256x256, 17 joints (training step: forward + JS/Euclidean loss + backward + SGD)',
        'value': images / dt, 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32 (3xfp16 split operands: 22-bit operand pieces, fp32 accumulate; same-piece gradient gate 1e-4 vs fp64 at this size with p99 <= 4x the fp32 oracle\'s)', 'f16': 'f16 (operands rounded to fp16, fp32 accumulate; BatchNorm / loss fp32)'}[args.conv_dtype], 'data': 'synthetic',
        'config': {'workload': '%s: training step, per-GPU batch %d, %d-stage MargiPose, %dx%d input, 17 joints, '
                               '%dx%d heatmaps, JS + Euclidean loss, SGD(momentum 0.9)' % (
                                   'BASELINE configs[4] (reduced-precision convolutions)' if args.conv_dtype != 'f32' and args.stages == 5 and args.size == 384
                                   else 'BASELINE configs[2]', B, args.stages, args.size, args.size, args.size // 8, args.size // 8),
                   'global_batch': world * B, 'n_stages': args.stages,
                   'stem': ('inceptionv4 (reference default; restated from SURVEY Appendix B, third-party original unavailable: '
                            'unpinned, random init)' if args.stem == 'inceptionv4' else
                            '%s (reference option, models/margipose_model.py:119-137; torchvision layers restated, unpinned, '
                            'random init)' % args.stem if args.stem.startswith('resnet') else
                            'patch8 (in-repo deterministic stem; the InceptionV4 stem is available with --stem inceptionv4)'), 'parallelism': 'dp%d' % world, 'overlap_wgrad': (not args.no_overlap_wgrad) and (world == 1 or model.inner.engine().dp_overlap()),
                   'step_dispatch': 'hip graph replay (train_helpers.GraphedTrainStep)' if use_graph else (
                       'launch plan replay (train_helpers.PlannedTrainStep, csrc/plan.hip): %d recorded launches and %d cross-stream waits '
                       're-issued per step from one C loop, same two-stream schedule as the eager step' % (planned.n_launches, planned.n_waits)
                       if planned is not None else 'eager launches (Python / ctypes)'),
                   'conv_engine': {2: 'conv_igemm_k / conv_wgrad_rows_k (conv.hip, wgrad.hip), three fp16 products per fp32 multiply-add of per-tensor-scaled, '
                                      'two-way split operands (MPOSE_CONV_F16X3: fp32-equivalent, tests/test_conv_f16x3_gpu.py)',
                                   3: 'three fp16 products per fp32 multiply-add (MPOSE_CONV_F16X3): the regular 128-channel blocks on producer-split '
                                      'fp16 planes end to end -- conv_h2r_k (conv_h.hip: DMA-fed shared LDS tiles, two workgroups per CU) for their '
                                      'forward and BOTH data gradients, conv_wgrad_rows_k reading the same planes for both weight gradients, the '
                                      'BatchNorm-backward applications writing planes only; conv_igemm_k / conv_wgrad_rows_k on fp32 tensors for '
                                      'everything else'}[
                                       model.inner.engine().conv_mode_for(True, True)],
                   'final_loss': loss_value},
    }
    products = 3.0
    if args.conv_dtype == 'f16':
        products = 1.0                 # one MFMA product per multiply-add: priced against the full dense 16-bit MFMA peak
    peak_equiv = PEAK_BF16_MFMA_TFLOPS / products
    if survey is not None and top_label is not None:
        in_region = timer.summary().get(top_label) if timer is not None else None
        sv = sconv[top_label]
        tf = sv['work_per_launch'] / (sv['avg_us'] * 1e-6) / 1e12
        all_flops = sum(v['work'] for v in sconv.values())
        all_ms = sum(v['total_ms'] for v in sconv.values())
        tr_bytes, tr_detail = traffic_fields(top_label)
        meas = sv
        res['roofline'] = {'bound': 'mfma', 'achieved': tf, 'peak': peak_equiv, 'unit': 'TFLOP/s',
                           'frac': tf / peak_equiv, 'traffic': tr_bytes,
                           'traffic_source': 'profiles/*_pmc_traffic.json: a SEPARATE rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass of this '
                                             'command (tools/profile.sh), not measured in this run' if tr_bytes is not None else None,
                           'traffic_detail': tr_detail, 'kernel': top_label,
                           'mfma_busy_frac_pmc': (tr_detail or {}).get('mfma_busy_frac'),
                           'avg_launch_us': sv['avg_us'], 'launches': sv['n'],
                           'measured': 'HIP events around every launch of ONE training step run in this process between the warm-up and the '
                                       'timed steps, serial stream schedule: the kernel alone on the GPU (what rocprofv3 of tools/profile.sh '
                                       'sees).  The step is outside the timed region because bracketing ~440 launches makes it take ~45 ms: '
                                       'inside, it added 1 ms per step to a 20-step run',
                           'in_timed_region': None if not in_region else {
                               'avg_launch_us': in_region['avg_us'], 'launches': in_region['n'],
                               'achieved': in_region['work_per_launch'] / (in_region['avg_us'] * 1e-6) / 1e12,
                               'frac': in_region['work_per_launch'] / (in_region['avg_us'] * 1e-6) / 1e12 / peak_equiv,
                               'note': 'the same kernel bracketed at every launch of all %d timed steps, normal two-stream schedule: the '
                                       'weight-gradient launches of the side stream share the CUs with it for most of its launches, so this '
                                       'duration measures the overlap, not the kernel' % args.steps},
                           'event_bracket_overhead_us_not_subtracted': 1e3 * survey.bracket_cal_ms,
                           'flops_per_launch': meas['work_per_launch'],
                           'note': 'achieved = algorithmic fp32 FLOPs / launch duration; the kernel executes %d 16-bit MFMA FLOPs '
                                   'per algorithmic FLOP (split operands), so peak = dense 16-bit MFMA peak 2500 / %d (fp16 and bf16 '
                                   'MFMA run at the same rate: tools/probe/f16_probe); the plain fp32 MFMA peak is %.1f'
                                   % (products, products, PEAK_FP32_MFMA_TFLOPS),
                           'mfma_tflops_executed': products * tf,
                           'all_conv_kernels_tflops': all_flops / (all_ms * 1e-3) / 1e12,
                           'all_conv_kernels_frac_note': 'every convolution of the survey step (columns and feature extractor, forward, data- and '
                                                         'weight-gradient: all in the three-product form) over the same peak',
                           'all_conv_kernels_frac': all_flops / (all_ms * 1e-3) / 1e12 / peak_equiv,
                           'conv_share_of_step_gpu_time': all_ms / (1e3 * dt / args.steps)}
        res['kernel_time_breakdown_ms_per_step'] = {k: round(v['total_ms'], 3) for k, v in
                                                    sorted(ssum.items(), key=lambda kv: -kv[1]['total_ms'])[:12]}
    if rank == 0 and not args.no_kernel_timing:
        # The soft-argmax path the metric also names.  No subtraction of a calibration term: N back-to-back launches / N.
        # At the configuration sizes the whole working set (13-27 MB) lives in the 256 MB Infinity Cache and a launch is a few
        # microseconds: latency-bound, the GB/s there say little; the cache-defeating size is the HBM-roofline point.
        res['tail_roofline'] = dict(tail_microbench(device, 2048), note='B=2048 fp32: 856 MB per launch, exceeds the 256 MB Infinity Cache '
                                                                        '(the HBM-roofline point of the metric)')
        res['tail_config_sizes'] = {
            'configs[2] training, B=%d fp32' % B: dict(tail_microbench(device, B), note='latency-bound: working set in Infinity Cache'),
            'configs[1] inference, B=64 bf16 heatmaps': dict(tail_microbench(device, 64, bf16_out=True), note='latency-bound: working set in Infinity Cache'),
            'configs[2] training, B=%d fp32, as launched by the step (fused with the residual sum)' % B: dict(
                fused_tail_microbench(device, B), note='replaces bn_add_nchw_k + softmax_dsnt_fwd_k; latency-bound'),
            'configs[1] inference, B=64 bf16 heatmaps, as launched by the step (fused with the residual sum)': dict(
                fused_tail_microbench(device, 64, bf16_out=True), note='latency-bound'),
            'B=2048 fp32, fused with the residual sum (beyond the Infinity Cache: the fused form\'s HBM-roofline point)': dict(
                fused_tail_microbench(device, 2048, launches=30),
                note='the all-joints form of bn_add_softmax_k (one workgroup per image and column, every 128-byte channel line of the two '
                     'padded-to-32-channel inputs read once): 17 of the 32 stored channels are algorithmic bytes, so the line fetches '
                     'bound it at 204 / 324 of the rate a dense stream reaches')}
        res['tail_training'] = {
            'configs[2] training, B=%d fp32 (latency-bound: working set in Infinity Cache)' % B: train_tail_microbench(device, B),
            'B=2048 fp32 (beyond the Infinity Cache: the HBM-roofline point)': train_tail_microbench(device, 2048, launches=50)}
    if allreduce is not None:
        res['allreduce_buckets'] = {'buckets': allreduce, 'xgmi_peak_GBps_per_gpu': 7 * 153.0, 'total_bytes': sum(b['bytes'] for b in allreduce),
                                    'total_ms_if_serial': sum(b['ms'] for b in allreduce)}
    if world == 1 and not args.no_inference:
        res['inference'] = inference_microbench(model, device, args.size)
    if world == 1 and not args.no_inference and not args.no_configs4 and args.conv_dtype == 'f32' and args.stages == 3:
        del graphed
        res['configs4_f16'] = configs4_leg(device)
    if world == 1 and not args.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(args.stages, args.size, args.cpu_batch, args.stem)
    print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()


if __name__ == '__main__':
    main()
