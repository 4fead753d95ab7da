"""Host-side pieces of the reference's training step that sit directly around the hot path
(reference src/margipose/bin/train_3d.py:126-196,338-340,374-382 and hyperparam_scheduler.py:6-42), so that the
per-step sequence  scheduler.batch_step -> model(x) -> forward_loss -> zero_grad -> backward -> optimiser.step
can be driven exactly like the reference drives it.  No dataset / telemetry code (out of scope)."""
import bisect

import torch

from . import _lib, dsntnn


class PiecewiseLinearSchedule:
    """Sets optimiser hyper-parameters from piecewise-linear milestones, once per batch."""

    def __init__(self, optimizer, ts, hyperparam_milestones):
        for name, values in hyperparam_milestones.items():
            assert len(values) == len(ts), 'expected {} milestones for hyperparameter "{}"'.format(len(ts), name)
            for group in optimizer.param_groups:
                assert name in group, '"{}" is not an optimizer hyperparameter'.format(name)
        self.optimizer = optimizer
        self.ts = [float(t) for t in ts]
        self.milestones = {k: [float(v) for v in vs] for k, vs in hyperparam_milestones.items()}
        self.batch_count = 0

    def value_at(self, name, t):
        ts, vs = self.ts, self.milestones[name]
        if t <= ts[0]:
            return vs[0]
        if t >= ts[-1]:
            return vs[-1]
        i = bisect.bisect_right(ts, t) - 1
        w = (t - ts[i]) / (ts[i + 1] - ts[i])
        return vs[i] + w * (vs[i + 1] - vs[i])

    def batch_step(self):
        self.batch_count += 1
        for name in self.milestones:
            value = self.value_at(name, float(self.batch_count))
            for group in self.optimizer.param_groups:
                group[name] = value


def make_1cycle(optimizer, max_iters, lr_max, momentum=0):
    """1cycle policy with the reference's constants: lr_max/10 -> lr_max -> lr_max/10 -> lr_max/1e4 at
    iterations 1, 0.45*max, 0.9*max, max; momentum mirrors it between `momentum` and min(momentum, 0.85)."""
    lr_min = lr_max * 1e-1
    lr_nihil = lr_min * 1e-3
    t3 = max_iters
    t2 = 0.9 * t3
    t1 = t2 / 2
    m_min = min(momentum, 0.85)
    return PiecewiseLinearSchedule(optimizer, ts=[1, t1, t2, t3],
                                   hyperparam_milestones={'lr': [lr_min, lr_max, lr_min, lr_nihil],
                                                          'momentum': [momentum, m_min, momentum, momentum]})


def forward_loss(model, out_var, target_var, mask_var, valid_depth):
    """3D loss, 2D loss, or the per-sample selection by `valid_depth` (train_3d.py:126-142), then the masked mean."""
    target_var = target_var.narrow(-1, 0, 3)
    flags = [int(v) for v in valid_depth]
    if 0 not in flags:
        losses = model.forward_3d_losses(out_var, target_var)
    elif 1 not in flags:
        losses = model.forward_2d_losses(out_var, target_var)
    else:
        sel = torch.tensor(flags, dtype=torch.float32, device=out_var.device)[:, None]
        losses = sel * model.forward_3d_losses(out_var, target_var) + (1.0 - sel) * model.forward_2d_losses(out_var, target_var)
    return dsntnn.average_loss(losses, mask_var)


class StepTimes:
    """The reference's wall-clock meters around one iteration (train_3d.py:44-49: `forward_time`, `backward_time`, `optim_time`,
    each a mean over the epoch's batches; tele's MeanValueMeter): .add(name, seconds), .mean(name), .reset().
    The reference times host-side `perf_counter` intervals around asynchronous launches -- its `forward_time` ends with
    `loss.sum().item()`, a device synchronisation, the other two do not wait for the GPU -- and so does training_step when it is
    given a StepTimes: the forward bucket ends with the same `.item()`, nothing else synchronises."""
    NAMES = ('forward_time', 'backward_time', 'optim_time')

    def __init__(self):
        self.reset()

    def reset(self):
        self.total = {k: 0.0 for k in self.NAMES}
        self.count = {k: 0 for k in self.NAMES}
        self.train_loss = 0.0

    def add(self, name, seconds):
        self.total[name] += seconds
        self.count[name] += 1

    def mean(self, name):
        return self.total[name] / self.count[name] if self.count[name] else 0.0


def training_step(model, scheduler, in_var, target_var, mask_var, valid_depth, times=None):
    """One iteration of do_training_pass (train_3d.py:154-186) without data loading / metrics.
    times: an optional StepTimes that receives the reference's three timing buckets (and, like the reference's
    `tel['train_loss'].add(loss.sum().item())`, the loss value -- which is what makes `forward_time` include the GPU's work)."""
    import time
    if hasattr(scheduler, 'batch_step'):
        scheduler.batch_step()
    optimiser = scheduler.optimizer
    t0 = time.perf_counter()
    out_var = model(in_var)
    loss = forward_loss(model, out_var, target_var, mask_var, valid_depth)
    if times is not None:
        times.train_loss += loss.sum().item()
        t1 = time.perf_counter()
        times.add('forward_time', t1 - t0)
        t0 = t1
    optimiser.zero_grad()
    loss.backward()
    if times is not None:
        t1 = time.perf_counter()
        times.add('backward_time', t1 - t0)
        t0 = t1
    optimiser.step()
    if times is not None:
        times.add('optim_time', time.perf_counter() - t0)
    return out_var, loss


def save_checkpoint(path, model, model_desc, optimiser=None, epoch=0, train_datasets=()):
    """Checkpoint wire format of train_3d.py:374-382 / export_model.py:44-50 (plain tensors + dicts only)."""
    state = {'state_dict': {k: v.detach().cpu() for k, v in model.state_dict().items()}, 'model_desc': model_desc,
             'train_datasets': list(train_datasets)}
    if optimiser is not None:
        state['optimizer'] = optimiser.state_dict()
        state['epoch'] = epoch
    torch.save(state, path)
    return state


SGD_JOB_DT = None


def _sgd_job_dtype():
    global SGD_JOB_DT
    if SGD_JOB_DT is None:
        import numpy as np
        SGD_JOB_DT = np.dtype([('p', 'u8'), ('g', 'u8'), ('buf', 'u8'), ('n', 'i8')], align=True)
    return SGD_JOB_DT


class DeviceSGD:
    """torch.optim.SGD(params, lr, momentum) -- the reference's optimiser (bin/train_3d.py:339) -- as ONE launch over all
    parameters (csrc/optim.hip) with lr / momentum read from device memory, so that the 1cycle schedule
    (hyperparam_scheduler.py: lr AND momentum move every batch) keeps working when the whole iteration is replayed from a
    HIP graph (GraphedTrainStep).  Same arithmetic as torch (no dampening / weight decay / Nesterov) up to fused-multiply-add
    rounding, checked in tests/test_train_graph_gpu.py.  Duck-types what the reference's loop touches: .param_groups, .zero_grad(), .step(),
    .state_dict()."""

    def __init__(self, params, lr, momentum=0.0):
        import numpy as np
        from . import _lib
        self.params = [p for p in params]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 for p in self.params):
            raise _lib.MposeError('DeviceSGD needs float32 parameters on a ROCm device')
        if not all(p.is_contiguous() and p.data_ptr() % 16 == 0 for p in self.params):       # (sgd_step_k reads and writes float4)
            raise _lib.MposeError('DeviceSGD needs contiguous, 16-byte aligned parameters')
        if _lib.lib().mpose_sizeof(11) != _sgd_job_dtype().itemsize:
            raise _lib.MposeError('ABI struct 11 (mpose_sgd_job): library and binding disagree')
        self.param_groups = [{'params': self.params, 'lr': float(lr), 'momentum': float(momentum)}]
        dev = self.params[0].device
        offs, tot = [], 0
        for p in self.params:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        self._bufs = torch.zeros(tot, dtype=torch.float32, device=dev)          # momentum buffers, one arena
        self._buf_ptr = [self._bufs.data_ptr() + 4 * o for o in offs]
        self._max_n = max(p.numel() for p in self.params)
        self._hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self._steps = 0
        self._np = np
        self._eager_table = torch.zeros(len(self.params) * _sgd_job_dtype().itemsize, dtype=torch.uint8, device=dev)
        self._uploaded = {}          # id(table) -> bytes it holds
        self._table_key = {}         # id(table) -> gradient and parameter addresses it was built from (alignment / contiguity checked then)
        self._graph_table = torch.zeros_like(self._eager_table)                  # filled by finish_capture() (allocated HERE: an
        #   allocation inside the capture would come from the graph's pool and its zero-fill would be replayed before every step)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _fill_table(self, table):
        # (the gradients are usually where they were one or two steps ago: compare their addresses before rebuilding the table)
        # -- and the parameters' own addresses with them: the table bakes those in too, and storage that moved (p.data = ...,
        # model.to(), an engine arena rebuild) while the allocator handed the gradients back at the same addresses must not
        # leave sgd_step_k writing through stale pointers
        try:
            gptrs = tuple([p.grad.data_ptr() for p in self.params] + [p.data_ptr() for p in self.params])
        except AttributeError:
            raise RuntimeError('DeviceSGD.step(): a parameter has no gradient')
        if self._table_key.get(id(table)) == gptrs:
            return
        jobs = self._np.zeros(len(self.params), dtype=_sgd_job_dtype())
        for i, p in enumerate(self.params):
            if p.grad is None:
                raise RuntimeError('DeviceSGD.step(): a parameter has no gradient')
            g = p.grad
            if not g.is_contiguous() or g.data_ptr() % 16:
                raise RuntimeError('DeviceSGD needs contiguous, 16-byte aligned gradients')
            jobs[i] = (p.data_ptr(), g.data_ptr(), self._buf_ptr[i], p.numel())
        # The engine hands out views of a per-step clone of its flat gradient buffer; the caching allocator usually returns the same
        # one or two blocks, so the table mostly repeats: upload it only when an address moved, and then from pinned memory,
        # stream-ordered.  (A copy from pageable memory every step was a
        # host-device synchronisation per iteration: the eager loop lost 1.1 ms of a 34 ms step to the bubble behind it.)
        raw = jobs.tobytes()
        if self._uploaded.get(id(table)) != raw:
            staged = torch.from_numpy(jobs.view(self._np.uint8)).pin_memory()
            table.copy_(staged, non_blocking=True)
            self._uploaded[id(table)] = raw
        self._table_key[id(table)] = gptrs

    def upload_hyper(self):
        """Hand {lr, momentum, first-step flag} of the next step() to the device (values travel as kernel arguments)."""
        from . import _lib
        g = self.param_groups[0]
        f = _lib.c_float
        _lib.check(_lib.lib().mpose_set4(_lib.ptr(self._hyper), f(g['lr']), f(g['momentum']), f(1.0 if self._steps == 0 else 0.0), f(0.0),
                                         _lib.stream_ptr()), 'mpose_set4')

    def step(self):
        from . import _lib
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # nothing executes during capture: the job table is filled by finish_capture() with the gradients' addresses inside
            # the graph's memory pool, and the hyper-parameters are uploaded by the caller before every replay
            table = self._graph_table
        elif _lib.plan_recording():
            # a launch plan is being recorded (PlannedTrainStep): this iteration executes, so the table is bound to its gradient
            # tensors now; the hyper-parameters were uploaded before the recording began (they must not be part of it: a replay
            # would write this iteration's values over the ones before_replay() just uploaded)
            self._fill_table(self._graph_table)
            table = self._graph_table
            self._steps += 1
            torch.autograd.graph.increment_version(self.params)
        else:
            self.upload_hyper()
            self._fill_table(self._eager_table)
            table = self._eager_table
            self._steps += 1
            torch.autograd.graph.increment_version(self.params)       # the kernel writes the parameters behind autograd's back
        _lib.check(_lib.lib().mpose_sgd_step(_lib.ptr(table), len(self.params), _lib.c_int64(self._max_n), _lib.ptr(self._hyper),
                                             _lib.stream_ptr()), 'mpose_sgd_step')

    def finish_capture(self):
        """After torch.cuda.graph(...) captured a step(): bind the captured launch to the captured gradient tensors."""
        self._fill_table(self._graph_table)

    def before_replay(self):
        self.upload_hyper()
        self._steps += 1
        torch.autograd.graph.increment_version(self.params)

    def state_dict(self):
        return {'state': {'momentum_buffers': self._bufs.detach().cpu(), 'steps': self._steps},
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._bufs.copy_(sd['state']['momentum_buffers'])
        self._steps = int(sd['state']['steps'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)


class GraphedTrainStep:
    """One training iteration of the reference (bin/train_3d.py:154-186: model(x) -> forward_loss -> zero_grad -> backward ->
    optimiser.step) captured ONCE as a HIP graph and replayed: every buffer of a step has a fixed address (engine.py: arenas,
    device-resident job tables), so a replay costs one graph launch of host time instead of ~1000 kernel launches.

        step = GraphedTrainStep(model, optimiser, x, target, mask)          # example tensors fix the shapes
        out, loss = step(x, target, mask)                                    # copies the batch in, replays

    `optimiser`: DeviceSGD (hyper-parameters may change every step, e.g. driven by make_1cycle(...).batch_step()) or any torch
    optimiser whose step() is capturable with FIXED hyper-parameters (torch.optim.SGD(..., fused=True)).
    `valid_depth` (3D vs 2D loss per sample, train_3d.py:126-142) is fixed at capture time.

    Side effects of the constructor (torch's capture rule needs eager warm-up runs): it executes `warmup` REAL iterations on the
    example batch and the capture pass itself records one more -- optimiser steps that move the weights, the momentum buffers,
    the BatchNorm running statistics and num_batches_tracked, and consume DeviceSGD's first-step flag.  Construct it before
    training starts (or on a throw-away copy of the batch with lr = 0); a non-DeviceSGD optimiser's hyper-parameters are frozen
    into the graph."""

    def __init__(self, model, optimiser, x, target, mask, valid_depth=None, warmup=2):
        self.model, self.opt = model, optimiser
        self.x, self.target, self.mask = x.clone(), target.clone(), mask.clone()
        self.valid_depth = [1] * x.shape[0] if valid_depth is None else [int(v) for v in valid_depth]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                   # torch's capture rule: warm up on a side stream first
            for _ in range(warmup):
                self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out, self.loss = self._iteration()
        if hasattr(optimiser, 'finish_capture'):
            optimiser.finish_capture()
        torch.cuda.synchronize()

    def _iteration(self):
        out = self.model(self.x)
        loss = forward_loss(self.model, out, self.target, self.mask, self.valid_depth)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return out, loss

    def __call__(self, x=None, target=None, mask=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if mask is not None:
            self.mask.copy_(mask, non_blocking=True)
        if hasattr(self.opt, 'before_replay'):
            self.opt.before_replay()
        self.graph.replay()
        return self.out, self.loss


def _check_stamp(eng, stamp, who, table_keys=None, backward=True):
    """A launch plan replays raw device addresses.  Those inside its private allocator pool cannot move; the engine's arenas, job
    tables and workspaces and the model's parameters can (model.to(), load_state_dict(assign=True), `p.data = ...`, an eager
    step at a larger batch, a switch flipped on the engine): a replay after such a change would write through stale pointers."""
    now = eng.plan_stamp(table_keys, backward)
    if now != stamp:
        names = ('device', 'parameter / buffer addresses', 'engine arenas', 'reduction workspace', 'job tables', 'engine switches')
        what = [n for n, a, b in zip(names, now or (None,) * 6, stamp or (None,) * 6) if a != b]
        raise _lib.MposeError('%s: the recorded launch plan is stale (%s changed since it was recorded); build a new one'
                              % (who, ', '.join(what) or 'the engine was rebuilt'))


class PlannedTrainStep:
    """One training iteration (the same sequence as GraphedTrainStep: model(x) -> forward_loss -> zero_grad -> backward ->
    optimiser.step, reference bin/train_3d.py:154-186) recorded ONCE as a launch plan (csrc/plan.hip) and re-issued from one C
    loop: the eager schedule exactly -- same kernels, same argument values, the weight-gradient launches on the side stream
    with the same dependencies -- for ~3 ms of host time instead of 15-28 ms of Python.  (A HIP graph of the iteration replays
    its two streams serially on this runtime: 25.3 against 23.9 ms, profiles/r5_graph_replay_ab.txt; on the pool's slow hosts
    the eager loop is host-bound: 28.7 ms.)

        step = PlannedTrainStep(model, optimiser, x, target, mask)          # example tensors fix the shapes
        out, loss = step(x, target, mask)                                    # copies the batch in, replays

    What makes a replay valid: every tensor of the recorded iteration lives in a private allocator pool that stays reserved for
    this object (the replayed launches write the same addresses); everything the iteration computes is a launch of this library
    (the engine's fills and copies, the stage-loss sum, average_loss's backward, DeviceSGD) -- kernels of the tensor library
    are not recorded; the backward pass is seeded with a persistent tensor of ones.  `optimiser` must be a DeviceSGD (its
    hyper-parameters live in device memory and may change every step); `valid_depth` must select one loss for the whole batch.
    Data parallel: the all-reduces of the gradient buckets are host actions -- the plan breaks at each, a replay issues them again
    between two segments (the same places the eager backward pass issues them).
    Side effects of the constructor: like GraphedTrainStep it runs `warmup` + 1 real iterations on the example batch."""

    def __init__(self, model, optimiser, x, target, mask, valid_depth=None, warmup=2):
        import ctypes
        if not isinstance(optimiser, DeviceSGD):
            raise _lib.MposeError('PlannedTrainStep needs a DeviceSGD optimiser (torch optimisers launch kernels a plan cannot record)')
        self.model, self.opt = model, optimiser
        self.x, self.target, self.mask = x.clone(), target.clone(), mask.clone()
        self.valid_depth = [1] * x.shape[0] if valid_depth is None else [int(v) for v in valid_depth]
        if 0 in self.valid_depth and 1 in self.valid_depth:
            raise _lib.MposeError('PlannedTrainStep: a per-sample mix of 2D and 3D losses is composed with tensor-library kernels')
        eng = model.inner.engine()
        self._one = torch.ones((), dtype=torch.float32, device=x.device)
        self._plan = None
        # the recorded iteration's .grad tensors are views of the engine's flat gradient buffer (no per-step copy of it: the
        # optimiser launch reads them before the next backward pass overwrites them); eager iterations afterwards copy again
        views_before, eng.grad_views = eng.grad_views, True
        self._restore_views = lambda: setattr(eng, 'grad_views', views_before)
        try:
            for _ in range(max(1, warmup)):      # (creates the engine's arenas, job tables and side stream; raises kernels' LDS limits)
                self._iteration()
        except BaseException:
            self._restore_views()
            raise
        torch.cuda.synchronize()
        L = _lib.lib()
        self._sides = [s for s in ((eng.side_stream if eng.overlap_wgrad else None),) if s is not None]
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        self._pool = torch.cuda.MemPool()
        optimiser.upload_hyper()                  # (outside the recording, see DeviceSGD.step)
        arr = self._stream_array()
        # every thread's allocations (autograd runs the backward pass on its own) go to the private pool while recording
        torch._C._cuda_beginAllocateToPool(dev, self._pool.id)
        try:
            _lib.check(L.mpose_plan_begin(arr, len(arr)), 'mpose_plan_begin')
            # data parallel: the all-reduces of the gradient buckets and the wait for them are host actions (Engine._finish_bucket):
            # the plan breaks there and a replay calls them again, in order
            self._host_ops = _lib.PLAN_HOST_OPS = []
            try:
                self.out, self.loss = self._iteration()
            except BaseException:
                L.mpose_plan_abort()
                raise
            finally:
                _lib.PLAN_HOST_OPS = None
            plan = ctypes.c_void_p()
            _lib.check(L.mpose_plan_end(ctypes.byref(plan)), 'mpose_plan_end (a launch went to a stream outside the plan?)')
            self._plan = plan
        finally:
            self._restore_views()
            torch._C._cuda_endAllocateToPool(dev, self._pool.id)
            torch._C._cuda_releasePool(dev, self._pool.id)
        n = [ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)]
        L.mpose_plan_size(self._plan, ctypes.byref(n[0]), ctypes.byref(n[1]), ctypes.byref(n[2]))
        self.n_launches, self.n_waits = n[0].value, n[1].value
        if n[2].value != len(self._host_ops):
            raise _lib.MposeError('PlannedTrainStep: %d breaks recorded for %d host actions' % (n[2].value, len(self._host_ops)))
        torch.cuda.synchronize()
        self._eng = eng
        self._table_keys = tuple(sorted(eng._tables))
        self._cmode = eng._packed_for             # (the packing the recorded iteration leaves in the weight arena)
        self._stamp = eng.plan_stamp(self._table_keys)           # (what the recorded launches point at outside the private pool: checked per replay)

    def _stream_array(self):
        import ctypes
        streams = [torch.cuda.current_stream()] + self._sides
        return (ctypes.c_void_p * len(streams))(*[s.cuda_stream for s in streams])

    def _iteration(self):
        out = self.model(self.x)
        loss = forward_loss(self.model, out, self.target, self.mask, self.valid_depth)
        self.opt.zero_grad(set_to_none=True)
        loss.backward(self._one)
        self.opt.step()
        return out, loss

    def __call__(self, x=None, target=None, mask=None):
        import ctypes
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if target is not None:
            self.target.copy_(target, non_blocking=True)
        if mask is not None:
            self.mask.copy_(mask, non_blocking=True)
        _check_stamp(self._eng, self._stamp, 'PlannedTrainStep', self._table_keys)
        self._eng.before_replay()
        self.opt.before_replay()
        arr = self._stream_array()
        nxt = ctypes.c_int(0)
        replay = _lib.lib().mpose_plan_replay
        _lib.check(replay(self._plan, arr, len(arr), 0, ctypes.byref(nxt)), 'mpose_plan_replay')
        for host_op in self._host_ops:              # (data parallel: issue / wait for the bucket's all-reduce, then go on)
            host_op()
            _lib.check(replay(self._plan, arr, len(arr), nxt.value, ctypes.byref(nxt)), 'mpose_plan_replay')
        self._eng._packed_for = self._cmode       # (the replayed pack launches rewrote the arena: an eager pass in another mode repacks)
        self._eng._pack_epoch += 1
        return self.out, self.loss

    def __del__(self):
        try:
            if self._plan is not None:
                torch.cuda.synchronize()
                _lib.lib().mpose_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass


class PlannedInference:
    """`model(x)` in eval mode under torch.no_grad() (reference bin/infer_single.py:66, bin/eval_3d.py:61) recorded once as a launch
    plan and re-issued from one C loop, like PlannedTrainStep: the forward's ~250 launches for a fraction of a millisecond of host
    time.  Returns the recorded output tensor (coordinates); `model.xy_heatmaps / zy_heatmaps / xz_heatmaps` are the recorded
    heatmap tensors and follow the replays.  The model's weights and running statistics may change between calls (they are read at
    replay time), its shapes, modes (`heatmap_dtype`, `conv_dtype`) and device may not.
    frozen_weights=True leaves the weight measuring / packing launches (0.27 ms of a 12 ms forward at batch 64) out of the recording:
    the packed arena is reused as long as nobody repacked it and no parameter's version counter moved (checked per call; a change
    packs once, eagerly, before the replay).  Weights rebound through `p.data = ...` are not seen: call refresh() after that."""

    def __init__(self, model, x, warmup=2, frozen_weights=False):
        import ctypes
        if model.training:
            raise _lib.MposeError('PlannedInference records an eval-mode forward: call model.eval() first')
        self.model = model
        self.x = x.clone()
        self._plan = None
        with torch.no_grad():
            for _ in range(max(1, warmup)):
                model(self.x)
        torch.cuda.synchronize()
        L = _lib.lib()
        eng = model.inner.engine() if hasattr(model, 'inner') else model.engine()
        self._eng, self._frozen = eng, bool(frozen_weights)
        self._cmode = eng._packed_for          # (the warm-up forward's engine mode: what the recording will run)
        self._stamp = None
        eng.pack_frozen = self._frozen
        self._sides = []           # (a forward runs on one stream)
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        self._pool = torch.cuda.MemPool()
        arr = self._stream_array()
        torch._C._cuda_beginAllocateToPool(dev, self._pool.id)
        try:
            _lib.check(L.mpose_plan_begin(arr, len(arr)), 'mpose_plan_begin')
            try:
                with torch.no_grad():
                    self.out = model(self.x)
                self._heatmaps = tuple(getattr(model, k, None) for k in ('xy_heatmaps', 'zy_heatmaps', 'xz_heatmaps'))
            except BaseException:
                L.mpose_plan_abort()
                raise
            plan = ctypes.c_void_p()
            _lib.check(L.mpose_plan_end(ctypes.byref(plan)), 'mpose_plan_end (a launch went to a stream outside the plan?)')
            self._plan = plan
        finally:
            eng.pack_frozen = False
            torch._C._cuda_endAllocateToPool(dev, self._pool.id)
            torch._C._cuda_releasePool(dev, self._pool.id)
        n = [ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)]
        L.mpose_plan_size(self._plan, ctypes.byref(n[0]), ctypes.byref(n[1]), ctypes.byref(n[2]))
        self.n_launches, self.n_waits = n[0].value, n[1].value
        self._stamp = self._weights_stamp()
        self._table_keys = tuple(sorted(eng._tables))
        self._addr_stamp = eng.plan_stamp(self._table_keys, backward=False)
        torch.cuda.synchronize()

    def _stream_array(self):
        import ctypes
        streams = [torch.cuda.current_stream()] + self._sides
        return (ctypes.c_void_p * len(streams))(*[s.cuda_stream for s in streams])

    def _weights_stamp(self):
        eng = self._eng
        return (eng._pack_epoch, eng._packed_for, sum(p._version for p in eng.param_list()))

    def refresh(self):
        """frozen_weights=True: pack the model's current weights again (after `p.data = ...`, which no counter records)."""
        if self._frozen:
            self._eng.pack_weights(self._cmode)
            self._stamp = self._weights_stamp()

    def __call__(self, x=None):
        import ctypes
        if self._frozen and self._weights_stamp() != self._stamp:      # somebody repacked (another engine mode) or a weight changed
            self.refresh()
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        _check_stamp(self._eng, self._addr_stamp, 'PlannedInference', self._table_keys, backward=False)
        self._eng.before_replay()
        arr = self._stream_array()
        nxt = ctypes.c_int(0)
        _lib.check(_lib.lib().mpose_plan_replay(self._plan, arr, len(arr), 0, ctypes.byref(nxt)), 'mpose_plan_replay')
        if not self._frozen:       # the replayed pack launches rewrote the weight arena in this plan's layout: a pending eager
            self._eng._packed_for = self._cmode       # backward pass of another engine mode repacks (Engine.backward checks)
            self._eng._pack_epoch += 1
            self._stamp = self._weights_stamp()
        for k, v in zip(('xy_heatmaps', 'zy_heatmaps', 'xz_heatmaps'), self._heatmaps):      # (an eager forward in between re-bound them)
            if v is not None:
                setattr(self.model, k, v)
        return self.out

    def __del__(self):
        try:
            if self._plan is not None:
                torch.cuda.synchronize()
                _lib.lib().mpose_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass


class BatchStager:
    """Host -> device staging of training batches (reference bin/train_3d.py:158-161: `batch['input'].to(device, float32)`,
    `batch['target']...`, `batch['joint_mask']...`, synchronous and from pageable memory), done the way the device wants it:
    pinned double buffers, one asynchronous copy per tensor on a dedicated copy stream, overlapped with the previous
    iteration's kernels (the copy stream waits for the consumer's position at the stage() call depth-1 calls back -- by then the
    last reader of this slot's device buffers had been enqueued -- not for the work enqueued since; stage(k) must be called
    BEFORE iteration k is enqueued and after iteration k-1 was);
    the consumer stream waits on an event, never on the host.  Frames may stay uint8 across PCIe (a
    quarter of the bytes: 6.3 MB instead of 25 MB per 32 frames): MargiPoseModel normalises them on the device
    (`ImageSpecs.convert` fused into the feature extractor's first load).

        stager = BatchStager(device)
        for batch in loader:                         # batch: dict of CPU tensors, as the reference's DataLoader yields
            dev = stager.stage(batch)                # returns immediately; copies run on the copy stream
            out = model(dev['input']) ...
    """

    def __init__(self, device, keys=('input', 'target', 'joint_mask'), depth=2):
        self.device = torch.device(device)
        self.keys, self.depth = tuple(keys), depth
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = [dict() for _ in range(depth)]          # key -> (pinned host tensor, device tensor)
        self._events = [None] * depth
        self._marks = []                                       # the consumer stream's position at the last `depth` stage() calls
        if depth < 1:
            raise ValueError('BatchStager: depth must be >= 1')
        self._i = 0

    def _buffers(self, slot, key, t, dtype):
        cur = slot.get(key)
        if cur is None or cur[0].shape != t.shape or cur[0].dtype != dtype:
            cur = (torch.empty(t.shape, dtype=dtype).pin_memory(), torch.empty(t.shape, dtype=dtype, device=self.device))
            slot[key] = cur
        return cur

    def stage(self, batch):
        i = self._i
        self._i = (i + 1) % self.depth
        slot = self._slots[i]
        if self._events[i] is not None:
            self._events[i].synchronize()                      # the copy that last read this slot's pinned buffers is done
        out = dict(batch)
        consumer = torch.cuda.current_stream(self.device)
        # The device buffers of this slot were last read by the iteration fed by the stage() call `depth` calls ago, which (order
        # contract: stage(k), then enqueue iteration k, then stage(k+1)) was fully enqueued when the NEXT call, depth-1 calls ago,
        # took its mark of the consumer stream: that mark is the one to wait for.  depth 2: the previous call's mark (the
        # iteration just enqueued is NOT waited for: the copy overlaps it); depth 1: the mark taken right now, i.e. the whole
        # consumer stream (safe, no overlap).  (Waiting for the whole consumer stream at every depth would put the copy BEHIND
        # the iteration that was just enqueued.)  A prefetching caller -- stage(k+1) BEFORE enqueuing iteration k -- breaks the
        # contract: use depth >= 3 then, which leaves one more iteration of slack.
        mark = torch.cuda.Event()
        mark.record(consumer)
        self._marks.append(mark)
        if len(self._marks) > self.depth:
            self._marks.pop(0)
        if len(self._marks) == self.depth:            # (before that the slot has never been used)
            self.stream.wait_event(self._marks[0])
        with torch.cuda.stream(self.stream):
            for key in self.keys:
                if key not in batch:
                    continue
                t = batch[key]
                dtype = torch.uint8 if (key == 'input' and t.dtype == torch.uint8) else torch.float32
                host, dev = self._buffers(slot, key, t, dtype)
                host.copy_(t)                                  # (pageable -> pinned, with the dtype conversion, on the host)
                dev.copy_(host, non_blocking=True)
                out[key] = dev
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._events[i] = ev
        consumer.wait_event(ev)
        return out
