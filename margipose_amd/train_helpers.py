"""Host-side pieces of the reference's training step that sit directly around the hot path
(reference src/margipose/bin/train_3d.py:126-196,338-340,374-382 and hyperparam_scheduler.py:6-42), so that the
per-step sequence  scheduler.batch_step -> model(x) -> forward_loss -> zero_grad -> backward -> optimiser.step
can be driven exactly like the reference drives it.  No dataset / telemetry code (out of scope)."""
import bisect

import torch

from . import dsntnn


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


def training_step(model, scheduler, in_var, target_var, mask_var, valid_depth):
    """One iteration of do_training_pass (train_3d.py:154-186) without data loading / metrics."""
    if hasattr(scheduler, 'batch_step'):
        scheduler.batch_step()
    optimiser = scheduler.optimizer
    out_var = model(in_var)
    loss = forward_loss(model, out_var, target_var, mask_var, valid_depth)
    optimiser.zero_grad()
    loss.backward()
    optimiser.step()
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
