#!/usr/bin/env python
"""Layer-by-layer check of the GPU engine: every saved intermediate is recomputed on the CPU (fp64) from the
GPU's OWN inputs to that layer, so the first wrong kernel is pinpointed."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import weights as W
from oracle import model_ref as R


def nchw(t):
    return t.detach().cpu().double().permute(0, 3, 1, 2).contiguous()


def err(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def main(train):
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    T, seed, B = 1, 401, 2
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, 'patch8', 'jsd')
    sd32 = W.make_state_dict(T, seed)
    m.load_state_dict(sd32, strict=True)
    m = m.cuda()
    m.train(train)
    sd = W.make_state_dict(T, seed, torch.float64)
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    if os.environ.get('CALIB'):
        from collections import OrderedDict
        R.calibrate_running_stats(sd, x.double(), T)
        m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    eng = m.inner.engine()
    with torch.no_grad():
        hms, xyz, ctx = eng.forward(x.cuda(), train, save=True)
    torch.cuda.synchronize()

    def bn(key, t):
        return F.batch_norm(t, sd[key + '.running_mean'].clone(), sd[key + '.running_var'].clone(), sd[key + '.weight'], sd[key + '.bias'],
                            training=train, momentum=0.1, eps=1e-5)
    # stem
    raw = F.conv2d(x.double(), sd['inner.in_cnn.0.weight'], stride=8)
    print('stem_raw', err(nchw(ctx['stem_raw']), raw))
    print('stem_out', err(nchw(ctx['stem_out']), F.relu(bn('inner.in_cnn.1', nchw(ctx['stem_raw'])))))
    kinds = ['regular', 'regular', 'down', 'regular', 'regular', 'regular', 'regular', 'up', 'regular', 'regular']
    planes = ('xy', 'zy', 'xz')
    for i in range(10):
        sv = ctx['blocks'][0][i]
        for c in range(3):
            seq = 'down_layers.%d' % i if i < 5 else 'up_layers.%d' % (i - 5)
            pre = 'inner.%s_hm_cnns.0.%s' % (planes[c], seq)
            xin = nchw(sv['x'][c])
            if i == 9:
                pass
            c1 = R._conv_in(sd, pre + '.module.0', xin, kinds[i], 3)
            scr = R._conv_in(sd, pre + '.shortcut.0', xin, kinds[i], 1)
            g_c1, g_sc, g_c2 = nchw(sv['c1'][c]), nchw(sv['sc'][c]), nchw(sv['c2'][c])
            nc = c1.shape[1]
            e1, e2 = err(g_c1[:, :nc], c1), err(g_sc[:, :nc], scr)
            pad1 = float(g_c1[:, nc:].abs().max()) if g_c1.shape[1] > nc else 0.0
            a1 = F.relu(bn(pre + '.module.1', g_c1[:, :nc]))
            c2 = F.conv2d(a1, sd[pre + '.module.3.weight'], padding=1)
            e3 = err(g_c2[:, :nc], c2)
            out = F.relu(bn(pre + '.module.4', g_c2[:, :nc])) + bn(pre + '.shortcut.1', g_sc[:, :nc])
            if i < 9:
                nxt = ctx['blocks'][0][i + 1]['x'][c]
                if i == 4:
                    out = R.axis_permute(out, planes[c])
                e4 = err(nchw(nxt), out)
            else:
                hm_ref = R.flat_softmax(out)
                e4 = err(hms[c][0].detach().cpu().double(), hm_ref)
                print('   logits-ref range', float(out.min()), float(out.max()), 'hm sum', float(hms[c][0].sum()), 'hm[0,0,0,:4]', hms[c][0][0,0,0,:4].tolist(), hm_ref[0,0,0,:4].tolist())
            print('blk %d col %d (%s): c1 %.2e sc %.2e c2 %.2e out %.2e pad %.1e' % (i, c, kinds[i], e1, e2, e3, e4, pad1))


def model_level():
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    from collections import OrderedDict
    T, seed, B = 1, 401, 2
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64), x.double(), T)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, 'patch8', 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(x.cuda())
        xy, zy, xz = R.inner_forward(sd, x.double(), T, False)
    print('rv sample gpu', m.state_dict()['inner.in_cnn.1.running_var'][:4].tolist(), 'cpu', sd['inner.in_cnn.1.running_var'][:4].tolist())
    for name, a, b in (('xy', m.xy_heatmaps[0], xy[0]), ('zy', m.zy_heatmaps[0], zy[0]), ('xz', m.xz_heatmaps[0], xz[0])):
        a = a.cpu().double()
        print(name, 'gpu max', float(a.max()), 'ref max', float(b.max()), 'err', err(a, b), 'nan', bool(torch.isnan(a).any()))
    eng = m.inner.engine()
    with torch.no_grad():
        hms, xyz, ctx = eng.forward(x.cuda(), False, save=True)
    print('save=True path: xy err', err(hms[0][0].cpu().double(), xy[0]))
    with torch.no_grad():
        hms2, xyz, ctx2 = eng.forward(x.cuda(), False, save=False)
    print('save=False path: xy err', err(hms2[0][0].cpu().double(), xy[0]))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'global':
        pass
    elif len(sys.argv) > 1 and sys.argv[1] == 'model':
        model_level()
        sys.exit(0)
    else:
        main(len(sys.argv) > 1 and sys.argv[1] == 'train')


def global_check():
    from margipose_amd.models import CanonicalSkeletonDesc, MargiPoseModel
    from collections import OrderedDict
    T, seed, B = 1, 401, 2
    x, target, mask = W.seeded_inputs(seed + 1000, B)
    sd = R.calibrate_running_stats(W.make_state_dict(T, seed, torch.float64), x.double(), T)
    m = MargiPoseModel(CanonicalSkeletonDesc, T, True, 'patch8', 'jsd')
    m.load_state_dict(OrderedDict((k, v.float() if v.is_floating_point() else v) for k, v in sd.items()), strict=True)
    m = m.cuda().eval()
    eng = m.inner.engine()
    with torch.no_grad():
        hms, xyz, ctx = eng.forward(x.cuda(), False, save=True)
    rec = []
    orig = R.residual_block
    def spy(sd_, prefix, xx, kind, train):
        rec.append((prefix, xx.clone()))
        return orig(sd_, prefix, xx, kind, train)
    R.residual_block = spy
    with torch.no_grad():
        R.inner_forward(sd, x.double(), T, False)
    R.residual_block = orig
    planes = ('xy', 'zy', 'xz')
    for prefix, xx in rec:
        parts = prefix.split('.')
        c = planes.index(parts[1][:2]); i = int(parts[-1]) + (5 if parts[3] == 'up_layers' else 0)
        g = nchw(ctx['blocks'][0][i]['x'][c])
        print(prefix, 'input err', err(g[:, :xx.shape[1]], xx), 'absmax ref', float(xx.abs().max()), 'gpu', float(g.abs().max()))


if len(sys.argv) > 1 and sys.argv[1] == 'global':
    global_check()
