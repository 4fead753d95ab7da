"""Golden vectors for the reference's _ChatterboxCnn (models/chatterbox_model.py:87-221), generated HERE by importing the
read-only reference (tools/_reference_import.py) -- run in the build container only:

    python tools/make_golden_chatterbox.py

Writes tests/golden/chatterbox_cnn.npz (inputs' seeds, the reference's outputs, input gradients and per-parameter gradient
norms, both orientations, train and eval mode) and tests/golden/chatterbox_keys.json (its state_dict keys and shapes).  The
weights are oracle/weights.py's deterministic stream, so no tensor file is committed.  ChatterboxModel as a whole cannot be
built from the reference here: its constructor calls torchvision's resnet34, which is not in the image.
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import weights as W                      # noqa: E402
from tools._reference_import import import_reference  # noqa: E402

SEED_W, SEED_X = 4101, 4102


def main():
    import_reference()
    import margipose.models.chatterbox_model as cm
    out = {}
    keys = {}
    for sw in (True, False):
        tag = 'w' if sw else 'h'
        torch.manual_seed(0)
        net = cm._ChatterboxCnn(17, shrink_width=sw)
        keys[tag] = [(k, list(v.shape)) for k, v in net.state_dict().items()]
        sd = W.fill_like(OrderedDict(W.chatterbox_cnn_entries('', sw)), SEED_W)
        net.load_state_dict(sd)
        rng = np.random.default_rng(SEED_X)
        x = torch.from_numpy(rng.standard_normal((1, 128, 32, 32))).float().requires_grad_(True)
        probe = torch.from_numpy(rng.standard_normal((1, 17, 32, 32))).float()
        net.train()
        y = net(x)
        (y * probe).sum().backward()
        out['train_out_' + tag] = y.detach().numpy()
        out['train_dx_' + tag] = x.grad.numpy().copy()
        names = [n for n, _ in net.named_parameters()]
        out['train_gnorm_' + tag] = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()])
        out['running_mean_k8_' + tag] = net.down_convs[5].running_mean.numpy().copy()     # updated running statistics of one layer
        out['running_var_k8_' + tag] = net.down_convs[5].running_var.numpy().copy()
        net.load_state_dict(sd)
        net.eval()
        with torch.no_grad():
            out['eval_out_' + tag] = net(x).numpy()
        # the same class on NON-NEGATIVE features (what ResNet's final ReLU hands it): the fixture the HIP heads are compared with
        # directly (tests/test_golden_direct_gpu.py), since the launch graph's feature node is read through a ReLU
        xp = x.detach().abs()
        with torch.no_grad():
            out['eval_out_pos_' + tag] = net(xp).numpy()
            net.train()
            out['train_out_pos_' + tag] = net(xp).numpy()
        keys['params_' + tag] = names
    out['seeds'] = np.array([SEED_W, SEED_X])
    root = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden')
    np.savez_compressed(os.path.join(root, 'chatterbox_cnn.npz'), **out)
    with open(os.path.join(root, 'chatterbox_keys.json'), 'w') as f:
        json.dump(keys, f, indent=0)
    print('wrote', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
