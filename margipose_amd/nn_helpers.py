"""Initial values of the parameters this package owns.

The MargiPose hot path holds exactly two kinds of learnable layers -- bias-free (transposed)
convolutions and BatchNorm2d -- and the reference seeds them with the rule of its
`init_parameters` (src/margipose/nn_helpers.py:7-21): He/Kaiming normal with the fan counted on
the *leading* weight dimension (`mode='fan_out'`, so Cin*k*k for a ConvTranspose2d whose weight
is (Cin, Cout, k, k)), zero conv bias where one exists, and an identity BatchNorm (gamma 1,
beta 0).  Only the columns and combiners are seeded this way (margipose_model.py:62,146); the
feature extractor keeps its own initialisation.
"""
import math

import torch
from torch import nn


def _he_normal_leading_(weight):
    """N(0, 2 / (size(0) * receptive field)) in place -- what kaiming_normal_(w, 0, 'fan_out') draws."""
    receptive = weight[0][0].numel() if weight.dim() > 2 else 1
    std = math.sqrt(2.0 / (weight.size(0) * receptive))
    with torch.no_grad():
        return weight.normal_(0.0, std)


def init_parameters(net):
    convs = [m for m in net.modules() if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
    norms = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    for conv in convs:
        _he_normal_leading_(conv.weight)
        if conv.bias is not None:
            nn.init.zeros_(conv.bias)
    for bn in norms:
        nn.init.ones_(bn.weight)
        nn.init.zeros_(bn.bias)
    return net
