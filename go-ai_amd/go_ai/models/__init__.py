"""go_ai.models -- the reference's model surface (/root/reference/go_ai/models/__init__.py)
backed by the HIP CNN forward of libgoai_hip.so.

`RLNet.create_numpy(mode)` returns the same kind of callable the reference's does
(models/__init__.py:45-49): numpy states [B,6,s,s] in, numpy float32 logits out, with the
invalid-move mask and the "assist" overrides applied (models/__init__.py:90-105).  Weights
come in through `load_state_dict` with the reference's own key layout, so a checkpoint
saved by the reference's train.py (utils.py:146-153) loads unchanged.
"""
import numpy as np
import torch

CHANNELS = 128
LAYERS = 3


def random_state_dict(kind, size, seed=0, layers=LAYERS):
    """Random-init weights in the reference's state_dict layout (there are no pretrained
    weights to load: the reference git-ignores *.pt).  BN statistics are randomised too."""
    g = torch.Generator().manual_seed(seed)
    P = size * size
    sd = {}

    def conv(prefix, cout, cin, k):
        fan_in = cin * k * k
        sd[prefix + ".weight"] = torch.randn((cout, cin, k, k), generator=g) * (2.0 / fan_in) ** 0.5
        sd[prefix + ".bias"] = 0.05 * torch.randn(cout, generator=g)

    def bn(prefix, c):
        sd[prefix + ".weight"] = 0.5 + torch.rand(c, generator=g)
        sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[prefix + ".running_var"] = 0.5 + torch.rand(c, generator=g)

    def linear(prefix, cout, cin):
        sd[prefix + ".weight"] = torch.randn((cout, cin), generator=g) / cin ** 0.5
        sd[prefix + ".bias"] = 0.05 * torch.randn(cout, generator=g)

    conv("main.0", CHANNELS, 6, 3)
    bn("main.1", CHANNELS)
    for i in range(layers):
        b = f"main.{3 + i}.main"
        conv(b + ".0", CHANNELS, CHANNELS, 3)
        bn(b + ".1", CHANNELS)
        conv(b + ".3", CHANNELS, CHANNELS, 3)
        bn(b + ".4", CHANNELS)
    if kind == "ac":
        conv("act_head.0", 2, CHANNELS, 1)
        bn("act_head.1", 2)
        linear("act_head.4", P + 1, 2 * P)
        head = "crit_head"
    else:
        head = "convs"
    conv(head + ".0", 1, CHANNELS, 1)
    bn(head + ".1", 1)
    linear(head + ".4", 256, P)
    linear(head + ".6", 1, 256)
    return sd
