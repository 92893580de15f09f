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


class RLNet:
    """Counterpart of the reference's RLNet (/root/reference/go_ai/models/__init__.py:13-113),
    inference half: weights live on the GPU inside a go_ai.engine.Context."""
    kind = None
    MAX_SEARCH_CONTEXTS = 4      # tree storage is kept for the last few (n_games, max_sims, mode) combinations only

    def __init__(self, size, device=0, n_blocks=LAYERS, max_batch=512):
        from ..engine import Context
        self.size = size
        self.layers = n_blocks
        self.channels = CHANNELS
        self.assist = True
        self.requires_children = False
        self.device_index = device
        self.max_batch = max_batch
        self.ctx = Context(size, device)
        self.ctx.nn_init(Context.NET_AC if self.kind == "ac" else Context.NET_VAL, n_blocks, max_batch)
        self._sd = None
        self._search = {}
        self.load_state_dict(random_state_dict(self.kind, size, seed=0, layers=n_blocks))

    # -- torch.nn.Module look-alikes used by the reference's callers --------------------
    def load_state_dict(self, state_dict, strict=False):
        self._sd = {k: (v.detach().cpu().float().clone() if torch.is_tensor(v) else torch.as_tensor(v).float())
                    for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        self.ctx.load_weights(self._sd)
        for c in self._search.values():
            c.load_weights(self._sd)

    def state_dict(self):
        return dict(self._sd)

    def eval(self):
        return self

    def to(self, device):
        return self

    def search_context(self, n_games, max_sims, mode=0):
        """A context with tree storage for (n_games, max_sims, search mode), sharing this net's weights."""
        from ..engine import Context
        key = (n_games, max_sims, mode)
        if key in self._search:
            self._search[key] = self._search.pop(key)            # most recently used last
        else:
            while len(self._search) >= self.MAX_SEARCH_CONTEXTS:  # drop the least recently used one (frees its trees)
                self._search.pop(next(iter(self._search))).close()
            c = Context(self.size, self.device_index)
            batch = n_games if mode == Context.SEARCH_AC else min(n_games * (self.size ** 2 + 2), 16384)
            c.nn_init(Context.NET_AC if self.kind == "ac" else Context.NET_VAL, self.layers, max(batch, 1))
            c.load_weights(self._sd)
            c.mcts_init(n_games, max_sims, mode)
            self._search[key] = c
        return self._search[key]

    # -- the numpy shim -----------------------------------------------------------------
    def create_numpy(self, mode):
        def np_func(states):
            return self._numpy(states, mode)
        np_func._goai_net = self          # mct_search keeps the whole search on the GPU for callables that carry this
        np_func._goai_mode = mode
        return np_func

    def _numpy(self, states, mode, children=None):
        """RLNet._numpy (models/__init__.py:54-113): numpy in, numpy float32 logits out."""
        states = np.asarray(states)
        assert states.ndim == 4, states.shape
        if mode == "critic":
            _, val = self.ctx.nn_forward(states, mode=self.ctx.NN_NUMPY_SHIM, want_pi=False)
            return val.cpu().numpy()
        if mode not in ("actor", "actor_critic"):
            raise Exception(f"Unknown mode: {mode}")
        if self.kind != "ac":
            raise Exception("Not Implemented")
        pi, val = self.ctx.nn_forward(states, mode=self.ctx.NN_NUMPY_SHIM)
        if mode == "actor":
            return pi.cpu().numpy()
        return pi.cpu().numpy(), val.cpu().numpy()


def get_modelpath(args, savetype):
    # /root/reference/go_ai/models/__init__.py:285-294
    import os
    if savetype == 'checkpoint':
        d = args.checkdir
    elif savetype == 'baseline':
        d = 'bin/baselines/'
    else:
        raise Exception(f"Unknown location type: {savetype}")
    return os.path.join(d, f'{args.model}{args.size}.pt')
