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


class Adam:
    """What the reference's callers build with torch.optim.Adam(model.parameters(), args.lr, weight_decay=1e-4)
    (/root/reference/train.py:52) and hand to optimize(): here only the two numbers -- the moments and the step
    count live on the GPU next to the parameters (goai_train_step)."""

    def __init__(self, params=None, lr=1e-3, weight_decay=1e-4):
        self.lr, self.weight_decay = float(lr), float(weight_decay)

    def zero_grad(self):
        pass


class ModelMetrics:
    """Losses and accuracies of one optimize() call (/root/reference/go_ai/models/__init__.py:254-276)."""

    def __init__(self, cl=None, ca=None, al=None, aa=None, gl=None):
        self.crit_loss, self.crit_acc = cl, ca
        self.act_loss, self.act_acc = al, aa
        self.game_loss = gl

    @staticmethod
    def _has(x):
        return x is not None and not np.isnan(x)

    def __str__(self):
        ret = ''
        if self._has(self.crit_loss):
            ret += f'C[{self.crit_acc * 100 :.1f}% {self.crit_loss:.3f}L] '
        if self._has(self.act_loss):
            ret += f'A[{self.act_acc * 100 :.1f}% {self.act_loss:.3f}L] '
        if self._has(self.game_loss):
            ret += f'G[{self.game_loss:.3f}L] '
        return ret

    __repr__ = __str__


def average_model(comm, model):
    """Every rank ends up with the mean of the ranks' parameters; BatchNorm statistics are not touched
    (/root/reference/go_ai/models/__init__.py:279-282).  One all-reduce of the flat parameter buffer (RCCL on the
    GPU view when the group runs on nccl) instead of the reference's one per tensor."""
    world_size = comm.Get_size()
    if world_size == 1:
        return
    flat = model.flat_parameters()
    # data.TorchComm sums in place and returns its argument; mpi4py's lowercase allreduce (what the reference
    # passes) and any comm with its semantics return a NEW object holding the sum and leave `flat` alone
    summed = comm.allreduce(flat)
    if summed is not None and summed is not flat:
        summed = torch.as_tensor(summed)
        if tuple(summed.shape) != tuple(flat.shape):
            raise RuntimeError(f"average_model: comm.allreduce returned shape {tuple(summed.shape)}, expected {tuple(flat.shape)}")
        flat.copy_(summed.to(device=flat.device, dtype=flat.dtype))
    flat.div_(world_size)


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

    def __init__(self, size, device=0, n_blocks=LAYERS, max_batch=512, train_batch=256):
        from ..engine import Context
        self.train_batch = train_batch       # largest batch train_step accepts (its buffers are allocated on first use)
        self._train = None
        self._train_dirty = False
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
        self._push_inference()
        if self._train is not None:          # like nn.Module.load_state_dict under a live optimiser: Adam's state stays
            self._train.train_load(self._sd)
        self._train_dirty = False

    def _push_inference(self):
        self.ctx.load_weights(self._sd)
        for c in self._search.values():
            c.load_weights(self._sd)

    def state_dict(self):
        self._pull_trained()
        return dict(self._sd)

    def eval(self):
        self._pull_trained()
        return self

    def parameters(self):
        """The trainable tensors in the reference's parameters() order (weights and biases, not BatchNorm statistics):
        what train.py hands to the optimiser (train.py:52) and utils.count_parameters sums.  Host copies."""
        self._pull_trained()
        for k, v in self._sd.items():
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                yield v.clone().requires_grad_(True)

    # -- optimisation (models/__init__.py:115-231, ac_net.py:67-86, val_net.py:29-43) -----
    def train(self):
        """Training state on the GPU (goai_train_init): raw parameters, gradients, Adam moments, BatchNorm statistics."""
        if self._train is None:
            from ..engine import Context
            self._train = Context(self.size, self.device_index)
            self._train.train_init(Context.NET_AC if self.kind == "ac" else Context.NET_VAL, self.layers, self.train_batch)
            self._train.train_load(self._sd)
        return self

    def _grow_train_batch(self, n):
        """Training buffers for batches of up to n events.  An existing training context is rebuilt around the current
        parameters and statistics; Adam's moments start again (as after a fresh torch.optim.Adam)."""
        if self._train is not None:
            import warnings
            self._pull_trained()
            self._train.close()
            self._train = None
            warnings.warn(f"go_ai.models: training batch capacity {self.train_batch} -> {n}: Adam's moments were reset",
                          RuntimeWarning)
        self.train_batch = int(n)

    def _tensor_shapes(self):
        return {k: tuple(v.shape) for k, v in self._sd.items() if not k.startswith("game_head.")}

    def _pull_trained(self):
        """Trained parameters and statistics back into the state_dict and the inference / search contexts."""
        if self._train is not None and self._train_dirty:
            for k, v in self._train.train_export(self._tensor_shapes()).items():
                self._sd[k] = torch.from_numpy(v)
            self._push_inference()
            self._train_dirty = False

    def flat_parameters(self):
        """All parameters as one flat GPU tensor (a view of the training buffer): what average_model all-reduces."""
        self.train()
        self._train_dirty = True
        return self._train.train_params()

    def train_step(self, optimizer, states, actions, reward, children, terminal, wins, pi):
        """One optimisation step on a batch of events (ac_net.py:67-86 / val_net.py:29-43): the critic learns -wins on
        next_states = children[actions] under random board symmetries, the actor (actor-critic net) argmax(pi) on
        states.  Returns (critic loss, critic accuracy, actor loss, actor accuracy), None where the net has no actor."""
        from .. import data
        self.train()
        children = np.asarray(children)
        bsz = len(children)
        if bsz < 2:
            raise ValueError(f"train_step: a batch of {bsz} event(s); BatchNorm needs at least 2")
        if bsz > self.train_batch:
            self._grow_train_batch(bsz)
            self.train()
        next_states = children[np.arange(bsz), np.asarray(actions)]
        next_states = data.batch_random_symmetries(next_states)          # train_step's own
        next_states = data.batch_random_symmetries(next_states)          # critic_step's (models/__init__.py:119)
        self._train_dirty = True
        return self._train.train_step(states if self.kind == "ac" else None, next_states, np.asarray(wins, dtype=np.float32),
                                      pi if self.kind == "ac" else None, lr=optimizer.lr, weight_decay=optimizer.weight_decay)

    def optimize(self, comm, batched_data, optimizer):
        """RLNet.optimize (models/__init__.py:210-231): train_step over the batches, then parameters and metrics
        averaged over the ranks."""
        import warnings
        from ..data import SoloComm
        comm = comm if comm is not None else SoloComm()
        # data.mpi_sample_eventdata's np.array_split hands out batches of up to 2 * batchsize - 1 events (and a
        # single event when the replay is tiny): size the training buffers for the largest BEFORE any step moves the
        # weights, and refuse what the step cannot take while the parameters are still untouched
        batched_data = list(batched_data)
        sizes = [len(batch[3]) for batch in batched_data]
        if sizes and min(sizes) < 2:
            raise ValueError(f"optimize: a batch of {min(sizes)} event(s); the optimisation step needs at least 2 "
                             "(BatchNorm over a batch); nothing was changed")
        if sizes and max(sizes) > self.train_batch:
            self._grow_train_batch(max(sizes))
        self.train()
        raw_metrics = [self.train_step(optimizer, *batch) for batch in batched_data]
        average_model(comm, self)
        world_size = comm.Get_size()
        raw_metrics = np.array(raw_metrics, dtype=float).reshape(-1, 4)          # None -> nan
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            mean_metrics = np.nanmean(raw_metrics, axis=0) if len(raw_metrics) else np.full(4, np.nan)
        reduced_metrics = np.asarray(comm.allreduce(mean_metrics)) / world_size
        self._pull_trained()
        return ModelMetrics(*reduced_metrics)

    def to(self, device):
        return self

    def search_context(self, n_games, max_sims, mode=0):
        """A context with tree storage for (n_games, max_sims, search mode), sharing this net's weights."""
        from ..engine import Context
        self._pull_trained()
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
        self._pull_trained()
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
