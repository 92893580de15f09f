"""Thin object wrapper around a `goai_ctx` (include/goai_hip.h).

Holds one GPU context per (device, board size) and exposes the C-ABI entry points on torch
device tensors.  numpy inputs are accepted for convenience and copied to the GPU; results
come back as torch device tensors (callers `.cpu().numpy()` them where the reference's
surface promises numpy).
"""
import ctypes

import numpy as np
import torch

from . import _hip

NUM_CHNLS = 6


def _dev_u8_states(states, device):
    if isinstance(states, np.ndarray):
        if states.dtype != np.uint8:
            states = (states != 0).astype(np.uint8)
        states = torch.from_numpy(np.ascontiguousarray(states))
    if states.dtype != torch.uint8:
        states = (states != 0).to(torch.uint8)
    return states.to(device, non_blocking=True).contiguous()


class Context:
    def __init__(self, board_size, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("go_ai.engine needs a ROCm GPU; there is no CPU fallback")
        self.size = int(board_size)
        self.P = self.size * self.size
        self.A = self.P + 1
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        h = ctypes.c_void_p()
        _hip.check(_hip.lib.goai_ctx_create(ctypes.byref(h), self.device_index, self.size))
        self._h = h
        self.net_kind = -1
        self.n_games = self.max_sims = self.search_mode = 0
        self.search_generation = 0           # bumped by every mcts_reset: tree views check it before reading the tree
        self.set_stream(torch.cuda.current_stream(self.device))

    def close(self):
        if getattr(self, "_h", None) is not None:
            _hip.lib.goai_ctx_destroy(self._h)
            self._h = None

    @property
    def closed(self):
        return getattr(self, "_h", None) is None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        self._stream = stream
        _hip.check(_hip.lib.goai_ctx_set_stream(self._h, ctypes.c_void_p(stream.cuda_stream)))

    def _bind(self):
        """Every call enqueues on torch's CURRENT stream of this device: the tensors a call allocates,
        the copies it makes and its kernels are then ordered on one stream, and the caching allocator
        never hands a block back while a kernel of this context still reads it.  When the current stream
        differs from the one the context last used (a `with torch.cuda.stream(...)` block), the old
        stream is drained first, so the context's own device state (trees, activations) stays ordered."""
        if self._h is None:
            raise RuntimeError("this go_ai.engine.Context was closed (evicted from a context cache, or close() was called): "
                               "its device state is gone")
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self._stream.cuda_stream:
            self._stream.synchronize()
            self.set_stream(cur)

    def sync(self):
        _hip.check(_hip.lib.goai_ctx_sync(self._h))

    # ---- Go rules ------------------------------------------------------------------
    def transition(self, states, actions, canonical=True):
        self._bind()
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        acts = torch.as_tensor(np.asarray(actions) if not torch.is_tensor(actions) else actions)
        acts = acts.to(self.device, dtype=torch.int32).contiguous()
        out = torch.empty_like(st)
        status = torch.empty(n, dtype=torch.int32, device=self.device)
        _hip.check(_hip.lib.goai_transition(self._h, _hip.ptr(st), _hip.ptr(acts), int(bool(canonical)),
                                            _hip.ptr(out), _hip.ptr(status), n))
        return out, status

    def children(self, states, canonical=True):
        self._bind()
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        kids = torch.empty((n, self.A, NUM_CHNLS, self.size, self.size), dtype=torch.uint8, device=self.device)
        valid = torch.empty((n, self.A), dtype=torch.uint8, device=self.device)
        win = torch.empty((n, self.A), dtype=torch.int8, device=self.device)
        _hip.check(_hip.lib.goai_children(self._h, _hip.ptr(st), int(bool(canonical)), _hip.ptr(kids),
                                          _hip.ptr(valid), _hip.ptr(win), n))
        return kids, valid, win

    def score(self, states):
        self._bind()
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        outs = [torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(4)]
        _hip.check(_hip.lib.goai_score(self._h, _hip.ptr(st), *[_hip.ptr(o) for o in outs], n))
        return tuple(outs)   # black_area, white_area, ended, winner

    def liberties(self, states):
        self._bind()
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        b, w = [torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(2)]
        _hip.check(_hip.lib.goai_liberties(self._h, _hip.ptr(st), _hip.ptr(b), _hip.ptr(w), n))
        return b, w

    # ---- CNN -----------------------------------------------------------------------
    NET_AC, NET_VAL = 0, 1
    NN_RAW, NN_NUMPY_SHIM = 0, 1

    DTYPE_F32, DTYPE_BF16X3, DTYPE_BF16X6, DTYPE_F16X3, DTYPE_F16 = 0, 1, 2, 3, 4      # DTYPE_F16: reduced precision, opt-in

    def nn_init(self, net_kind, n_blocks=3, max_batch=1024, dtype=0):
        self._bind()
        self.net_kind = int(net_kind)
        _hip.check(_hip.lib.goai_nn_init(self._h, self.net_kind, int(n_blocks), int(max_batch), int(dtype)))

    CONV_DIRECT, CONV_WINOGRAD, CONV_BF16X3, CONV_WINOGRAD_F3, CONV_WINOGRAD_F3_PAIRED, CONV_WINOGRAD_F3_BF16X6, CONV_WINOGRAD_BF16X6 = 0, 1, 2, 3, 4, 5, 6
    CONV_WINOGRAD_F3_F16X3, CONV_WINOGRAD_F16X3 = 7, 8
    CONV_WINOGRAD_F3_F16, CONV_WINOGRAD_F16 = 9, 10

    def nn_conv_algo(self):
        """The 3x3 convolution kernel nn_init chose for this context (CONV_*)."""
        rc = _hip.lib.goai_nn_conv_algo(self._h)
        if rc < 0:
            _hip.check(rc)
        return rc

    def nn_convs_per_launch(self):
        """Trunk convolutions per kernel launch: 2 * blocks when the F(3x3,3x3) trunk runs as one launch, else 1."""
        rc = _hip.lib.goai_nn_convs_per_launch(self._h)
        if rc < 0:
            _hip.check(rc)
        return rc

    def load_weights(self, state_dict):
        """state_dict: mapping of the reference's keys to float tensors / arrays (host)."""
        self._bind()
        names, arrays = [], []
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            if a.dtype.kind != "f":
                continue                      # num_batches_tracked
            names.append(k.encode())
            arrays.append(np.ascontiguousarray(a, dtype=np.float32))
        n = len(names)
        c_names = (ctypes.c_char_p * n)(*names)
        c_ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in arrays])
        c_numels = (ctypes.c_int64 * n)(*[a.size for a in arrays])
        _hip.check(_hip.lib.goai_load_weights(self._h, c_names, c_ptrs, c_numels, n))

    # ---- result gather over ranks (goai_gather_stats) --------------------------------------
    @staticmethod
    def comm_unique_id():
        """128 bytes from rank 0 (ncclGetUniqueId) to be handed to every rank's comm_init."""
        buf = ctypes.create_string_buffer(128)
        _hip.check(_hip.lib.goai_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, n_ranks, rank):
        self._bind()
        comm = ctypes.c_void_p()
        _hip.check(_hip.lib.goai_comm_init(self._h, ctypes.c_char_p(bytes(unique_id)), int(n_ranks), int(rank), ctypes.byref(comm)))
        return comm

    def comm_destroy(self, comm):
        _hip.check(_hip.lib.goai_comm_destroy(self._h, comm))

    GATHER_FIELDS = ("finished", "black_wins", "white_wins", "finished_moves", "leaf_evals", "transitions", "errors", "forwards")

    def gather_stats(self, comm=None, extra=()):
        """Sums over the communicator's ranks (utils.py:201-204) of this context's game statistics and search counters,
        plus the caller's own additive figures `extra`.  Returns (dict of GATHER_FIELDS, list of summed extras)."""
        self._bind()
        n = len(extra)
        ex = (ctypes.c_double * max(n, 1))(*[float(x) for x in extra])
        tot = (ctypes.c_int64 * 8)()
        _hip.check(_hip.lib.goai_gather_stats(self._h, comm, ex if n else None, n, tot))
        return dict(zip(self.GATHER_FIELDS, [int(x) for x in tot])), [float(ex[i]) for i in range(n)]

    # ---- optimisation step (include/goai_hip.h: goai_train_*) ---------------------------
    @staticmethod
    def _tensor_table(items):
        names = [k.encode() for k, _ in items]
        n = len(names)
        return (n, (ctypes.c_char_p * n)(*names), (ctypes.c_void_p * n)(*[a.ctypes.data for _, a in items]),
                (ctypes.c_int64 * n)(*[a.size for _, a in items]))

    def train_init(self, net_kind, n_blocks=3, max_batch=32):
        self._bind()
        self.train_kind = int(net_kind)
        _hip.check(_hip.lib.goai_train_init(self._h, self.train_kind, int(n_blocks), int(max_batch)))

    def train_load(self, state_dict, strict=True):
        """Parameters and BatchNorm statistics from a state_dict in the reference's key layout (host tensors / arrays)."""
        self._bind()
        items = []
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            if a.dtype.kind == "f":             # not num_batches_tracked
                items.append((k, np.ascontiguousarray(a, dtype=np.float32)))
        n, names, ptrs, numels = self._tensor_table(items)
        _hip.check(_hip.lib.goai_train_load(self._h, names, ptrs, numels, n, int(bool(strict))))

    TRAIN_VALUES, TRAIN_GRADS, TRAIN_ADAM_M, TRAIN_ADAM_V = 0, 1, 2, 3

    def train_export(self, shapes, what=0):
        """shapes: mapping key -> shape; returns key -> float32 numpy array of that shape (`what`: TRAIN_*)."""
        self._bind()
        items = [(k, np.empty(tuple(shape), dtype=np.float32)) for k, shape in shapes.items()]
        n, names, ptrs, numels = self._tensor_table(items)
        _hip.check(_hip.lib.goai_train_export(self._h, int(what), names, ptrs, numels, n))
        return dict(items)

    def train_reset_optimizer(self):
        self._bind()
        _hip.check(_hip.lib.goai_train_reset_optimizer(self._h))

    def train_step(self, states, next_states, wins, pi, lr=1e-3, weight_decay=1e-4):
        """goai_train_step; returns (critic_loss, critic_acc, actor_loss, actor_acc), the last two None for the value net."""
        self._bind()
        nxt = _dev_u8_states(next_states, self.device)
        n = nxt.shape[0]
        w = torch.as_tensor(np.asarray(wins, dtype=np.float32) if not torch.is_tensor(wins) else wins)
        w = w.to(self.device, dtype=torch.float32).contiguous()
        st = p = None
        if self.train_kind == self.NET_AC:
            st = _dev_u8_states(states, self.device)
            p = torch.as_tensor(np.asarray(pi, dtype=np.float32) if not torch.is_tensor(pi) else pi)
            p = p.to(self.device, dtype=torch.float32).contiguous()
            assert st.shape[0] == n and tuple(p.shape) == (n, self.A), (st.shape, p.shape)
        assert w.numel() == n
        m = (ctypes.c_float * 4)()
        _hip.check(_hip.lib.goai_train_step(self._h, _hip.ptr(st), _hip.ptr(nxt), _hip.ptr(w), _hip.ptr(p), n,
                                            float(lr), float(weight_decay), m))
        vals = [float(x) for x in m]
        return tuple(None if np.isnan(v) else v for v in vals)

    def train_params(self):
        """The flat fp32 parameter buffer as a torch DEVICE tensor (a view, not a copy): all-reduce it in place to
        average the model over ranks."""
        self._bind()
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        _hip.check(_hip.lib.goai_train_params(self._h, ctypes.byref(p), ctypes.byref(n)))

        class _View:
            __cuda_array_interface__ = {"shape": (n.value,), "typestr": "<f4", "data": (p.value, False), "version": 2}
        return torch.as_tensor(_View(), device=self.device)

    def train_sync_inference(self):
        self._bind()
        _hip.check(_hip.lib.goai_train_sync_inference(self._h))

    def nn_forward(self, states, mode=0, want_pi=True):
        self._bind()
        st = _dev_u8_states(states, self.device)
        n = st.shape[0]
        want_pi = want_pi and self.net_kind == self.NET_AC
        pi = torch.empty((n, self.A), dtype=torch.float32, device=self.device) if want_pi else None
        val = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        _hip.check(_hip.lib.goai_nn_forward(self._h, _hip.ptr(st), _hip.ptr(pi), _hip.ptr(val), n, int(mode)))
        return pi, val

    NN_STATIC_MAX_CELLS = 8192

    def nn_forward_static(self, states, pi, val, mode=0):
        """goai_nn_forward_static: the forward of `states` (a uint8 DEVICE tensor) into the caller's
        `pi` ([n, A] float32 or None) and `val` ([n, 1] float32) without any allocation: the
        activations live in the library's static device array (n * size**2 <= NN_STATIC_MAX_CELLS)."""
        self._bind()
        assert states.is_cuda and states.dtype == torch.uint8 and states.is_contiguous()
        n = states.shape[0]
        _hip.check(_hip.lib.goai_nn_forward_static(self._h, _hip.ptr(states), _hip.ptr(pi), _hip.ptr(val), n, int(mode)))

    # ---- batched search ------------------------------------------------------------
    SEARCH_AC, SEARCH_Q = 0, 1

    def mcts_init(self, n_games, max_sims, mode=0):
        self._bind()
        _hip.check(_hip.lib.goai_mcts_init(self._h, int(n_games), int(max_sims), int(mode)))
        self.n_games, self.max_sims, self.search_mode = int(n_games), int(max_sims), int(mode)
        self.capacity = int(n_games)

    def mcts_set_active(self, n_active):
        """Work on the first n_active games only (goai_mcts_set_active); mcts_reset must follow."""
        _hip.check(_hip.lib.goai_mcts_set_active(self._h, int(n_active)))
        self.n_games = int(n_active)

    def sample_actions_keyed(self, pi, seed, game_ids, steps):
        """One draw per row of pi with the Philox key (seed, game_ids[i], steps[i])."""
        self._bind()
        n = pi.shape[0]
        ids = torch.as_tensor(game_ids).to(self.device, dtype=torch.int64).contiguous()
        st = torch.as_tensor(steps).to(self.device, dtype=torch.int32).contiguous()
        actions = torch.empty(n, dtype=torch.int32, device=self.device)
        _hip.check(_hip.lib.goai_sample_actions_keyed(self._h, _hip.ptr(pi), int(seed), _hip.ptr(ids), _hip.ptr(st), n,
                                                      _hip.ptr(actions)))
        return actions

    def mcts_reset(self, root_states):
        self._bind()
        st = _dev_u8_states(root_states, self.device)
        assert self.n_games == 0 or st.shape == (self.n_games, NUM_CHNLS, self.size, self.size), st.shape
        self.search_generation += 1
        _hip.check(_hip.lib.goai_mcts_reset(self._h, _hip.ptr(st)))

    def mcts_select(self, want_leaves=True):
        self._bind()
        leaves = None
        if want_leaves:
            leaves = torch.empty((self.n_games, NUM_CHNLS, self.size, self.size), dtype=torch.uint8,
                                 device=self.device)
        _hip.check(_hip.lib.goai_mcts_select(self._h, _hip.ptr(leaves)))
        return leaves

    def mcts_backup(self, val_logits, pi_logits=None, prior=None, logits_are_raw=False):
        self._bind()
        def dev32(x):
            if x is None:
                return None
            x = torch.as_tensor(x)
            return x.to(self.device, dtype=torch.float32).contiguous()
        val, pi, pr = dev32(val_logits), dev32(pi_logits), dev32(prior)
        assert val.numel() == self.n_games
        _hip.check(_hip.lib.goai_mcts_backup(self._h, _hip.ptr(pi), _hip.ptr(pr), _hip.ptr(val),
                                             int(bool(logits_are_raw))))

    def mcts_expand(self, want_children=True):
        """Q mode: make_children of every selected leaf -> (children [G,A,6,s,s], valid [G,A])."""
        self._bind()
        kids = valid = None
        if want_children:
            kids = torch.empty((self.n_games, self.A, NUM_CHNLS, self.size, self.size), dtype=torch.uint8,
                               device=self.device)
            valid = torch.empty((self.n_games, self.A), dtype=torch.uint8, device=self.device)
        _hip.check(_hip.lib.goai_mcts_expand(self._h, _hip.ptr(kids), _hip.ptr(valid)))
        return kids, valid

    def mcts_backup_q(self, leaf_val, child_vals, vals_are_raw=False):
        self._bind()
        lv = torch.as_tensor(leaf_val).to(self.device, dtype=torch.float64).contiguous()
        cv = torch.as_tensor(child_vals).to(self.device, dtype=torch.float64).contiguous()
        assert lv.numel() == self.n_games and cv.numel() == self.n_games * self.A
        _hip.check(_hip.lib.goai_mcts_backup_q(self._h, _hip.ptr(lv), _hip.ptr(cv), int(bool(vals_are_raw))))

    def mcts_child_values(self):
        self._bind()
        out = torch.empty((self.n_games, self.A), dtype=torch.float64, device=self.device)
        _hip.check(_hip.lib.goai_mcts_child_values(self._h, _hip.ptr(out)))
        return out

    def mcts_policy_value(self, temp=1.0, from_values=False):
        self._bind()
        pi = torch.empty((self.n_games, self.A), dtype=torch.float64, device=self.device)
        _hip.check(_hip.lib.goai_mcts_policy_value(self._h, float(temp), int(bool(from_values)), _hip.ptr(pi)))
        return pi

    def mcts_run(self, n_sims):
        self._bind()
        _hip.check(_hip.lib.goai_mcts_run(self._h, int(n_sims)))

    def mcts_steps(self, n_steps):
        self._bind()
        _hip.check(_hip.lib.goai_mcts_steps(self._h, int(n_steps)))

    def mcts_steps_in_one_launch(self):
        """Trunk passes per launch of the last mcts_steps (k_search_steps), 0 = separate launches."""
        return int(_hip.lib.goai_mcts_steps_in_one_launch(self._h))

    def mcts_readout(self):
        self._bind()
        visits = torch.empty((self.n_games, self.A), dtype=torch.int32, device=self.device)
        prior = torch.empty((self.n_games, self.A), dtype=torch.float64, device=self.device)
        root_visits = torch.empty(self.n_games, dtype=torch.int32, device=self.device)
        _hip.check(_hip.lib.goai_mcts_readout(self._h, _hip.ptr(visits), _hip.ptr(prior), _hip.ptr(root_visits)))
        return visits, prior, root_visits

    def mcts_policy(self, temp=1.0):
        self._bind()
        pi = torch.empty((self.n_games, self.A), dtype=torch.float64, device=self.device)
        _hip.check(_hip.lib.goai_mcts_policy(self._h, float(temp), _hip.ptr(pi)))
        return pi

    def uniform_policy(self, allow_pass=True):
        self._bind()
        pi = torch.empty((self.n_games, self.A), dtype=torch.float64, device=self.device)
        _hip.check(_hip.lib.goai_uniform_policy(self._h, int(bool(allow_pass)), _hip.ptr(pi)))
        return pi

    def sample_actions(self, pi, seed, step, game_id_offset=0):
        self._bind()
        actions = torch.empty(self.n_games, dtype=torch.int32, device=self.device)
        _hip.check(_hip.lib.goai_sample_actions(self._h, _hip.ptr(pi), int(seed), int(step), int(game_id_offset),
                                                _hip.ptr(actions)))
        return actions

    def mcts_advance(self, actions):
        self._bind()
        acts = torch.as_tensor(actions).to(self.device, dtype=torch.int32).contiguous()
        _hip.check(_hip.lib.goai_mcts_advance(self._h, _hip.ptr(acts)))

    def mcts_export_tree(self, game=0):
        """The whole tree of one game as host arrays: dict(parent, action, visits, value, raw_sum, prior, states),
        one entry per node, node 0 = the root (goai_mcts_export_tree)."""
        self._bind()
        cap = _hip.lib.goai_mcts_node_capacity(self._h)
        if cap < 0:
            _hip.check(cap)
        dev = self.device
        n = torch.zeros(1, dtype=torch.int32, device=dev)
        parent, action, visits = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3)]
        value, raw_sum = [torch.empty(cap, dtype=torch.float64, device=dev) for _ in range(2)]
        prior = torch.empty((cap, self.A), dtype=torch.float64, device=dev)
        states = torch.empty((cap, NUM_CHNLS, self.size, self.size), dtype=torch.uint8, device=dev)
        _hip.check(_hip.lib.goai_mcts_export_tree(self._h, int(game), _hip.ptr(n), _hip.ptr(parent), _hip.ptr(action),
                                                  _hip.ptr(visits), _hip.ptr(value), _hip.ptr(raw_sum), _hip.ptr(prior),
                                                  _hip.ptr(states)))
        k = int(n.item())
        return dict(parent=parent[:k].cpu().numpy(), action=action[:k].cpu().numpy(), visits=visits[:k].cpu().numpy(),
                    value=value[:k].cpu().numpy(), raw_sum=raw_sum[:k].cpu().numpy(), prior=prior[:k].cpu().numpy(),
                    states=states[:k].cpu().numpy())

    def mcts_set_refill(self, on=True):
        """Finished games are replaced by fresh ones (empty board) in their slot by mcts_advance."""
        _hip.check(_hip.lib.goai_mcts_set_refill(self._h, int(bool(on))))

    def mcts_game_stats(self):
        """Totals over the games finished since the last reset: finished, black_wins, white_wins, moves."""
        self._bind()
        out = (ctypes.c_int64 * 4)()
        _hip.check(_hip.lib.goai_mcts_game_stats(self._h, out))
        return dict(finished=out[0], black_wins=out[1], white_wins=out[2], finished_moves=out[3])

    def mcts_roots(self):
        self._bind()
        states = torch.empty((self.n_games, NUM_CHNLS, self.size, self.size), dtype=torch.uint8, device=self.device)
        turn, moves, result = [torch.empty(self.n_games, dtype=torch.int32, device=self.device) for _ in range(3)]
        _hip.check(_hip.lib.goai_mcts_roots(self._h, _hip.ptr(states), _hip.ptr(turn), _hip.ptr(moves),
                                            _hip.ptr(result)))
        return states, turn, moves, result

    def mcts_counters(self):
        self._bind()
        out = (ctypes.c_int64 * 4)()
        _hip.check(_hip.lib.goai_mcts_counters(self._h, out))
        return dict(leaf_evals=out[0], transitions=out[1], errors=out[2], forwards=out[3])

    # ---- measurement -----------------------------------------------------------------
    def profile_enable(self, on=True):
        _hip.check(_hip.lib.goai_profile_enable(self._h, int(bool(on))))

    def profile_read(self):
        self._bind()
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _hip.check(_hip.lib.goai_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
