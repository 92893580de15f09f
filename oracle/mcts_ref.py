"""oracle/mcts_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Sequential, one-leaf-at-a-time CPU restatement of the reference's search and policy code,
keeping its execution shape (Python tree of nodes, per-node value lists, NN called on a
batch of one):

  find_next_node / mct_step / mct_search   /root/reference/go_ai/search/mct.py:10-62
  Node (get_ucbs, step, backprop, ...)      /root/reference/go_ai/search/tree.py:22-195
  invert_vals / temp_norm / greedy_pi       /root/reference/go_ai/search/__init__.py:9-10,46-52,78-87
  ActorCritic.__call__ / Value.__call__     /root/reference/go_ai/policies/actorcritic.py:22-69,
                                            /root/reference/go_ai/policies/value.py:19-45

Arithmetic follows the reference's NumPy-1.x semantics: priors widened to float64 before
tree.py:182, `np.mean(np.tanh(list))` with numpy's pairwise summation, `np.nanargmax`
(first maximum wins).  `math='numpy'` uses numpy/scipy for tanh and softmax exactly as the
reference does; `math='det'` swaps in the deterministic restatements of oracle/detmath.c,
which the HIP kernels reproduce bit for bit.  Pinned by tests/golden/mcts_*.npz, which were
produced by the reference's own tree.py / mct.py (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes

import numpy as np
from scipy import special

from . import gogame
from .gogame import GoGame


def _det():
    lib = gogame.lib()
    if not getattr(lib, "_det_ready", False):
        lib.goai_oracle_tanh64.restype = ctypes.c_double
        lib.goai_oracle_tanh64.argtypes = [ctypes.c_double]
        lib.goai_oracle_exp32.restype = ctypes.c_float
        lib.goai_oracle_exp32.argtypes = [ctypes.c_float]
        lib.goai_oracle_exp64.restype = ctypes.c_double
        lib.goai_oracle_exp64.argtypes = [ctypes.c_double]
        lib.goai_oracle_softmax32.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_float)]
        lib.goai_oracle_softmax64_masked.argtypes = [ctypes.POINTER(ctypes.c_double),
                                                     ctypes.POINTER(ctypes.c_uint8), ctypes.c_int,
                                                     ctypes.POINTER(ctypes.c_double)]
        lib.goai_oracle_pairwise_sum64.restype = ctypes.c_double
        lib.goai_oracle_pairwise_sum64.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        lib._det_ready = True
    return lib


def det_tanh(x):
    lib = _det()
    return np.array([lib.goai_oracle_tanh64(float(v)) for v in np.atleast_1d(x)], dtype=np.float64)


def det_softmax32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    _det().goai_oracle_softmax32(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), len(x),
                                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def det_softmax64_masked(x, valid):
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = np.ascontiguousarray(valid != 0, dtype=np.uint8)
    out = np.empty_like(x)
    _det().goai_oracle_softmax64_masked(x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                        v.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(x),
                                        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def pairwise_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return _det().goai_oracle_pairwise_sum64(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(a))


def invert_vals(vals):
    return -vals


class Node:
    def __init__(self, state, parent=None, math="numpy"):
        self.state = state
        self.parent = parent
        self.math = math
        self.child_nodes = [None] * GoGame.action_size(state)
        self.level = 0 if parent is None else parent.level + 1
        self.val = None
        self.visits = 0
        self.prior_pi = None
        self.post_vals = []

    def terminal(self):
        return GoGame.game_ended(self.state)

    def actionsize(self):
        return len(self.child_nodes)

    def valid_moves(self):
        return GoGame.valid_moves(self.state)

    def make_childnode(self, action, state):
        child = Node(state, self, self.math)
        self.child_nodes[action] = child
        return child

    def make_children(self):
        kids = GoGame.children(self.state, canonical=True, padded=True)
        for a in np.argwhere(self.valid_moves()).flatten():
            self.make_childnode(int(a), kids[a])
        return kids

    def get_child_nodes(self):
        return [c for c in self.child_nodes if c is not None]

    def step(self, move):
        child = self.child_nodes[move]
        if child is None:
            child = self.make_childnode(move, GoGame.next_state(self.state, move, canonical=True))
        return child

    def inverted_children_values(self):
        return np.array([invert_vals(c.val) if c is not None else 0 for c in self.child_nodes])

    def backprop(self, val):
        node = self
        while node is not None:           # the reference recurses; same order, same values
            node.post_vals.append(val)
            node.visits += 1
            val = invert_vals(val)
            node = node.parent

    def set_prior_pi(self, prior_pi):
        if prior_pi is not None:
            self.prior_pi = np.asarray(prior_pi, dtype=np.float64)   # NumPy-1.x promotion at tree.py:182
        else:
            valid = self.valid_moves()
            q_logits = self.inverted_children_values().astype(np.float64)
            if self.math == "det":
                self.prior_pi = det_softmax64_masked(q_logits, valid)
            else:
                self.prior_pi = np.zeros(self.actionsize())
                where_valid = np.argwhere(valid).flatten()
                self.prior_pi[where_valid] = special.softmax(q_logits[where_valid])
            assert not np.isnan(self.prior_pi).any()

    def get_visit_counts(self):
        return np.array([0 if c is None else c.visits for c in self.child_nodes])

    def mean_tanh(self):
        if self.math == "det":
            t = det_tanh(np.asarray(self.post_vals, dtype=np.float64))
            return pairwise_sum(t) / len(t)
        return np.mean(np.tanh(self.post_vals))

    def get_ucbs(self):
        ucbs = np.full(self.actionsize(), np.nan, dtype=float)
        for a in np.argwhere(self.valid_moves()).flatten():
            avg_q, n = 0, 0
            prior_q = self.prior_pi[a]
            child = self.child_nodes[a]
            if child is not None and child.visits > 0:
                n = child.visits
                avg_q = invert_vals(child.mean_tanh())
            u = 1.5 * prior_q * np.sqrt(self.visits) / (1 + n)
            ucbs[a] = avg_q + u
        return ucbs


def find_next_node(node):
    curr = node
    while curr.visits > 0 and not curr.terminal():
        curr = curr.step(int(np.nanargmax(curr.get_ucbs())))
    return curr


def mct_step(rootnode, actor_critic, critic):
    node = find_next_node(rootnode)
    if actor_critic is not None:
        pi_logits, val_logits = actor_critic(node.state[np.newaxis])
    else:
        pi_logits, val_logits = None, critic(node.state[np.newaxis])
    node.backprop(val_logits.item())
    if node.terminal():
        return node
    if pi_logits is not None:
        flat = pi_logits.flatten()
        pi = det_softmax32(flat) if rootnode.math == "det" else special.softmax(flat)
        node.set_prior_pi(pi)
    else:
        node.make_children()
        kids = node.get_child_nodes()
        vals = critic(np.array([k.state for k in kids]))
        for v, k in zip(vals, kids):
            k.val = v.item()
        node.set_prior_pi(None)
    return node


def mct_search(go_env, num_searches, actor_critic=None, critic=None, math="numpy"):
    rootnode = Node(go_env.canonical_state(), math=math)
    for _ in range(num_searches + 1):      # the first iteration "doesn't count" (mct.py:55-56)
        mct_step(rootnode, actor_critic, critic)
    return rootnode


# ---- policies ---------------------------------------------------------------------------
def _l1(x):
    s = np.abs(x).sum()
    return x / s if s != 0 else x


def greedy_pi(qvals, valid_moves):
    expq = np.exp(qvals - np.max(qvals)) * valid_moves
    return _l1((expq == np.max(expq)).astype(float))


def temp_norm(qs, temp, valid_moves):
    if temp <= 0:
        return greedy_pi(qs, valid_moves)
    pi = np.zeros(valid_moves.shape)
    where_valid = np.where(valid_moves)
    pi[where_valid] = _l1(qs[where_valid])
    return _l1(pi ** (1 / temp))


def actor_critic_policy(go_env, ac_func, mcts, temp, math="numpy"):
    """ActorCritic.__call__ for mcts > 0: returns (pi, visit_counts, rootnode)."""
    root = mct_search(go_env, mcts, actor_critic=ac_func, math=math)
    counts = root.get_visit_counts().astype(float)
    pi = counts ** (1 / temp)
    pi = pi / np.sum(pi)
    valid = go_env.valid_moves()
    pi = 0.98 * pi + 0.02 * (valid / valid.sum())
    return pi, counts, root


def value_policy(go_env, val_func, mcts, temp, math="numpy"):
    """Value.__call__: visit counts (mcts > 0) or exp(-child values) (mcts == 0) -> temp_norm."""
    root = mct_search(go_env, mcts, critic=val_func, math=math)
    if mcts > 0:
        qs = root.get_visit_counts().astype(float)
    elif math == "det":
        lib = _det()
        qs = np.array([lib.goai_oracle_exp64(float(v)) for v in root.inverted_children_values()])
    else:
        qs = np.exp(root.inverted_children_values())
    return temp_norm(qs, temp, root.valid_moves()), root
