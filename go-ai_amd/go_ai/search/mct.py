"""mct_search with the reference's signature (/root/reference/go_ai/search/mct.py:50-62),
executed by the GPU-resident tree kernels.

`actor_critic` may be
  * the callable returned by `RLNet.create_numpy('actor_critic')` of a go_ai.models net: the
    whole search (select, CNN forward, expand, backup) then stays on the GPU
    (goai_mcts_run), or
  * any other callable f(states[B,6,s,s]) -> (pi_logits[B,A], val_logits[B,1]) with
    RLNet._numpy semantics (e.g. the reference's torch model): the tree runs on the GPU and f
    is called once per step on the selected leaf (goai_mcts_select / goai_mcts_backup).

`critic` (Q-value search, mct.py:27-30,43-47) works the same way: a create_numpy('critic')
closure of a go_ai.models net keeps everything on the GPU (leaf + all valid children in one
compacted batch per step); any other callable f(states[B,6,s,s]) -> val_logits[B,1] is called
on the leaf and then on its valid children, exactly as tree.set_state_vals does.
"""
import numpy as np

from ..engine import Context
from .tree import RootView

_plain_ctx = {}
_MAX_PLAIN_CTX = 4       # tree-only contexts for foreign callables: the last few (size, searches, mode) are kept


def _tree_context(size, num_searches, mode=Context.SEARCH_AC):
    key = (size, num_searches, mode)
    if key in _plain_ctx:
        _plain_ctx[key] = _plain_ctx.pop(key)                    # most recently used last
    else:
        while len(_plain_ctx) >= _MAX_PLAIN_CTX:
            _plain_ctx.pop(next(iter(_plain_ctx))).close()
        c = Context(size)
        c.mcts_init(1, num_searches, mode)
        _plain_ctx[key] = c
    return _plain_ctx[key]


def _critic_search(root_state, num_searches, critic):
    size = root_state.shape[-1]
    net = _owning_net(critic)
    if net is not None:
        ctx = net.search_context(1, num_searches, Context.SEARCH_Q)
        ctx.mcts_reset(root_state[None])
        ctx.mcts_run(num_searches)
    else:
        ctx = _tree_context(size, num_searches, Context.SEARCH_Q)
        ctx.mcts_reset(root_state[None])
        for _ in range(num_searches + 1):
            leaf = ctx.mcts_select().cpu().numpy().astype(np.float64)
            kids, valid = ctx.mcts_expand()
            valid = valid[0].cpu().numpy().astype(bool)
            leaf_val = np.asarray(critic(leaf), dtype=np.float64).reshape(1)
            child_vals = np.zeros((1, ctx.A), dtype=np.float64)
            if valid.any():
                kid_states = kids[0].cpu().numpy()[valid].astype(np.float64)
                child_vals[0, valid] = np.asarray(critic(kid_states), dtype=np.float64).reshape(-1)
            ctx.mcts_backup_q(leaf_val, child_vals, vals_are_raw=False)
    visits, prior, root_visits = ctx.mcts_readout()
    inv = ctx.mcts_child_values()[0].cpu().numpy()
    valid_root = np.append(root_state[3].flatten() == 0, True)
    child_vals = np.where(valid_root, -inv, np.nan)
    root = RootView(root_state, visits[0].cpu().numpy(), prior[0].cpu().numpy(), root_visits[0].item(), child_vals)
    root.attach(ctx)
    return root


def _owning_net(fn):
    """The go_ai.models net behind a create_numpy callable: RLNet.create_numpy marks what it returns
    with `_goai_net` (functools.partial objects and bound methods of such a callable are looked through).
    None for any other callable: it is then evaluated on the host, one call per search step."""
    from ..models import RLNet
    for _ in range(4):
        net = getattr(fn, "_goai_net", None)
        if isinstance(net, RLNet):
            return net
        inner = getattr(fn, "func", None) or getattr(fn, "__wrapped__", None) or getattr(fn, "__func__", None)
        if inner is None:
            return None
        fn = inner
    return None


def mct_search(go_env, num_searches, actor_critic=None, critic=None):
    root_state = np.asarray(go_env.canonical_state())
    if actor_critic is None:
        assert critic is not None
        return _critic_search(root_state, num_searches, critic)
    size = root_state.shape[-1]
    net = _owning_net(actor_critic)
    if net is not None and net.kind == "ac":
        ctx = net.search_context(1, num_searches)
        ctx.mcts_reset(root_state[None])
        ctx.mcts_run(num_searches)
    else:
        ctx = _tree_context(size, num_searches)
        ctx.mcts_reset(root_state[None])
        for _ in range(num_searches + 1):
            leaf = ctx.mcts_select().cpu().numpy().astype(np.float64)
            pi_logits, val_logits = actor_critic(leaf)
            ctx.mcts_backup(np.asarray(val_logits, dtype=np.float32).reshape(1),
                            pi_logits=np.asarray(pi_logits, dtype=np.float32).reshape(1, -1), logits_are_raw=False)
    visits, prior, root_visits = ctx.mcts_readout()
    root = RootView(root_state, visits[0].cpu().numpy(), prior[0].cpu().numpy(), root_visits[0].item())
    root.attach(ctx)
    return root
