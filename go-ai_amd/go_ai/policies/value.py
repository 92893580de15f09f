"""Value (Q-value) policy (/root/reference/go_ai/policies/value.py:9-48)."""
import numpy as np

from .. import search
from ..env import context
from ..search.tree import RootView
from . import Policy


def children_values(state, val_func):
    """One expansion of the root in critic mode (mct.py:44-47 with num_searches = 0): all
    children by the HIP transition kernels, one batched value call, -value per valid move."""
    size = state.shape[-1]
    kids, valid, _ = context(size).children(state[None], canonical=True)
    valid = valid[0].cpu().numpy().astype(bool)
    kid_states = kids[0][np.flatnonzero(valid)].cpu().numpy().astype(np.float64)
    vals = np.asarray(val_func(kid_states), dtype=np.float64).reshape(-1)
    child_vals = np.full(len(valid), np.nan)
    child_vals[valid] = vals
    root = RootView(state, np.zeros(len(valid), dtype=np.int64), None, 1, child_vals)
    return root.inverted_children_values(), root


class Value(Policy):
    def __init__(self, name, engine, args=None):
        super().__init__(name, args.temp if args is not None else 0)
        from ..models import RLNet
        if isinstance(engine, RLNet):
            self.pt_model = engine
            self.val_func = engine.create_numpy('critic')
        else:
            self.val_func = engine
        self.mcts = args.mcts if args is not None else 0

    def __call__(self, go_env, **kwargs):
        debug = kwargs.get('debug', False)
        from ..search import mct
        rootnode = mct.mct_search(go_env, self.mcts, critic=self.val_func)
        if self.mcts > 0:
            qs = rootnode.get_visit_counts()
            assert np.sum(qs) > 0, rootnode
        else:
            qs = np.exp(rootnode.inverted_children_values())
        # host fp64, as the reference; the batched self-play path has the same normalisation as
        # a kernel (goai_mcts_policy_value)
        pi = search.temp_norm(np.asarray(qs, dtype=np.float64), self.temp, rootnode.valid_moves())
        if debug:
            q_logits = rootnode.inverted_children_values()
            return pi, [q_logits, q_logits], rootnode
        rootnode.destroy()
        return pi

    def __str__(self):
        return f"{self.__class__.__name__}[{self.mcts}S {self.temp:.2f}T]-{self.name}"
