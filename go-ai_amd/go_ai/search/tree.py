"""Read-only views of a GPU-resident search tree with the attribute surface callers of the
reference's tree.Node use (/root/reference/go_ai/search/tree.py:22-195;
/root/reference/go_ai/policies/actorcritic.py:71-76, /root/reference/go_ai/policies/value.py:32-43,
/root/reference/go_ai/search/plot.py:39-61 walks `child_nodes` recursively and prints str(node)).

The tree itself lives in HBM (csrc/mcts.h).  `mct_search` returns the root as a `RootView` built from
the small per-root read-outs; the rest of the tree is fetched on first use of `child_nodes`
(goai_mcts_export_tree) -- before the next search on the same context, which overwrites it.
"""
import numpy as np

from . import invert_vals


class Node:
    """One node of an exported tree (tree.Node's read side)."""

    def __init__(self, state, parent, level, visits, prior_pi, val, post_sum, first_action, actionsize):
        self.state = state
        self.parent = parent
        self.level = level
        self.visits = int(visits)
        self.prior_pi = prior_pi
        self.val = val
        self.first_action = first_action
        self.child_states = None
        self._post_sum = post_sum                     # sum(post_vals); the list itself stays on the GPU as running sums
        self.child_nodes = np.empty(actionsize, dtype=object)

    # ---- basic tree API (tree.py:65-108) ----
    def terminal(self):
        return int(np.count_nonzero(self.state[5] == 1) == self.state.shape[1] * self.state.shape[2])

    def winning(self):
        from ..env import GoGame
        return GoGame.winning(self.state)

    def isleaf(self):
        return bool((self.child_nodes == None).all())       # noqa: E711  (elementwise, as the reference)

    def isroot(self):
        return self.parent is None

    def get_child_nodes(self):
        return [c for c in self.child_nodes if c is not None]

    def actionsize(self):
        return len(self.child_nodes)

    def valid_moves(self):
        return np.append(1.0 - (self.state[3] != 0).astype(np.float64).flatten(), 1.0)

    # ---- value (tree.py:122-135) ----
    def get_value(self):
        return self.val

    def inverted_children_values(self):
        return np.array([invert_vals(c.val) if c is not None and c.val is not None else 0 for c in self.child_nodes])

    # ---- MCT read-outs (tree.py:161-168, 186-195) ----
    def get_visit_counts(self):
        return np.array([0 if c is None else c.visits for c in self.child_nodes])

    def post_mean(self):
        """np.mean(node.post_vals) (tree.py:191), from the sum kept on the GPU."""
        return self._post_sum / self.visits if self.visits > 0 else None

    def destroy(self):
        for child in self.child_nodes:
            if child is not None:
                child.destroy()
        self.state = self.parent = None
        self.child_nodes = np.empty(0, dtype=object)

    def __str__(self):
        result = ''
        if self.val is not None:
            result += f'{self.val:.2f}V'
        if self.visits > 0 and self._post_sum is not None:
            result += f' {self.post_mean():.2f}AV'
        result += f' {self.level}L {self.visits}N'
        return result


def build_tree(export, root=None):
    """Node objects of a goai_mcts_export_tree dump (engine.Context.mcts_export_tree); `root`, if given,
    is used as node 0 (its child_nodes are filled in).  Returns the root."""
    n, A = len(export["parent"]), export["prior"].shape[1]
    nodes = [None] * n
    for i in range(n):
        par = int(export["parent"][i])
        prior = export["prior"][i]
        prior = None if np.isnan(prior).all() else prior.copy()
        val = export["value"][i]
        val = None if np.isnan(val) else float(val)
        state = export["states"][i].astype(np.float64)
        if i == 0:
            node = root if root is not None else Node(state, None, 0, export["visits"][0], prior, val,
                                                      float(export["raw_sum"][0]), None, A)
            node._post_sum = float(export["raw_sum"][0])
        else:
            parent = nodes[par]
            a = int(export["action"][i])
            node = Node(state, parent, parent.level + 1, export["visits"][i], prior, val, float(export["raw_sum"][i]),
                        a if parent.level == 0 else parent.first_action, A)
            parent.child_nodes[a] = node
        nodes[i] = node
    return nodes[0]


class RootView(Node):
    """The root of a search: counts, prior and child values come from the per-root read-outs
    (goai_mcts_readout / goai_mcts_child_values); `child_nodes` loads the whole tree on demand."""

    def __init__(self, state, visit_counts, prior_pi, visits, child_vals=None):
        self.state = state
        self.prior_pi = prior_pi
        self.visits = int(visits)
        self.level = 0
        self.parent = None
        self.val = None
        self.first_action = None
        self.child_states = None
        self._post_sum = None
        self._counts = visit_counts
        self._child_vals = child_vals      # Q-mode: value of every valid child (nan where invalid)
        self._ctx = self._gen = None
        self._children = None

    def attach(self, ctx, game=0):
        """Remember where the tree lives so that child_nodes can fetch it later."""
        self._ctx, self._game, self._gen = ctx, game, ctx.search_generation

    @property
    def child_nodes(self):
        if self._children is None:
            if self._counts is None:
                return np.empty(0, dtype=object)
            if self._ctx is None:                     # a root without a tree behind it (mcts = 0 helpers)
                self._children = np.empty(self.actionsize(), dtype=object)
            else:
                if self._ctx.closed:
                    raise RuntimeError("the context that held this search tree was closed (evicted by later searches of "
                                       "other shapes); read child_nodes (or call load()) right after the search")
                if self._ctx.search_generation != self._gen:
                    raise RuntimeError("this search tree was overwritten by a later search on the same context; "
                                       "read child_nodes (or call load()) before searching again")
                self._children = np.empty(self.actionsize(), dtype=object)
                build_tree(self._ctx.mcts_export_tree(self._game), root=self)
        return self._children

    def load(self):
        """Fetch the whole tree now (it is overwritten by the next search on the same context)."""
        self.child_nodes
        return self

    def actionsize(self):
        return len(self._counts)

    def isleaf(self):
        return not (np.asarray(self._counts) > 0).any() and self._child_vals is None

    def get_visit_counts(self):
        return np.array(self._counts)

    def inverted_children_values(self):
        if self._child_vals is None:
            return np.zeros(self.actionsize())
        return np.array([invert_vals(v) if not np.isnan(v) else 0 for v in self._child_vals])

    def destroy(self):
        self._counts = self._child_vals = self.prior_pi = None
        self._children = self._ctx = None
