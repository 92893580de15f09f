"""Read-only view of a GPU-resident search tree's root, with the attribute surface callers of
the reference's tree.Node use (/root/reference/go_ai/policies/actorcritic.py:71-76,
/root/reference/go_ai/policies/value.py:32-43, /root/reference/go_ai/search/plot.py:43-60)."""
import numpy as np

from . import invert_vals


class RootView:
    def __init__(self, state, visit_counts, prior_pi, visits, child_vals=None):
        self.state = state
        self.prior_pi = prior_pi
        self.visits = int(visits)
        self.level = 0
        self.parent = None
        self._counts = visit_counts
        self._child_vals = child_vals      # Q-mode: value of every valid child (nan where invalid)

    def actionsize(self):
        return len(self._counts)

    def valid_moves(self):
        return np.append(1.0 - (self.state[3] != 0).astype(np.float64).flatten(), 1.0)

    def get_visit_counts(self):
        return np.array(self._counts)

    def inverted_children_values(self):
        if self._child_vals is None:
            return np.zeros(self.actionsize())
        return np.array([invert_vals(v) if not np.isnan(v) else 0 for v in self._child_vals])

    def terminal(self):
        return int(np.count_nonzero(self.state[5] == 1) == self.state.shape[1] * self.state.shape[2])

    def isroot(self):
        return True

    def destroy(self):
        self._counts = self._child_vals = self.prior_pi = None

    def __str__(self):
        return f'{self.level}L {self.visits}N'
