"""ActorCritic policy (/root/reference/go_ai/policies/actorcritic.py:8-79)."""
import numpy as np

from .. import search
from ..search import mct
from . import Policy


class ActorCritic(Policy):
    def __init__(self, name, model, args=None):
        super().__init__(name, temp=args.temp)
        self.pt_model = model
        self.ac_func = model.create_numpy('actor_critic')
        self.val_func = model.create_numpy('critic')
        self.pi_func = model.create_numpy('actor')
        self.mcts = args.mcts

    def __call__(self, go_env, **kwargs):
        if self.mcts > 0:
            rootnode = mct.mct_search(go_env, self.mcts, actor_critic=self.ac_func)
            qs = np.empty((2, rootnode.actionsize()))
            qs[0] = rootnode.prior_pi
            qs[1] = rootnode.get_visit_counts()
            # counts ** (1/T), normalise, 2 % uniform noise over valid moves: on the GPU (fp64)
            pi = rootnode._ctx.mcts_policy(self.temp)[0].cpu().numpy()
            assert np.allclose(np.sum(pi), 1), np.sum(pi)
        elif self.mcts == 0:
            from .value import children_values
            state = np.asarray(go_env.canonical_state())
            q_logits, rootnode = children_values(state, self.val_func)
            pi = search.temp_norm(np.exp(q_logits), self.temp, rootnode.valid_moves())
            qs = [q_logits]
        else:
            # the reference's mcts < 0 branch builds a broken Node (actorcritic.py:54); the
            # arithmetic it intends is a temperature softmax of the actor logits
            from ..search.tree import RootView
            state = np.asarray(go_env.canonical_state())
            policy_scores = self.pi_func(state[np.newaxis])[0]
            valid_moves = np.append(1.0 - (state[3] != 0).flatten(), 1.0)
            pi = search.temp_softmax(policy_scores, self.temp, valid_moves)
            qs = [pi]
            rootnode = RootView(state, np.zeros(len(pi)), None, 0)
        if kwargs.get('debug', False):
            return pi, qs, rootnode
        return pi

    def __str__(self):
        return f"{self.__class__.__name__}[{self.mcts}S {self.temp:.2f}T]-{self.name}"
