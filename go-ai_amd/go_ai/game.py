"""Game loops.

`pit` / `play_games` keep the reference's signatures (/root/reference/go_ai/game.py:55-154) for
arbitrary Policy objects on one environment.  `BatchedSelfPlay` is the MI355X-native shape
of the same loop for the hot case -- one actor-critic net playing itself with MCTS -- where
thousands of games advance in lock step and everything (rules, search, CNN, policy,
sampling) stays on the GPU; games are independent, so a node shards them over its GPUs by
game id and only gathers results at the end (/root/reference/go_ai/utils.py:179-208).
"""
import numpy as np
import torch

from .engine import Context
from .env import GoGame, GoVars


class Trajectory:
    # /root/reference/go_ai/game.py:6-52
    def __init__(self):
        self.states, self.actions, self.rewards, self.children, self.pis = [], [], [], [], []

    def add_event(self, state, action, reward, children, pi):
        self.states.append(state)
        self.actions.append(action)
        self.rewards.append(reward)
        self.children.append(children)
        self.pis.append(pi)

    def set_win(self, black_won):
        self.rewards[-1] = black_won

    def get_winner(self):
        return self.rewards[-1]

    def get_events(self):
        black_won, n, events = self.get_winner(), len(self), []
        for i, (s, a, r, c, pi) in enumerate(zip(self.states, self.actions, self.rewards, self.children, self.pis)):
            won = black_won if i % 2 == 0 else -black_won
            events.append((s, a, r, c, i == n - 1, won, pi))
        return events

    def __len__(self):
        return len(self.states)


def pit(go_env, black_policy, white_policy):
    num_steps, done = 0, False
    state = go_env.canonical_state()
    max_steps = 2 * (go_env.size ** 2)
    traj = Trajectory()
    while not done:
        policy = black_policy if go_env.turn() == GoVars.BLACK else white_policy
        pi = policy(go_env, step=num_steps)
        action = GoGame.random_weighted_action(pi)
        padded_children = go_env.children(canonical=True, padded=True)
        _, reward, done, _ = go_env.step(action)
        if num_steps >= max_steps:
            done = True
        traj.add_event(state, action, reward, padded_children, pi)
        num_steps += 1
        state = padded_children[action]
    black_won = go_env.winning()
    traj.set_win(black_won)
    return black_won, num_steps, traj


def play_games(go_env, first_policy, second_policy, episodes, progress=False):
    replay, all_steps, first_wins, black_wins = [], [], 0, 0
    for i in range(1, episodes + 1):
        go_env.reset()
        if i % 2 == 0:
            black_won, steps, traj = pit(go_env, first_policy, second_policy)
            first_won = black_won
        else:
            black_won, steps, traj = pit(go_env, second_policy, first_policy)
            first_won = -black_won
        black_wins += int(black_won == 1)
        first_wins += int(first_won == 1)
        all_steps.append(steps)
        replay.append(traj)
    return first_wins / episodes, black_wins / episodes, replay, all_steps


class BatchedSelfPlay:
    """n_games concurrent self-play games of one actor-critic net with `mcts` searches/move."""

    def __init__(self, size, n_games, mcts, temp=1.0, state_dict=None, device=0, n_blocks=3, seed=1234,
                 game_id_offset=0):
        from .models import random_state_dict
        self.size, self.n_games, self.mcts, self.temp = size, n_games, mcts, temp
        self.seed, self.game_id_offset = seed, game_id_offset
        self.ctx = Context(size, device)
        self.ctx.nn_init(Context.NET_AC, n_blocks, n_games)
        self.ctx.load_weights(state_dict if state_dict is not None else random_state_dict("ac", size, 0, n_blocks))
        self.ctx.mcts_init(n_games, mcts)
        self.move_index = 0
        self.reset()

    def reset(self, root_states=None):
        if root_states is None:
            root_states = torch.zeros((self.n_games, 6, self.size, self.size), dtype=torch.uint8,
                                      device=self.ctx.device)
        self.ctx.mcts_reset(root_states)
        self.move_index = 0

    def randomize_starts(self, max_random_moves=40):
        """Advance game g by k_g ~ U{0..max_random_moves} uniformly random legal non-pass moves
        (Philox stream per game id), so that trees differ."""
        c = self.ctx
        gen = torch.Generator().manual_seed(self.seed)
        k = torch.randint(0, max_random_moves + 1, (1 << 20,), generator=gen)
        idx = (torch.arange(self.n_games) + self.game_id_offset) % (1 << 20)
        k = k[idx].to(c.device)
        states = c.mcts_roots()[0]
        for i in range(max_random_moves):
            c.mcts_reset(states)
            pi = c.uniform_policy(allow_pass=False)
            acts = c.sample_actions(pi, self.seed ^ 0x5eed, i, self.game_id_offset)
            nxt, status = c.transition(states, acts, canonical=True)
            take = (k > i) & (status == 0)
            states = torch.where(take[:, None, None, None], nxt, states)
        c.mcts_reset(states)
        self.move_index = 0

    def move(self):
        """One move for every running game: search, policy, temperature-sampled action, step."""
        c = self.ctx
        c.mcts_run(self.mcts)
        pi = c.mcts_policy(self.temp)
        acts = c.sample_actions(pi, self.seed, self.move_index, self.game_id_offset)
        c.mcts_advance(acts)
        self.move_index += 1
        return pi, acts

    def play(self, n_moves):
        for _ in range(n_moves):
            self.move()

    def stats(self):
        c = self.ctx
        _, turn, moves, result = [t.cpu().numpy() for t in c.mcts_roots()]
        cnt = c.mcts_counters()
        fin = result != 2
        return dict(games=self.n_games, finished=int(fin.sum()), black_wins=int((result == 1).sum()),
                    white_wins=int((result == -1).sum()), total_steps=int(moves.sum()), **cnt)


from .sharding import STAT_FIELDS, gather_stats, reduce_stats, shard_games  # noqa: E402,F401
