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
        # side to move of every event (0 black, 1 white) when the recorder knows it; None = the reference's
        # assumption that a game starts with black and the colours alternate (game.py:15-24)
        self.movers = None

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
            white = self.movers[i] if self.movers is not None else i % 2
            events.append((s, a, r, c, i == n - 1, -black_won if white else black_won, pi))
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
    """n_games concurrent self-play games of one actor-critic net with `mcts` searches/move.

    The games are split into `groups` independent groups, each with its own context and HIP
    stream.  Their search steps are enqueued interleaved, so while one group's convolutions
    occupy the matrix cores the other group's tree kernels, heads and kernel tails run beside
    them (the groups never exchange data; results do not depend on the grouping)."""

    def __init__(self, size, n_games, mcts, temp=1.0, state_dict=None, device=0, n_blocks=3, seed=1234,
                 game_id_offset=0, groups=1, search="ac", nn_batch=32768, dtype=Context.DTYPE_F32, refill=False):
        """search = "ac": actor-critic search (ActorCritic policy); "q": Q-value search with the
        value network (Value policy): every leaf expansion evaluates the leaf and all its valid
        children, in slices of at most nn_batch states.  refill=True: a slot whose game ends starts a
        new one from the empty board at once (play_games' episode loop, game.py:139-145, per slot), so
        the batch stays full however long the run; stats() counts every finished game either way."""
        from .models import random_state_dict
        from .sharding import shard_games
        assert search in ("ac", "q")
        self.size, self.n_games, self.mcts, self.temp, self.search = size, n_games, mcts, temp, search
        self.seed, self.game_id_offset = seed, game_id_offset
        groups = max(1, min(groups, n_games))
        kind = "ac" if search == "ac" else "val"
        sd = state_dict if state_dict is not None else random_state_dict(kind, size, 0, n_blocks)
        self.groups = []
        dev = torch.device("cuda", device)
        for gi in range(groups):
            lo, hi = shard_games(n_games, groups, gi)
            stream = torch.cuda.current_stream(dev) if groups == 1 else torch.cuda.Stream(dev)
            c = Context(size, device)
            c.set_stream(stream)
            if search == "ac":
                c.nn_init(Context.NET_AC, n_blocks, hi - lo, dtype=dtype)
            else:
                c.nn_init(Context.NET_VAL, n_blocks, min(nn_batch, (hi - lo) * (size * size + 2)), dtype=dtype)
            c.load_weights(sd)
            c.mcts_init(hi - lo, mcts, Context.SEARCH_AC if search == "ac" else Context.SEARCH_Q)
            c.mcts_set_refill(refill)
            self.groups.append((c, stream, game_id_offset + lo))
        self.ctx = self.groups[0][0]
        self.move_index = 0
        self._rec = None
        self.reset()

    def _each(self):
        for c, stream, off in self.groups:
            with torch.cuda.stream(stream):
                yield c, off

    def sync(self):
        for _, stream, _ in self.groups:
            stream.synchronize()

    def reset(self, root_states=None):
        lo = 0
        for c, _ in self._each():
            if root_states is None:
                roots = torch.zeros((c.n_games, 6, self.size, self.size), dtype=torch.uint8, device=c.device)
            else:
                roots = root_states[lo:lo + c.n_games]
            c.mcts_reset(roots)
            lo += c.n_games
        self.move_index = 0

    def randomize_starts(self, max_random_moves=40):
        """Advance game g by k_g ~ U{0..max_random_moves} uniformly random legal non-pass moves
        (Philox stream per game id), so that trees differ."""
        gen = torch.Generator().manual_seed(self.seed)
        ktab = torch.randint(0, max_random_moves + 1, (1 << 20,), generator=gen)
        for c, off in self._each():
            k = ktab[(torch.arange(c.n_games) + off) % (1 << 20)].to(c.device)
            states = c.mcts_roots()[0]
            for i in range(max_random_moves):
                c.mcts_reset(states)
                pi = c.uniform_policy(allow_pass=False)
                acts = c.sample_actions(pi, self.seed ^ 0x5eed, i, off)
                nxt, status = c.transition(states, acts, canonical=True)
                take = (k > i) & (status == 0)
                states = torch.where(take[:, None, None, None], nxt, states)
            c.mcts_reset(states)
        self.sync()
        self.move_index = 0

    def move(self):
        """One move for every running game: search, policy, temperature-sampled action, step."""
        if len(self.groups) == 1:
            self.ctx.mcts_run(self.mcts)
        else:
            for _ in range(self.mcts + 1):
                for c, _ in self._each():
                    c.mcts_steps(1)
        out = []
        rec = []
        for c, off in self._each():
            pi = c.mcts_policy(self.temp) if self.search == "ac" else c.mcts_policy_value(self.temp, self.mcts == 0)
            if self._rec is not None:
                states, turn, _, result = c.mcts_roots()
            acts = c.sample_actions(pi, self.seed, self.move_index, off)
            c.mcts_advance(acts)
            if self._rec is not None:
                # copied to the host on the group's own stream (.cpu() waits for that stream only)
                rec.append(tuple(t.cpu() for t in (states, turn, result, pi, acts)))
            out.append((pi, acts))
        if self._rec is not None:
            self._rec.append(tuple(torch.cat([r[i] for r in rec]) for i in range(5)))
        self.move_index += 1
        return out

    # ---- trajectories in the reference's replay format (game.py:6-52, data.py:85-105) -------------
    def record(self, on=True):
        """Keep (canonical state, side to move, pi, action) of every move so that `trajectories()` can
        hand the games to the reference's training code."""
        self._rec = [] if on else None

    def trajectories(self, children_batch=256):
        """One `Trajectory` per game with the events `game.pit` would have recorded
        (game.py:80-113): state = canonical state before the move, children =
        GoGame.children(state, canonical=True, padded=True) (one kernel launch per batch of
        states), reward 0 except the last = black's result.  Moves of a game that had already
        finished are dropped.  `won` of an event is from the mover's point of view; games here may
        start with white to move, so use `trajectory_events` (below) rather than index parity."""
        assert self._rec is not None, "call record() before playing"
        final = torch.cat([c.mcts_roots()[3].cpu() for c, _ in self._each()]).numpy()
        self.sync()
        trajs = [Trajectory() for _ in range(self.n_games)]
        movers = [[] for _ in range(self.n_games)]
        c0, stream0, _ = self.groups[0]
        for states, turn, result, pi, acts in self._rec:
            live = (result == 2).nonzero().flatten()
            for lo in range(0, len(live), children_batch):
                idx = live[lo:lo + children_batch]
                with torch.cuda.stream(stream0):
                    kids = c0.children(states[idx].to(c0.device), canonical=True)[0].cpu().numpy()
                for j, g in enumerate(idx.tolist()):
                    trajs[g].add_event(states[g].numpy().astype(np.float64), int(acts[g]), 0.0,
                                       kids[j].astype(np.float64), pi[g].numpy())
                    movers[g].append(int(turn[g]))
        for g, t in enumerate(trajs):
            if len(t) > 0:
                # unfinished games (move cap of this run not reached): winner by the current area score
                t.set_win(int(final[g]) if final[g] != 2 else 0)
            t.movers = movers[g]
        return trajs


    def play(self, n_moves):
        for _ in range(n_moves):
            self.move()

    def close(self):
        """Release the groups' contexts (trees, activations, weights)."""
        self.sync()
        for c, _, _ in self.groups:
            c.close()
        self.groups = []

    def counters(self):
        tot = dict(leaf_evals=0, transitions=0, errors=0, forwards=0)
        for c, _ in self._each():
            for k, v in c.mcts_counters().items():
                tot[k] += v
        return tot

    def stats(self):
        """games = slots; finished / black_wins / white_wins = games completed since the last reset (in
        refill mode a slot completes several); total_steps = moves played in them and in the running games."""
        tot = dict(finished=0, black_wins=0, white_wins=0, finished_moves=0)
        running_moves = 0
        for c, _ in self._each():
            _, _, moves, result = [t.cpu().numpy() for t in c.mcts_roots()]
            running_moves += int(moves[result == 2].sum())
            for k, v in c.mcts_game_stats().items():
                tot[k] += v
        return dict(games=self.n_games, finished=tot["finished"], black_wins=tot["black_wins"],
                    white_wins=tot["white_wins"], total_steps=tot["finished_moves"] + running_moves, **self.counters())

    def profile_transition(self, n_boards=262144, iters=20):
        """Time goai_transition (k_transition_staged: GoGame.next_state, one board per lane) over n_boards boards -- the games' current
        positions, repeated, each with a uniformly drawn legal non-pass move -- with HIP events on the
        launch stream.  Returns the `roofline` object of the transition side: algorithmic bytes are
        6P read + 6P written per board against the HBM peak."""
        from . import _hip
        c, stream, off = self.groups[0]
        P = self.size * self.size
        with torch.cuda.stream(stream):
            states = c.mcts_roots()[0]
            acts = c.sample_actions(c.uniform_policy(allow_pass=False), self.seed ^ 0x7a, 0, off)
            reps = -(-n_boards // c.n_games)
            st = states.repeat(reps, 1, 1, 1)[:n_boards].contiguous()
            ac = acts.repeat(reps)[:n_boards].contiguous()
            out, status = torch.empty_like(st), torch.empty(n_boards, dtype=torch.int32, device=c.device)

            def call():
                _hip.check(_hip.lib.goai_transition(c._h, _hip.ptr(st), _hip.ptr(ac), 1, _hip.ptr(out), _hip.ptr(status),
                                                    n_boards))
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(iters):
                call()
            e1.record(stream)
            e1.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / iters
            legal = float((status == 0).float().mean())
            stones = float((states[:, 0] | states[:, 1]).sum()) / max(1, states.shape[0])
        nbytes = n_boards * 12 * P
        return {"bound": "hbm", "kernel": f"k_transition_staged<{self.size}>", "achieved": round(nbytes / sec / 1e9, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / sec / 8e12, 4), "traffic": None,
                "avg_launch_us": round(sec * 1e6, 2), "launches_timed": iters, "boards_per_launch": n_boards,
                "algorithmic_bytes_per_board": 12 * P, "boards_per_s": round(n_boards / sec, 1), "legal_fraction": round(legal, 4),
                "stones_per_board": round(stones, 1),
                "note": "one board per lane as bit boards in registers, byte records staged through LDS in 16-byte lines; "
                        "bound by vector-ALU issue (pack / unpack of the uint8 planes and the flood fills), not by HBM "
                        "(DESIGN.md section 3; the one-board-per-wavefront kernel it replaces here: 0.083 at 65,536 boards)"}

    def profile_conv(self):
        """Time every conv3x3 launch of one more move with HIP events on the launch streams."""
        for c, _ in self._each():
            c.profile_enable(True)
        self.move()
        ms, n = 0.0, 0
        for c, _ in self._each():
            m, k = c.profile_read()
            c.profile_enable(False)
            ms, n = ms + m, n + k
        return ms, n


class BatchedPlay:
    """`play_games(go_env, first_policy, second_policy, episodes)` (/root/reference/go_ai/game.py:120-154) with
    the episodes played `slots` at a time on the GPU: the first policy takes black in the even-numbered
    episodes and white in the odd-numbered ones (game.py:141-146, episodes count from 1), a slot whose game
    ends takes the next episode at once, and `first_wins`, `black_wins` and the step counts are accumulated
    as the reference's loop does (game.py:147-149).  This is what evaluation needs
    (/root/reference/train.py:13-20: a checkpoint against an older one, the random and the greedy player).

    Every move, the running games are split by the policy that is to move (game.py:85-89); each policy's own
    context searches its share as one batch (goai_mcts_set_active / goai_mcts_reset / goai_mcts_run), turns the
    result into pi (goai_mcts_policy / goai_mcts_policy_value) and draws the action with the Philox key
    (seed, episode number, move number) -- a game does not depend on the slot, the batch or the GPU it is played
    in.  The moves themselves are goai_transition / goai_score on the whole batch.

    Policies: go_ai.policies.ActorCritic (mcts > 0: actor-critic search; mcts == 0: the critic's child values),
    go_ai.policies.Value (Q-value search, or child values for mcts == 0) on go_ai.models nets, and
    go_ai.policies.baselines.Random."""

    def __init__(self, size, first_policy, second_policy, slots=256, device=0, seed=1234, episode_offset=0):
        self.size, self.slots, self.seed, self.episode_offset = size, int(slots), seed, int(episode_offset)
        self.P, self.A = size * size, size * size + 1
        self.device = torch.device("cuda", device)
        self.policies = [self._prepare(first_policy, device), self._prepare(second_policy, device)]
        self.rules = Context(size, device)
        self._record = False

    def _prepare(self, policy, device):
        from .policies.actorcritic import ActorCritic
        from .policies.baselines import Random
        from .policies.value import Value
        if isinstance(policy, Random):
            c = Context(self.size, device)
            c.mcts_init(self.slots, 0)
            return dict(kind="random", ctx=c, policy=policy)
        net = getattr(policy, "pt_model", None)
        if net is None and isinstance(policy, Value):
            from .policies import baselines
            batched = {baselines.greedy_val_func: self._greedy_values, baselines.smart_greedy_val_func: self._smart_greedy_values}
            if policy.val_func in batched:
                c = Context(self.size, device)
                c.mcts_init(self.slots, max(policy.mcts, 0), Context.SEARCH_Q)
                return dict(kind="valfunc", ctx=c, policy=policy, values=batched[policy.val_func])
        if net is None or not hasattr(net, "search_context"):
            raise TypeError(f"BatchedPlay needs policies on go_ai.models nets (or baselines' Random / greedy players), got {policy}")
        if isinstance(policy, ActorCritic) and policy.mcts > 0:
            kind, mode = "ac", Context.SEARCH_AC
        elif isinstance(policy, (ActorCritic, Value)) and policy.mcts >= 0:
            kind, mode = "q", Context.SEARCH_Q
        else:
            raise TypeError(f"unsupported policy for batched play: {policy}")
        return dict(kind=kind, ctx=net.search_context(self.slots, max(policy.mcts, 0), mode), policy=policy)

    # baselines.greedy_val_func / smart_greedy_val_func (/root/reference/go_ai/policies/baselines.py:11-55) on device tensors
    def _greedy_values(self, states):
        b, w, ended, _ = self.rules.score(states)
        d = (b - w).to(torch.float64)
        return torch.where(ended > 0, 100.0 * torch.sign(d), d / self.P)

    def _smart_greedy_values(self, states):
        b, w, ended, _ = self.rules.score(states)
        bl, wl = self.rules.liberties(states)
        area = (b - w).to(torch.float64) / self.P
        libs = (bl - wl).to(torch.float64) / self.P
        return torch.where(ended > 0, torch.sign(area), (6 * area + libs) / 7)

    def record(self, on=True):
        """Keep every move so that play() also returns the reference's Trajectory objects."""
        self._record = bool(on)

    def _pi(self, pol, states):
        """pi of one policy for `states` (k canonical roots on the GPU), as a device tensor [k, A]."""
        c = pol["ctx"]
        k = states.shape[0]
        c.mcts_set_active(k)
        c.mcts_reset(states)
        if pol["kind"] == "random":
            return c.uniform_policy(allow_pass=True)
        policy = pol["policy"]
        if pol["kind"] == "valfunc":
            # mct_step with a value function that is not a network (mct.py:27-30,43-47): the kernels hand out the
            # leaf and its children, the function values them (here: area / liberty counts by the rule kernels)
            for _ in range(max(policy.mcts, 0) + 1):
                leaves = c.mcts_select()
                kids, _ = c.mcts_expand()
                c.mcts_backup_q(pol["values"](leaves), pol["values"](kids.reshape(k * self.A, 6, self.size, self.size)).reshape(k, self.A))
            return c.mcts_policy_value(policy.temp, from_values=(policy.mcts == 0))
        c.mcts_run(max(policy.mcts, 0))
        if pol["kind"] == "ac":
            return c.mcts_policy(policy.temp)
        return c.mcts_policy_value(policy.temp, from_values=(policy.mcts == 0))

    def play(self, episodes):
        """Returns (first_wins / episodes, black_wins / episodes, replay, all_steps) like play_games
        (game.py:154); replay is a list of Trajectory (empty unless record() was called), all_steps the
        number of moves of every episode, both in episode order.  Also sets self.results:
        per episode (black_won, first_won, steps)."""
        S, P, dev = self.size, self.P, self.device
        n = min(self.slots, episodes)
        max_steps = 2 * P                                            # game.py:74
        states = torch.zeros((n, 6, S, S), dtype=torch.uint8, device=dev)
        turn = torch.zeros(n, dtype=torch.int64, device=dev)         # 0 black to move
        moves = torch.zeros(n, dtype=torch.int64, device=dev)
        episode = torch.arange(1, n + 1, dtype=torch.int64, device=dev)     # 1-based, as the reference counts
        active = torch.ones(n, dtype=torch.bool, device=dev)
        next_episode = n + 1
        black_won = np.zeros(episodes + 1, dtype=np.int64)
        steps = np.zeros(episodes + 1, dtype=np.int64)
        trajs = {i: Trajectory() for i in range(1, episodes + 1)} if self._record else None
        self.leaf_evals = 0
        while bool(active.any()):
            idx = active.nonzero().flatten()
            # the first policy is black in even episodes (game.py:141-146); who moves now?
            first_is_black = (episode[idx] % 2 == 0)
            mover = torch.where(first_is_black == (turn[idx] == 0), 0, 1)
            actions = torch.empty(len(idx), dtype=torch.int32, device=dev)
            pis = torch.empty((len(idx), self.A), dtype=torch.float64, device=dev) if self._record else None
            for which in (0, 1):
                sel = (mover == which).nonzero().flatten()
                if len(sel) == 0:
                    continue
                sub = idx[sel]
                pol = self.policies[which]
                pi = self._pi(pol, states[sub].contiguous())
                acts = pol["ctx"].sample_actions_keyed(pi, self.seed, episode[sub] + self.episode_offset, moves[sub])
                actions[sel] = acts
                if pis is not None:
                    pis[sel] = pi
                if pol["kind"] != "random":
                    self.leaf_evals += len(sel) * (max(pol["policy"].mcts, 0) + 1)
            cur = states[idx].contiguous()
            nxt, status = self.rules.transition(cur, actions, canonical=True)
            assert bool((status == 0).all()), "a policy chose an illegal move"
            if self._record:
                kids = self.rules.children(cur, canonical=True)[0].cpu().numpy().astype(np.float64)
                cur_h, pis_h, acts_h, turn_h, ep_h = cur.cpu().numpy(), pis.cpu().numpy(), actions.cpu().numpy(), \
                    turn[idx].cpu().numpy(), episode[idx].cpu().numpy()
                for j in range(len(idx)):
                    t = trajs[int(ep_h[j])]
                    t.add_event(cur_h[j].astype(np.float64), int(acts_h[j]), 0.0, kids[j], pis_h[j])
                    if t.movers is None:
                        t.movers = []
                    t.movers.append(int(turn_h[j]))
            states[idx] = nxt
            turn[idx] ^= 1
            moves[idx] += 1
            over = (nxt[:, 5, 0, 0] != 0) | (moves[idx] > max_steps)          # two passes, or the step cap (game.py:97-99)
            if bool(over.any()):
                fin = idx[over]
                b, w, _, _ = self.rules.score(states[fin].contiguous())
                sign = torch.sign(b - w).to(torch.int64)                       # canonical frame: plane 0 = the side to move
                res = torch.where(turn[fin] == 1, -sign, sign).cpu().numpy()   # black's result (go_env.winning(), game.py:113)
                ep_h, mv_h = episode[fin].cpu().numpy(), moves[fin].cpu().numpy()
                for j, e in enumerate(ep_h):
                    black_won[e], steps[e] = res[j], mv_h[j]
                    if trajs is not None:
                        trajs[int(e)].set_win(int(res[j]))
                # the slot takes the next episode, if any is left
                for slot in fin.tolist():
                    if next_episode <= episodes:
                        states[slot] = 0
                        turn[slot] = 0
                        moves[slot] = 0
                        episode[slot] = next_episode
                        next_episode += 1
                    else:
                        active[slot] = False
        eps = np.arange(1, episodes + 1)
        first_won = np.where(eps % 2 == 0, black_won[1:], -black_won[1:])
        self.results = dict(black_won=black_won[1:].copy(), first_won=first_won, steps=steps[1:].copy())
        replay = [trajs[i] for i in range(1, episodes + 1)] if trajs is not None else []
        return float((first_won == 1).sum()) / episodes, float((black_won[1:] == 1).sum()) / episodes, replay, \
            [int(x) for x in steps[1:]]


from .sharding import STAT_FIELDS, gather_stats, reduce_stats, shard_games  # noqa: E402,F401


def trajectory_events(traj):
    """`Trajectory.get_events()`: `won` is from the mover's point of view, the mover taken from the
    recorded side to move (`traj.movers`) where the recorder knows it -- batched games may start with
    white to move -- else from the event index as in the reference (game.py:15-24)."""
    return traj.get_events()
