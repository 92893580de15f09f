"""The reference-shaped Python surface (go_ai.env / models / search / policies / game) on the GPU:
the reference's own KAT, golden policies, and a short drop-in game."""
import types

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, ReplayNet, load_mcts, load_nn
from oracle import nn_ref

pytestmark = pytest.mark.gpu


def test_reference_kat_greedy_and_env():
    # /root/reference/tests/test_policy.py:16-21,34-43 on the product's env + Value policy
    from go_ai.env import GoEnv
    from go_ai.policies import baselines
    env = GoEnv(5)
    env.reset()
    for a in (0, 1, 2, None):
        env.step(a)
    args = types.SimpleNamespace(size=5, model='greedy', temp=1, mcts=0)
    policy, _ = baselines.create_policy(args)
    pi = policy(env)
    assert pi[-1] == 1 and np.allclose(pi[:-1], 0)
    z = np.load(GOLDEN_DIR + "/policy_kat_greedy5.npz")
    assert np.array_equal(env.state.astype(np.uint8), z["state"])
    assert np.array_equal(pi, z["pi"])           # the reference's own policy output, bit for bit


def test_mct_search_with_foreign_callable_matches_golden():
    """mct_search driven by a host callable (here: the reference's recorded NN outputs)."""
    from go_ai.env import GoEnv
    from go_ai.search import mct
    size, sims, temp, cases = load_mcts("mcts_ac5_n40")
    for c in cases[:3]:
        env = GoEnv(size)
        env.state = c["root"].astype(np.float64)
        root = mct.mct_search(env, sims, actor_critic=ReplayNet(c))
        assert np.array_equal(root.get_visit_counts(), c["visits"])
        assert root.visits == c["root_visits"]


def test_tree_views_match_the_reference_node_surface():
    """rootnode.child_nodes / winning() / str(node) (tree.py:65-69,100-108,186-195), what search/plot.py:39-61 and the
    policies' debug return walk: the whole GPU tree exported once and compared node by node with the sequential oracle's
    tree built from the same recorded network outputs (actor-critic and Q-value searches)."""
    from golden_util import ReplayCritic
    from go_ai.env import GoEnv
    from go_ai.search import mct
    from oracle import mcts_ref
    from oracle.gogame import GoEnv as OEnv, GoGame as OG

    def same(view, ref, depth=0):
        assert view.visits == ref.visits and view.level == ref.level == depth
        assert np.array_equal(view.state, ref.state)
        assert view.terminal() == ref.terminal() and view.isroot() == (ref.parent is None)
        if ref.prior_pi is None:
            assert view.prior_pi is None
        else:
            assert np.allclose(view.prior_pi, ref.prior_pi, rtol=2e-6, atol=1e-30)   # fp32 det-softmax vs scipy's
        assert (view.val is None) == (ref.val is None) and (ref.val is None or view.val == ref.val)
        assert np.array_equal(view.get_visit_counts(), [0 if c is None else c.visits for c in ref.child_nodes])
        assert len(view.get_child_nodes()) == len([c for c in ref.child_nodes if c is not None])
        want = (f'{ref.val:.2f}V' if ref.val is not None else '') + \
               (f' {np.mean(ref.post_vals):.2f}AV' if len(ref.post_vals) > 0 else '') + f' {ref.level}L {ref.visits}N'
        assert str(view) == want, (str(view), want)                                   # tree.py:186-195
        n = 1
        for a, (cv, cr) in enumerate(zip(view.child_nodes, ref.child_nodes)):
            assert (cv is None) == (cr is None), (depth, a)
            if cv is not None:
                assert cv.parent is view and cv.first_action == (a if depth == 0 else view.first_action)
                n += same(cv, cr, depth + 1)
        return n

    size, sims, temp, cases = load_mcts("mcts_ac5_n40")
    for c in cases[:2]:
        env = GoEnv(size)
        env.state = c["root"].astype(np.float64)
        root = mct.mct_search(env, sims, actor_critic=ReplayNet(c))
        oenv = OEnv(size)
        oenv.state = c["root"].astype(np.float64)
        _, _, oroot = mcts_ref.actor_critic_policy(oenv, ReplayNet(c), sims, temp, math="numpy")
        def count(n):
            return 1 + sum(count(k) for k in n.child_nodes if k is not None)
        assert same(root, oroot) == count(oroot) <= sims + 1      # at most one new node per search step
        assert root.winning() == OG.winning(c["root"].astype(np.float64))
        assert not root.isleaf() and root.actionsize() == size * size + 1
    size, sims, temp, cases = load_mcts("mctsq_val5_n12")
    c = cases[0]
    env = GoEnv(size)
    env.state = c["root"].astype(np.float64)
    root = mct.mct_search(env, sims, critic=ReplayCritic(c))
    oenv = OEnv(size)
    oenv.state = c["root"].astype(np.float64)
    _, oroot = mcts_ref.value_policy(oenv, ReplayCritic(c), sims, temp, math="numpy")
    assert same(root, oroot) > sims + 1               # every expansion made all valid children
    assert np.array_equal(root.inverted_children_values(), c["inv_child_vals"])
    # a later search on the same context overwrites the tree: a view that was not loaded says so
    stale = mct.mct_search(env, sims, critic=ReplayCritic(c))
    mct.mct_search(env, sims, critic=ReplayCritic(c))
    with pytest.raises(RuntimeError, match="overwritten"):
        stale.child_nodes


def test_mct_search_keeps_wrapped_gpu_nets_on_the_gpu():
    """mct_search recognises a go_ai.models net behind create_numpy's callable by the attribute the callable
    carries, also through functools.partial (no closure poking): the search then runs in the net's own context."""
    import functools
    from go_ai.env import GoEnv
    from go_ai.models.ac_net import ActorCriticNet
    from go_ai.search import mct
    net = ActorCriticNet(5, max_batch=32)
    f = net.create_numpy('actor_critic')
    assert mct._owning_net(f) is net and mct._owning_net(functools.partial(f)) is net
    assert mct._owning_net(lambda s: f(s)) is None
    env = GoEnv(5)
    env.reset()
    a = mct.mct_search(env, 10, actor_critic=f).get_visit_counts()
    b = mct.mct_search(env, 10, actor_critic=functools.partial(f)).get_visit_counts()
    c = mct.mct_search(env, 10, actor_critic=lambda s: f(s)).get_visit_counts()      # host callback path, same numbers
    assert np.array_equal(a, b) and np.array_equal(a, c) and a.sum() == 10
    # the net keeps tree storage for a few (games, searches, mode) combinations only
    for sims in range(1, 8):
        mct.mct_search(env, sims, actor_critic=f)
    assert len(net._search) <= net.MAX_SEARCH_CONTEXTS


def test_actorcritic_policy_on_gpu_net():
    from go_ai.env import GoEnv
    from go_ai.models.ac_net import ActorCriticNet
    from go_ai.policies.actorcritic import ActorCritic
    g = load_nn("nn_ac5")
    net = ActorCriticNet(5, max_batch=32)
    sd = nn_ref.make_state_dict("ac", 5, int(g["seed"]))
    net.load_state_dict(sd)
    assert set(net.state_dict()) == set(sd)
    # create_numpy keeps the reference's return conventions
    pi, val = net.create_numpy('actor_critic')(g["states"].astype(np.float64))
    assert pi.shape == (20, 26) and val.shape == (20, 1) and pi.dtype == np.float32
    assert np.allclose(val, g["val_np"], rtol=1e-3, atol=1e-3)
    assert net.create_numpy('critic')(g["states"]).shape == (20, 1)
    assert net.create_numpy('actor')(g["states"]).shape == (20, 26)
    policy = ActorCritic("t", net, types.SimpleNamespace(temp=1, mcts=12))
    env = GoEnv(5)
    env.reset()
    env.step(7)
    pi, qs, root = policy(env, debug=True)
    assert pi.shape == (26,) and abs(pi.sum() - 1) < 1e-12 and qs.shape == (2, 26)
    assert root.get_visit_counts().sum() == 12 and root.visits == 13
    assert pi[7] == 0                                   # occupied point
    assert str(policy).startswith("ActorCritic[12S 1.00T]")


def test_pit_plays_a_legal_game():
    from go_ai import game
    from go_ai.env import GoEnv
    from go_ai.policies.baselines import Random
    np.random.seed(0)
    env = GoEnv(5)
    black_won, steps, traj = game.pit(env, Random(), Random())
    assert black_won in (-1, 0, 1) and 2 <= steps <= 2 * 25 + 1 and len(traj) == steps
    ev = traj.get_events()
    assert ev[-1][4] is True and ev[0][3].shape == (26, 6, 5, 5)


def test_batched_selfplay_runs_games_to_the_end():
    from go_ai.game import BatchedSelfPlay
    sp = BatchedSelfPlay(5, 64, 6, seed=3)
    sp.randomize_starts(10)
    sp.play(2 * 25 + 2)
    st = sp.stats()
    assert st["finished"] == 64 and st["errors"] == 0
    assert st["black_wins"] + st["white_wins"] <= 64
    assert 0 < st["leaf_evals"] <= 64 * 7 * 52


def test_batched_selfplay_refills_finished_slots():
    """refill=True: a slot whose game ends starts the next one from the empty board (play_games' episode
    loop, game.py:139-145), so every slot searches at every move; totals count every finished game and
    do not depend on how the slots are sharded over contexts (Philox streams keyed by slot id + episode)."""
    from go_ai.game import BatchedSelfPlay
    n, sims, moves = 64, 6, 130
    runs = []
    for shards in (1, 1, 2):
        tot = None
        for k in range(shards):
            m = n // shards
            sp = BatchedSelfPlay(5, m, sims, seed=3, refill=True, game_id_offset=k * m)
            sp.play(moves)
            st = sp.stats()
            states, turn, mv, result = [t.cpu().numpy() for t in sp.ctx.mcts_roots()]
            assert (result == 2).all() and (mv <= 2 * 25).all()
            assert st["errors"] == 0 and st["leaf_evals"] == m * (sims + 1) * moves      # no idle slot, ever
            assert st["total_steps"] == m * moves
            tot = st if tot is None else {k2: tot[k2] + st[k2] for k2 in st}
        runs.append(tot)
    assert runs[0] == runs[1] == runs[2]
    assert runs[0]["finished"] > n and runs[0]["black_wins"] + runs[0]["white_wins"] <= runs[0]["finished"]
    # without refill the same seeds give the same first games, and finished slots idle
    sp = BatchedSelfPlay(5, n, sims, seed=3)
    sp.play(moves)
    st = sp.stats()
    assert st["finished"] == n and st["leaf_evals"] < runs[0]["leaf_evals"]


def test_batched_selfplay_19x19_runs_games_to_the_end():
    """BASELINE.json configs[4]'s board: k_mcts_select / backup / policy / sample / advance at S = 19
    (362 actions, six actions per lane) over whole games, to the 2 * 361 move cap (game.py:74,98); the
    final result of every game is the area score of its final position on the CPU oracle's rules."""
    from go_ai.game import BatchedSelfPlay
    from oracle.gogame import GoGame as OG
    n = 6
    sp = BatchedSelfPlay(19, n, 2, seed=5)
    sp.randomize_starts(250)
    sp.play(2 * 361 + 2)
    st = sp.stats()
    assert st["finished"] == n and st["errors"] == 0
    states, turn, moves, result = [t.cpu().numpy() for t in sp.ctx.mcts_roots()]
    assert (moves <= 2 * 361 + 1).all() and (moves > 0).all()
    for g in range(n):
        b, w = OG.areas(states[g].astype(np.float64))          # canonical: plane 0 = the side to move
        sign = int(np.sign(b - w))
        assert result[g] == (-sign if turn[g] else sign)
        assert OG.game_ended(states[g].astype(np.float64)) or moves[g] == 2 * 361 + 1


def test_value_policy_q_search_matches_golden():
    """policies.Value with mcts > 0 driven by a host critic (the reference's recorded outputs)."""
    from golden_util import ReplayCritic
    from go_ai.env import GoEnv
    from go_ai.policies.value import Value
    size, sims, temp, cases = load_mcts("mctsq_val5_n12")
    for c in cases[:3]:
        env = GoEnv(size)
        env.state = c["root"].astype(np.float64)
        policy = Value("golden", ReplayCritic(c), types.SimpleNamespace(temp=temp, mcts=sims))
        pi, qs, root = policy(env, debug=True)
        assert np.array_equal(root.get_visit_counts(), c["visits"]) and root.visits == c["root_visits"]
        assert np.array_equal(qs[0], c["inv_child_vals"])
        assert np.array_equal(pi, c["pi"])


def test_value_policy_on_gpu_net():
    from go_ai.env import GoEnv
    from go_ai.models.val_net import ValueNet
    from go_ai.policies.value import Value
    net = ValueNet(5, max_batch=64)
    policy = Value("v", net, types.SimpleNamespace(temp=1, mcts=6))
    env = GoEnv(5)
    env.reset()
    env.step(12)
    pi, qs, root = policy(env, debug=True)
    assert pi.shape == (26,) and abs(pi.sum() - 1) < 1e-12 and pi[12] == 0
    assert root.get_visit_counts().sum() == 6 and root.visits == 7
    assert str(policy).startswith("Value[6S 1.00T]")


def test_data_batch_helpers_match_oracle():
    """go_ai.data.batch_* (/root/reference/go_ai/data.py:15-60) against the CPU oracle's rules."""
    from helpers import random_positions
    from go_ai import data
    from oracle.gogame import GoGame as OG
    states = random_positions(7, 12, seed=5, canonical=True, pass_prob=0.2).astype(np.float64)
    states[0] = OG.next_state(OG.next_state(states[0], 49, canonical=True), 49, canonical=True)   # a finished game
    inv = data.batch_invalid_moves(states)
    assert inv.shape == (12, 50) and (inv[:, -1] == 0).all()
    assert np.array_equal(data.batch_valid_moves(states), np.stack([OG.valid_moves(s) for s in states]))
    vals = data.batch_invalid_values(states)
    assert (vals[inv == 1] == np.finfo(np.float32).min).all() and (vals[inv == 0] == 0).all()
    kids = data.batch_padded_children(states)
    want = [np.asarray(OG.children(s, canonical=True, padded=True)) for s in states]
    assert all(np.array_equal(k, w) for k, w in zip(kids, want))
    win = data.batch_win_children(kids)
    want_win = np.array([[OG.winning(c) if OG.game_ended(c) else 0 for c in ch] for ch in want])
    assert np.array_equal(win, want_win)


def test_liberties_and_smart_greedy_match_oracle():
    """GoGame.get_num_liberties + baselines.smart_greedy_val_func (baselines.py:33-55) vs the oracle."""
    from helpers import random_positions
    from go_ai.env import context
    from go_ai.policies.baselines import greedy_val_func, smart_greedy_val_func
    from oracle.gogame import GoGame as OG
    for size in (5, 9, 19):
        states = random_positions(size, 40, seed=30 + size, max_moves=2 * size * size, canonical=False, pass_prob=0.08)
        bl, wl = [t.cpu().numpy() for t in context(size).liberties(states)]
        want = np.array([OG.get_num_liberties(s) for s in states])
        assert np.array_equal(bl, want[:, 0]) and np.array_equal(wl, want[:, 1])
        area = size * size
        wg, ws = [], []
        for s, (b_l, w_l) in zip(states, want):
            b, w = OG.areas(s)
            if OG.game_ended(s):
                wg.append(100 * np.sign(b - w)); ws.append(np.sign(b - w))
            else:
                wg.append((b - w) / area); ws.append((6 * ((b - w) / area) + (b_l - w_l) / area) / 7)
        assert np.array_equal(greedy_val_func(states)[:, 0], np.array(wg, dtype=float))
        assert np.array_equal(smart_greedy_val_func(states)[:, 0], np.array(ws, dtype=float))


def test_batched_selfplay_trajectories_replay_on_the_oracle():
    """BatchedSelfPlay.record()/trajectories(): the recorded (state, action, children) chains are a
    legal game on the CPU oracle's rules, in the reference's replay layout (game.py:6-52)."""
    from go_ai.game import BatchedSelfPlay, trajectory_events
    from oracle.gogame import GoGame as OG
    sp = BatchedSelfPlay(5, 12, 4, seed=11)
    sp.randomize_starts(6)
    sp.record()
    sp.play(8)
    trajs = sp.trajectories(children_batch=5)
    assert len(trajs) == 12 and all(len(t) >= 1 for t in trajs)
    for t in trajs:
        for i in range(len(t)):
            st, a = t.states[i], t.actions[i]
            assert OG.valid_moves(st)[a] == 1 and abs(t.pis[i].sum() - 1) < 1e-12
            want = np.asarray(OG.children(st, canonical=True, padded=True))
            assert np.array_equal(t.children[i], want)
            if i + 1 < len(t):
                assert np.array_equal(t.states[i + 1], t.children[i][a])      # game.py:108
                assert t.rewards[i] == 0
        ev = trajectory_events(t)
        assert ev[-1][4] is True and ev[0][3].shape == (26, 6, 5, 5)
        assert all(e[5] == (-t.get_winner() if m else t.get_winner()) for e, m in zip(ev, t.movers))


def test_stream_groups_do_not_change_recorded_trajectories():
    """groups=2 runs two game groups on their own HIP streams (contexts follow torch's current
    stream, engine.Context._bind); records and trajectories must equal the single-stream run."""
    from go_ai.game import BatchedSelfPlay
    runs = []
    for groups in (1, 2):
        sp = BatchedSelfPlay(5, 12, 4, seed=11, groups=groups)
        sp.randomize_starts(6)
        sp.record()
        sp.play(8)
        runs.append(sp.trajectories(children_batch=5))
    for a, b in zip(*runs):
        assert len(a) == len(b) and a.get_winner() == b.get_winner() and a.movers == b.movers
        assert a.actions == b.actions
        assert all(np.array_equal(x, y) for x, y in zip(a.states, b.states))
        assert all(np.array_equal(x, y) for x, y in zip(a.pis, b.pis))
        assert all(np.array_equal(x, y) for x, y in zip(a.children, b.children))
    # a context follows the stream it is called under
    import torch
    from go_ai.engine import Context
    c = Context(5)
    side = torch.cuda.Stream()
    st = np.zeros((3, 6, 5, 5), np.uint8)
    with torch.cuda.stream(side):
        out, status = c.transition(st, [0, 1, 25])
        assert c._stream.cuda_stream == side.cuda_stream
    out2, status2 = c.transition(st, [0, 1, 25])
    assert c._stream.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert torch.equal(out, out2) and torch.equal(status, status2)
    c.close()


def test_q_helpers_and_replay_conversions_match_oracle():
    """search.qs_from_valfunc / batchqs_from_valfunc / batch_greedy_pi / batch_temperate_pi
    (/root/reference/go_ai/search/__init__.py:13-61,90-97) and data.replay_to_events /
    events_to_numpy / batch_combine_state_actions / batch_random_symmetries (data.py:63-105)."""
    from helpers import random_positions
    from go_ai import data, search
    from go_ai.env import GoGame
    from go_ai.game import BatchedSelfPlay
    from go_ai.policies.baselines import greedy_val_func
    from oracle.gogame import GoGame as OG
    states = random_positions(7, 6, seed=9, canonical=True, pass_prob=0.1).astype(np.float64)
    bq, bkids = search.batchqs_from_valfunc(states, greedy_val_func)
    assert bq.shape == (6, 50)
    for st, q, kids in zip(states, bq, bkids):
        valid = OG.valid_moves(st)
        want_kids = OG.children(st, canonical=True, padded=False)
        assert len(kids) == int(valid.sum()) == len(want_kids)
        assert all(np.array_equal(k, w) for k, w in zip(kids, want_kids))
        want_q = np.zeros(50)
        want_q[np.where(valid)] = [-((b - w) / 49 if not OG.game_ended(c) else 100 * np.sign(b - w))
                                   for c in want_kids for b, w in [OG.areas(c)]]
        assert np.array_equal(q, want_q)
        q1, kids1 = search.qs_from_valfunc(st, greedy_val_func, None)      # group_map by position, as the reference
        assert np.array_equal(q1, q) and len(kids1) == len(kids)
        assert np.array_equal(search.batch_greedy_pi(q[None], valid[None])[0], search.greedy_pi(q, valid))
    valids = np.stack([OG.valid_moves(s) for s in states])
    pi = search.batch_temperate_pi(bq, 0.5, valids)
    assert np.allclose(pi.sum(axis=1), 1) and np.array_equal(search.batch_temperate_pi(bq, 0, valids),
                                                             search.batch_greedy_pi(bq, valids))
    # symmetries: 8 images of the same stones
    syms = GoGame.get_symmetries(states[1])
    assert len(syms) == 8 and all(s.shape == states[1].shape and s.sum() == states[1].sum() for s in syms)
    assert np.array_equal(syms[0], states[1]) and np.array_equal(syms[3], states[1][:, ::-1, ::-1])
    assert data.batch_random_symmetries(states).shape == states.shape
    sa = data.batch_combine_state_actions(states, [0, 10, 49, 48, 7, 49])
    assert sa.shape == (6, 7, 7, 7) and sa[1, 6, 1, 3] == 1 and sa[2, 6].sum() == 0 and sa[:, 6].sum() == 4
    # replay -> events -> numpy columns
    sp = BatchedSelfPlay(5, 6, 2, seed=3)
    sp.record()
    sp.play(4)
    replay = sp.trajectories()
    events = data.replay_to_events(replay)
    assert len(events) == sum(len(t) for t in replay)
    cols = data.events_to_numpy(events)
    assert cols[0].shape == (len(events), 6, 5, 5) and cols[0].dtype == np.float32
    assert cols[3].shape == (len(events), 26, 6, 5, 5) and cols[4].dtype == np.uint8 and cols[6].shape == (len(events), 26)
    assert data.events_to_numpy([]) == ([], [], [], [], [], [])


def test_batched_play_two_policies_matches_sequential_oracle():
    """go_ai.game.BatchedPlay = play_games(go_env, first, second, episodes) (game.py:120-154) on the GPU: colours alternate
    by episode, results are accumulated as the reference does, and every recorded move is what the sequential oracle
    search computes for the policy that was to move, fed that policy's own network (pi bit for bit), with the action
    the Philox stream drew; winner and step count by the oracle's rules.  Results do not depend on the slot count."""
    import types
    from go_ai.game import BatchedPlay
    from go_ai.models.ac_net import ActorCriticNet
    from go_ai.models.val_net import ValueNet
    from go_ai.policies.actorcritic import ActorCritic
    from go_ai.policies.value import Value
    from oracle import mcts_ref
    from oracle.gogame import GoEnv as OEnv, GoGame as OG
    size, episodes = 5, 10
    net_a, net_b = ActorCriticNet(size, max_batch=64), ValueNet(size, max_batch=64)
    net_a.load_state_dict(nn_ref.make_state_dict("ac", size, 3))
    net_b.load_state_dict(nn_ref.make_state_dict("val", size, 4))
    first = ActorCritic("a", net_a, types.SimpleNamespace(temp=1, mcts=6))
    second = Value("b", net_b, types.SimpleNamespace(temp=1, mcts=3))
    runs = []
    for slots in (4, 16):
        bp = BatchedPlay(size, first, second, slots=slots, seed=7)
        bp.record()
        fw, bw, replay, steps = bp.play(episodes)
        runs.append((fw, bw, steps, bp.results, replay))
    fw, bw, steps, res, replay = runs[0]
    assert runs[1][0] == fw and runs[1][1] == bw and runs[1][2] == steps          # slot count does not matter
    assert all(np.array_equal(a.actions, b.actions) and all(np.array_equal(x, y) for x, y in zip(a.pis, b.pis))
               for a, b in zip(replay, runs[1][4]))
    assert len(replay) == episodes and [len(t) for t in replay] == steps
    assert fw == (res["first_won"] == 1).mean() and bw == (res["black_won"] == 1).mean()

    ac_fn, val_fn = net_a.create_numpy('actor_critic'), net_b.create_numpy('critic')
    for e, traj in enumerate(replay, start=1):
        first_is_black = e % 2 == 0                                              # game.py:141-146
        env = OEnv(size)
        env.reset()
        assert traj.movers == [i % 2 for i in range(len(traj))]
        for i in range(len(traj)):
            assert np.array_equal(traj.states[i], env.canonical_state())
            mover_is_first = (traj.movers[i] == 0) == first_is_black
            if mover_is_first:
                pi, _, _ = mcts_ref.actor_critic_policy(env, ac_fn, 6, 1.0, math="det")
            else:
                pi, _ = mcts_ref.value_policy(env, val_fn, 3, 1.0, math="det")
            assert np.array_equal(pi, traj.pis[i]), (e, i, np.abs(pi - traj.pis[i]).max())
            assert pi[traj.actions[i]] > 0
            assert np.array_equal(traj.children[i], np.asarray(OG.children(env.canonical_state(), canonical=True, padded=True)))
            _, _, done, _ = env.step(traj.actions[i])
            assert done == (i == len(traj) - 1) or len(traj) == 2 * size * size + 1
        assert env.winning() == res["black_won"][e - 1] == traj.get_winner()
        assert res["first_won"][e - 1] == (res["black_won"][e - 1] if first_is_black else -res["black_won"][e - 1])
        assert steps[e - 1] == len(traj) <= 2 * size * size + 1


def test_batched_play_against_random_and_greedy():
    """The evaluation opponents of /root/reference/train.py:13-20: baselines.Random and the greedy player
    (Value on baselines.greedy_val_func, mcts = 0) in batched play; the greedy player's pi equals the host policy's."""
    import types
    from go_ai.env import GoEnv
    from go_ai.game import BatchedPlay
    from go_ai.models.ac_net import ActorCriticNet
    from go_ai.policies import baselines
    from go_ai.policies.actorcritic import ActorCritic
    from go_ai.policies.value import Value
    size = 5
    net = ActorCriticNet(size, max_batch=64)
    ac = ActorCritic("a", net, types.SimpleNamespace(temp=1, mcts=4))
    greedy = Value('Greedy', baselines.greedy_val_func, types.SimpleNamespace(temp=1, mcts=0))
    bp = BatchedPlay(size, ac, baselines.Random(), slots=8, seed=1)
    fw, bw, _, steps = bp.play(12)
    assert 0 <= fw <= 1 and len(steps) == 12 and all(2 <= s <= 51 for s in steps)
    bp = BatchedPlay(size, greedy, baselines.Random(), slots=8, seed=2)
    bp.record()
    fw, bw, replay, steps = bp.play(6)
    assert fw >= 0.5                                   # greedy area-grabbing beats uniform random play
    for e, traj in enumerate(replay, start=1):
        greedy_is_black = e % 2 == 0
        for i in range(min(len(traj), 6)):
            if (traj.movers[i] == 0) == greedy_is_black:
                env = GoEnv(size)
                env.state = traj.states[i].copy()      # canonical: the mover is "black"
                assert np.allclose(greedy(env), traj.pis[i], rtol=1e-12, atol=0)


def test_mpi_play_and_count_parameters():
    """go_ai.utils.mpi_play (/root/reference/go_ai/utils.py:179-208) = this rank's share as one BatchedPlay run + the sums
    over the ranks; with one rank it is BatchedPlay itself."""
    import types
    from go_ai import data, utils
    from go_ai.env import GoEnv
    from go_ai.game import BatchedPlay
    from go_ai.models.ac_net import ActorCriticNet
    from go_ai.policies.actorcritic import ActorCritic
    from go_ai.policies.baselines import Random
    size = 5
    net = ActorCriticNet(size, max_batch=64)
    net.load_state_dict(nn_ref.make_state_dict("ac", size, 3))
    assert utils.count_parameters(net) == sum(v.numel() for k, v in net.state_dict().items() if not k.endswith(("running_mean", "running_var")))
    first, second = ActorCritic("a", net, types.SimpleNamespace(temp=1, mcts=4)), Random()
    wr, bwr, replay = utils.mpi_play(data.SoloComm(), GoEnv(size), first, second, 6, slots=4, seed=5)
    bp = BatchedPlay(size, first, second, slots=4, seed=5)
    bp.record()
    fw, bw, want_replay, steps = bp.play(6)
    assert (wr, bwr) == (fw, bw) and len(replay) == 6
    assert all(np.array_equal(a.actions, b.actions) for a, b in zip(replay, want_replay))

    class TwoRanks(data.SoloComm):              # rank 1 of 2: its three episodes are keyed 4, 5, 6 -- a different stream
        def Get_rank(self):
            return 1

        def Get_size(self):
            return 2

        def allreduce(self, x):
            return 2 * x
    wr2, _, replay2 = utils.mpi_play(TwoRanks(), GoEnv(size), first, second, 6, slots=4, seed=5, device=0)
    assert len(replay2) == 3 and 0 <= wr2 <= 1
    assert any(not np.array_equal(a.actions, b.actions) for a, b in zip(replay2, want_replay))
