"""Call-sequence and argument errors of the batched search entry points (the reference asserts;
the C ABI returns GOAI_ERR_* with a message), and behaviour on finished games."""
import numpy as np
import pytest

from oracle import nn_ref

pytestmark = pytest.mark.gpu


def test_search_call_sequence_errors():
    from go_ai import _hip
    from go_ai.engine import Context
    c = Context(5)
    with pytest.raises(_hip.GoaiError, match="nn_init"):
        c.nn_conv_algo()
    with pytest.raises(_hip.GoaiError, match="nn_init"):
        c.nn_convs_per_launch()
    with pytest.raises(_hip.GoaiError, match="mcts_init"):
        c.mcts_reset(np.zeros((1, 6, 5, 5), np.uint8))
    with pytest.raises(_hip.GoaiError, match="max_sims"):
        c.mcts_init(2, 2049)             # GOAI_MAX_SIMS = 2048
    c.mcts_init(2, 3)
    with pytest.raises(_hip.GoaiError, match="already"):
        c.mcts_init(2, 3)
    with pytest.raises(_hip.GoaiError, match="mcts_reset"):
        c.mcts_select()
    c.mcts_reset(np.zeros((2, 6, 5, 5), np.uint8))
    with pytest.raises(_hip.GoaiError, match="without a preceding"):
        c.mcts_backup(np.zeros(2, np.float32), pi_logits=np.zeros((2, 26), np.float32))
    c.mcts_select()
    with pytest.raises(_hip.GoaiError, match="twice"):
        c.mcts_select()
    with pytest.raises(_hip.GoaiError, match="exactly one"):
        _hip.check(_hip.lib.goai_mcts_backup(c._h, None, None, _hip.ptr(c.mcts_readout()[2].float()), 0))
    with pytest.raises(_hip.GoaiError, match="GOAI_SEARCH_Q"):
        c.mcts_expand()
    c.mcts_backup(np.zeros(2, np.float32), pi_logits=np.zeros((2, 26), np.float32))
    for _ in range(3):
        c.mcts_select()
        c.mcts_backup(np.zeros(2, np.float32), pi_logits=np.zeros((2, 26), np.float32))
    with pytest.raises(_hip.GoaiError, match="max_sims"):
        c.mcts_select()                      # 1 + max_sims steps were done
    with pytest.raises(_hip.GoaiError, match="network"):
        c.mcts_reset(np.zeros((2, 6, 5, 5), np.uint8))
        c.mcts_run(3)                        # no weights loaded
    assert c.mcts_counters()["errors"] == 0
    c.close()

    q = Context(5)
    q.mcts_init(1, 2, mode=Context.SEARCH_Q)
    q.mcts_reset(np.zeros((1, 6, 5, 5), np.uint8))
    with pytest.raises(_hip.GoaiError, match="goai_mcts_backup_q"):
        q.mcts_select()
        q.mcts_backup(np.zeros(1, np.float32), pi_logits=np.zeros((1, 26), np.float32))
    with pytest.raises(_hip.GoaiError, match="must follow"):
        q.mcts_backup_q(np.zeros(1), np.zeros((1, 26)))      # expand was skipped
    q.mcts_expand()
    with pytest.raises(_hip.GoaiError, match="must follow"):
        q.mcts_expand()
    q.mcts_backup_q(np.zeros(1), np.zeros((1, 26)))
    q.close()


def test_finished_games_stay_finished_and_are_not_counted():
    """Two passes end a game (game.py:98, test_policy.py:16-21); a finished game's root is terminal,
    every further step re-evaluates it (mct.py:12,36) without growing the tree or the leaf counter."""
    from go_ai.engine import Context
    size, n = 5, 6
    c = Context(size)
    c.nn_init(Context.NET_AC, 3, n)
    c.load_weights(nn_ref.make_state_dict("ac", size, 2))
    c.mcts_init(n, 4)
    c.mcts_reset(np.zeros((n, 6, size, size), np.uint8))
    passes = np.full(n, size * size, np.int32)
    passes[3:] = 7                                     # games 3.. play a stone instead
    c.mcts_advance(passes)
    passes[3:] = 8
    c.mcts_advance(passes)
    _, turn, moves, result = [t.cpu().numpy() for t in c.mcts_roots()]
    assert (result[:3] == 0).all() and (result[3:] == 2).all() and (moves == 2).all()
    before = c.mcts_counters()["leaf_evals"]
    c.mcts_run(4)
    assert c.mcts_counters()["leaf_evals"] - before == 3 * 5        # only the running games count
    visits, _, root_visits = [t.cpu().numpy() for t in c.mcts_readout()]
    assert (visits[:3] == 0).all() and (root_visits[:3] == 5).all()
    assert (visits[3:].sum(axis=1) == 4).all()
    pi = c.mcts_policy(1.0).cpu().numpy()
    assert np.allclose(pi.sum(axis=1), 1.0)
    c.mcts_advance(c.sample_actions(c.mcts_policy(1.0), 1, 0))
    _, _, moves2, result2 = [t.cpu().numpy() for t in c.mcts_roots()]
    assert (moves2[:3] == 2).all() and (result2[:3] == 0).all() and (moves2[3:] == 3).all()
    assert c.mcts_counters()["errors"] == 0
    c.close()


def test_failed_init_can_be_retried():
    """An init that runs out of device memory frees what it had allocated and leaves the context as it was: the same
    call with a size that fits then succeeds, and the context works."""
    from go_ai import _hip
    from go_ai.engine import Context
    c = Context(19)
    with pytest.raises(_hip.GoaiError):
        c.mcts_init(1 << 24, 128)                        # ~2^24 games x 130 nodes x 362 actions: terabytes
    with pytest.raises(_hip.GoaiError):
        c.nn_init(Context.NET_AC, 3, 1 << 22)            # 3 activation buffers of 2^22 x 361 x 128 floats
    with pytest.raises(_hip.GoaiError):
        c.train_init(Context.NET_AC, 64, 40000)          # 260 activation buffers of 40000 x 361 x 128 floats
    c.nn_init(Context.NET_AC, 1, 4)
    c.load_weights(nn_ref.make_state_dict("ac", 19, 1, layers=1))
    c.mcts_init(4, 3)
    c.train_init(Context.NET_AC, 1, 4)
    c.mcts_reset(np.zeros((4, 6, 19, 19), np.uint8))
    c.mcts_run(3)
    assert int(c.mcts_readout()[0].sum()) == 4 * 3 and c.mcts_counters()["errors"] == 0
    c.close()


def test_game_stats_count_the_active_slots_only():
    """goai_mcts_game_stats / goai_gather_stats are totals 'since the last goai_mcts_reset' -- and that reset clears the
    ACTIVE slots only.  Games finished while all slots were active must not show up after goai_mcts_set_active shrank
    the set and the roots were reset."""
    from go_ai.engine import Context
    from helpers import from_ascii
    size, n = 5, 8
    root = from_ascii(["XX...", ".X...", "XX...", ".....", "....."], to_move=1, prev_pass=True).astype(np.uint8)
    c = Context(size)
    c.mcts_init(n, 2)
    c.mcts_reset(np.repeat(root[None], n, 0))
    c.mcts_advance(np.full(n, size * size, np.int32))       # second pass in a row: every game ends
    assert c.mcts_game_stats()["finished"] == n
    assert c.gather_stats(None)[0]["finished"] == n
    c.mcts_set_active(3)
    c.mcts_reset(np.repeat(root[None], 3, 0))
    st = c.mcts_game_stats()
    assert st == dict(finished=0, black_wins=0, white_wins=0, finished_moves=0), st
    assert c.gather_stats(None)[0]["finished"] == 0
    c.mcts_advance(np.full(3, size * size, np.int32))
    assert c.mcts_game_stats()["finished"] == 3
    # a half-done step does not survive a change of the active set
    c.mcts_set_active(2)
    c.mcts_reset(np.zeros((2, 6, size, size), np.uint8))
    c.mcts_select()
    c.mcts_set_active(2)
    c.mcts_reset(np.zeros((2, 6, size, size), np.uint8))
    c.mcts_select()                                         # would be "called twice" if the flag had survived
    c.close()


def test_closed_context_raises_clearly():
    """Context caches (search/mct.py, RLNet.search_context) evict by close(); a tree view or a policy that still holds
    the context must get a clear error, not a call on a dead handle."""
    from go_ai.engine import Context
    from go_ai.search.tree import RootView
    c = Context(5)
    c.mcts_init(1, 2)
    c.mcts_reset(np.zeros((1, 6, 5, 5), np.uint8))
    for _ in range(3):
        c.mcts_select()
        c.mcts_backup(np.zeros(1, np.float32), pi_logits=np.zeros((1, 26), np.float32))
    visits, prior, root_visits = c.mcts_readout()
    view = RootView(np.zeros((6, 5, 5)), visits[0].cpu().numpy(), prior[0].cpu().numpy(), int(root_visits[0]))
    view.attach(c, 0)
    c.close()
    assert c.closed
    with pytest.raises(RuntimeError, match="closed"):
        view.child_nodes
    with pytest.raises(RuntimeError, match="closed"):
        c.mcts_readout()
    c.close()                                               # closing twice is fine
