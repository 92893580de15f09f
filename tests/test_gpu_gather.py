"""goai_gather_stats: the end-of-run sums over ranks (/root/reference/go_ai/utils.py:201-204) as an RCCL all-reduce
through the C-ABI.  One GPU here, so the communicator has one rank (RCCL does not allow two ranks on one device):
the collective really runs, and its result must equal the context's own statistics."""
import numpy as np
import pytest

from oracle import nn_ref

pytestmark = pytest.mark.gpu


def test_gather_stats_one_rank_communicator():
    from go_ai import _hip
    from go_ai.engine import Context
    size, games, sims = 5, 16, 4
    c = Context(size, 0)
    c.nn_init(Context.NET_AC, 1, games)
    c.load_weights(nn_ref.make_state_dict("ac", size, 2, layers=1))
    c.mcts_init(games, sims)
    with pytest.raises(_hip.GoaiError, match="mcts_reset"):
        c.gather_stats()
    c.mcts_set_refill(True)
    c.mcts_reset(np.zeros((games, 6, size, size), np.uint8))
    for move in range(40):
        c.mcts_run(sims)
        pi = c.mcts_policy(1.0)
        c.mcts_advance(c.sample_actions(pi, seed=5, step=move))
    stats, counters = c.mcts_game_stats(), c.mcts_counters()
    assert stats["finished"] > 0
    local, extra = c.gather_stats()
    want = {**stats, **counters}
    assert local == want and extra == []

    comm = c.comm_init(Context.comm_unique_id(), 1, 0)
    summed, extra = c.gather_stats(comm, extra=[1.5, -2.0, 3e9])
    assert summed == want and extra == [1.5, -2.0, 3e9]
    summed2, _ = c.gather_stats(comm)                      # reusable; no extras
    assert summed2 == want
    with pytest.raises(_hip.GoaiError):
        c.gather_stats(comm, extra=[0.0] * 65)
    c.comm_destroy(comm)
    c.close()
