"""Host logic of the replay format (CPU): `won` signs of events, the disk replay and its balanced
sampling (/root/reference/go_ai/game.py:15-52, /root/reference/go_ai/data.py:85-182)."""
import types

import numpy as np


def _traj(n, winner, movers=None, size=5):
    from go_ai.game import Trajectory
    t = Trajectory()
    A = size * size + 1
    for i in range(n):
        t.add_event(np.full((6, size, size), i, dtype=np.float64), i % A, 0.0, np.zeros((A, 6, size, size)), np.full(A, 1.0 / A))
    t.set_win(winner)
    t.movers = movers
    return t


def test_won_sign_follows_the_recorded_mover():
    from go_ai import data
    # the reference's assumption: black moves first, colours alternate (game.py:15-24)
    ev = _traj(4, 1).get_events()
    assert [e[5] for e in ev] == [1, -1, 1, -1] and [e[4] for e in ev] == [False, False, False, True]
    # a batched game that starts with WHITE to move: index parity would give every sign wrong
    t = _traj(3, -1, movers=[1, 0, 1])
    assert [e[5] for e in t.get_events()] == [1, -1, 1]
    assert [e[5] for e in data.replay_to_events([t, _traj(2, 1, movers=[0, 1])])] == [1, -1, 1, 1, -1]


def test_disk_replay_append_reset_and_balanced_sampling(tmp_path):
    import random
    from go_ai import data
    args = types.SimpleNamespace(replay_path=str(tmp_path / "replay.pkl"), replaysize=6)
    data.reset_replay(args)
    assert len(data.load_replay(args.replay_path)) == 0
    data.mpi_disk_append_replay(data.SoloComm(), args, [_traj(4, 1), _traj(5, -1), _traj(3, 0)])
    data.mpi_disk_append_replay(None, args, [_traj(6, 1)])
    replay = data.load_replay(args.replay_path)
    assert [len(t) for t in replay] == [4, 5, 3, 6] and replay.maxlen == 6
    # black wins: 10 events, non-wins: 8 events -> 8 of each at most; asked for 2 batches x 4 -> 4 + 4
    random.seed(0)
    batches, n_games = data.mpi_sample_eventdata(data.SoloComm(), args.replay_path, batches=2, batchsize=4)
    assert n_games == 4 and len(batches) == 2
    states, actions, rewards, children, terminals, wins, pis = batches[0]
    assert states.shape == (4, 6, 5, 5) and states.dtype == np.float32 and children.shape == (4, 26, 6, 5, 5)
    assert actions.dtype.kind == "i" and terminals.dtype == np.uint8 and pis.shape == (4, 26)
    assert sum(len(b[0]) for b in batches) == 8
    # the deque drops the oldest games once more than `replaysize` are stored (data.py:180)
    data.mpi_disk_append_replay(None, args, [_traj(1, 1), _traj(1, 1), _traj(2, -1)])
    assert [len(t) for t in data.load_replay(args.replay_path)] == [5, 3, 6, 1, 1, 2]
    # without a file the list itself becomes the replay (data.py:169-170)
    args2 = types.SimpleNamespace(replay_path=str(tmp_path / "fresh.pkl"), replaysize=3)
    data.mpi_disk_append_replay(None, args2, [_traj(2, 1)])
    assert isinstance(data.load_replay(args2.replay_path), list)
