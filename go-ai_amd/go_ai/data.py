"""The mask / children half of /root/reference/go_ai/data.py:15-60 on the HIP Go-rule kernels.

Same names, argument meaning and return shapes as the reference.  Of the replay half
(data.py:63-105) the in-memory conversions are here (symmetries, state+action planes, replay ->
events -> numpy batches); the pickle files and the MPI sampling (data.py:108-182) are storage and
out of this build's scope.
"""
import numpy as np

from .env import GoGame, GoVars, context  # noqa: F401  (GoGame / GoVars are re-exported like the reference's)


def batch_invalid_moves(states):
    """1 where a move is invalid, 0 where it is valid; the pass is always valid (data.py:15-24)."""
    states = np.asarray(states)
    assert len(states.shape) == 4
    batchsize, board_size = states.shape[0], states.shape[2]
    invalid_moves = states[:, GoVars.INVD_CHNL].reshape((batchsize, -1))
    return np.insert(invalid_moves, board_size ** 2, 0, axis=1)


def batch_valid_moves(states):
    return 1 - batch_invalid_moves(states)


def batch_invalid_values(states):
    """np.finfo(float32).min at invalid moves, (-)0 at valid ones (data.py:31-38)."""
    return np.finfo(np.float32).min * batch_invalid_moves(states)


def batch_padded_children(states):
    """GoGame.children(state, canonical=True, padded=True) for every state (data.py:54-60): one
    launch of the all-actions transition kernel instead of B x A Python calls."""
    states = np.asarray(states)
    kids, _, _ = context(states.shape[-1]).children(states, canonical=True)
    return list(kids.cpu().numpy().astype(np.float64))


def batch_win_children(batch_children):
    """winning(child) where the child ended the game, else 0 (data.py:41-51), by the scoring
    kernel; zero-padded (invalid) children give 0."""
    kids = np.asarray(batch_children)
    b, a = kids.shape[:2]
    _, _, ended, winner = context(kids.shape[-1]).score(kids.reshape((b * a,) + kids.shape[2:]))
    win = np.where(ended.cpu().numpy() > 0, winner.cpu().numpy(), 0)
    return win.reshape(b, a)


def batch_random_symmetries(states):
    """A random one of the 8 board symmetries per state (data.py:63-68)."""
    states = np.asarray(states)
    assert len(states.shape) == 4
    return np.array([GoGame.random_symmetry(state) for state in states])


def batch_combine_state_actions(states, actions):
    """[B, 7, s, s]: the state planes plus a one-hot plane of the action (all zero for the pass),
    data.py:71-82."""
    states = np.asarray(states)
    new_shape = np.array(states.shape)
    new_shape[1] = GoVars.NUM_CHNLS + 1
    size = new_shape[-1]
    state_actions = np.zeros(new_shape)
    state_actions[:, :-1] = states
    for i, a in enumerate(actions):
        if a < size ** 2:
            state_actions[i, -1, a // size, a % size] = 1
    return state_actions


def replay_to_events(replay):
    """Flatten trajectories (go_ai.game.Trajectory, e.g. BatchedSelfPlay.trajectories()) into the
    event tuples (state, action, reward, children, terminal, won, pi) of data.py:85-89."""
    events = []
    for traj in replay:
        events.extend(traj.get_events())
    return events


def events_to_numpy(events):
    """Column arrays with the reference's dtypes (data.py:92-105)."""
    if len(events) == 0:
        return [], [], [], [], [], []
    cols = list(zip(*events))
    states = np.array(list(cols[0]), dtype=np.float32)
    actions = np.array(list(cols[1]), dtype=int)
    rewards = np.array(list(cols[2]), dtype=np.float32).reshape((-1,))
    next_states = np.array(list(cols[3]), dtype=np.float32)
    terminals = np.array(list(cols[4]), dtype=np.uint8)
    wins = np.array(list(cols[5]), dtype=int)
    pis = np.array(list(cols[6]), dtype=np.float32)
    return states, actions, rewards, next_states, terminals, wins, pis
