"""The mask / children half of /root/reference/go_ai/data.py:15-60 on the HIP Go-rule kernels.

Same names, argument meaning and return shapes as the reference; the replay / pickle half of
that file (data.py:85-182) is storage and out of this build's scope.
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
