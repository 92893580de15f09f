"""GymGo's surface as the reference uses it, served by the HIP Go-rule
kernels through the C ABI.  `GoGame.*` are the static functions the reference calls as
`gym.make('gym_go:go-v0', size=0).gogame.*`; `GoEnv` is the environment object of
/root/reference/go_ai/game.py:72-113.  Single states ride through the batched entry points
as batches of one; the batched search never goes through this module.
"""
import numpy as np

from .engine import Context

_ctx = {}


def context(size, device=0):
    key = (int(size), int(device))
    if key not in _ctx:
        _ctx[key] = Context(size, device)
    return _ctx[key]


class GoVars:
    BLACK, WHITE, TURN_CHNL, INVD_CHNL, PASS_CHNL, DONE_CHNL, NUM_CHNLS = 0, 1, 2, 3, 4, 5, 6


class GoGame:
    @staticmethod
    def get_init_board(size):
        return np.zeros((GoVars.NUM_CHNLS, size, size))

    @staticmethod
    def action_size(state=None, board_size=None):
        if state is not None:
            return int(state.shape[1] * state.shape[2] + 1)
        return int(board_size ** 2 + 1)

    @staticmethod
    def next_state(state, action1d, canonical=False):
        c = context(state.shape[-1])
        out, status = c.transition(np.asarray(state)[None], np.asarray([action1d], dtype=np.int32), canonical)
        assert status.item() == 0, ("Invalid move", action1d)
        return out[0].cpu().numpy().astype(np.float64)

    @staticmethod
    def children(state, group_map=None, canonical=False, padded=None):
        """All A next states.  The reference calls this two ways: keywords only, padded=True
        (tree.py:92, data.py:58) and with GymGo's group_map as second positional argument and no
        `padded`, expecting the valid children only, in ascending action order
        (search/__init__.py:14).  group_map was a host-side cache of GymGo's; the kernel
        recomputes groups, so it is accepted and ignored."""
        if isinstance(group_map, (bool, np.bool_)):          # children(state, canonical[, padded]) by position
            group_map, canonical, padded = None, bool(group_map), (True if padded is None else padded)
        if padded is None:
            padded = False
        c = context(state.shape[-1])
        kids, valid, _ = c.children(np.asarray(state)[None], canonical)
        kids = kids[0].cpu().numpy().astype(np.float64)
        if padded:
            return kids
        return [kids[a] for a in np.flatnonzero(valid[0].cpu().numpy())]

    @staticmethod
    def valid_moves(state):
        return np.append(1.0 - (np.asarray(state[GoVars.INVD_CHNL]) != 0).astype(np.float64).flatten(), 1.0)

    @staticmethod
    def game_ended(state):
        return int(np.count_nonzero(np.asarray(state[GoVars.DONE_CHNL]) == 1) == state.shape[1] * state.shape[2])

    @staticmethod
    def areas(state):
        b, w, _, _ = context(state.shape[-1]).score(np.asarray(state)[None])
        return int(b.item()), int(w.item())

    @staticmethod
    def get_num_liberties(state):
        b, w = context(state.shape[-1]).liberties(np.asarray(state)[None])
        return int(b.item()), int(w.item())

    @staticmethod
    def winning(state):
        _, _, _, winner = context(state.shape[-1]).score(np.asarray(state)[None])
        return int(winner.item())

    @staticmethod
    def turn(state):
        return int(np.max(state[GoVars.TURN_CHNL]))

    @staticmethod
    def canonical_form(state):
        state = np.array(state, dtype=np.float64)
        if GoGame.turn(state) == GoVars.WHITE:
            state[[0, 1]] = state[[1, 0]]
            state[GoVars.TURN_CHNL] = 0
        return state

    @staticmethod
    def action_2d_to_1d(action2d, state):
        size = state.shape[-1]
        if action2d is None:
            return size * size
        return int(action2d[0]) * size + int(action2d[1])

    @staticmethod
    def get_symmetries(chunk):
        """The 8 symmetries of a [C, s, s] chunk (views): bit 0 flips rows, bit 1 flips columns,
        bit 2 rotates a quarter turn (data.py:67, measurements.py:190 use them for augmentation)."""
        syms = []
        for i in range(8):
            x = np.asarray(chunk)
            if i & 1:
                x = np.flip(x, 1)
            if i & 2:
                x = np.flip(x, 2)
            if i & 4:
                x = np.rot90(x, 1, axes=(1, 2))
            syms.append(x)
        return syms

    @staticmethod
    def random_symmetry(chunk):
        return GoGame.get_symmetries(chunk)[np.random.randint(8)]

    @staticmethod
    def random_weighted_action(move_weights):
        p = np.asarray(move_weights, dtype=np.float64)
        return int(np.random.choice(len(p), p=p / np.sum(p)))


class GoEnv:
    gogame = GoGame
    govars = GoVars

    def __init__(self, size, reward_method='real', **_):
        self.size = size
        self.reward_method = reward_method
        self.state = GoGame.get_init_board(size)
        self.done = False

    def reset(self):
        self.state = GoGame.get_init_board(self.size)
        self.done = False
        return np.copy(self.state)

    def get_state(self):
        return np.copy(self.state)

    def canonical_state(self):
        return GoGame.canonical_form(self.state)

    def turn(self):
        return GoGame.turn(self.state)

    def valid_moves(self):
        return GoGame.valid_moves(self.state)

    def game_ended(self):
        return GoGame.game_ended(self.state)

    def winning(self):
        return GoGame.winning(self.state)

    def children(self, canonical=False, padded=True):
        return GoGame.children(self.state, canonical=canonical, padded=padded)

    def step(self, action):
        assert not self.done
        if isinstance(action, (tuple, list, np.ndarray)):
            action = int(action[0]) * self.size + int(action[1])
        elif action is None:
            action = self.size ** 2
        self.state = GoGame.next_state(self.state, int(action), canonical=False)
        self.done = bool(GoGame.game_ended(self.state))
        reward = float(self.winning()) if self.done else 0.0
        return np.copy(self.state), reward, self.done, {}
