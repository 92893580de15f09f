"""oracle/gogame.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes face of oracle/go_rules.c under the names the reference calls on GymGo
(`GoGame.next_state`, `.children`, `.valid_moves`, `.game_ended`, `.winning`, `.areas`, ...;
call sites listed in SURVEY.md section 8a-G) plus a `GoEnv` with the environment methods
/root/reference/go_ai/game.py:72-113 uses (`canonical_state`, `turn`, `step`, `children`,
`winning`, `valid_moves`, `reset`, `size`, `state`).  States are float64 [6,s,s] arrays as
GymGo returns them.  PARITY UNPINNED versus GymGo's bits (see go_rules.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_SOURCES = [os.path.join(_HERE, f) for f in ("go_rules.c", "detmath.c")]


def build(force=False):
    """Compile the C restatement (gcc, no GPU involved)."""
    srcs = [s for s in _SOURCES if os.path.exists(s)]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
           "-o", _LIB_PATH] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.goai_oracle_next_state.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, u8p]
        _lib.goai_oracle_children.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p, u8p]
        _lib.goai_oracle_areas.argtypes = [u8p, ctypes.c_int, ip, ip]
        _lib.goai_oracle_num_liberties.argtypes = [u8p, ctypes.c_int, ip, ip]
        _lib.goai_oracle_winning.argtypes = [u8p, ctypes.c_int]
        _lib.goai_oracle_game_ended.argtypes = [u8p, ctypes.c_int]
        _lib.goai_oracle_turn.argtypes = [u8p, ctypes.c_int]
        _lib.goai_oracle_canonical_form.argtypes = [u8p, ctypes.c_int, u8p]
        _lib.goai_oracle_refresh_invalid.argtypes = [u8p, ctypes.c_int]
        _lib.goai_oracle_next_states.argtypes = [u8p, ctypes.c_int, i32p, ctypes.c_int, ctypes.c_int,
                                                 u8p, i32p]
    return _lib


def _u8(state):
    a = np.ascontiguousarray(state)
    if a.dtype != np.uint8:
        a = (a != 0).astype(np.uint8)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


class GoVars:
    BLACK = 0
    WHITE = 1
    TURN_CHNL = 2
    INVD_CHNL = 3
    PASS_CHNL = 4
    DONE_CHNL = 5
    NUM_CHNLS = 6


class GoGame:
    """Static functions with GymGo's names, backed by the C restatement."""

    @staticmethod
    def get_init_board(size):
        return np.zeros((GoVars.NUM_CHNLS, size, size))

    init_state = get_init_board

    @staticmethod
    def action_size(state=None, board_size=None):
        if state is not None:
            return int(state.shape[1] * state.shape[2] + 1)
        return int(board_size * board_size + 1)

    @staticmethod
    def next_state(state, action1d, canonical=False):
        s = state.shape[-1]
        st = _u8(state)
        out = np.empty_like(st)
        rc = lib().goai_oracle_next_state(_ptr(st), s, int(action1d), int(bool(canonical)), _ptr(out))
        assert rc == 0, ("Invalid move", int(action1d), rc)
        return out.astype(np.float64)

    @staticmethod
    def next_states_u8(states, actions, canonical=True):
        """Batched uint8 variant used by the parity tests: returns (out, status)."""
        st = _u8(states)
        n, _, s, _ = st.shape
        acts = np.ascontiguousarray(actions, dtype=np.int32)
        out = np.empty_like(st)
        status = np.empty(n, dtype=np.int32)
        lib().goai_oracle_next_states(_ptr(st), s, acts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n,
                                      int(bool(canonical)), _ptr(out),
                                      status.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        return out, status

    @staticmethod
    def children_u8(state, canonical=True):
        st = _u8(state)
        s = st.shape[-1]
        A = s * s + 1
        out = np.empty((A,) + st.shape, dtype=np.uint8)
        valid = np.empty(A, dtype=np.uint8)
        lib().goai_oracle_children(_ptr(st), s, int(bool(canonical)), _ptr(out), _ptr(valid))
        return out, valid

    @staticmethod
    def children(state, canonical=False, padded=True):
        out, valid = GoGame.children_u8(state, canonical)
        out = out.astype(np.float64)
        if padded:
            return out
        return [out[a] for a in np.flatnonzero(valid)]

    @staticmethod
    def valid_moves(state):
        return np.append(1.0 - (np.asarray(state[GoVars.INVD_CHNL]) != 0).astype(np.float64).flatten(), 1.0)

    @staticmethod
    def invalid_moves(state):
        return 1.0 - GoGame.valid_moves(state)

    @staticmethod
    def game_ended(state):
        st = _u8(state)
        return int(lib().goai_oracle_game_ended(_ptr(st), st.shape[-1]))

    @staticmethod
    def winning(state):
        st = _u8(state)
        return int(lib().goai_oracle_winning(_ptr(st), st.shape[-1]))

    @staticmethod
    def areas(state):
        st = _u8(state)
        b, w = ctypes.c_int(), ctypes.c_int()
        lib().goai_oracle_areas(_ptr(st), st.shape[-1], ctypes.byref(b), ctypes.byref(w))
        return b.value, w.value

    @staticmethod
    def get_num_liberties(state):
        st = _u8(state)
        b, w = ctypes.c_int(), ctypes.c_int()
        lib().goai_oracle_num_liberties(_ptr(st), st.shape[-1], ctypes.byref(b), ctypes.byref(w))
        return b.value, w.value

    num_liberties = get_num_liberties

    @staticmethod
    def turn(state):
        return int(np.max(state[GoVars.TURN_CHNL]))

    @staticmethod
    def prev_player_passed(state):
        return bool(np.max(state[GoVars.PASS_CHNL]) == 1)

    @staticmethod
    def canonical_form(state):
        st = _u8(state)
        out = np.empty_like(st)
        lib().goai_oracle_canonical_form(_ptr(st), st.shape[-1], _ptr(out))
        return out.astype(np.float64)

    @staticmethod
    def refresh_invalid(state):
        """Not a GymGo name: rebuild INVD from the stones (no ko) for hand-made test positions."""
        st = _u8(state).copy()
        lib().goai_oracle_refresh_invalid(_ptr(st), st.shape[-1])
        return st.astype(np.float64)

    @staticmethod
    def action_2d_to_1d(action2d, state):
        size = state.shape[-1]
        if action2d is None:
            return size * size
        return int(action2d[0]) * size + int(action2d[1])

    @staticmethod
    def random_weighted_action(move_weights):
        move_weights = np.asarray(move_weights, dtype=np.float64)
        p = move_weights / np.sum(move_weights)
        return int(np.random.choice(np.arange(len(p)), p=p))

    @staticmethod
    def get_symmetries(chunk):
        syms = []
        for i in range(8):
            x = chunk
            if (i >> 0) % 2:
                x = np.flip(x, 1)
            if (i >> 1) % 2:
                x = np.flip(x, 2)
            if (i >> 2) % 2:
                x = np.rot90(x, 1, axes=(1, 2))
            syms.append(x)
        return syms

    @staticmethod
    def random_symmetry(chunk):
        return GoGame.get_symmetries(chunk)[np.random.randint(8)]


class GoEnv:
    """The slice of GymGo's GoEnv that /root/reference/go_ai/game.py and the policies touch."""
    gogame = GoGame
    govars = GoVars

    def __init__(self, size, reward_method='real', **_):
        self.size = size
        self.reward_method = reward_method
        self.state = GoGame.get_init_board(size) if size > 0 else None
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
        return GoGame.children(self.state, canonical, padded)

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
