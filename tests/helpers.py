"""Shared test helpers (CPU side)."""
import numpy as np

from oracle.gogame import GoGame, GoVars


def from_ascii(rows, to_move=0, prev_pass=False):
    """'X' black, 'O' white, '.' empty.  INVD is rebuilt from the stones (no ko)."""
    size = len(rows)
    st = np.zeros((6, size, size))
    for r, row in enumerate(rows):
        assert len(row) == size
        for c, ch in enumerate(row):
            if ch == 'X':
                st[GoVars.BLACK, r, c] = 1
            elif ch == 'O':
                st[GoVars.WHITE, r, c] = 1
    st[GoVars.TURN_CHNL] = to_move
    if prev_pass:
        st[GoVars.PASS_CHNL] = 1
    return GoGame.refresh_invalid(st)


def to_ascii(st):
    size = st.shape[-1]
    return ["".join('X' if st[0, r, c] else 'O' if st[1, r, c] else '.' for c in range(size)) for r in range(size)]


def random_positions(size, n, seed, min_moves=0, max_moves=None, canonical=True, pass_prob=0.03):
    """Seeded random legal playouts from the empty board -> uint8 [n,6,s,s] (canonical) states."""
    rng = np.random.default_rng(seed)
    if max_moves is None:
        max_moves = size * size
    out = []
    while len(out) < n:
        st = GoGame.get_init_board(size)
        k = int(rng.integers(min_moves, max_moves + 1))
        for _ in range(k):
            valid = GoGame.valid_moves(st)
            p = valid.copy()
            p[-1] = pass_prob * max(1.0, p[:-1].sum())
            p /= p.sum()
            a = int(rng.choice(len(p), p=p))
            nxt = GoGame.next_state(st, a, canonical=False)
            if GoGame.game_ended(nxt):
                break
            st = nxt
        out.append(GoGame.canonical_form(st) if canonical else st)
    return np.asarray(out).astype(np.uint8)
