"""CPU tests of the Go-rules oracle (oracle/go_rules.c) -- known-answer positions, the
reference's own KAT at the GymGo boundary, and a cross-check against the independent
numpy/scipy restatement of GymGo's published algorithm (oracle/gogame_np.py).

These pin the BUILD's rules; versus GymGo's bits parity stays "unpinned" except for the
reference's KAT (/root/reference/tests/test_policy.py:16-43), see SURVEY.md section 8c.
"""
import numpy as np
import pytest

from helpers import from_ascii, to_ascii
from oracle import gogame_np as gn
from oracle.gogame import GoEnv, GoGame, GoVars


def test_reference_kat_obvious_pass(oracle_lib):
    # /root/reference/tests/test_policy.py:16-21: black 0, white 1, black 2, white pass
    env = GoEnv(5)
    env.reset()
    env.step(0)
    env.step(1)
    env.step(2)
    env.step(None)
    assert env.turn() == GoVars.BLACK and not env.game_ended()
    kids = env.children(canonical=True, padded=True)
    assert kids.shape == (26, 6, 5, 5)
    # passing now ends the game and black (2 stones) beats white (1 stone): the child is
    # in white's canonical frame, so white "loses" there.
    assert GoGame.game_ended(kids[-1]) == 1
    assert GoGame.winning(kids[-1]) == -1
    for a in range(25):
        assert GoGame.game_ended(kids[a]) == 0


def test_single_stone_capture_and_ko(oracle_lib):
    st = from_ascii([".XO..",
                     "XO.O.",
                     ".XO..",
                     ".....",
                     "....."], to_move=0)
    # black plays (1,2): captures the single white stone (1,1); the new stone is surrounded -> ko
    nxt = GoGame.next_state(st, 1 * 5 + 2)
    assert to_ascii(nxt)[1] == "X.XO."
    assert GoGame.turn(nxt) == 1
    assert nxt[GoVars.INVD_CHNL, 1, 1] == 1          # ko point is illegal for white
    # white plays elsewhere, black answers elsewhere -> ko point is open again for white
    nxt2 = GoGame.next_state(nxt, 4 * 5 + 4)
    nxt3 = GoGame.next_state(nxt2, 4 * 5 + 0)
    assert nxt3[GoVars.INVD_CHNL, 1, 1] == 0
    back = GoGame.next_state(nxt3, 1 * 5 + 1)         # white retakes
    assert to_ascii(back)[1] == "XO.O."
    assert back[GoVars.INVD_CHNL, 1, 2] == 1          # and now black is barred by ko


def test_capture_without_ko_when_not_surrounded(oracle_lib):
    st = from_ascii(["OX...",
                     ".....",
                     ".....",
                     ".....",
                     "....."], to_move=0)
    nxt = GoGame.next_state(st, 1 * 5 + 0)            # black (1,0) captures corner stone
    assert to_ascii(nxt)[0] == ".X..."
    # (0,0) has no empty neighbour and both adjacent black groups keep >= 2 liberties:
    # it is suicide for white, but that is the suicide rule, not ko.
    assert nxt[GoVars.INVD_CHNL, 0, 0] == 1
    far = GoGame.next_state(GoGame.next_state(nxt, 24), 23)
    assert far[GoVars.INVD_CHNL, 0, 0] == 1


def test_multi_stone_capture(oracle_lib):
    st = from_ascii(["XOO.X",
                     ".XX..",
                     ".....",
                     ".....",
                     "....."], to_move=0)
    nxt = GoGame.next_state(st, 3)                    # black (0,3) removes both white stones
    assert to_ascii(nxt)[0] == "X..XX"
    assert np.sum(nxt[GoVars.WHITE]) == 0
    # two stones captured -> no ko
    assert nxt[GoVars.INVD_CHNL, 0, 1] == 0 and nxt[GoVars.INVD_CHNL, 0, 2] == 0


def test_suicide_illegal_but_capturing_suicide_legal(oracle_lib):
    st = from_ascii([".X...",
                     "X....",
                     ".....",
                     ".....",
                     "....."], to_move=1)
    assert st[GoVars.INVD_CHNL, 0, 0] == 1            # white in the corner = suicide
    st2 = from_ascii([".XO..",
                      "XO...",
                      "O....",
                      ".....",
                      "....."], to_move=1)
    # white (0,0) now captures (0,1) and (1,0), each in atari -> legal
    assert st2[GoVars.INVD_CHNL, 0, 0] == 0
    nxt = GoGame.next_state(st2, 0)
    assert to_ascii(nxt)[:2] == ["O.O..", ".O..."]


def test_own_last_liberty_fill_is_suicide(oracle_lib):
    st = from_ascii(["X.XO.",
                     "XXXO.",
                     "OOOO.",
                     ".....",
                     "....."], to_move=0)
    assert st[GoVars.INVD_CHNL, 0, 1] == 1            # black filling its own last liberty
    st_w = from_ascii(["X.XO.",
                       "XXXO.",
                       "OOOO.",
                       ".....",
                       "....."], to_move=1)
    assert st_w[GoVars.INVD_CHNL, 0, 1] == 0          # white captures there
    nxt = GoGame.next_state(st_w, 1)
    assert np.sum(nxt[GoVars.BLACK]) == 0


def test_pass_pass_ends_and_areas(oracle_lib):
    st = from_ascii(["..X.O",
                     "..XOO",
                     "XXXO.",
                     "OOOO.",
                     "....."], to_move=0)
    assert GoGame.areas(st) == (9, 15)   # (0,3) touches both colours
    p1 = GoGame.next_state(st, 25)
    assert not GoGame.game_ended(p1) and GoGame.prev_player_passed(p1)
    p2 = GoGame.next_state(p1, 25)
    assert GoGame.game_ended(p2)
    assert GoGame.winning(p2) == -1
    # a move after a pass clears the pass plane
    p1b = GoGame.next_state(p1, 24)
    assert not GoGame.prev_player_passed(p1b)


def test_neutral_territory_counts_for_nobody(oracle_lib):
    st = from_ascii(["X.O..",
                     "X.O..",
                     "X.O..",
                     "X.O..",
                     "X.O.."], to_move=0)
    # column 1 touches both colours -> nobody; columns 3-4 are white's
    assert GoGame.areas(st) == (5, 15)
    assert GoGame.areas(GoGame.get_init_board(5)) == (0, 0)
    assert GoGame.winning(GoGame.get_init_board(5)) == 0


def test_edge_and_corner_liberties(oracle_lib):
    st = from_ascii(["X...O",
                     ".....",
                     "..X..",
                     ".....",
                     "....."], to_move=0)
    assert GoGame.get_num_liberties(st) == (2 + 4, 2)


def test_canonical_form_and_children_padding(oracle_lib):
    st = from_ascii([".X...",
                     "X....",
                     ".....",
                     ".....",
                     "....."], to_move=1)
    can = GoGame.canonical_form(st)
    assert np.array_equal(can[0], st[1]) and np.array_equal(can[1], st[0])
    assert can[GoVars.TURN_CHNL].max() == 0
    kids = GoGame.children(st, canonical=True, padded=True)
    assert np.all(kids[0] == 0)                        # the suicide move is zero padded
    assert np.array_equal(kids[7], GoGame.next_state(st, 7, canonical=True))
    assert kids[7][GoVars.TURN_CHNL].max() == 0
    assert len(GoGame.children(st, canonical=True, padded=False)) == int(GoGame.valid_moves(st).sum())
    with pytest.raises(AssertionError):
        GoGame.next_state(st, 0)


@pytest.mark.parametrize("size,games", [(5, 6), (9, 3), (13, 1)])
def test_c_matches_numpy_restatement(oracle_lib, size, games):
    rng = np.random.default_rng(100 + size)
    for g in range(games):
        st = GoGame.get_init_board(size)
        for mv in range(2 * size * size):
            if GoGame.game_ended(st):
                break
            valid = GoGame.valid_moves(st)
            assert np.array_equal(valid, gn.valid_moves(st))
            p = valid.copy()
            p[-1] = 0.02 if mv < size * size else 0.4
            p /= p.sum()
            a = int(rng.choice(len(p), p=p))
            canonical = bool(rng.integers(2)) and False
            s1 = GoGame.next_state(st, a, canonical=canonical)
            s2 = gn.next_state(st, a, canonical=canonical)
            assert np.array_equal(s1, s2), (size, g, mv, a)
            assert GoGame.areas(s1) == gn.areas(s2)
            assert GoGame.winning(s1) == gn.winning(s2)
            assert GoGame.get_num_liberties(s1) == gn.get_num_liberties(s2)
            assert np.array_equal(GoGame.canonical_form(s1), gn.canonical_form(s2))
            if mv % 11 == 0:
                assert np.array_equal(GoGame.children(st, canonical=True), gn.children(st, canonical=True))
            st = s1
