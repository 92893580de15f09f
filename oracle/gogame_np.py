"""oracle/gogame_np.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/scipy restatement of the *published* GymGo algorithm (gym_go.gogame and
gym_go.state_utils of github.com/aigagror/GymGo; un-vendored, un-pinned dependency of the
reference, /root/reference/README.md:4-6, absent from this container).  GymGo labels groups
with scipy.ndimage and derives the invalid-move plane from per-group liberty counts; this
file follows that formulation (ndimage.label / binary_dilation / convolve) so that it is an
independent second statement of the rules, cross-checked against the scalar C restatement
in oracle/go_rules.c by tests/test_oracle_go.py.  PARITY UNPINNED versus GymGo's bits except
for the reference's own known-answer test (/root/reference/tests/test_policy.py:16-43).

Only tests/ may import this module.
"""
import numpy as np
from scipy import ndimage

BLACK, WHITE, TURN_CHNL, INVD_CHNL, PASS_CHNL, DONE_CHNL, NUM_CHNLS = 0, 1, 2, 3, 4, 5, 6

surround_struct = np.array([[0, 1, 0],
                            [1, 0, 1],
                            [0, 1, 0]])


def init_state(size):
    return np.zeros((NUM_CHNLS, size, size))


def turn(state):
    return int(np.max(state[TURN_CHNL]))


def prev_player_passed(state):
    return np.max(state[PASS_CHNL] == 1) == 1


def game_ended(state):
    m, n = state.shape[1:]
    return int(np.count_nonzero(state[DONE_CHNL] == 1) == m * n)


def action_size(state=None, board_size=None):
    if state is not None:
        m, n = state.shape[1:]
    else:
        m = n = board_size
    return m * n + 1


def compute_invalid_moves(state, player, ko_protect=None):
    """Invalid moves for the opponent of `player` (GymGo state_utils.compute_invalid_moves)."""
    all_pieces = np.sum(state[[BLACK, WHITE]], axis=0)
    empties = 1 - all_pieces

    possible_invalid = np.zeros(state.shape[1:])
    definite_valid = np.zeros(state.shape[1:])

    own_groups, n_own = ndimage.label(state[player])
    opp_groups, n_opp = ndimage.label(state[1 - player])

    for groups, n, own in ((own_groups, n_own, True), (opp_groups, n_opp, False)):
        for i in range(1, n + 1):
            grp = groups == i
            libs = empties * ndimage.binary_dilation(grp, surround_struct)
            count = np.sum(libs)
            if own:
                # liberties of the mover's groups: capturing point if it is the last one
                if count > 1:
                    possible_invalid += libs
                elif count == 1:
                    definite_valid += libs
            else:
                # liberties of the next player's groups: self-atari fill is suicide-prone
                if count == 1:
                    possible_invalid += libs
                elif count > 1:
                    definite_valid += libs

    surrounded = ndimage.convolve(all_pieces, surround_struct, mode='constant', cval=1) == 4
    invalid = all_pieces + possible_invalid * (definite_valid == 0) * surrounded
    if ko_protect is not None:
        invalid[ko_protect[0], ko_protect[1]] = 1
    return invalid > 0


def canonical_form(state):
    state = np.copy(state)
    if turn(state) == WHITE:
        channels = np.arange(NUM_CHNLS)
        channels[BLACK] = WHITE
        channels[WHITE] = BLACK
        state = state[channels]
        state[TURN_CHNL] = 0
    return state


def next_state(state, action1d, canonical=False):
    state = np.copy(state)
    size = state.shape[1]
    pass_idx = size * size
    player = turn(state)
    previously_passed = prev_player_passed(state)
    ko_protect = None

    if action1d == pass_idx:
        state[PASS_CHNL] = 1
        if previously_passed:
            state[DONE_CHNL] = 1
    else:
        r, c = action1d // size, action1d % size
        state[PASS_CHNL] = 0
        assert state[INVD_CHNL, r, c] == 0, ("Invalid move", (r, c))
        state[player, r, c] = 1

        # neighbours of the new stone; "surrounded" = every one of them is an enemy stone
        adj = [(r + dr, c + dc) for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1))
               if 0 <= r + dr < size and 0 <= c + dc < size]
        surrounded = all(state[1 - player, ar, ac] > 0 for ar, ac in adj)

        # remove enemy groups adjacent to the stone that have no liberty left
        all_pieces = np.sum(state[[BLACK, WHITE]], axis=0)
        empties = 1 - all_pieces
        opp_groups, _ = ndimage.label(state[1 - player])
        killed_groups = []
        for lab in sorted({opp_groups[ar, ac] for ar, ac in adj if opp_groups[ar, ac] > 0}):
            grp = opp_groups == lab
            libs = empties * ndimage.binary_dilation(grp, surround_struct)
            if np.sum(libs) == 0:
                killed_groups.append(np.argwhere(grp))
        for grp_cells in killed_groups:
            for kr, kc in grp_cells:
                state[1 - player, kr, kc] = 0

        if len(killed_groups) == 1 and surrounded and len(killed_groups[0]) == 1:
            ko_protect = tuple(killed_groups[0][0])

    state[INVD_CHNL] = compute_invalid_moves(state, player, ko_protect)
    state[TURN_CHNL] = 1 - player
    if canonical:
        state = canonical_form(state)
    return state


def valid_moves(state):
    return np.append(1 - state[INVD_CHNL].flatten(), 1)


def children(state, canonical=False, padded=True):
    valid = valid_moves(state)
    n = len(valid)
    kids = [next_state(state, a, canonical) for a in np.argwhere(valid > 0).flatten()]
    if not padded:
        return kids
    out = np.zeros((n, *state.shape))
    out[np.argwhere(valid > 0).flatten()] = kids
    return out


def areas(state):
    all_pieces = np.sum(state[[BLACK, WHITE]], axis=0)
    empties = 1 - all_pieces
    labels, n = ndimage.label(empties)
    black_area, white_area = np.sum(state[BLACK]), np.sum(state[WHITE])
    for lab in range(1, n + 1):
        region = labels == lab
        neighbours = ndimage.binary_dilation(region)
        black_claim = (state[BLACK] * neighbours > 0).any()
        white_claim = (state[WHITE] * neighbours > 0).any()
        if black_claim and not white_claim:
            black_area += np.sum(region)
        elif white_claim and not black_claim:
            white_area += np.sum(region)
    return int(black_area), int(white_area)


def winning(state):
    b, w = areas(state)
    return int(np.sign(b - w))


def get_num_liberties(state):
    all_pieces = np.sum(state[[BLACK, WHITE]], axis=0)
    out = []
    for pieces in (state[BLACK], state[WHITE]):
        libs = ndimage.binary_dilation(pieces, surround_struct) * (1 - all_pieces)
        out.append(int(np.sum(libs)))
    return out[0], out[1]
