/*
 * oracle/go_rules.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, one board at a time) of the Go rules that
 * huangeddie/Go-AI delegates to the third-party package GymGo
 * (gym_go.gogame / gym_go.state_utils, github.com/aigagror/GymGo, un-vendored and
 * un-pinned by the reference: /root/reference/README.md:4-6,
 * /root/reference/go_ai/data.py:10-12).  GymGo's source is absent from the build
 * container, so this file restates GymGo's *published* algorithm and is anchored on
 * the reference's own call sites:
 *   next_state(state, a, canonical=True)        /root/reference/go_ai/search/tree.py:115
 *   children(state, canonical=True, padded=True) /root/reference/go_ai/search/tree.py:92,
 *                                                /root/reference/go_ai/data.py:58
 *   valid_moves / INVD channel                   /root/reference/go_ai/data.py:15-24
 *   game_ended / winning                         /root/reference/go_ai/search/tree.py:65-69,
 *                                                /root/reference/go_ai/data.py:41-51
 *   areas / get_num_liberties                    /root/reference/go_ai/policies/baselines.py:18,41
 * and on the one known-answer test the reference holds at this boundary
 * (/root/reference/tests/test_policy.py:16-43).  Everything else is
 * "PARITY UNPINNED" with respect to GymGo's bits.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * State layout (GymGo govars): uint8 planes [6][s][s]
 *   0 BLACK stones, 1 WHITE stones, 2 TURN (all 0 = black to move, all 1 = white),
 *   3 INVD (1 = illegal for the side to move), 4 PASS (all 1 = previous move was a
 *   pass), 5 DONE (all 1 = game over).  Action a in [0, s*s) = (a / s, a % s); a == s*s = pass.
 */
#include <stdint.h>
#include <string.h>

#define MAXS 25
#define MAXP (MAXS * MAXS)

enum { CH_BLACK = 0, CH_WHITE = 1, CH_TURN = 2, CH_INVD = 3, CH_PASS = 4, CH_DONE = 5, NCH = 6 };

static int nbrs(int s, int p, int out[4])
{
    int r = p / s, c = p % s, n = 0;
    if (r > 0) out[n++] = p - s;
    if (r < s - 1) out[n++] = p + s;
    if (c > 0) out[n++] = p - 1;
    if (c < s - 1) out[n++] = p + 1;
    return n;
}

/* Connected components of equal-valued cells (4-connectivity).  board: 0 empty, 1 black,
 * 2 white.  label[p] = smallest cell index in p's component.  For stone components
 * libs[root] = number of distinct empty cells adjacent to the group; for empty components
 * libs[root] = bit mask of adjacent stone colours (1 black, 2 white). */
static void analyze(int s, const uint8_t *board, int *label, int *libs)
{
    int P = s * s, stack[MAXP], nb[4];
    static const int UNSET = -1;
    for (int p = 0; p < P; p++) { label[p] = UNSET; libs[p] = 0; }
    for (int p = 0; p < P; p++) {
        if (label[p] != UNSET) continue;
        int sp = 0, colour = board[p];
        uint8_t seen_lib[MAXP];
        memset(seen_lib, 0, (size_t)P);
        stack[sp++] = p; label[p] = p;
        while (sp) {
            int q = stack[--sp];
            int n = nbrs(s, q, nb);
            for (int i = 0; i < n; i++) {
                int t = nb[i];
                if (board[t] == colour) {
                    if (label[t] == UNSET) { label[t] = p; stack[sp++] = t; }
                } else if (colour != 0) {
                    if (board[t] == 0 && !seen_lib[t]) { seen_lib[t] = 1; libs[p]++; }
                } else {
                    libs[p] |= board[t]; /* empty region: colours touching it */
                }
            }
        }
    }
}

static void to_board(int s, const uint8_t *st, uint8_t *board)
{
    int P = s * s;
    for (int p = 0; p < P; p++) board[p] = st[CH_BLACK * P + p] ? 1 : (st[CH_WHITE * P + p] ? 2 : 0);
}

static int plane_any(const uint8_t *pl, int P)
{
    for (int p = 0; p < P; p++) if (pl[p]) return 1;
    return 0;
}
static int plane_all(const uint8_t *pl, int P)
{
    for (int p = 0; p < P; p++) if (!pl[p]) return 0;
    return 1;
}

int goai_oracle_turn(const uint8_t *st, int s) { return plane_any(st + CH_TURN * s * s, s * s); }
int goai_oracle_prev_passed(const uint8_t *st, int s) { return plane_any(st + CH_PASS * s * s, s * s); }
int goai_oracle_game_ended(const uint8_t *st, int s) { return plane_all(st + CH_DONE * s * s, s * s); }

/* GymGo state_utils.compute_invalid_moves: moves illegal for the player who moves AFTER
 * `mover` (colour index 0 black / 1 white): occupied cells, the ko point, and empty
 * cells with no empty neighbour that neither join a next-player group having >= 2
 * liberties nor take the last liberty of a mover group. */
static void compute_invalid(int s, const uint8_t *board, int mover, int ko, uint8_t *inv)
{
    int P = s * s, label[MAXP], libs[MAXP], nb[4];
    int mover_col = mover + 1, next_col = 2 - mover;
    analyze(s, board, label, libs);
    for (int p = 0; p < P; p++) {
        if (board[p]) { inv[p] = 1; continue; }
        int n = nbrs(s, p, nb), ok = 0;
        for (int i = 0; i < n && !ok; i++) {
            int t = nb[i];
            if (board[t] == 0) ok = 1;
            else if (board[t] == next_col && libs[label[t]] >= 2) ok = 1;
            else if (board[t] == mover_col && libs[label[t]] == 1) ok = 1;
        }
        inv[p] = ok ? 0 : 1;
    }
    if (ko >= 0) inv[ko] = 1;
}

void goai_oracle_canonical_form(const uint8_t *st, int s, uint8_t *out)
{
    int P = s * s;
    if (out != st) memcpy(out, st, (size_t)NCH * P);
    if (plane_any(out + CH_TURN * P, P)) {
        uint8_t tmp[MAXP];
        memcpy(tmp, out + CH_BLACK * P, (size_t)P);
        memcpy(out + CH_BLACK * P, out + CH_WHITE * P, (size_t)P);
        memcpy(out + CH_WHITE * P, tmp, (size_t)P);
        memset(out + CH_TURN * P, 0, (size_t)P);
    }
}

/* GymGo gogame.next_state.  Returns 0, or -1 for an illegal move (GymGo asserts),
 * -2 for an out-of-range action. */
int goai_oracle_next_state(const uint8_t *st, int s, int action, int canonical, uint8_t *out)
{
    int P = s * s, nb[4];
    if (action < 0 || action > P) return -2;
    uint8_t tmp[NCH * MAXP];
    memcpy(tmp, st, (size_t)NCH * P);
    int player = goai_oracle_turn(st, s);
    int prev_pass = goai_oracle_prev_passed(st, s);
    int ko = -1;
    uint8_t board[MAXP];
    if (action == P) {
        memset(tmp + CH_PASS * P, 1, (size_t)P);
        if (prev_pass) memset(tmp + CH_DONE * P, 1, (size_t)P);
        to_board(s, tmp, board);
    } else {
        if (st[CH_INVD * P + action]) return -1;
        memset(tmp + CH_PASS * P, 0, (size_t)P);
        to_board(s, tmp, board);
        if (board[action]) return -1;
        int me = player + 1, opp = 2 - player;
        board[action] = (uint8_t)me;
        int n = nbrs(s, action, nb), surrounded = 1;
        for (int i = 0; i < n; i++) if (board[nb[i]] != opp) surrounded = 0;
        int label[MAXP], libs[MAXP];
        analyze(s, board, label, libs);
        int killed_roots[4], nk = 0;
        for (int i = 0; i < n; i++) {
            int t = nb[i];
            if (board[t] == opp && libs[label[t]] == 0) {
                int r = label[t], dup = 0;
                for (int j = 0; j < nk; j++) if (killed_roots[j] == r) dup = 1;
                if (!dup) killed_roots[nk++] = r;
            }
        }
        int last_size = 0, last_cell = -1;
        for (int j = 0; j < nk; j++) {
            last_size = 0;
            for (int p = 0; p < P; p++)
                if (board[p] == opp && label[p] == killed_roots[j]) { board[p] = 0; last_size++; last_cell = p; }
        }
        if (nk == 1 && last_size == 1 && surrounded) ko = last_cell;
        for (int p = 0; p < P; p++) {
            tmp[CH_BLACK * P + p] = board[p] == 1;
            tmp[CH_WHITE * P + p] = board[p] == 2;
        }
    }
    compute_invalid(s, board, player, ko, tmp + CH_INVD * P);
    memset(tmp + CH_TURN * P, player ? 0 : 1, (size_t)P);
    if (canonical) goai_oracle_canonical_form(tmp, s, tmp);
    memcpy(out, tmp, (size_t)NCH * P);
    return 0;
}

/* GymGo gogame.children(state, canonical, padded=True): out is [A][6][s][s]; children of
 * illegal moves are all-zero states.  valid_out (optional) gets the [A] validity flags. */
int goai_oracle_children(const uint8_t *st, int s, int canonical, uint8_t *out, uint8_t *valid_out)
{
    int P = s * s, A = P + 1, nvalid = 0;
    for (int a = 0; a < A; a++) {
        uint8_t *o = out + (size_t)a * NCH * P;
        int valid = (a == P) || !st[CH_INVD * P + a];
        if (valid && goai_oracle_next_state(st, s, a, canonical, o) == 0) nvalid++;
        else { valid = 0; memset(o, 0, (size_t)NCH * P); }
        if (valid_out) valid_out[a] = (uint8_t)valid;
    }
    return nvalid;
}

/* GymGo gogame.areas: stones plus empty regions bordered by one colour only. */
void goai_oracle_areas(const uint8_t *st, int s, int *black_area, int *white_area)
{
    int P = s * s, label[MAXP], libs[MAXP], b = 0, w = 0;
    uint8_t board[MAXP];
    to_board(s, st, board);
    analyze(s, board, label, libs);
    for (int p = 0; p < P; p++) {
        if (board[p] == 1) b++;
        else if (board[p] == 2) w++;
        else if (libs[label[p]] == 1) b++;
        else if (libs[label[p]] == 2) w++;
    }
    *black_area = b; *white_area = w;
}

/* GymGo gogame.winning (komi 0): sign(black_area - white_area) of planes 0 / 1 as given. */
int goai_oracle_winning(const uint8_t *st, int s)
{
    int b, w;
    goai_oracle_areas(st, s, &b, &w);
    return (b > w) - (b < w);
}

/* GymGo gogame.get_num_liberties: distinct empty cells adjacent to any stone of a colour. */
void goai_oracle_num_liberties(const uint8_t *st, int s, int *black_libs, int *white_libs)
{
    int P = s * s, nb[4], bl = 0, wl = 0;
    uint8_t board[MAXP];
    to_board(s, st, board);
    for (int p = 0; p < P; p++) {
        if (board[p]) continue;
        int n = nbrs(s, p, nb), tb = 0, tw = 0;
        for (int i = 0; i < n; i++) { tb |= board[nb[i]] == 1; tw |= board[nb[i]] == 2; }
        bl += tb; wl += tw;
    }
    *black_libs = bl; *white_libs = wl;
}

/* Recompute the INVD plane of a state from its stones for the side to move, without ko
 * (used to build arbitrary test positions).  */
void goai_oracle_refresh_invalid(uint8_t *st, int s)
{
    uint8_t board[MAXP];
    to_board(s, st, board);
    int to_move = goai_oracle_turn(st, s);
    compute_invalid(s, board, 1 - to_move, -1, st + CH_INVD * s * s);
}

/* Batched helpers so Python test loops stay cheap. */
int goai_oracle_next_states(const uint8_t *st, int s, const int32_t *actions, int n, int canonical,
                            uint8_t *out, int32_t *status)
{
    size_t stride = (size_t)NCH * s * s;
    int bad = 0;
    for (int i = 0; i < n; i++) {
        int rc = goai_oracle_next_state(st + i * stride, s, actions[i], canonical, out + i * stride);
        if (rc) { memset(out + i * stride, 0, stride); bad++; }
        if (status) status[i] = rc;
    }
    return bad;
}
