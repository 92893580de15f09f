// go_bits.h -- the Go rules of go_board.h on bit boards, ONE BOARD PER LANE.
//
// go_board.h runs one board per 64-lane wavefront (board in LDS, flood fill by label propagation): the right shape
// inside k_mcts_select, where a wave owns one game.  Where many independent boards are transformed at once
// (goai_transition, goai_children, goai_score, the Q search's make_children) that shape leaves the kernel waiting on
// dependent LDS round trips (0.08 of the HBM roofline, profiles/r02_go_*).  Here a position is three bit planes of
// P = S*S bits in the lane's own registers -- the packed state format of the search (bit i = cell i, exactly the
// __ballot() order) -- and every rule is a handful of 64-bit shifts and masks:
//   neighbours(x) = ((x & ~last column) << 1) | ((x & ~first column) >> 1) | (x << S) | (x >> S)
//   group of a stone = fixed point of g |= neighbours(g) & colour;   liberties = neighbours(g) & empty
// Same results as go_board.h bit for bit (GymGo's next_state / compute_invalid_moves / areas as restated there):
// tests/test_gpu_go.py runs both against oracle/go_rules.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "go_board.h"

namespace goai {

template <int S>
struct BitGeom {
    static constexpr int P = S * S, NW = (P + 63) / 64;
    struct Tab { uint64_t board[NW], not_first[NW], not_last[NW]; };
    static constexpr Tab make()
    {
        Tab t{};
        for (int i = 0; i < P; i++) {
            t.board[i >> 6] |= 1ull << (i & 63);
            if (i % S != 0) t.not_first[i >> 6] |= 1ull << (i & 63);
            if (i % S != S - 1) t.not_last[i >> 6] |= 1ull << (i & 63);
        }
        return t;
    }
};
template <int S>
__device__ constexpr typename BitGeom<S>::Tab kBitTab = BitGeom<S>::make();

template <int S>
struct BB {
    static constexpr int NW = BitGeom<S>::NW;
    uint64_t w[NW];
};

#define BB_FOR(k) _Pragma("unroll") for (int k = 0; k < BB<S>::NW; k++)

template <int S> __device__ __forceinline__ BB<S> bb_zero() { BB<S> r; BB_FOR(k) r.w[k] = 0; return r; }
template <int S> __device__ __forceinline__ BB<S> bb_and(const BB<S>& a, const BB<S>& b) { BB<S> r; BB_FOR(k) r.w[k] = a.w[k] & b.w[k]; return r; }
template <int S> __device__ __forceinline__ BB<S> bb_or(const BB<S>& a, const BB<S>& b) { BB<S> r; BB_FOR(k) r.w[k] = a.w[k] | b.w[k]; return r; }
template <int S> __device__ __forceinline__ BB<S> bb_andnot(const BB<S>& a, const BB<S>& b) { BB<S> r; BB_FOR(k) r.w[k] = a.w[k] & ~b.w[k]; return r; }
template <int S> __device__ __forceinline__ bool bb_any(const BB<S>& a) { uint64_t o = 0; BB_FOR(k) o |= a.w[k]; return o != 0; }
template <int S> __device__ __forceinline__ bool bb_eq(const BB<S>& a, const BB<S>& b) { uint64_t o = 0; BB_FOR(k) o |= a.w[k] ^ b.w[k]; return o == 0; }
template <int S> __device__ __forceinline__ int bb_count(const BB<S>& a) { int n = 0; BB_FOR(k) n += __popcll(a.w[k]); return n; }
template <int S> __device__ __forceinline__ BB<S> bb_board() { BB<S> r; BB_FOR(k) r.w[k] = kBitTab<S>.board[k]; return r; }

// the single bit `idx` (0 <= idx < P); idx is a per-lane value: no dynamic register indexing
template <int S>
__device__ __forceinline__ BB<S> bb_bit(int idx)
{
    BB<S> r;
    BB_FOR(k) r.w[k] = ((idx >> 6) == k) ? (1ull << (idx & 63)) : 0ull;
    return r;
}
template <int S>
__device__ __forceinline__ bool bb_test(const BB<S>& a, int idx)
{
    uint64_t word = 0;
    BB_FOR(k) word = ((idx >> 6) == k) ? a.w[k] : word;
    return (word >> (idx & 63)) & 1ull;
}
// the lowest set bit of a (a != 0)
template <int S>
__device__ __forceinline__ BB<S> bb_lowest(const BB<S>& a)
{
    BB<S> r;
    bool found = false;
    BB_FOR(k) {
        const uint64_t low = a.w[k] & (0ull - a.w[k]);
        r.w[k] = found ? 0ull : low;
        found = found || a.w[k] != 0;
    }
    return r;
}

// the on-board 4-neighbourhood of every cell of x
template <int S>
__device__ __forceinline__ BB<S> bb_nbr(const BB<S>& x)
{
    constexpr int NW = BB<S>::NW;
    BB<S> r;
    BB_FOR(k) {
        const uint64_t e = x.w[k] & kBitTab<S>.not_last[k], w = x.w[k] & kBitTab<S>.not_first[k];
        uint64_t v = (e << 1) | (w >> 1) | (x.w[k] << S) | (x.w[k] >> S);
        if (k > 0) v |= ((x.w[k - 1] & kBitTab<S>.not_last[k - 1]) >> 63) | (x.w[k - 1] >> (64 - S));
        if (k < NW - 1) v |= ((x.w[k + 1] & kBitTab<S>.not_first[k + 1]) << 63) | (x.w[k + 1] << (64 - S));
        r.w[k] = v & kBitTab<S>.board[k];
    }
    return r;
}

// the connected component of `seed` inside `mask` (seed is a subset of mask)
template <int S>
__device__ __forceinline__ BB<S> bb_flood(const BB<S>& seed, const BB<S>& mask)
{
    BB<S> g = seed;
    for (;;) {
        const BB<S> n = bb_or(g, bb_and(bb_nbr(g), mask));
        if (bb_eq(n, g)) return g;
        g = n;
    }
}

// Grow g (a subset of one group of `mask`) until it is the whole group or has at least `enough` liberties; returns the
// number of liberties found (the exact count when the group is complete), the liberties themselves in `libs`.
template <int S>
__device__ __forceinline__ int bb_flood_libs(BB<S>& g, const BB<S>& mask, const BB<S>& emp, int enough, BB<S>& libs)
{
    for (;;) {
        const BB<S> nb = bb_nbr(g);
        libs = bb_and(nb, emp);
        const int c = bb_count(libs);
        if (c >= enough) return c;
        const BB<S> n = bb_or(g, bb_and(nb, mask));
        if (bb_eq(n, g)) return c;
        g = n;
    }
}

// areas of a position as given (planes 0 / 1 = a / b): stones plus empty regions touching one colour only
template <int S>
__device__ __forceinline__ void bits_areas(const BB<S>& a, const BB<S>& b, int& area_a, int& area_b)
{
    const BB<S> emp = bb_andnot(bb_board<S>(), bb_or(a, b));
    int a1 = bb_count(a), a2 = bb_count(b);
    BB<S> rest = emp;
    while (bb_any(rest)) {
        const BB<S> r = bb_flood(bb_lowest(rest), emp);
        rest = bb_andnot(rest, r);
        const BB<S> edge = bb_nbr(r);
        const bool ta = bb_any(bb_and(edge, a)), tb = bb_any(bb_and(edge, b));
        const int n = bb_count(r);
        a1 += (ta && !tb) ? n : 0;
        a2 += (tb && !ta) ? n : 0;
    }
    area_a = a1;
    area_b = a2;
}

// GymGo next_state in the mover-relative frame, one board in this lane (the counterpart of board_next_state,
// go_board.h: same inputs, same outputs, same return value).  inv_at_action: the state's invalid-plane bit at `action`.
// AREAS: 0 no area score (sign_out = 0), 1 always, 2 only where the search reads it (a position after a pass, or a finished game)
template <int S, int AREAS = 1>
__device__ __forceinline__ int bits_next_state(const BB<S>& me, const BB<S>& opp, bool inv_at_action, uint32_t flags, int action,
                                               BB<S>& nme, BB<S>& nopp, BB<S>& ninv, uint32_t& nflags, int& sign_out)
{
    constexpr int P = BitGeom<S>::P;
    if (action < 0 || action > P) return -1;
    BB<S> m = me, o = opp, ko = bb_zero<S>();
    if (action == P) {
        nflags = FLAG_PASS | ((flags & FLAG_PASS) ? FLAG_DONE : 0u) | (flags & FLAG_DONE);
    } else {
        const BB<S> a = bb_bit<S>(action);
        if (inv_at_action || bb_any(bb_and(a, bb_or(m, o)))) return -1;
        nflags = flags & FLAG_DONE;
        const BB<S> around = bb_nbr(a);
        const bool surrounded = !bb_any(bb_andnot(around, o));       // every on-board neighbour is an opponent stone
        m = bb_or(m, a);
        const BB<S> emp = bb_andnot(bb_board<S>(), bb_or(m, o));
        // opponent groups next to the new stone that are left without a liberty are captured
        BB<S> cand = bb_and(around, o), cap = bb_zero<S>();
        int ncap = 0;
        while (bb_any(cand)) {
            BB<S> g = bb_lowest(cand), libs;
            const int c = bb_flood_libs(g, o, emp, 1, libs);        // stops at the first liberty (the group may be partial then)
            cand = bb_andnot(cand, g);
            if (c == 0) { cap = bb_or(cap, g); ncap++; }
        }
        o = bb_andnot(o, cap);
        if (ncap == 1 && bb_count(cap) == 1 && surrounded) ko = cap;   // one stone taken by a stone that was surrounded: ko
    }
    // the next mover (stones o) may play on an empty point that has an empty neighbour, or touches one of its own
    // groups with two or more liberties, or touches a group of the other side with exactly one; never on the ko point
    const BB<S> emp = bb_andnot(bb_board<S>(), bb_or(m, o));
    const BB<S> open = bb_nbr(emp);
    BB<S> valid = bb_and(emp, open);
    const BB<S> need = bb_andnot(emp, open);                 // empty points with no empty neighbour: the groups decide
    if (bb_any(need)) {
        const BB<S> near = bb_nbr(need);
        BB<S> cand = bb_and(near, o);
        // (a flood stops as soon as two liberties are known; stones of the same group that it has not reached stay in
        // `cand` and start a flood of their own, so every point next to the group is still seen)
        while (bb_any(cand)) {
            BB<S> g = bb_lowest(cand), libs;
            const int c = bb_flood_libs(g, o, emp, 2, libs);
            cand = bb_andnot(cand, g);
            if (c >= 2) valid = bb_or(valid, libs);
        }
        cand = bb_and(near, m);
        while (bb_any(cand)) {
            BB<S> g = bb_lowest(cand), libs;
            const int c = bb_flood_libs(g, m, emp, 2, libs);
            cand = bb_andnot(cand, g);
            if (c == 1) valid = bb_or(valid, libs);
        }
    }
    ninv = bb_or(bb_andnot(bb_board<S>(), valid), ko);
    sign_out = 0;
    if (AREAS == 1 || (AREAS == 2 && (nflags & (FLAG_PASS | FLAG_DONE)))) {   // area score from the next mover's point of view
        int a1, a2;
        bits_areas<S>(m, o, a1, a2);
        sign_out = (a2 > a1) - (a2 < a1);
    }
    nme = o;
    nopp = m;
    return 0;
}

// ---- the boundary's uint8 records ([6][P] bytes per state) in LDS, one record per lane -----------------------------------
// A record sits at an arbitrary (even) LDS address and its planes at odd offsets: everything below moves four cells per
// LDS access through unaligned ds_read_b32 / ds_write_b32 (gfx950 takes them at any alignment; memcpy lets the compiler
// choose).  P = 1 (mod 4) for every board size, so a plane is (P - 1) / 4 whole words and one last byte.
__device__ __forceinline__ uint32_t lds_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void lds_put_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
// eight bytes at any alignment (one unaligned ds_read_b64 / ds_write_b64)
__device__ __forceinline__ uint64_t lds_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void lds_put_u64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
// bit 7 of every byte of x that is not zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }
// the four flags (bit 7 of each byte) as bits 0..3
__device__ __forceinline__ uint32_t gather4(uint32_t f)
{
    const uint32_t x = f >> 7;                   // flags at bits 0, 8, 16, 24
    const uint32_t t = x | (x >> 7);             // bits 0, 1 and 16, 17
    return (t | (t >> 14)) & 0xfu;
}
// four cells (bits 0..3 of n) as four bytes of 0 / 1
__device__ __forceinline__ uint32_t spread4(uint32_t n) { return __umul24(n & 0xfu, 0x204081u) & 0x01010101u; }

// planes 0 / 1 as bit boards, any(turn plane), any(pass plane), all(done plane) -- GymGo's turn / prev_player_passed /
// game_ended -- and the invalid-plane byte at `action`
template <int S>
__device__ __forceinline__ int record_load(const uint8_t* rec, int action, BB<S>& me, BB<S>& opp, bool& inv_at, uint32_t& flags)
{
    constexpr int P = BitGeom<S>::P, NW = BitGeom<S>::NW, ND = P / 8;      // whole 8-byte pieces per plane; byte P - 1 is left over
    static_assert(P % 8 == 1, "odd board sizes");
    uint32_t c0[2 * NW], c1[2 * NW];
    uint32_t any_turn = 0, any_pass = 0, all_done = 0x80808080u;
    // The stone planes, eight cells per LDS access.  Fast form: for bytes that are 0 / 1 (what every producer of such records
    // writes) the eight cells are two dot products with the bytes (1, 2, 4, 8) / (16, 32, 64, 128) -- v_dot4_u32_u8, three
    // instructions per eight cells instead of ~20.  A record with any other nonzero byte in these planes (the boundary's
    // rule is "nonzero = stone") is redone by the general form below.
    uint64_t seen = 0, dn_and = ~0ull, dn_or = 0;
#pragma unroll
    for (int h = 0; h < 2 * NW; h++) { c0[h] = 0; c1[h] = 0; }
#pragma unroll
    for (int h = 0; h < 2 * NW; h++) {
#pragma unroll
        for (int n = 0; n < 4; n++) {                            // cells 32 h + 8 n .. + 7 -> bits 8 n .. of word h
            const int j = 4 * h + n;
            if (j < ND) {
                const uint64_t a = lds_u64(rec + 8 * j), b = lds_u64(rec + P + 8 * j), t = lds_u64(rec + 2 * P + 8 * j),
                               ps = lds_u64(rec + 4 * P + 8 * j), dn = lds_u64(rec + 5 * P + 8 * j);
                seen |= a | b;
                c0[h] |= __builtin_amdgcn_udot4((uint32_t)(a >> 32), 0x80402010u, __builtin_amdgcn_udot4((uint32_t)a, 0x08040201u, 0u, false), false) << (8 * n);
                c1[h] |= __builtin_amdgcn_udot4((uint32_t)(b >> 32), 0x80402010u, __builtin_amdgcn_udot4((uint32_t)b, 0x08040201u, 0u, false), false) << (8 * n);
                any_turn |= (uint32_t)t | (uint32_t)(t >> 32);
                any_pass |= (uint32_t)ps | (uint32_t)(ps >> 32);
                dn_and &= dn;
                dn_or |= dn;
            }
        }
    }
    // all(done plane != 0): for 0 / 1 bytes the AND of the words; else byte by byte
    if (!(dn_or & 0xfefefefefefefefeull)) {
        all_done = dn_and == 0x0101010101010101ull ? 0x80808080u : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < ND; j++) {
            const uint64_t dn = lds_u64(rec + 5 * P + 8 * j);
            all_done &= nonzero_bytes((uint32_t)dn) & nonzero_bytes((uint32_t)(dn >> 32));
        }
    }
    if (seen & 0xfefefefefefefefeull) {                          // general form: any nonzero byte is a stone
#pragma unroll
        for (int h = 0; h < 2 * NW; h++) { c0[h] = 0; c1[h] = 0; }
#pragma unroll
        for (int h = 0; h < 2 * NW; h++) {
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const int j = 4 * h + n;
                if (j < ND) {
                    const uint64_t a = lds_u64(rec + 8 * j), b = lds_u64(rec + P + 8 * j);
                    c0[h] |= (gather4(nonzero_bytes((uint32_t)a)) | gather4(nonzero_bytes((uint32_t)(a >> 32))) << 4) << (8 * n);
                    c1[h] |= (gather4(nonzero_bytes((uint32_t)b)) | gather4(nonzero_bytes((uint32_t)(b >> 32))) << 4) << (8 * n);
                }
            }
        }
    }
    {   // the last cell of every plane
        constexpr int i = P - 1;
        c0[i >> 5] |= (rec[i] != 0 ? 1u : 0u) << (i & 31);
        c1[i >> 5] |= (rec[P + i] != 0 ? 1u : 0u) << (i & 31);
        any_turn |= rec[2 * P + i];
        any_pass |= rec[4 * P + i];
        if (rec[5 * P + i] == 0) all_done = 0;
    }
    const int player = any_turn ? 1 : 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint64_t a = ((uint64_t)c0[2 * k + 1] << 32) | c0[2 * k], b = ((uint64_t)c1[2 * k + 1] << 32) | c1[2 * k];
        me.w[k] = player ? b : a;
        opp.w[k] = player ? a : b;
    }
    inv_at = action >= 0 && action < P && rec[3 * P + action] != 0;
    flags = (any_pass ? FLAG_PASS : 0u) | (all_done == 0x80808080u ? FLAG_DONE : 0u);
    return player;
}

// a child position (or zeros for an illegal move) as its uint8 record: planes ch0, ch1, turn, invalid, pass, done
template <int S>
__device__ __forceinline__ void record_store(uint8_t* rec, bool legal, const BB<S>& ch0, const BB<S>& ch1, const BB<S>& inv, int turn,
                                             uint32_t flags)
{
    constexpr int P = BitGeom<S>::P, NW = BitGeom<S>::NW, ND = P / 8;
    const uint32_t t = (legal && turn) ? 0x01010101u : 0u, ps = (legal && (flags & FLAG_PASS)) ? 0x01010101u : 0u,
                   dn = (legal && (flags & FLAG_DONE)) ? 0x01010101u : 0u;
    const uint64_t t8 = ((uint64_t)t << 32) | t, ps8 = ((uint64_t)ps << 32) | ps, dn8 = ((uint64_t)dn << 32) | dn;
    auto spread8 = [](uint32_t n) { return ((uint64_t)spread4(n >> 4) << 32) | spread4(n); };
    uint32_t a[2 * NW], b[2 * NW], v[2 * NW];
#pragma unroll
    for (int h = 0; h < 2 * NW; h++) {
        a[h] = legal ? (uint32_t)(ch0.w[h >> 1] >> (32 * (h & 1))) : 0u;
        b[h] = legal ? (uint32_t)(ch1.w[h >> 1] >> (32 * (h & 1))) : 0u;
        v[h] = legal ? (uint32_t)(inv.w[h >> 1] >> (32 * (h & 1))) : 0u;
    }
#pragma unroll
    for (int h = 0; h < 2 * NW; h++) {
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const int j = 4 * h + n;
            if (j < ND) {
                lds_put_u64(rec + 8 * j, spread8(a[h] >> (8 * n)));
                lds_put_u64(rec + P + 8 * j, spread8(b[h] >> (8 * n)));
                lds_put_u64(rec + 2 * P + 8 * j, t8);
                lds_put_u64(rec + 3 * P + 8 * j, spread8(v[h] >> (8 * n)));
                lds_put_u64(rec + 4 * P + 8 * j, ps8);
                lds_put_u64(rec + 5 * P + 8 * j, dn8);
            }
        }
    }
    constexpr int i = P - 1;
    rec[i] = (uint8_t)((a[i >> 5] >> (i & 31)) & 1u);
    rec[P + i] = (uint8_t)((b[i >> 5] >> (i & 31)) & 1u);
    rec[2 * P + i] = (uint8_t)(t & 1u);
    rec[3 * P + i] = (uint8_t)((v[i >> 5] >> (i & 31)) & 1u);
    rec[4 * P + i] = (uint8_t)(ps & 1u);
    rec[5 * P + i] = (uint8_t)(dn & 1u);
}

}  // namespace goai
