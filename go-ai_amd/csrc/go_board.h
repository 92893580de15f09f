// go_board.h -- one Go board per 64-lane wavefront, staged in LDS.
//
// Replaces the GymGo routines the reference calls on the MCTS path
// (/root/reference/go_ai/search/tree.py:92,108,115; /root/reference/go_ai/data.py:15-60):
// next_state, invalid-move plane (occupied + ko + suicide), captures, area scoring.
//
// Design (gfx950): a workgroup is exactly ONE wavefront (64 lanes), so every barrier is a
// single-wave barrier and all control flow below is wave-uniform.  Lane l owns cells
// l, l+64, l+128, ...  A bit plane of the board is therefore K = ceil(S*S/64) __ballot()
// words; those words are the packed state format kept in HBM (struct-of-words per node:
// me[K], opp[K], inv[K], flags).  The board itself lives in LDS as a (S+2)x(S+2) byte grid
// with a border so neighbour reads never branch; groups are found by min-label propagation
// with pointer jumping over that grid, liberties / territory claims by LDS atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace goai {

enum : uint32_t { FLAG_PASS = 1u, FLAG_DONE = 2u, FLAG_TURN = 4u };

template <int S>
struct Geom {
    static constexpr int P = S * S;
    static constexpr int A = P + 1;
    static constexpr int K = (P + 63) / 64;   // ballot words per plane
    static constexpr int PS = S + 2;          // padded row stride
    static constexpr int NPAD = PS * PS;
    static constexpr int NPAD4 = (NPAD + 3) / 4 * 4;
    static constexpr int SW = 3 * K + 1;      // u64 words per packed state
};

template <int S>
struct BoardLds {
    uint8_t col[Geom<S>::NPAD4];     // 0 empty, 1 mover ("me"), 2 opponent, 3 border
    uint16_t lab[Geom<S>::NPAD4];    // component label = smallest padded index in the component
    uint32_t aux[Geom<S>::NPAD4];    // stone root: liberty count; empty root: colour-claim mask
};

template <int S>
__device__ __forceinline__ int cell_q(int idx)
{
    return (idx / S + 1) * Geom<S>::PS + (idx % S) + 1;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Write the border once per wave.
template <int S>
__device__ __forceinline__ void board_init_border(BoardLds<S>& L)
{
    for (int i = lane_id(); i < Geom<S>::NPAD4; i += 64) L.col[i] = 3;
    __syncthreads();
}

// Fill the interior from packed bit planes (mover-relative).
template <int S>
__device__ __forceinline__ void board_load(BoardLds<S>& L, const uint64_t* me, const uint64_t* opp)
{
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < Geom<S>::K; k++) {
        int idx = lane + 64 * k;
        if (idx < Geom<S>::P) {
            uint32_t m = (uint32_t)(me[k] >> lane) & 1u, o = (uint32_t)(opp[k] >> lane) & 1u;
            L.col[cell_q<S>(idx)] = (uint8_t)(m ? 1 : (o ? 2 : 0));
        }
    }
    __syncthreads();
}

// Connected components of equal colour (empty cells included), liberties of stone groups
// and colour claims of empty regions.
template <int S>
__device__ __forceinline__ void board_analyze(BoardLds<S>& L)
{
    using G = Geom<S>;
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        int idx = lane + 64 * k;
        if (idx < G::P) { int q = cell_q<S>(idx); L.lab[q] = (uint16_t)q; L.aux[q] = 0; }
    }
    __syncthreads();
    // Colours do not change during the propagation: which neighbours belong to the cell's component
    // is decided once, and every sweep then issues its five label reads back to back (no branches
    // around LDS reads: a chain of dependent conditional reads was most of this function's time).
    int qs[G::K];
    uint32_t same[G::K];
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int idx = lane + 64 * k;
        qs[k] = -1;
        same[k] = 0;
        if (idx < G::P) {
            const int q = cell_q<S>(idx);
            const uint8_t c = L.col[q], c0 = L.col[q - 1], c1 = L.col[q + 1], c2 = L.col[q - G::PS], c3 = L.col[q + G::PS];
            qs[k] = q;
            same[k] = (c0 == c ? 1u : 0u) | (c1 == c ? 2u : 0u) | (c2 == c ? 4u : 0u) | (c3 == c ? 8u : 0u);
        }
    }
    bool any_changed;
    do {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            if (qs[k] >= 0) {
                const int q = qs[k];
                const int cur = L.lab[q];
                const int l0 = L.lab[q - 1], l1 = L.lab[q + 1], l2 = L.lab[q - G::PS], l3 = L.lab[q + G::PS];
                int m = cur;
                m = (same[k] & 1u) ? min(m, l0) : m;
                m = (same[k] & 2u) ? min(m, l1) : m;
                m = (same[k] & 4u) ? min(m, l2) : m;
                m = (same[k] & 8u) ? min(m, l3) : m;
                m = L.lab[m];   // pointer jump: lab[m] <= m and lies in the same component
                if (m < cur) { L.lab[q] = (uint16_t)m; ch = true; }
            }
        }
        __syncthreads();
        any_changed = __ballot(ch) != 0ull;
    } while (any_changed);

#pragma unroll
    for (int k = 0; k < G::K; k++) {
        if (qs[k] >= 0) {
            const int q = qs[k];
            const uint8_t c = L.col[q];
            const int nq[4] = { q - 1, q + 1, q - G::PS, q + G::PS };
            uint8_t cn[4];
            int ln[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { cn[i] = L.col[nq[i]]; ln[i] = L.lab[nq[i]]; }   // (border labels are never used)
            if (c == 0) {
                int r[4];
#pragma unroll
                for (int i = 0; i < 4; i++) r[i] = (cn[i] == 1 || cn[i] == 2) ? ln[i] : -1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bool dup = r[i] < 0;
#pragma unroll
                    for (int j = 0; j < i; j++) dup = dup || (r[j] == r[i]);
                    if (!dup) atomicAdd(&L.aux[r[i]], 1u);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (cn[i] == 0) atomicOr(&L.aux[ln[i]], (uint32_t)c);
            }
        }
    }
    __syncthreads();
}

// Area score of the analysed position: stones plus empty regions claimed by one colour.
template <int S>
__device__ __forceinline__ void board_areas(BoardLds<S>& L, int& me_area, int& opp_area)
{
    using G = Geom<S>;
    const int lane = lane_id();
    int a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        int idx = lane + 64 * k;
        bool m = false, o = false;
        if (idx < G::P) {
            int q = cell_q<S>(idx);
            uint8_t c = L.col[q];
            uint32_t claim = (c == 0) ? L.aux[L.lab[q]] : (uint32_t)c;
            m = claim == 1u; o = claim == 2u;
        }
        a1 += __popcll(__ballot(m));
        a2 += __popcll(__ballot(o));
    }
    me_area = a1; opp_area = a2;
}

// Invalid-move plane for the player with colour code `next_col` (the other colour is
// 3 - next_col) on the analysed position; `ko_q` = padded index of the ko point or -1.
template <int S>
__device__ __forceinline__ void board_invalid(BoardLds<S>& L, int next_col, int ko_q, uint64_t* inv)
{
    using G = Geom<S>;
    const int lane = lane_id();
    const int other = 3 - next_col;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        int idx = lane + 64 * k;
        bool bad = false;
        if (idx < G::P) {
            int q = cell_q<S>(idx);
            const int nq[4] = { q - 1, q + 1, q - G::PS, q + G::PS };
            uint8_t cn[4];
            uint32_t an[4];
#pragma unroll
            for (int i = 0; i < 4; i++) cn[i] = L.col[nq[i]];
#pragma unroll
            for (int i = 0; i < 4; i++) {                    // liberties of the neighbouring group (own cell for non-stones)
                const int lb = L.lab[nq[i]];
                const int root = (cn[i] == 1 || cn[i] == 2) ? lb : q;
                an[i] = L.aux[root];
            }
            if (L.col[q] != 0 || q == ko_q) bad = true;
            else {
                bool ok = false;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (cn[i] == 0) ok = true;
                    else if (cn[i] == next_col) ok = ok || (an[i] >= 2u);
                    else if (cn[i] == other) ok = ok || (an[i] == 1u);
                }
                bad = !ok;
            }
        }
        inv[k] = __ballot(bad);
    }
}

// GymGo next_state in the mover-relative frame.  Inputs: packed planes of the state
// (me = stones of the side to move) and its flags; action in [0, P].  Outputs are in the
// NEXT mover's frame (canonical form).  Leaves the analysed child position in LDS with
// colour codes of the OLD frame (1 = old mover, 2 = old opponent = next mover) and returns
// the child's area score sign from the next mover's point of view in `sign_out`.
// Returns 0, or -1 if the action is illegal (nothing written).
template <int S>
__device__ __forceinline__ int board_next_state(BoardLds<S>& L, const uint64_t* me, const uint64_t* opp,
                                                const uint64_t* inv, uint32_t flags, int action,
                                                uint64_t* nme, uint64_t* nopp, uint64_t* ninv,
                                                uint32_t& nflags, int& sign_out)
{
    using G = Geom<S>;
    const int lane = lane_id();
    int ko_q = -1;
    if (action < 0 || action > G::P) return -1;
    if (action < G::P) {
        int k = action >> 6, b = action & 63;
        if (((inv[k] | me[k] | opp[k]) >> b) & 1ull) return -1;
    }
    board_load<S>(L, me, opp);
    if (action == G::P) {
        nflags = FLAG_PASS | ((flags & FLAG_PASS) ? FLAG_DONE : 0u) | (flags & FLAG_DONE);
        board_analyze<S>(L);
    } else {
        nflags = flags & FLAG_DONE;
        const int qa = cell_q<S>(action);
        const int nq[4] = { qa - 1, qa + 1, qa - G::PS, qa + G::PS };
        bool surrounded = true;
#pragma unroll
        for (int i = 0; i < 4; i++) surrounded = surrounded && (L.col[nq[i]] >= 2);
        __syncthreads();
        if (lane == 0) L.col[qa] = 1;
        __syncthreads();
        board_analyze<S>(L);
        int roots[4], nk = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int r = -1;
            if (L.col[nq[i]] == 2) { int lr = L.lab[nq[i]]; if (L.aux[lr] == 0u) r = lr; }
#pragma unroll
            for (int j = 0; j < i; j++) if (roots[j] == r) r = -1;
            roots[i] = r;
            nk += (r >= 0);
        }
        if (nk > 0) {
            int total = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < G::K; k++) {
                int idx = lane + 64 * k;
                bool dead = false;
                if (idx < G::P) {
                    int q = cell_q<S>(idx);
                    if (L.col[q] == 2) {
                        int lr = L.lab[q];
                        dead = (lr == roots[0]) || (lr == roots[1]) || (lr == roots[2]) || (lr == roots[3]);
                    }
                    if (dead) L.col[q] = 0;
                }
                total += __popcll(__ballot(dead));
            }
            __syncthreads();
            if (nk == 1 && total == 1 && surrounded) {
#pragma unroll
                for (int i = 0; i < 4; i++) if (roots[i] >= 0) ko_q = roots[i];
            }
            board_analyze<S>(L);
        }
    }
    board_invalid<S>(L, 2, ko_q, ninv);
    int a1, a2;
    board_areas<S>(L, a1, a2);
    sign_out = (a2 > a1) - (a2 < a1);
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        int idx = lane + 64 * k;
        uint8_t c = (idx < G::P) ? L.col[cell_q<S>(idx)] : (uint8_t)3;
        nme[k] = __ballot(c == 2);
        nopp[k] = __ballot(c == 1);
    }
    return 0;
}

// Write a packed state as six uint8 planes [6][P] (the reference's GymGo layout).
// ch0/ch1 ordering and the turn plane are chosen by the caller.
template <int S>
__device__ __forceinline__ void planes_store(uint8_t* out, const uint64_t* ch0, const uint64_t* ch1,
                                             const uint64_t* inv, int turn, uint32_t flags)
{
    using G = Geom<S>;
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        int idx = lane + 64 * k;
        if (idx < G::P) {
            out[0 * G::P + idx] = (uint8_t)((ch0[k] >> lane) & 1ull);
            out[1 * G::P + idx] = (uint8_t)((ch1[k] >> lane) & 1ull);
            out[2 * G::P + idx] = (uint8_t)turn;
            out[3 * G::P + idx] = (uint8_t)((inv[k] >> lane) & 1ull);
            out[4 * G::P + idx] = (uint8_t)((flags & FLAG_PASS) ? 1 : 0);
            out[5 * G::P + idx] = (uint8_t)((flags & FLAG_DONE) ? 1 : 0);
        }
    }
}

// Read six uint8 planes into packed mover-relative words.  Returns the colour index
// (0 black / 1 white) of the side to move.
template <int S>
__device__ __forceinline__ int planes_load(const uint8_t* in, uint64_t* me, uint64_t* opp, uint64_t* inv,
                                           uint32_t& flags)
{
    using G = Geom<S>;
    const int lane = lane_id();
    uint64_t t = 0, pz = 0, c0[G::K], c1[G::K];
    int ncell_done = 0;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        int idx = lane + 64 * k;
        bool in_b = idx < G::P;
        c0[k] = __ballot(in_b && in[0 * G::P + idx] != 0);
        c1[k] = __ballot(in_b && in[1 * G::P + idx] != 0);
        t |= __ballot(in_b && in[2 * G::P + idx] != 0);
        inv[k] = __ballot(in_b && in[3 * G::P + idx] != 0);
        pz |= __ballot(in_b && in[4 * G::P + idx] != 0);
        uint64_t d = __ballot(in_b && in[5 * G::P + idx] != 0);
        ncell_done += __popcll(d);
    }
    int player = t ? 1 : 0;
#pragma unroll
    for (int k = 0; k < G::K; k++) { me[k] = player ? c1[k] : c0[k]; opp[k] = player ? c0[k] : c1[k]; }
    flags = (pz ? FLAG_PASS : 0u) | ((ncell_done == G::P) ? FLAG_DONE : 0u);
    return player;
}

}  // namespace goai
