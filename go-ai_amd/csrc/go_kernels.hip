// go_kernels.hip -- stateless Go-rule kernels behind goai_transition / goai_children /
// goai_score.  One board (or one (board, action) pair) per 64-lane workgroup; see go_board.h.
//
// Boundary format = the reference's GymGo state: uint8 planes [n][6][S][S]
// (/root/reference/go_ai/data.py:15-24 reads plane 3 as the invalid-move mask).
#include <cstdlib>
#include <string>

#include "go_bits.h"
#include "go_board.h"
#include "goai_internal.h"

namespace goai {

// GoGame.next_state(state, action, canonical) -- /root/reference/go_ai/search/tree.py:115
template <int S>
__global__ __launch_bounds__(64) void k_transition(const uint8_t* __restrict__ states,
                                                   const int32_t* __restrict__ actions, int canonical,
                                                   uint8_t* __restrict__ out, int32_t* __restrict__ status, int n)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const uint8_t* in = states + (size_t)i * 6 * G::P;
        uint8_t* o = out + (size_t)i * 6 * G::P;
        uint64_t me[G::K], opp[G::K], inv[G::K], nme[G::K], nopp[G::K], ninv[G::K];
        uint32_t flags, nflags = 0;
        int sign;
        int player = planes_load<S>(in, me, opp, inv, flags);
        int rc = board_next_state<S>(L, me, opp, inv, flags, actions[i], nme, nopp, ninv, nflags, sign);
        if (rc == 0) {
            int np = 1 - player;   // colour of the next mover
            if (canonical || np == 0) planes_store<S>(o, nme, nopp, ninv, canonical ? 0 : np, nflags);
            else planes_store<S>(o, nopp, nme, ninv, np, nflags);
        } else {
            for (int j = lane_id(); j < 6 * G::P; j += 64) o[j] = 0;
        }
        if (lane_id() == 0 && status) status[i] = rc;
        __syncthreads();
    }
}

// GoGame.children(state, canonical, padded=True) + data.batch_win_children
// (/root/reference/go_ai/data.py:41-60): one workgroup per (state, action).
template <int S>
__global__ __launch_bounds__(64) void k_children(const uint8_t* __restrict__ states, int canonical,
                                                 uint8_t* __restrict__ children, uint8_t* __restrict__ valid,
                                                 int8_t* __restrict__ win, int n)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const long total = (long)n * G::A;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const int i = (int)(w / G::A), a = (int)(w % G::A);
        const uint8_t* in = states + (size_t)i * 6 * G::P;
        uint8_t* o = children + (size_t)w * 6 * G::P;
        uint64_t me[G::K], opp[G::K], inv[G::K], nme[G::K], nopp[G::K], ninv[G::K];
        uint32_t flags, nflags = 0;
        int sign = 0;
        int player = planes_load<S>(in, me, opp, inv, flags);
        int rc = board_next_state<S>(L, me, opp, inv, flags, a, nme, nopp, ninv, nflags, sign);
        if (rc == 0) {
            int np = 1 - player;
            if (canonical || np == 0) planes_store<S>(o, nme, nopp, ninv, canonical ? 0 : np, nflags);
            else planes_store<S>(o, nopp, nme, ninv, np, nflags);
        } else {
            for (int j = lane_id(); j < 6 * G::P; j += 64) o[j] = 0;
        }
        if (lane_id() == 0) {
            if (valid) valid[w] = rc == 0;
            if (win) {
                // winning(child) reads planes 0/1 as stored: the next mover's frame if
                // canonical (or if the next mover is black), else the absolute frame.
                int s_abs = (canonical || player == 1) ? sign : -sign;
                win[w] = (int8_t)((rc == 0 && (nflags & FLAG_DONE)) ? s_abs : 0);
            }
        }
        __syncthreads();
    }
}

// GoGame.areas / game_ended / winning -- /root/reference/go_ai/search/tree.py:65-69
template <int S>
__global__ __launch_bounds__(64) void k_score(const uint8_t* __restrict__ states, int32_t* __restrict__ black_area,
                                              int32_t* __restrict__ white_area, int32_t* __restrict__ ended,
                                              int32_t* __restrict__ winner, int n)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const uint8_t* in = states + (size_t)i * 6 * G::P;
        uint64_t me[G::K], opp[G::K], inv[G::K];
        uint32_t flags;
        int player = planes_load<S>(in, me, opp, inv, flags);
        board_load<S>(L, me, opp);
        board_analyze<S>(L);
        int a_me, a_opp;
        board_areas<S>(L, a_me, a_opp);
        if (lane_id() == 0) {
            int b = player ? a_opp : a_me, w = player ? a_me : a_opp;   // planes 0 / 1 as given
            if (black_area) black_area[i] = b;
            if (white_area) white_area[i] = w;
            if (ended) ended[i] = (flags & FLAG_DONE) ? 1 : 0;
            if (winner) winner[i] = (b > w) - (b < w);
        }
        __syncthreads();
    }
}

// GoGame.get_num_liberties(state) -- /root/reference/go_ai/policies/baselines.py:41: per colour, the
// number of distinct empty points next to any stone of that colour (planes 0 / 1 as given).
template <int S>
__global__ __launch_bounds__(64) void k_liberties(const uint8_t* __restrict__ states, int32_t* __restrict__ black_libs,
                                                  int32_t* __restrict__ white_libs, int n)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const int lane = lane_id();
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const uint8_t* in = states + (size_t)i * 6 * G::P;
        uint64_t me[G::K], opp[G::K], inv[G::K];
        uint32_t flags;
        const int player = planes_load<S>(in, me, opp, inv, flags);
        board_load<S>(L, me, opp);
        int n1 = 0, n2 = 0;                       // empty points next to colour code 1 ("me") / 2 ("opp")
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int idx = lane + 64 * k;
            bool a1 = false, a2 = false;
            if (idx < G::P) {
                const int q = cell_q<S>(idx);
                if (L.col[q] == 0) {
                    const uint8_t c0 = L.col[q - 1], c1 = L.col[q + 1], c2 = L.col[q - G::PS], c3 = L.col[q + G::PS];
                    a1 = c0 == 1 || c1 == 1 || c2 == 1 || c3 == 1;
                    a2 = c0 == 2 || c1 == 2 || c2 == 2 || c3 == 2;
                }
            }
            n1 += __popcll(__ballot(a1));
            n2 += __popcll(__ballot(a2));
        }
        if (lane == 0) {
            if (black_libs) black_libs[i] = player ? n2 : n1;
            if (white_libs) white_libs[i] = player ? n1 : n2;
        }
        __syncthreads();
    }
}

// ---- one board per LANE (go_bits.h) -----------------------------------------------------------------------------------
// A wave owns 64 consecutive work items.  The uint8 planes of the boundary are read and written by the whole wave,
// one board at a time (ballots in, bit tests out: coalesced accesses), the rules run in the lanes' registers.
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int j)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), j);
    return ((uint64_t)hi << 32) | lo;
}

// lane j's child position -> its six planes (or all zeros if the move was illegal)
template <int S>
__device__ __forceinline__ void lane_child_store(uint8_t* o, int j, const BB<S>& nme, const BB<S>& nopp, const BB<S>& ninv,
                                                 uint32_t nflags, int rc, int player, int canonical)
{
    using G = Geom<S>;
    uint64_t a[G::K], b[G::K], iv[G::K];
#pragma unroll
    for (int k = 0; k < G::K; k++) { a[k] = readlane64(nme.w[k], j); b[k] = readlane64(nopp.w[k], j); iv[k] = readlane64(ninv.w[k], j); }
    const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)nflags, j);
    const int r = __builtin_amdgcn_readlane(rc, j), np = 1 - __builtin_amdgcn_readlane(player, j);
    if (r == 0) {
        if (canonical || np == 0) planes_store<S>(o, a, b, iv, canonical ? 0 : np, f);
        else planes_store<S>(o, b, a, iv, np, f);
    } else {
        for (int q = lane_id(); q < 6 * G::P; q += 64) o[q] = 0;
    }
}

template <int S>
__global__ __launch_bounds__(256) void k_transition_lanes(const uint8_t* __restrict__ states, const int32_t* __restrict__ actions,
                                                          int canonical, uint8_t* __restrict__ out, int32_t* __restrict__ status, int n)
{
    using G = Geom<S>;
    const int lane = lane_id();
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (first >= n) return;
    const int nb = (int)min((long)64, (long)n - first);
    const int action = lane < nb ? actions[first + lane] : -1;
    BB<S> me = bb_zero<S>(), opp = bb_zero<S>();
    bool inv_at = false;
    uint32_t flags = 0;
    int player = 0;
    for (int j = 0; j < nb; j++) {
        uint64_t m[G::K], o[G::K], iv[G::K];
        uint32_t f;
        const int pl = planes_load<S>(states + (size_t)(first + j) * 6 * G::P, m, o, iv, f);
        if (lane == j) {
            BB<S> ivb;
#pragma unroll
            for (int k = 0; k < G::K; k++) { me.w[k] = m[k]; opp.w[k] = o[k]; ivb.w[k] = iv[k]; }
            inv_at = action >= 0 && action < G::P && bb_test(ivb, action);
            flags = f;
            player = pl;
        }
    }
    BB<S> nme = bb_zero<S>(), nopp = bb_zero<S>(), ninv = bb_zero<S>();
    uint32_t nflags = 0;
    int sign = 0;
    const int rc = bits_next_state<S, 0>(me, opp, inv_at, flags, action, nme, nopp, ninv, nflags, sign);
    for (int j = 0; j < nb; j++)
        lane_child_store<S>(out + (size_t)(first + j) * 6 * G::P, j, nme, nopp, ninv, nflags, rc, player, canonical);
    if (lane < nb && status) status[first + lane] = rc;
}

// one lane per (state, action)
template <int S>
__global__ __launch_bounds__(256) void k_children_lanes(const uint8_t* __restrict__ states, int canonical, uint8_t* __restrict__ children,
                                                        uint8_t* __restrict__ valid, int8_t* __restrict__ win, int n)
{
    using G = Geom<S>;
    const int lane = lane_id();
    const long total = (long)n * G::A;
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (first >= total) return;
    const int nb = (int)min((long)64, total - first);
    const long w = first + lane;
    const int my_board = lane < nb ? (int)(w / G::A) : -1, action = lane < nb ? (int)(w % G::A) : -1;
    BB<S> me = bb_zero<S>(), opp = bb_zero<S>();
    bool inv_at = false;
    uint32_t flags = 0;
    int player = 0;
    const int b0 = (int)(first / G::A), b1 = (int)((first + nb - 1) / G::A);
    for (int b = b0; b <= b1; b++) {                       // the few parents this wave's items come from
        uint64_t m[G::K], o[G::K], iv[G::K];
        uint32_t f;
        const int pl = planes_load<S>(states + (size_t)b * 6 * G::P, m, o, iv, f);
        if (my_board == b) {
            BB<S> ivb;
#pragma unroll
            for (int k = 0; k < G::K; k++) { me.w[k] = m[k]; opp.w[k] = o[k]; ivb.w[k] = iv[k]; }
            inv_at = action < G::P && bb_test(ivb, action);
            flags = f;
            player = pl;
        }
    }
    BB<S> nme = bb_zero<S>(), nopp = bb_zero<S>(), ninv = bb_zero<S>();
    uint32_t nflags = 0;
    int sign = 0;
    const int rc = bits_next_state<S, 2>(me, opp, inv_at, flags, action, nme, nopp, ninv, nflags, sign);
    for (int j = 0; j < nb; j++)
        lane_child_store<S>(children + (size_t)(first + j) * 6 * G::P, j, nme, nopp, ninv, nflags, rc, player, canonical);
    if (lane < nb) {
        if (valid) valid[w] = rc == 0;
        if (win) {
            const int s_abs = (canonical || player == 1) ? sign : -sign;       // as k_children
            win[w] = (int8_t)((rc == 0 && (nflags & FLAG_DONE)) ? s_abs : 0);
        }
    }
}

// areas / ended / winner (score != 0) or the liberty counts of get_num_liberties (score == 0), one lane per state
template <int S, bool SCORE>
__global__ __launch_bounds__(256) void k_score_lanes(const uint8_t* __restrict__ states, int32_t* __restrict__ o0, int32_t* __restrict__ o1,
                                                     int32_t* __restrict__ ended, int32_t* __restrict__ winner, int n)
{
    using G = Geom<S>;
    const int lane = lane_id();
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (first >= n) return;
    const int nb = (int)min((long)64, (long)n - first);
    BB<S> me = bb_zero<S>(), opp = bb_zero<S>();
    uint32_t flags = 0;
    int player = 0;
    for (int j = 0; j < nb; j++) {
        uint64_t m[G::K], o[G::K], iv[G::K];
        uint32_t f;
        const int pl = planes_load<S>(states + (size_t)(first + j) * 6 * G::P, m, o, iv, f);
        if (lane == j) {
#pragma unroll
            for (int k = 0; k < G::K; k++) { me.w[k] = m[k]; opp.w[k] = o[k]; }
            flags = f;
            player = pl;
        }
    }
    if (lane >= nb) return;
    int a_me, a_opp;
    if (SCORE) bits_areas<S>(me, opp, a_me, a_opp);
    else {
        const BB<S> emp = bb_andnot(bb_board<S>(), bb_or(me, opp));
        a_me = bb_count(bb_and(bb_nbr(me), emp));
        a_opp = bb_count(bb_and(bb_nbr(opp), emp));
    }
    const int b = player ? a_opp : a_me, w = player ? a_me : a_opp;      // planes 0 / 1 as given
    if (o0) o0[first + lane] = b;
    if (o1) o1[first + lane] = w;
    if (SCORE) {
        if (ended) ended[first + lane] = (flags & FLAG_DONE) ? 1 : 0;
        if (winner) winner[first + lane] = (b > w) - (b < w);
    }
}

// ---- one board per lane, records staged through LDS --------------------------------------------------------------------
// The same per-lane rules with the boundary's byte records moved as whole 16-byte lines: a wave copies the records of
// NBL consecutive states (one contiguous, 16-byte aligned stretch of HBM) into LDS, every lane packs / unpacks its own
// record there, and results leave the same way.  For 16-byte aligned `states` / `out` (the launchers check).
// NBL = records per LDS pass.  A pass costs every lane of the wave the pack / unpack instructions of one record, whichever
// lanes own a record of it, so few large passes are cheapest per wave; but LDS per wave bounds the waves per CU, and a
// lone wave per SIMD spends most of its time waiting for its own instructions.  The launchers take large passes for small
// launches (every wave gets a SIMD to itself anyway) and small ones when there are waves enough to fill the SIMDs.
template <int S, int NBL_>
struct Staged {
    static constexpr int REC = 6 * S * S;
    static constexpr int NBL = NBL_;
    static constexpr int LDS_U4 = (NBL * REC + 15) / 16 + 1;               // + 16 bytes: the last record's last word reads 3 bytes on
    static_assert((NBL * REC) % 16 == 0, "a pass is whole 16-byte units");
};

// HBM -> LDS: `bytes` bytes from src (16-byte aligned); only the last 16-byte unit of the whole array can be partial
__device__ __forceinline__ void stage_in(uint4* lds, const uint8_t* src, long bytes)
{
    const int lane = lane_id();
    for (long c = lane; c * 16 < bytes; c += 64) {
        if ((c + 1) * 16 <= bytes) lds[c] = *reinterpret_cast<const uint4*>(src + c * 16);
        else
            for (long t = c * 16; t < bytes; t++) reinterpret_cast<uint8_t*>(lds)[t] = src[t];
    }
}
__device__ __forceinline__ void stage_out(uint8_t* dst, const uint4* lds, long bytes)
{
    const int lane = lane_id();
    for (long c = lane; c * 16 < bytes; c += 64) {
        if ((c + 1) * 16 <= bytes) *reinterpret_cast<uint4*>(dst + c * 16) = lds[c];
        else
            for (long t = c * 16; t < bytes; t++) dst[t] = reinterpret_cast<const uint8_t*>(lds)[t];
    }
}

template <int S, int NBL>
__global__ __launch_bounds__(64) void k_transition_staged(const uint8_t* __restrict__ states, const int32_t* __restrict__ actions,
                                                          int canonical, uint8_t* __restrict__ out, int32_t* __restrict__ status, int n)
{
    using T = Staged<S, NBL>;
    __shared__ uint4 lds[T::LDS_U4];
    uint8_t* bytes = reinterpret_cast<uint8_t*>(lds);
    const int lane = lane_id();
    const long first = (long)blockIdx.x * 64;
    const int nb = (int)min((long)64, (long)n - first);
    const int action = lane < nb ? actions[first + lane] : -1;
    BB<S> me = bb_zero<S>(), opp = bb_zero<S>();
    bool inv_at = false;
    uint32_t flags = 0;
    int player = 0;
#pragma unroll 1
    for (int q = 0; q * T::NBL < nb; q++) {
        const int cnt = min(T::NBL, nb - q * T::NBL);
        stage_in(lds, states + (size_t)(first + q * T::NBL) * T::REC, (long)cnt * T::REC);
        __syncthreads();
        const int l = lane - q * T::NBL;
        if (l >= 0 && l < cnt) player = record_load<S>(bytes + l * T::REC, action, me, opp, inv_at, flags);
        __syncthreads();
    }
    BB<S> nme = bb_zero<S>(), nopp = bb_zero<S>(), ninv = bb_zero<S>();
    uint32_t nflags = 0;
    int sign = 0;
#ifdef GOAI_GO_DIAG_NO_RULES      // diagnostic build only (wrong results): the records' way in and out without the rules between them
    const int rc = 0;
    nme = opp; nopp = me; ninv = me; nflags = flags; sign = inv_at;
#else
    const int rc = bits_next_state<S, 0>(me, opp, inv_at, flags, action, nme, nopp, ninv, nflags, sign);
#endif
    const int np = 1 - player;
    const bool keep = canonical || np == 0;                     // the next mover's stones in plane 0
#pragma unroll 1
    for (int q = 0; q * T::NBL < nb; q++) {
        const int cnt = min(T::NBL, nb - q * T::NBL);
        const int l = lane - q * T::NBL;
        if (l >= 0 && l < cnt) record_store<S>(bytes + l * T::REC, rc == 0, keep ? nme : nopp, keep ? nopp : nme, ninv, canonical ? 0 : np, nflags);
        __syncthreads();
        stage_out(out + (size_t)(first + q * T::NBL) * T::REC, lds, (long)cnt * T::REC);
        __syncthreads();
    }
    if (lane < nb && status) status[first + lane] = rc;
}

template <int S, int NBL>
__global__ __launch_bounds__(64) void k_children_staged(const uint8_t* __restrict__ states, int canonical, uint8_t* __restrict__ children,
                                                        uint8_t* __restrict__ valid, int8_t* __restrict__ win, int n)
{
    using G = Geom<S>;
    using T = Staged<S, NBL>;
    __shared__ uint4 lds[T::LDS_U4];
    uint8_t* bytes = reinterpret_cast<uint8_t*>(lds);
    const int lane = lane_id();
    const long total = (long)n * G::A;
    const long first = (long)blockIdx.x * 64;
    const int nb = (int)min((long)64, total - first);
    const long w = first + lane;
    const int my_board = lane < nb ? (int)(w / G::A) : -1, action = lane < nb ? (int)(w % G::A) : -1;
    BB<S> me = bb_zero<S>(), opp = bb_zero<S>();
    bool inv_at = false;
    uint32_t flags = 0;
    int player = 0;
    const int b0 = (int)(first / G::A), b1 = (int)((first + nb - 1) / G::A);
    for (int b = b0; b <= b1; b++) {                       // the few parents this wave's items come from
        uint64_t m[G::K], o[G::K], iv[G::K];
        uint32_t f;
        const int pl = planes_load<S>(states + (size_t)b * 6 * G::P, m, o, iv, f);
        if (my_board == b) {
            BB<S> ivb;
#pragma unroll
            for (int k = 0; k < G::K; k++) { me.w[k] = m[k]; opp.w[k] = o[k]; ivb.w[k] = iv[k]; }
            inv_at = action < G::P && bb_test(ivb, action);
            flags = f;
            player = pl;
        }
    }
    BB<S> nme = bb_zero<S>(), nopp = bb_zero<S>(), ninv = bb_zero<S>();
    uint32_t nflags = 0;
    int sign = 0;
    const int rc = bits_next_state<S, 2>(me, opp, inv_at, flags, action, nme, nopp, ninv, nflags, sign);
    const int np = 1 - player;
    const bool keep = canonical || np == 0;
#pragma unroll 1
    for (int q = 0; q * T::NBL < nb; q++) {
        const int cnt = min(T::NBL, nb - q * T::NBL);
        const int l = lane - q * T::NBL;
        if (l >= 0 && l < cnt) record_store<S>(bytes + l * T::REC, rc == 0, keep ? nme : nopp, keep ? nopp : nme, ninv, canonical ? 0 : np, nflags);
        __syncthreads();
        stage_out(children + (size_t)(first + q * T::NBL) * T::REC, lds, (long)cnt * T::REC);
        __syncthreads();
    }
    if (lane < nb) {
        if (valid) valid[w] = rc == 0;
        if (win) {
            const int s_abs = (canonical || player == 1) ? sign : -sign;       // as k_children
            win[w] = (int8_t)((rc == 0 && (nflags & FLAG_DONE)) ? s_abs : 0);
        }
    }
}

// SCORE != 0: areas / ended / winner; SCORE == 0: the liberty counts of get_num_liberties
template <int S, int NBL>
__global__ __launch_bounds__(64) void k_score_staged(const uint8_t* __restrict__ states, int32_t* __restrict__ o0, int32_t* __restrict__ o1,
                                                     int32_t* __restrict__ ended, int32_t* __restrict__ winner, int n, int SCORE)
{
    using T = Staged<S, NBL>;
    __shared__ uint4 lds[T::LDS_U4];
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(lds);
    const int lane = lane_id();
    const long first = (long)blockIdx.x * 64;
    const int nb = (int)min((long)64, (long)n - first);
    BB<S> me = bb_zero<S>(), opp = bb_zero<S>();
    bool inv_at = false;
    uint32_t flags = 0;
    int player = 0;
#pragma unroll 1
    for (int q = 0; q * T::NBL < nb; q++) {
        const int cnt = min(T::NBL, nb - q * T::NBL);
        stage_in(lds, states + (size_t)(first + q * T::NBL) * T::REC, (long)cnt * T::REC);
        __syncthreads();
        const int l = lane - q * T::NBL;
        if (l >= 0 && l < cnt) player = record_load<S>(bytes + l * T::REC, -1, me, opp, inv_at, flags);
        __syncthreads();
    }
    if (lane >= nb) return;
    int a_me, a_opp;
    if (SCORE) bits_areas<S>(me, opp, a_me, a_opp);
    else {
        const BB<S> emp = bb_andnot(bb_board<S>(), bb_or(me, opp));
        a_me = bb_count(bb_and(bb_nbr(me), emp));
        a_opp = bb_count(bb_and(bb_nbr(opp), emp));
    }
    const int b = player ? a_opp : a_me, w = player ? a_me : a_opp;      // planes 0 / 1 as given
    if (o0) o0[first + lane] = b;
    if (o1) o1[first + lane] = w;
    if (SCORE) {
        if (ended) ended[first + lane] = (flags & FLAG_DONE) ? 1 : 0;
        if (winner) winner[first + lane] = (b > w) - (b < w);
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int staged_grid(long units) { return (int)((units + 63) / 64); }
// records per LDS pass for a launch of `waves` waves: 0 = the largest (64 up to 9x9, 32 at 13x13, 16 at 19x19), 1 = 16 (8 at
// 19x19).  GOAI_GO_NBL=0/1 forces one (tuning aid).
static int staged_small_passes(int S, long waves)
{
    if (const char* e = getenv("GOAI_GO_NBL")) return atoi(e) != 0;
    return S <= 13 && waves >= 1280;      // measured: the small passes win from ~1.25 waves per SIMD on (profiles/r02_go_passes.txt)
}
#define GOAI_STAGED_LAUNCH(S_RT, SMALL, KERNEL, ...)                                                        \
    switch (S_RT) {                                                                                         \
        case 5: if (SMALL) KERNEL<5, 16> __VA_ARGS__; else KERNEL<5, 64> __VA_ARGS__; break;                  \
        case 7: if (SMALL) KERNEL<7, 16> __VA_ARGS__; else KERNEL<7, 64> __VA_ARGS__; break;                  \
        case 9: if (SMALL) KERNEL<9, 16> __VA_ARGS__; else KERNEL<9, 64> __VA_ARGS__; break;                  \
        case 13: if (SMALL) KERNEL<13, 16> __VA_ARGS__; else KERNEL<13, 32> __VA_ARGS__; break;               \
        case 19: if (SMALL) KERNEL<19, 8> __VA_ARGS__; else KERNEL<19, 16> __VA_ARGS__; break;                \
        default: set_error("unsupported board size"); return GOAI_ERR_ARG;                                  \
    }

// GOAI_GO_KERNELS=wave (read at every call) keeps the one-board-per-wavefront kernels above for these entry points (A/B
// timing, and the tests run all variants); the search's own transitions (k_mcts_select) always use them.
static bool go_per_lane()
{
    const char* e = getenv("GOAI_GO_KERNELS");
    return !(e && std::string(e) == "wave");
}
static inline int lanes_grid(long units) { return (int)((units + 255) / 256); }

static inline int grid_for(long units) { return (int)(units < 65536 ? (units > 0 ? units : 1) : 65536); }

int launch_transition(int S_rt, const uint8_t* states, const int32_t* actions, int canonical, uint8_t* out,
                      int32_t* status, int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    if (go_per_lane() && aligned16(states) && aligned16(out)) {
        const bool small = staged_small_passes(S_rt, staged_grid(n));
        GOAI_STAGED_LAUNCH(S_rt, small, k_transition_staged, <<<dim3(staged_grid(n)), dim3(64), 0, st>>>(states, actions, canonical, out, status, n));
    } else if (go_per_lane()) {
        GOAI_DISPATCH_SIZE(S_rt, k_transition_lanes<S><<<dim3(lanes_grid(n)), dim3(256), 0, st>>>(states, actions, canonical, out, status, n));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, k_transition<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, actions, canonical, out, status, n));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_children(int S_rt, const uint8_t* states, int canonical, uint8_t* children, uint8_t* valid, int8_t* win,
                    int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    if (go_per_lane() && aligned16(children)) {
        const long items = (long)n * (S_rt * S_rt + 1);
        const bool small = staged_small_passes(S_rt, staged_grid(items));
        GOAI_STAGED_LAUNCH(S_rt, small, k_children_staged, <<<dim3(staged_grid(items)), dim3(64), 0, st>>>(states, canonical, children, valid, win, n));
    } else if (go_per_lane()) {
        GOAI_DISPATCH_SIZE(S_rt, k_children_lanes<S><<<dim3(lanes_grid((long)n * (S * S + 1))), dim3(256), 0, st>>>(states, canonical, children, valid, win, n));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, k_children<S><<<dim3(grid_for((long)n * (S * S + 1))), dim3(64), 0, st>>>(states, canonical, children, valid, win, n));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_liberties(int S_rt, const uint8_t* states, int32_t* black_libs, int32_t* white_libs, int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    if (go_per_lane() && S_rt > 9) {
        GOAI_DISPATCH_SIZE(S_rt, k_liberties<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, black_libs, white_libs, n));
    } else if (go_per_lane() && aligned16(states)) {
        const bool small = staged_small_passes(S_rt, staged_grid(n));
        GOAI_STAGED_LAUNCH(S_rt, small, k_score_staged, <<<dim3(staged_grid(n)), dim3(64), 0, st>>>(states, black_libs, white_libs, nullptr, nullptr, n, 0));
    } else if (go_per_lane()) {
        GOAI_DISPATCH_SIZE(S_rt, (k_score_lanes<S, false><<<dim3(lanes_grid(n)), dim3(256), 0, st>>>(states, black_libs, white_libs, nullptr, nullptr, n)));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, k_liberties<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, black_libs, white_libs, n));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_score(int S_rt, const uint8_t* states, int32_t* black_area, int32_t* white_area, int32_t* ended,
                 int32_t* winner, int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    if (go_per_lane() && S_rt > 9) {
        // big empty regions flood slowly on 3..6-word bit boards: one board per wavefront is faster for scoring there
        GOAI_DISPATCH_SIZE(S_rt, k_score<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, black_area, white_area, ended, winner, n));
    } else if (go_per_lane() && aligned16(states)) {
        const bool small = staged_small_passes(S_rt, staged_grid(n));
        GOAI_STAGED_LAUNCH(S_rt, small, k_score_staged, <<<dim3(staged_grid(n)), dim3(64), 0, st>>>(states, black_area, white_area, ended, winner, n, 1));
    } else if (go_per_lane()) {
        GOAI_DISPATCH_SIZE(S_rt, (k_score_lanes<S, true><<<dim3(lanes_grid(n)), dim3(256), 0, st>>>(states, black_area, white_area, ended, winner, n)));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, k_score<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, black_area, white_area, ended, winner, n));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
