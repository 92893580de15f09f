// go_kernels.hip -- stateless Go-rule kernels behind goai_transition / goai_children /
// goai_score.  One board (or one (board, action) pair) per 64-lane workgroup; see go_board.h.
//
// Boundary format = the reference's GymGo state: uint8 planes [n][6][S][S]
// (/root/reference/go_ai/data.py:15-24 reads plane 3 as the invalid-move mask).
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

static inline int grid_for(long units) { return (int)(units < 65536 ? (units > 0 ? units : 1) : 65536); }

int launch_transition(int S_rt, const uint8_t* states, const int32_t* actions, int canonical, uint8_t* out,
                      int32_t* status, int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    GOAI_DISPATCH_SIZE(S_rt, k_transition<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, actions, canonical, out, status, n));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_children(int S_rt, const uint8_t* states, int canonical, uint8_t* children, uint8_t* valid, int8_t* win,
                    int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    GOAI_DISPATCH_SIZE(S_rt, k_children<S><<<dim3(grid_for((long)n * (S * S + 1))), dim3(64), 0, st>>>(states, canonical, children, valid, win, n));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_liberties(int S_rt, const uint8_t* states, int32_t* black_libs, int32_t* white_libs, int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    GOAI_DISPATCH_SIZE(S_rt, k_liberties<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, black_libs, white_libs, n));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_score(int S_rt, const uint8_t* states, int32_t* black_area, int32_t* white_area, int32_t* ended,
                 int32_t* winner, int n, hipStream_t st)
{
    if (n <= 0) return GOAI_OK;
    GOAI_DISPATCH_SIZE(S_rt, k_score<S><<<dim3(grid_for(n)), dim3(64), 0, st>>>(states, black_area, white_area, ended, winner, n));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
