// mcts_kernels.hip -- PUCT selection, expansion and value backup on GPU-resident trees,
// one game per 64-lane wavefront, all games of a context in lock step.
//
// Reference arithmetic being reproduced (fp64, NumPy-1.x semantics):
//   find_next_node   /root/reference/go_ai/search/mct.py:10-17    descend by np.nanargmax(get_ucbs())
//   Node.get_ucbs    /root/reference/go_ai/search/tree.py:170-184  avg_q + 1.5*prior*sqrt(N)/(1+n)
//   Node.step        /root/reference/go_ai/search/tree.py:110-117  existing child or GoGame.next_state
//   Node.backprop    /root/reference/go_ai/search/tree.py:141-146  append value, negate up to the root
//   mct_step         /root/reference/go_ai/search/mct.py:20-47     prior = softmax(pi_logits)
//   ActorCritic.__call__  /root/reference/go_ai/policies/actorcritic.py:29-42  visit counts -> pi
#include "detmath.h"
#include "go_board.h"
#include "goai_internal.h"
#include "mcts.h"

namespace goai {

template <int S>
__device__ __forceinline__ void node_load(const Tree& t, size_t node, uint64_t* me, uint64_t* opp, uint64_t* inv,
                                          uint64_t& meta)
{
    using G = Geom<S>;
    const uint64_t* s = t.st + node * G::SW;
#pragma unroll
    for (int k = 0; k < G::K; k++) { me[k] = s[k]; opp[k] = s[G::K + k]; inv[k] = s[2 * G::K + k]; }
    meta = s[3 * G::K];
}

template <int S>
__device__ __forceinline__ void node_store(const Tree& t, size_t node, const uint64_t* me, const uint64_t* opp,
                                           const uint64_t* inv, uint64_t meta)
{
    using G = Geom<S>;
    uint64_t* s = t.st + node * G::SW;
#pragma unroll
    for (int k = 0; k < G::K; k++) { s[k] = me[k]; s[G::K + k] = opp[k]; s[2 * G::K + k] = inv[k]; }
    s[3 * G::K] = meta;
}

// (Re)start every game's tree from the given root states.
template <int S>
__global__ __launch_bounds__(64) void k_mcts_reset(Tree t, const uint8_t* __restrict__ roots)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const int g = blockIdx.x;
    const size_t nb = (size_t)g * t.ncap;
    uint64_t me[G::K], opp[G::K], inv[G::K];
    uint32_t flags;
    const int player = planes_load<S>(roots + (size_t)g * 6 * G::P, me, opp, inv, flags);
    board_load<S>(L, me, opp);
    board_analyze<S>(L);
    int a_me, a_opp;
    board_areas<S>(L, a_me, a_opp);
    const int sign = (a_me > a_opp) - (a_me < a_opp);
    if (lane_id() == 0) {
        node_store<S>(t, nb, me, opp, inv, pack_meta(flags, sign));
        t.parent[nb] = -1;
        t.pact[nb] = -1;
        t.visits[nb] = 0;
        t.nnodes[g] = 1;
        t.turn[g] = player;
        t.moves[g] = 0;
        t.result[g] = (flags & FLAG_DONE) ? (player ? -sign : sign) : 2;
    }
}

// One mct_step, first half: descend from the root to the next node to evaluate, creating it
// (one Go transition) if the chosen edge is new.  Writes the leaf's six input planes.
template <int S>
__global__ __launch_bounds__(64) void k_mcts_select(Tree t, uint8_t* __restrict__ leaf_out)
{
    using G = Geom<S>;
    static_assert((G::A + 63) / 64 == G::K, "action words must match plane words");
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    int32_t* path = t.path + nb;
    int cur = 0, d = 0;
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    if (lane == 0) path[0] = 0;
    while (true) {
        node_load<S>(t, nb + cur, me, opp, inv, meta);
        const int N = t.visits[nb + cur];
        if (N == 0 || (meta & FLAG_DONE)) break;
        const size_t row = (nb + cur) * G::A;
        const double sqrtN = t.sqrt_tab[N];
        double best = -INFINITY;
        int besta = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            if (a < G::A) {
                const bool valid = (a == G::P) || !((inv[k] >> lane) & 1ull);
                if (valid) {
                    const double p = t.prior[row + a];
                    const int n = t.cN[row + a];
                    const double q = t.cQ[row + a];
                    const double u = ((1.5 * p) * sqrtN) / (double)(1 + n);
                    const double ucb = q + u;
                    if (ucb > best) { best = ucb; besta = a; }
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int oa = __shfl_xor(besta, off);
            if (ob > best || (ob == best && oa < besta)) { best = ob; besta = oa; }
        }
        const int a = besta;
        const int ch = t.child[row + a];
        if (ch >= 0) {
            cur = ch;
            d++;
            if (lane == 0) path[d] = cur;
            continue;
        }
        uint64_t nme[G::K], nopp[G::K], ninv[G::K];
        uint32_t nflags = 0;
        int sgn = 0;
        const int rc = board_next_state<S>(L, me, opp, inv, (uint32_t)(meta & 7u), a, nme, nopp, ninv, nflags, sgn);
        const int idx = t.nnodes[g];
        if (rc != 0 || idx >= t.ncap) {          // cannot happen for a consistent tree; never write out of bounds
            if (lane == 0) atomicAdd(&t.counters[2], 1ull);
            break;
        }
        meta = pack_meta(nflags, sgn);
        if (lane == 0) {
            node_store<S>(t, nb + idx, nme, nopp, ninv, meta);
            t.parent[nb + idx] = cur;
            t.pact[nb + idx] = a;
            t.visits[nb + idx] = 0;
            t.child[row + a] = idx;
            t.nnodes[g] = idx + 1;
            path[d + 1] = idx;
            atomicAdd(&t.counters[1], 1ull);
        }
#pragma unroll
        for (int k = 0; k < G::K; k++) { me[k] = nme[k]; opp[k] = nopp[k]; inv[k] = ninv[k]; }
        cur = idx;
        d++;
        break;
    }
    if (lane == 0) {
        t.leaf[g] = cur;
        t.depth[g] = d;
        if (t.result[g] == 2) atomicAdd(&t.counters[0], 1ull);
    }
    planes_store<S>(t.leaf_planes + (size_t)g * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
    if (leaf_out) planes_store<S>(leaf_out + (size_t)g * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
}

// One mct_step, second half: back the leaf value up the path and, for a non-terminal leaf,
// store its prior.  `prior_in` (fp32, as scipy produced it) bypasses the softmax: that is the
// "fed the reference's NN outputs" mode of the parity tests.  With `raw` set the logits are
// straight from the network and RLNet._numpy's post-processing
// (/root/reference/go_ai/models/__init__.py:90-105) is fused in here.
template <int S>
__global__ __launch_bounds__(64) void k_mcts_backup(Tree t, const float* __restrict__ pi_logits,
                                                    const float* __restrict__ prior_in,
                                                    const float* __restrict__ val_in, int raw)
{
    using G = Geom<S>;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    const int leaf = t.leaf[g], d = t.depth[g];
    const int32_t* path = t.path + nb;
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb + leaf, me, opp, inv, meta);
    const bool done = meta & FLAG_DONE;
    const int sign = meta_sign(meta);
    float v = val_in[g];
    if (raw && done) v = (float)(100 * sign);
    const double T = det_tanh64((double)v);

    for (int j = lane; j <= d; j += 64) {
        const int X = path[d - j];
        const size_t xi = nb + X;
        const double tv = (j & 1) ? -T : T;
        const int n = t.visits[xi];
        double* R = t.accR + xi * 8;
        double* Tl = t.accT + xi * 8;
        Tl[n & 7] = tv;
        const int n1 = n + 1;
        if ((n1 & 7) == 0) {
            if (n1 == 8) { for (int i = 0; i < 8; i++) R[i] = Tl[i]; }
            else { for (int i = 0; i < 8; i++) R[i] += Tl[i]; }
        }
        t.visits[xi] = n1;
        if (X != 0) {
            double res;
            if (n1 < 8) {
                res = 0.0;
                for (int i = 0; i < n1; i++) res += Tl[i];
            } else {
                res = ((R[0] + R[1]) + (R[2] + R[3])) + ((R[4] + R[5]) + (R[6] + R[7]));
                for (int i = 0; i < (n1 & 7); i++) res += Tl[i];
            }
            const double mean = res / (double)n1;
            const size_t prow = (nb + t.parent[xi]) * G::A + t.pact[xi];
            t.cN[prow] = n1;
            t.cQ[prow] = -mean;
        }
    }

    if (done) return;
    const size_t row = (nb + leaf) * G::A;
    double pr[G::K];
    if (prior_in) {
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            pr[k] = (a < G::A) ? (double)prior_in[(size_t)g * G::A + a] : 0.0;
        }
    } else {
        const double LOWEST = -3.4028234663852886e38;
        float x[G::K], e[G::K];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            float xv = -INFINITY;
            if (a < G::A) {
                xv = pi_logits[(size_t)g * G::A + a];
                if (raw) {
                    if (a < G::P && ((inv[k] >> lane) & 1ull)) xv = (float)((double)xv + LOWEST);
                    if (a == G::P && (meta & FLAG_PASS)) xv = (float)((double)xv + (double)(100 * sign));
                }
            }
            x[k] = xv;
            m = fmaxf(m, xv);
        }
        m = wave_max_f32(m);
        float part = 0.0f;
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            e[k] = (a < G::A) ? det_exp32(x[k] - m) : 0.0f;
            part += e[k];
        }
        const float sum = wave_sum_f32(part);
#pragma unroll
        for (int k = 0; k < G::K; k++) pr[k] = (double)(e[k] / sum);
    }
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        if (a < G::A) {
            t.prior[row + a] = pr[k];
            t.child[row + a] = -1;
            t.cN[row + a] = 0;
            t.cQ[row + a] = 0.0;
        }
    }
}

// rootnode.get_visit_counts() / rootnode.prior_pi / rootnode.visits
template <int S>
__global__ __launch_bounds__(64) void k_mcts_readout(Tree t, int32_t* __restrict__ visits, double* __restrict__ prior,
                                                     int32_t* __restrict__ root_visits)
{
    using G = Geom<S>;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap, row = nb * G::A;
    const bool expanded = t.visits[nb] > 0 && !(t.st[nb * G::SW + 3 * G::K] & FLAG_DONE);
    for (int a = lane; a < G::A; a += 64) {
        if (visits) visits[(size_t)g * G::A + a] = expanded ? t.cN[row + a] : 0;
        if (prior) prior[(size_t)g * G::A + a] = expanded ? t.prior[row + a] : 0.0;
    }
    if (root_visits && lane == 0) root_visits[g] = t.visits[nb];
}

// ActorCritic.__call__, mcts > 0 (/root/reference/go_ai/policies/actorcritic.py:29-42):
//   pi = counts ** (1/T); pi /= np.sum(pi); pi = 0.98 * pi + 0.02 * valid / valid.sum()
// np.sum is numpy's pairwise sum; T == 1 is exact, other T go through pow() (not bit-pinned).
// Finished games and unexpanded roots get the uniform-valid distribution.
template <int S>
__global__ __launch_bounds__(64) void k_mcts_policy(Tree t, double temp, int uniform_only, int allow_pass,
                                                    double* __restrict__ pi)
{
    using G = Geom<S>;
    __shared__ double buf[G::A];
    __shared__ double total;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap, row = nb * G::A;
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb, me, opp, inv, meta);
    const bool expanded = !uniform_only && t.visits[nb] > 0 && !(meta & FLAG_DONE);
    int nvalid = 0;
    bool valid[G::K];
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        valid[k] = a < G::A && ((a == G::P) ? (allow_pass != 0) : !((inv[k] >> lane) & 1ull));
        nvalid += __popcll(__ballot(valid[k]));
        if (a < G::A) {
            const double c = expanded ? (double)t.cN[row + a] : 0.0;
            buf[a] = (temp == 1.0) ? c : pow(c, 1.0 / temp);
        }
    }
    if (nvalid == 0) {           // only the pass is left
#pragma unroll
        for (int k = 0; k < G::K; k++) valid[k] = (lane + 64 * k) == G::P;
        nvalid = 1;
    }
    __syncthreads();
    if (lane == 0) total = np_pairwise_sum(buf, G::A);
    __syncthreads();
    const double sum = total, nv = (double)nvalid;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        if (a < G::A) {
            const double noise = (valid[k] ? 1.0 : 0.0) / nv;
            pi[(size_t)g * G::A + a] = (expanded && sum > 0.0) ? (0.98 * (buf[a] / sum) + 0.02 * noise) : noise;
        }
    }
}

// GoGame.random_weighted_action(pi) (/root/reference/go_ai/game.py:91): inverse-CDF draw with
// a Philox stream per (seed, game, step).  The reference draws from numpy's unseeded global
// generator, so parity is on pi, not on the drawn action.
template <int S>
__global__ __launch_bounds__(64) void k_sample(Tree t, const double* __restrict__ pi, uint64_t seed, uint64_t step,
                                               int game_offset, int32_t* __restrict__ actions)
{
    using G = Geom<S>;
    const int g = blockIdx.x;
    if (lane_id() != 0) return;
    const double* p = pi + (size_t)g * G::A;
    uint32_t r[4];
    const uint64_t gid = (uint64_t)(g + game_offset);
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const double u = (double)((((uint64_t)r[0] << 32) | r[1]) >> 11) * 0x1p-53;
    double total = 0.0;
    for (int a = 0; a < G::A; a++) total += p[a];
    const double target = u * total;
    double cum = 0.0;
    int pick = -1, last = G::P;
    for (int a = 0; a < G::A; a++) {
        if (p[a] > 0.0) {
            last = a;
            cum += p[a];
            if (pick < 0 && cum > target) pick = a;
        }
    }
    actions[g] = pick >= 0 ? pick : last;
}

// go_env.step(action) for every running game, then a fresh tree (the reference builds a
// new root every move, /root/reference/go_ai/search/mct.py:52-53).  Games end on two passes
// or once more than `max_moves` moves were played (/root/reference/go_ai/game.py:74,98).
template <int S>
__global__ __launch_bounds__(64) void k_mcts_advance(Tree t, const int32_t* __restrict__ actions, int max_moves)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    if (t.result[g] == 2) {
        uint64_t me[G::K], opp[G::K], inv[G::K], meta, nme[G::K], nopp[G::K], ninv[G::K];
        uint32_t nflags = 0;
        int sgn = 0;
        node_load<S>(t, nb, me, opp, inv, meta);
        const int rc = board_next_state<S>(L, me, opp, inv, (uint32_t)(meta & 7u), actions[g], nme, nopp, ninv, nflags, sgn);
        if (rc != 0) {
            if (lane == 0) atomicAdd(&t.counters[2], 1ull);
        } else if (lane == 0) {
            node_store<S>(t, nb, nme, nopp, ninv, pack_meta(nflags, sgn));
            const int nturn = t.turn[g] ^ 1, nmoves = t.moves[g] + 1;
            t.turn[g] = nturn;
            t.moves[g] = nmoves;
            if ((nflags & FLAG_DONE) || nmoves > max_moves) t.result[g] = nturn ? -sgn : sgn;
        }
    }
    if (lane == 0) {
        t.visits[nb] = 0;
        t.nnodes[g] = 1;
    }
}

template <int S>
__global__ __launch_bounds__(64) void k_mcts_roots(Tree t, uint8_t* __restrict__ states)
{
    using G = Geom<S>;
    const int g = blockIdx.x;
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, (size_t)g * t.ncap, me, opp, inv, meta);
    planes_store<S>(states + (size_t)g * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
}

// ---- launchers -----------------------------------------------------------------------------
#define LAUNCH_PER_GAME(KERNEL, ...)                                                              \
    GOAI_DISPATCH_SIZE(S_rt, KERNEL<S><<<dim3(t.n_games), dim3(64), 0, st>>>(__VA_ARGS__));       \
    GOAI_HIP_CHECK(hipGetLastError());                                                            \
    return GOAI_OK

int launch_mcts_reset(int S_rt, const Tree& t, const uint8_t* roots, hipStream_t st) { LAUNCH_PER_GAME(k_mcts_reset, t, roots); }
int launch_mcts_select(int S_rt, const Tree& t, uint8_t* leaf_out, hipStream_t st) { LAUNCH_PER_GAME(k_mcts_select, t, leaf_out); }
int launch_mcts_backup(int S_rt, const Tree& t, const float* pi_logits, const float* prior, const float* val, int raw,
                       hipStream_t st)
{
    LAUNCH_PER_GAME(k_mcts_backup, t, pi_logits, prior, val, raw);
}
int launch_mcts_readout(int S_rt, const Tree& t, int32_t* visits, double* prior, int32_t* root_visits, hipStream_t st)
{
    LAUNCH_PER_GAME(k_mcts_readout, t, visits, prior, root_visits);
}
int launch_mcts_policy(int S_rt, const Tree& t, double temp, double* pi, hipStream_t st)
{
    LAUNCH_PER_GAME(k_mcts_policy, t, temp, 0, 1, pi);
}
int launch_uniform_policy(int S_rt, const Tree& t, int allow_pass, double* pi, hipStream_t st)
{
    LAUNCH_PER_GAME(k_mcts_policy, t, 1.0, 1, allow_pass, pi);
}
int launch_sample(int S_rt, const Tree& t, const double* pi, uint64_t seed, uint64_t step, int game_offset,
                  int32_t* actions, hipStream_t st)
{
    LAUNCH_PER_GAME(k_sample, t, pi, seed, step, game_offset, actions);
}
int launch_mcts_advance(int S_rt, const Tree& t, const int32_t* actions, int max_moves, hipStream_t st)
{
    LAUNCH_PER_GAME(k_mcts_advance, t, actions, max_moves);
}
int launch_mcts_roots(int S_rt, const Tree& t, uint8_t* states, hipStream_t st) { LAUNCH_PER_GAME(k_mcts_roots, t, states); }

}  // namespace goai
