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
#include "go_bits.h"
#include "go_board.h"
#include "goai_internal.h"
#include "mcts.h"
#include "nn_groups.h"
#include "conv_wino3h_body.h"

namespace goai {
// board_next_state (go_board.h) on the packed words themselves: the bit-board rules of go_bits.h with wave-uniform
// operands (no LDS, no barriers).  Same results bit for bit (tests/test_gpu_go.py, tests/test_gpu_mcts.py).
template <int S>
__device__ __forceinline__ int uniform_next_state(const uint64_t* me, const uint64_t* opp, const uint64_t* inv, uint32_t flags, int action,
                                                  uint64_t* nme, uint64_t* nopp, uint64_t* ninv, uint32_t& nflags, int& sign_out)
{
    using G = Geom<S>;
    BB<S> m, o, iv, a, b, c;
#pragma unroll
    for (int k = 0; k < G::K; k++) { m.w[k] = me[k]; o.w[k] = opp[k]; iv.w[k] = inv[k]; }
    const bool inv_at = action >= 0 && action < G::P && bb_test(iv, action);
    // the area sign of a node is read only where the previous move was a pass or the game is over (backup_game)
    const int rc = bits_next_state<S, 2>(m, o, inv_at, flags, action, a, b, c, nflags, sign_out);
    if (rc == 0) {
#pragma unroll
        for (int k = 0; k < G::K; k++) { nme[k] = a.w[k]; nopp[k] = b.w[k]; ninv[k] = c.w[k]; }
    }
    return rc;
}


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
        t.rowof[nb] = -1;
        t.nnodes[g] = 1;
        t.nrows[g] = 0;
        if (t.nlog) t.nlog[g] = 0;
        t.turn[g] = player;
        t.moves[g] = 0;
        t.result[g] = (flags & FLAG_DONE) ? (player ? -sign : sign) : 2;
        t.episode[g] = 0;
        for (int k = 0; k < 4; k++) t.fin[(size_t)k * t.n_games + g] = 0;
    }
}

// One mct_step, first half: descend from the root to the next node to evaluate, creating it
// (one Go transition) if the chosen edge is new.  Writes the leaf's six input planes.
// Everything a level of the descent needs sits in the current node's row -- prior, child visit
// count (= the child's N), cQ, child index (-2 marks an invalid move) and the child's own row
// (crow, -1 while the child is unvisited or terminal) -- so a level costs ONE round of
// independent loads, not a chain of node -> visits -> row look-ups.
// BITS: the new node's position by the bit-board rules on the packed words (go_bits.h: no LDS, no barriers -- for
// kernels whose workgroup is more than this one wave); else by go_board.h on the wave's LDS board L.
template <int S, bool BITS>
__device__ __forceinline__ void select_game(const Tree& t, int g, uint8_t* __restrict__ leaf_out, BoardLds<S>* L)
{
    using G = Geom<S>;
    static_assert((G::A + 63) / 64 == G::K, "action words must match plane words");
    const int lane = lane_id();
    const size_t nb = (size_t)g * t.ncap, rb = (size_t)g * t.rcap;
    int32_t* path = t.path + rb;
    int cur = 0, d = 0;
    int N = t.visits[nb];
    int crow = t.rowof[nb];                  // >= 0 iff the root was expanded (visited and not terminal)
    long edge = -1;                          // row slot (row * A + action) of the edge into the leaf
    bool create = false;
    int act = -1;
    if (lane == 0) path[0] = 0;
    while (crow >= 0) {
        const size_t row = (rb + crow) * G::A;
        const double sqrtN = t.sqrt_tab[N];
        double best = -INFINITY;
        int besta = 0x7fffffff;
        int chv[G::K], crv[G::K], nv[G::K];
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            chv[k] = -2; crv[k] = -1; nv[k] = 0;
            if (a < G::A) {
                const int ch = t.child[row + a];
                const double p = t.prior[row + a];
                const int n = t.cN[row + a];
                const double q = t.cQ[row + a];
                chv[k] = ch; crv[k] = t.crow[row + a]; nv[k] = n;
                if (ch != -2) {
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
        const int a = besta, ak = a >> 6;
        int sch = chv[0], scr = crv[0], sn = nv[0];
#pragma unroll
        for (int k = 1; k < G::K; k++)
            if (ak == k) { sch = chv[k]; scr = crv[k]; sn = nv[k]; }
        const int ch = __shfl(sch, a & 63), cr = __shfl(scr, a & 63), cn = __shfl(sn, a & 63);
        edge = (long)(row + a);
        act = a;
        if (ch < 0) { create = true; break; }
        cur = ch;
        d++;
        if (lane == 0) path[d] = cur;
        crow = cr;
        N = cn;
    }
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb + cur, me, opp, inv, meta);
    if (create) {
        uint64_t nme[G::K], nopp[G::K], ninv[G::K];
        uint32_t nflags = 0;
        int sgn = 0;
        const int rc = BITS ? uniform_next_state<S>(me, opp, inv, (uint32_t)(meta & 7u), act, nme, nopp, ninv, nflags, sgn)
                            : board_next_state<S>(*L, me, opp, inv, (uint32_t)(meta & 7u), act, nme, nopp, ninv, nflags, sgn);
        const int idx = t.nnodes[g];
        if (rc != 0 || idx >= t.ncap) {          // cannot happen for a consistent tree; never write out of bounds
            if (lane == 0) atomicAdd(&t.counters[(size_t)2 * t.n_games + g], 1ull);
            edge = -1;                           // evaluate the parent again rather than corrupt the tree
        } else {
            meta = pack_meta(nflags, sgn);
            if (lane == 0) {
                node_store<S>(t, nb + idx, nme, nopp, ninv, meta);
                t.parent[nb + idx] = cur;
                t.pact[nb + idx] = act;
                t.visits[nb + idx] = 0;
                t.rowof[nb + idx] = -1;
                t.child[edge] = idx;
                t.nnodes[g] = idx + 1;
                path[d + 1] = idx;
                atomicAdd(&t.counters[(size_t)1 * t.n_games + g], 1ull);
            }
#pragma unroll
            for (int k = 0; k < G::K; k++) { me[k] = nme[k]; opp[k] = nopp[k]; inv[k] = ninv[k]; }
            cur = idx;
            d++;
        }
    }
    if (lane == 0) {
        t.leaf[g] = cur;
        t.depth[g] = d;
        if (t.result[g] == 2) atomicAdd(&t.counters[(size_t)0 * t.n_games + g], 1ull);
        // a non-terminal leaf is expanded by this step's second half: give it its row now
        const bool expand = !(meta & FLAG_DONE);
        if (expand && t.rowof[nb + cur] < 0) {
            const int r = t.nrows[g];
            if (r < t.rcap) {
                t.rowof[nb + cur] = r;
                t.nrows[g] = r + 1;
                if (edge >= 0) t.crow[edge] = r;
            } else atomicAdd(&t.counters[(size_t)2 * t.n_games + g], 1ull);
        }
        if (t.q_cnt) {
            int nvalid = 1;                        // the pass
#pragma unroll
            for (int k = 0; k < G::K; k++) {
                const int rem = G::P - 64 * k;
                const uint64_t cells = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                nvalid += __popcll(~inv[k] & cells);
            }
            t.q_cnt[g] = expand ? 1 + nvalid : 1;
        }
    }
    planes_store<S>(t.leaf_planes + (size_t)g * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
    if (leaf_out) planes_store<S>(leaf_out + (size_t)g * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
}

template <int S>
__global__ __launch_bounds__(64) void k_mcts_select(Tree t, uint8_t* __restrict__ leaf_out)
{
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    select_game<S, false>(t, blockIdx.x, leaf_out, &L);
}

// Node.backprop (/root/reference/go_ai/search/tree.py:141-146) for the selected leaf of game g:
// every node on the path gets tanh(value) with alternating sign appended to its running
// numpy-pairwise mean, one lane per ancestor; the parent's PUCT inputs (cN, cQ) are refreshed.
template <int S, bool LONG = false>
__device__ __forceinline__ void backprop_path(const Tree& t, int g, double v)
{
    using G = Geom<S>;
    const int lane = lane_id();
    const size_t nb = (size_t)g * t.ncap, rb = (size_t)g * t.rcap;
    const int d = t.depth[g];
    const int32_t* path = t.path + rb;
    const double T = det_tanh64(v);
    for (int j = lane; j <= d; j += 64) {
        const int X = path[d - j];
        const size_t xi = nb + X;
        const double tv = (j & 1) ? -T : T;
        const int n = t.visits[xi];
        t.rawsum[xi] = (n == 0 ? 0.0 : t.rawsum[xi]) + ((j & 1) ? -v : v);   // post_vals as appended (tree.py:142)
        double* R = t.accR + xi * 8;
        double* Tl = t.accT + xi * 8;
        Tl[n & 7] = tv;
        const int n1 = n + 1;
        if ((n1 & 7) == 0) {
            if (n1 == 8) { for (int i = 0; i < 8; i++) R[i] = Tl[i]; }
            else { for (int i = 0; i < 8; i++) R[i] += Tl[i]; }
        }
        t.visits[xi] = n1;
        const double* log = nullptr;
        if (LONG) {
            // the node's whole value list (max_sims > 128): only the leaf (j == 0, lane 0) can be on its first visit
            const size_t M = (size_t)t.max_sims + 1;
            int slot;
            if (n == 0) {
                slot = t.nlog[g];
                t.nlog[g] = slot + 1;
                t.logof[xi] = slot;
            } else {
                slot = t.logof[xi];
            }
            double* lg = t.vlog + (size_t)g * (M * (M + 1) / 2) + (size_t)slot * M - (size_t)slot * (slot - 1) / 2;
            lg[n] = tv;
            log = lg;
        }
        if (X != 0) {
            double res;
            if (n1 < 8) {
                res = 0.0;
                for (int i = 0; i < n1; i++) res += Tl[i];
            } else if (!LONG || n1 <= 128) {
                res = ((R[0] + R[1]) + (R[2] + R[3])) + ((R[4] + R[5]) + (R[6] + R[7]));
                for (int i = 0; i < (n1 & 7); i++) res += Tl[i];
            } else {
                res = np_pairwise_sum(log, n1);        // numpy's recursion over the halves (tree.py:180, any length)
            }
            const double mean = res / (double)n1;
            const size_t prow = (rb + t.rowof[nb + t.parent[xi]]) * G::A + t.pact[xi];
            t.cN[prow] = n1;
            t.cQ[prow] = -mean;
        }
    }
}

// One mct_step, second half: back the leaf value up the path and, for a non-terminal leaf,
// store its prior.  `prior_in` (fp32, as scipy produced it) bypasses the softmax: that is the
// "fed the reference's NN outputs" mode of the parity tests.  With `raw` set the logits are
// straight from the network and RLNet._numpy's post-processing
// (/root/reference/go_ai/models/__init__.py:90-105) is fused in here.
template <int S, bool LONG = false>
__device__ __forceinline__ void backup_game(const Tree& t, int g, const float* __restrict__ pi_logits,
                                            const float* __restrict__ prior_in, const float* __restrict__ val_in, int raw)
{
    using G = Geom<S>;
    const int lane = lane_id();
    const size_t nb = (size_t)g * t.ncap, rb = (size_t)g * t.rcap;
    const int leaf = t.leaf[g];
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb + leaf, me, opp, inv, meta);
    const bool done = meta & FLAG_DONE;
    const int sign = meta_sign(meta);
    float v = val_in[g];
    if (raw && done) v = (float)(100 * sign);
    backprop_path<S, LONG>(t, g, v);

    const int lrow = t.rowof[nb + leaf];
    if (done || lrow < 0) return;
    const size_t row = (rb + lrow) * G::A;
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
            const bool valid = (a == G::P) || !((inv[k] >> lane) & 1ull);
            t.prior[row + a] = pr[k];
            t.child[row + a] = valid ? -1 : -2;
            t.crow[row + a] = -1;
            t.cN[row + a] = 0;
            t.cQ[row + a] = 0.0;
        }
    }
}

template <int S, bool LONG>
__global__ __launch_bounds__(64) void k_mcts_backup(Tree t, const float* __restrict__ pi_logits,
                                                    const float* __restrict__ prior_in,
                                                    const float* __restrict__ val_in, int raw)
{
    backup_game<S, LONG>(t, blockIdx.x, pi_logits, prior_in, val_in, raw);
}

// Between two trunk launches of the lock-step search (actor-critic mode, the context's own network): everything one
// mct_step does after the trunk and the next one does before it, for the HB = NB0 = 2 games of a workgroup --
//   heads of this step's leaves -> k_mcts_backup (raw logits) -> k_mcts_select of the NEXT step -> first convolution
//   of the new leaves
// -- the same device functions as the four kernels it stands for (same bits out), in one launch: those kernels are
// latency-bound chains of a few microseconds each, and a workgroup here never waits for another one.  `act` is the
// trunk's output; the new leaves' first-layer activations go to `act0` (which may be `act`: a workgroup reads its two
// boards' rows before it overwrites them).
#ifdef GOAI_PF_STAMPS
// diagnostic build only (tools/pf_stamps.py): s_memrealtime (100 MHz) at the phase boundaries of every workgroup's last run
__device__ unsigned long long g_pf_stamps[8 * 8192];
extern "C" int goai_debug_pf_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pf_stamps), (size_t)n * sizeof(unsigned long long));
}
#define PF_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 8192) g_pf_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define PF_STAMP(i)
#endif

template <int S>
__global__ __launch_bounds__(256) void k_post_forward(Tree t, const float* act, HeadWeights hw, const uint4* __restrict__ w0s,
                                                      const float* __restrict__ b0, float shift, float* act0, bool act_is_h)
{
    static_assert(HB == 2 && NB0 == 2, "two games per workgroup");
    __shared__ HeadsLds<S> hl;
    __shared__ Conv0Lds<S> cl;
    const int n = t.n_active, b_first = blockIdx.x * 2, wave = threadIdx.x >> 6;
    PF_STAMP(0);
    heads_groups<S>(hl, act, hw, 1, t.pi_logits, t.val_logits, n, b_first, n, act_is_h);   // one group: b_first, b_first + 1
    // (workgroup-scope ordering is all that is needed: every datum handed from phase to phase stays inside the
    // workgroup, whose waves share the CU's L1; a device-scope fence would write the XCD's whole dirty L2 back)
    __threadfence_block();
    __syncthreads();
    PF_STAMP(1);
    if (wave < 2 && b_first + wave < n) {
        const int g = b_first + wave;
        backup_game<S>(t, g, t.pi_logits, nullptr, t.val_logits, 1);
        PF_STAMP(2);
        __threadfence_block();              // the descent below reads rows other lanes of this wave have just written
        select_game<S, true>(t, g, nullptr, nullptr);
        PF_STAMP(3);
    }
    __threadfence_block();
    __syncthreads();
    PF_STAMP(4);
    conv0_group<S>(cl, t.leaf_planes, w0s, b0, shift, act0, n, b_first);
    PF_STAMP(5);
}

// THE WHOLE LOCK-STEP LOOP IN ONE LAUNCH (9x9, fp16 F(3x3,3x3) trunk; round 3): a 512-thread workgroup owns four games for
// n_steps search steps -- trunk (conv_wino3h_body.inc: the statements of k_conv3x3_wino3h), then what k_post_forward does for
// two games per 256 threads, here by the workgroup's two halves -- and never waits for another workgroup: no game depends on
// another game.  What that buys (profiles/r03_wino3h_notes.txt section 9): with one launch per trunk pass every workgroup of
// the chip is in the same phase at the same time, and the passes' memory phases (a convolution's residual reads and output
// stores: 85 MB over the chip, 10 us at the full HBM rate) cost 35 us per pass, of which a start stagger recovers 15 at the
// price of its own 22 us tail per launch; here the workgroups start spread over a whole pass ONCE per launch of n_steps
// passes and stay spread.  Same device code as the separate kernels, same bits.  n_active must be a multiple of 4
// (both halves of every workgroup make the same barrier calls).
template <int NP>
__global__ __launch_bounds__(512) void k_search_steps(Tree tree, float* x, float* y, const uint4* wt_all,
                                                      const float* __restrict__ bias_all, int n_convs,
                                                      const int* __restrict__ wexp_all, float amax_first, HeadWeights hw,
                                                      float* hbuf, const uint4* __restrict__ w0s, const float* __restrict__ b0,
                                                      float shift, int n_steps, int spread)
{
    __shared__ union { HeadsLds<9> h[2]; Conv0Lds<9> c[2]; } pl;      // 7.9 KB beside the trunk's 152.6
    const int nboards = tree.n_active;
    const float* resid0 = nullptr;
    float* amax_io = nullptr;
    const float* head_w = hw.conv_w;
    const float* head_b = hw.conv_b;
    float* head_out = hbuf;
    // the workgroups' starts spread over `spread` sleep units of 64 clocks (a trunk pass), in the order of their index bits
    // reversed over 8 bits so that neighbours (same XCD, same memory channels) are far apart
    {
        const unsigned r = __builtin_bitreverse32(blockIdx.x & 255u) >> 24;
        for (int left = (int)(r * (unsigned)spread >> 8); left > 0; left -= 16) __builtin_amdgcn_s_sleep(16);
    }
#define SS_STAMP(i) if (step == 5) { PF_STAMP(i); }
    for (int step = 0; step < n_steps; step++) {
        SS_STAMP(0);
        {
#define W3H_WG blockIdx.x
#include "conv_wino3h_body.inc"
#undef W3H_WG
        }
        if (step + 1 == n_steps) break;
        __threadfence_block();
        __syncthreads();
        SS_STAMP(1);
        {
            const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, wave = tid >> 6;
            const int bg = blockIdx.x * 4 + 2 * half;                  // this half's two games
            heads_groups<9>(pl.h[half], hbuf, hw, 1, tree.pi_logits, tree.val_logits, nboards, bg, nboards, true, tid);
            __threadfence_block();
            __syncthreads();
            SS_STAMP(2);
            if (wave < 2 && bg + wave < nboards) {
                backup_game<9>(tree, bg + wave, tree.pi_logits, nullptr, tree.val_logits, 1);
                __threadfence_block();
                select_game<9, true>(tree, bg + wave, nullptr, nullptr);
            }
            __threadfence_block();
            __syncthreads();
            SS_STAMP(3);
            conv0_group<9>(pl.c[half], tree.leaf_planes, w0s, b0, shift, x, nboards, bg, tid);
            __threadfence_block();
            __syncthreads();
            SS_STAMP(4);
        }
    }
#undef SS_STAMP
}

// rootnode.get_visit_counts() / rootnode.prior_pi / rootnode.visits
template <int S>
__global__ __launch_bounds__(64) void k_mcts_readout(Tree t, int32_t* __restrict__ visits, double* __restrict__ prior,
                                                     int32_t* __restrict__ root_visits)
{
    using G = Geom<S>;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    const int rrow = t.rowof[nb];
    const size_t row = ((size_t)g * t.rcap + (rrow < 0 ? 0 : rrow)) * G::A;
    const bool expanded = t.visits[nb] > 0 && rrow >= 0 && !(t.st[nb * G::SW + 3 * G::K] & FLAG_DONE);
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
    const size_t nb = (size_t)g * t.ncap;
    const int rrow = t.rowof[nb];
    const size_t row = ((size_t)g * t.rcap + (rrow < 0 ? 0 : rrow)) * G::A;
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb, me, opp, inv, meta);
    const bool expanded = !uniform_only && t.visits[nb] > 0 && rrow >= 0 && !(meta & FLAG_DONE);
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
__device__ __forceinline__ int sample_action(const double* p, int A, int P, uint64_t gid, uint64_t step, uint64_t seed)
{
    uint32_t r[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const double u = (double)((((uint64_t)r[0] << 32) | r[1]) >> 11) * 0x1p-53;
    double total = 0.0;
    for (int a = 0; a < A; a++) total += p[a];
    const double target = u * total;
    double cum = 0.0;
    int pick = -1, last = P;
    for (int a = 0; a < A; a++) {
        if (p[a] > 0.0) {
            last = a;
            cum += p[a];
            if (pick < 0 && cum > target) pick = a;
        }
    }
    return pick >= 0 ? pick : last;
}

template <int S>
__global__ __launch_bounds__(64) void k_sample(Tree t, const double* __restrict__ pi, uint64_t seed, uint64_t step,
                                               int game_offset, int32_t* __restrict__ actions)
{
    using G = Geom<S>;
    const int g = blockIdx.x;
    if (lane_id() != 0) return;
    // global game id: the slot's id, plus (refill mode) how many games the slot has already finished
    const uint64_t gid = (uint64_t)(g + game_offset) + ((uint64_t)(uint32_t)t.episode[g] << 32);
    actions[g] = sample_action(pi + (size_t)g * G::A, G::A, G::P, gid, step, seed);
}

// The same draw with the Philox key given per row: game_ids[i] and steps[i] (the game's own move number), for
// callers that keep the games themselves and hand a context a different subset every move (go_ai.game.BatchedPlay).
template <int S>
__global__ __launch_bounds__(64) void k_sample_keyed(const double* __restrict__ pi, uint64_t seed,
                                                     const uint64_t* __restrict__ game_ids,
                                                     const int32_t* __restrict__ steps, int n,
                                                     int32_t* __restrict__ actions)
{
    using G = Geom<S>;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    actions[i] = sample_action(pi + (size_t)i * G::A, G::A, G::P, game_ids[i], (uint64_t)(uint32_t)steps[i], seed);
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
            if (lane == 0) atomicAdd(&t.counters[(size_t)2 * t.n_games + g], 1ull);
        } else if (lane == 0) {
            const int nturn = t.turn[g] ^ 1, nmoves = t.moves[g] + 1;
            const bool over = (nflags & FLAG_DONE) || nmoves > max_moves;
            const int res = nturn ? -sgn : sgn;                 // black's result (game.py:113)
            if (over) {
                const size_t G = (size_t)t.n_games;
                t.fin[0 * G + g] += 1;
                t.fin[1 * G + g] += res == 1;
                t.fin[2 * G + g] += res == -1;
                t.fin[3 * G + g] += nmoves;
            }
            if (over && t.refill) {
                // the slot starts its next game: go_env.reset() (game.py:139-140) = the empty board, black to move
#pragma unroll
                for (int k = 0; k < G::K; k++) { nme[k] = 0; nopp[k] = 0; ninv[k] = 0; }
                node_store<S>(t, nb, nme, nopp, ninv, pack_meta(0, 0));
                t.turn[g] = 0;
                t.moves[g] = 0;
                t.episode[g] += 1;
            } else {
                node_store<S>(t, nb, nme, nopp, ninv, pack_meta(nflags, sgn));
                t.turn[g] = nturn;
                t.moves[g] = nmoves;
                if (over) t.result[g] = res;
            }
        }
    }
    if (lane == 0) {
        t.visits[nb] = 0;
        t.rowof[nb] = -1;
        t.nnodes[g] = 1;
        t.nrows[g] = 0;
        if (t.nlog) t.nlog[g] = 0;
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

// ============================================================================================
// Q-value search (critic mode): mct_step with actor_critic=None
// (/root/reference/go_ai/search/mct.py:27-30,43-47): the leaf is evaluated, then ALL its valid
// children are created (Node.make_children, tree.py:88-98), evaluated in one batch
// (tree.set_state_vals, tree.py:14-19) and the leaf's prior is softmax(-child values) over the
// valid moves (Node.set_prior_pi(None), tree.py:152-157, fp64).
// ============================================================================================

// Exclusive scan of the per-game batch sizes -> q_off[0..G]; one 1024-thread workgroup.
__global__ __launch_bounds__(1024) void k_q_scan(Tree t)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x, G = t.n_active;
    const int per = (G + 1023) / 1024;
    const int lo = tid * per, hi = min(G, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += t.q_cnt[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int i = lo; i < hi; i++) { t.q_off[i] = run; run += t.q_cnt[i]; }
    if (tid == 1023) t.q_off[G] = part[1023];
}

// rank of action a among the valid actions of a node (ascending action order; the pass is last)
template <int S>
__device__ __forceinline__ int valid_rank(const uint64_t* inv, int a)
{
    using G = Geom<S>;
    int r = 0;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int lo = 64 * k;
        const int rem = G::P - lo;
        const uint64_t cells = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
        uint64_t m = ~inv[k] & cells;
        if (a < lo + 64) m &= (a <= lo) ? 0ull : (((a - lo) >= 64) ? ~0ull : ((1ull << (a - lo)) - 1ull));
        r += __popcll(m);
    }
    return r;
}

// One wavefront per (game, action): create the child node of the selected leaf for that action
// and write its planes into the evaluation batch.  The wave of the pass action also writes the
// leaf's own planes (entry 0 of the game's slice).  children_out / valid_out (optional) receive
// GoGame.children(leaf, canonical=True, padded=True) and the valid mask, for callers that
// evaluate with their own value function.
template <int S>
__global__ __launch_bounds__(64) void k_q_expand(Tree t, uint8_t* __restrict__ children_out,
                                                 uint8_t* __restrict__ valid_out)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const long total = (long)t.n_active * G::A;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const int g = (int)(w / G::A), a = (int)(w % G::A), lane = lane_id();
        const size_t nb = (size_t)g * t.ncap, rb = (size_t)g * t.rcap;
        const int leaf = t.leaf[g];
        uint64_t me[G::K], opp[G::K], inv[G::K], meta;
        node_load<S>(t, nb + leaf, me, opp, inv, meta);
        const bool done = meta & FLAG_DONE;
        const int off = t.q_off[g];
        if (a == G::P) planes_store<S>(t.q_planes + (size_t)off * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
        const int lrow = t.rowof[nb + leaf];
        const bool valid = !done && lrow >= 0 && ((a == G::P) || !((inv[a >> 6] >> (a & 63)) & 1ull));
        uint8_t* cout = children_out ? children_out + (size_t)w * 6 * G::P : nullptr;
        if (valid_out && lane == 0) valid_out[w] = valid ? 1 : 0;
        if (!valid) {
            if (lane == 0 && !done && lrow >= 0) t.child[(rb + lrow) * G::A + a] = -2;
            if (cout) for (int j = lane; j < 6 * G::P; j += 64) cout[j] = 0;
            continue;
        }
        const int j = valid_rank<S>(inv, a);
        const int idx = t.nnodes[g] + j;        // nnodes is advanced by the backup kernel
        uint64_t nme[G::K], nopp[G::K], ninv[G::K];
        uint32_t nflags = 0;
        int sgn = 0;
        const int rc = board_next_state<S>(L, me, opp, inv, (uint32_t)(meta & 7u), a, nme, nopp, ninv, nflags, sgn);
        if (rc != 0 || idx >= t.ncap) {
            if (lane == 0) { atomicAdd(&t.counters[(size_t)2 * t.n_games + g], 1ull); t.child[(rb + lrow) * G::A + a] = -2; }
            __syncthreads();
            continue;
        }
        if (lane == 0) {
            node_store<S>(t, nb + idx, nme, nopp, ninv, pack_meta(nflags, sgn));
            t.parent[nb + idx] = leaf;
            t.pact[nb + idx] = a;
            t.visits[nb + idx] = 0;
            t.rowof[nb + idx] = -1;
            t.child[(rb + lrow) * G::A + a] = idx;
            atomicAdd(&t.counters[(size_t)1 * t.n_games + g], 1ull);
        }
        planes_store<S>(t.q_planes + (size_t)(off + 1 + j) * 6 * G::P, nme, nopp, ninv, 0, nflags);
        if (cout) planes_store<S>(cout, nme, nopp, ninv, 0, nflags);
        __syncthreads();
    }
}

// Second half of a critic-mode mct_step.  Values come either from the compacted batch
// (t.q_vals laid out by t.q_off: the context's own network) or from the caller:
// leaf_val [G] and child_vals_padded [G][A] float64 (entry a = value of the child reached by
// action a; the reference keeps val.item(), a Python float, tree.py:14-19).
// raw != 0: values are straight from the network and RLNet._numpy's override for finished
// states (/root/reference/go_ai/models/__init__.py:101-105) is applied here.
template <int S, bool LONG>
__global__ __launch_bounds__(64) void k_q_backup(Tree t, const double* __restrict__ leaf_val,
                                                 const double* __restrict__ child_vals_padded, int raw)
{
    using G = Geom<S>;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap, rb = (size_t)g * t.rcap;
    const int leaf = t.leaf[g];
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb + leaf, me, opp, inv, meta);
    const bool done = meta & FLAG_DONE;
    const int off = t.q_off[g];
    double v = leaf_val ? leaf_val[g] : (double)t.q_vals[off];
    if (raw && done) v = (double)(100 * meta_sign(meta));
    backprop_path<S, LONG>(t, g, v);

    const int lrow = t.rowof[nb + leaf];
    if (done || lrow < 0) return;
    const size_t row = (rb + lrow) * G::A;
    double q[G::K];
    bool valid[G::K];
    double m = -INFINITY;
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        valid[k] = a < G::A && ((a == G::P) || !((inv[k] >> lane) & 1ull));
        q[k] = 0.0;
        int ch = -1;
        if (valid[k]) ch = t.child[row + a];
        valid[k] = valid[k] && ch >= 0;
        if (valid[k]) {
            double cv = child_vals_padded ? child_vals_padded[(size_t)g * G::A + a]
                                          : (double)t.q_vals[off + 1 + valid_rank<S>(inv, a)];
            const uint64_t cmeta = t.st[(nb + ch) * G::SW + 3 * G::K];
            if (raw && (cmeta & FLAG_DONE)) cv = (double)(100 * meta_sign(cmeta));
            t.nval[nb + ch] = cv;
            q[k] = -cv;            // search.invert_vals(child.val), tree.py:128-135
            m = fmax(m, q[k]);
        }
        nvalid += __popcll(__ballot(valid[k]));
    }
    m = wave_max_f64(m);
    double e[G::K], part = 0.0;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        e[k] = valid[k] ? det_exp64(q[k] - m) : 0.0;
        part += e[k];
    }
    const double sum = wave_sum_f64(part);
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        if (a < G::A) {
            t.prior[row + a] = valid[k] ? e[k] / sum : 0.0;
            t.crow[row + a] = -1;
            t.cN[row + a] = 0;
            t.cQ[row + a] = 0.0;
        }
    }
    if (lane == 0) {
        t.nnodes[g] += nvalid;
        atomicAdd(&t.counters[(size_t)3 * t.n_games + g], (unsigned long long)(1 + nvalid));
    }
}

// rootnode.inverted_children_values() (tree.py:128-135): -child.val, 0 where there is no child.
template <int S>
__global__ __launch_bounds__(64) void k_q_child_values(Tree t, double* __restrict__ out)
{
    using G = Geom<S>;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    const int rrow = t.rowof[nb];
    const bool expanded = t.visits[nb] > 0 && rrow >= 0 && !(t.st[nb * G::SW + 3 * G::K] & FLAG_DONE);
    const size_t row = ((size_t)g * t.rcap + (rrow < 0 ? 0 : rrow)) * G::A;
    for (int a = lane; a < G::A; a += 64) {
        const int ch = expanded ? t.child[row + a] : -1;
        out[(size_t)g * G::A + a] = ch >= 0 ? -t.nval[nb + ch] : 0.0;
    }
}

// Value.__call__ (/root/reference/go_ai/policies/value.py:30-37), temp > 0:
//   qs = visit counts (mcts > 0)  |  exp(-child values) (mcts == 0)
//   pi = search.temp_norm(qs, temp, valid)  (/root/reference/go_ai/search/__init__.py:78-87):
//        L1-normalise qs over the valid moves, ** (1/temp), L1-normalise over all A entries;
// the L1 norms are numpy pairwise sums over the COMPACTED valid entries, then over all A.
// Finished games and unexpanded roots get the uniform-valid distribution.
template <int S>
__global__ __launch_bounds__(64) void k_q_policy(Tree t, double temp, int from_values, double* __restrict__ pi)
{
    using G = Geom<S>;
    __shared__ double comp[G::A], full[G::A];
    __shared__ double total;
    const int g = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    const int rrow = t.rowof[nb];
    const size_t row = ((size_t)g * t.rcap + (rrow < 0 ? 0 : rrow)) * G::A;
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, nb, me, opp, inv, meta);
    const bool expanded = t.visits[nb] > 0 && rrow >= 0 && !(meta & FLAG_DONE);
    bool valid[G::K];
    double qv[G::K];
    int rank[G::K];
    int nvalid = 0;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        valid[k] = a < G::A && ((a == G::P) || !((inv[k] >> lane) & 1ull));
        const uint64_t b = __ballot(valid[k]);
        rank[k] = nvalid + __popcll(b & ((1ull << lane) - 1ull));
        nvalid += __popcll(b);
        qv[k] = 0.0;
        if (valid[k] && expanded) {
            if (from_values) {
                const int ch = t.child[row + a];
                qv[k] = ch >= 0 ? det_exp64(-t.nval[nb + ch]) : 1.0;
            } else {
                qv[k] = (double)t.cN[row + a];
            }
        }
        if (valid[k]) comp[rank[k]] = fabs(qv[k]);
    }
    if (temp <= 0.0) {
        // search.greedy_pi(qs, valid) (/root/reference/go_ai/search/__init__.py:46-52): expq = exp(qs - max(qs)) * valid
        // over ALL A entries (an invalid move's qs is 0 as a visit count, exp(-0) = 1 as a value), then uniform over the
        // entries equal to max(expq)
        double mx = -INFINITY;
        double full_q[G::K];
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            full_q[k] = (valid[k] && expanded) ? qv[k] : ((from_values && expanded) ? 1.0 : 0.0);
            if (a < G::A) mx = fmax(mx, full_q[k]);
        }
        mx = wave_max_f64(mx);
        double ex[G::K], emax = 0.0;
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            ex[k] = valid[k] ? det_exp64(full_q[k] - mx) : 0.0;
            emax = fmax(emax, ex[k]);
        }
        emax = wave_max_f64(emax);
        int hits = 0;
        bool hit[G::K];
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            hit[k] = (lane + 64 * k) < G::A && ex[k] == emax;
            hits += __popcll(__ballot(hit[k]));
        }
#pragma unroll
        for (int k = 0; k < G::K; k++) {
            const int a = lane + 64 * k;
            if (a < G::A) pi[(size_t)g * G::A + a] = hit[k] ? 1.0 / (double)hits : 0.0;
        }
        return;
    }
    __syncthreads();
    if (lane == 0) total = np_pairwise_sum(comp, nvalid);
    __syncthreads();
    const double s1 = total;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        if (a < G::A) {
            double x = valid[k] ? (s1 != 0.0 ? qv[k] / s1 : qv[k]) : 0.0;
            if (temp != 1.0) x = pow(x, 1.0 / temp);
            full[a] = x;
        }
    }
    __syncthreads();
    if (lane == 0) total = np_pairwise_sum(full, G::A);     // entries are >= 0: |x| = x
    __syncthreads();
    const double s2 = total;
#pragma unroll
    for (int k = 0; k < G::K; k++) {
        const int a = lane + 64 * k;
        if (a < G::A) {
            const double noise = (valid[k] ? 1.0 : 0.0) / (double)nvalid;
            pi[(size_t)g * G::A + a] = (expanded && s2 > 0.0) ? full[a] / s2 : noise;
        }
    }
}

// The whole tree of one game, node by node, for the host-side Node views (tree.Node's fields,
// /root/reference/go_ai/search/tree.py:22-52): parent, the action that led to the node, visits, node.val
// (Q mode; NaN where none), the sum of node.post_vals, the prior row (NaN row where the node has none) and
// node.state.  One wavefront per node slot; slots beyond the game's node count are left alone.
template <int S>
__global__ __launch_bounds__(64) void k_mcts_export(Tree t, int g, int32_t* __restrict__ n_nodes,
                                                    int32_t* __restrict__ parent, int32_t* __restrict__ action,
                                                    int32_t* __restrict__ visits, double* __restrict__ value,
                                                    double* __restrict__ raw_sum, double* __restrict__ prior,
                                                    uint8_t* __restrict__ states)
{
    using G = Geom<S>;
    const int n = blockIdx.x, lane = lane_id();
    const size_t nb = (size_t)g * t.ncap;
    const int count = t.nnodes[g];
    if (n == 0 && lane == 0) *n_nodes = count;
    if (n >= count) return;
    const size_t xi = nb + n;
    const int vis = t.visits[xi];
    const int row = t.rowof[xi];
    uint64_t me[G::K], opp[G::K], inv[G::K], meta;
    node_load<S>(t, xi, me, opp, inv, meta);
    // a row holds a prior once the node was evaluated as a non-terminal leaf (visited at least once)
    const bool has_prior = row >= 0 && vis > 0 && !(meta & FLAG_DONE);
    if (lane == 0) {
        parent[n] = t.parent[xi];
        action[n] = t.pact[xi];
        visits[n] = vis;
        value[n] = (t.nval && n > 0) ? t.nval[xi] : __longlong_as_double(0x7ff8000000000000ll);
        raw_sum[n] = vis > 0 ? t.rawsum[xi] : 0.0;
    }
    for (int a = lane; a < G::A; a += 64)
        prior[(size_t)n * G::A + a] = has_prior ? t.prior[((size_t)g * t.rcap + row) * G::A + a]
                                                : __longlong_as_double(0x7ff8000000000000ll);
    planes_store<S>(states + (size_t)n * 6 * G::P, me, opp, inv, 0, (uint32_t)(meta & 7u));
}

// ---- launchers -----------------------------------------------------------------------------
#define LAUNCH_PER_GAME(KERNEL, ...)                                                              \
    GOAI_DISPATCH_SIZE(S_rt, KERNEL<S><<<dim3(t.n_active), dim3(64), 0, st>>>(__VA_ARGS__));      \
    GOAI_HIP_CHECK(hipGetLastError());                                                            \
    return GOAI_OK

int launch_mcts_reset(int S_rt, const Tree& t, const uint8_t* roots, hipStream_t st) { LAUNCH_PER_GAME(k_mcts_reset, t, roots); }
int launch_mcts_select(int S_rt, const Tree& t, uint8_t* leaf_out, hipStream_t st) { LAUNCH_PER_GAME(k_mcts_select, t, leaf_out); }
int launch_post_forward(int S_rt, const Tree& t, const float* act, const HeadWeights& hw, const void* w0_split, const float* b0,
                        float shift, float* act0, hipStream_t st, bool act_is_h)
{
    const int grid = (t.n_active + 1) / 2;
    GOAI_DISPATCH_SIZE(S_rt, k_post_forward<S><<<dim3(grid), dim3(256), 0, st>>>(t, act, hw, (const uint4*)w0_split, b0, shift, act0, act_is_h));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}
// n_steps trunk passes with the search's work between them in one launch (k_search_steps): 9x9, n_active a multiple of 4,
// split fp16 weights of the whole trunk (2 * layers convolutions), parts = 2 (f16x3) or 1 (f16)
int launch_search_steps(const Tree& t, float* x, float* y, const void* wt_split, const float* bias, int n_convs, int parts,
                        const int* wexp, float amax_first, const HeadWeights& hw, float* hbuf, const void* w0_split,
                        const float* b0, float shift, int n_steps, int spread, hipStream_t st)
{
    if (t.n_active < 4 || (t.n_active & 3) || n_convs < 2 || (n_convs & 1) || (parts != 1 && parts != 2) || n_steps < 1) {
        set_error("k_search_steps: 9x9, a multiple of four games, a whole residual trunk");
        return GOAI_ERR_ARG;
    }
    const dim3 grid(t.n_active / 4), block(512);
    if (parts == 2)
        k_search_steps<2><<<grid, block, 0, st>>>(t, x, y, (const uint4*)wt_split, bias, n_convs, wexp, amax_first, hw, hbuf,
                                                   (const uint4*)w0_split, b0, shift, n_steps, spread);
    else
        k_search_steps<1><<<grid, block, 0, st>>>(t, x, y, (const uint4*)wt_split, bias, n_convs, wexp, amax_first, hw, hbuf,
                                                   (const uint4*)w0_split, b0, shift, n_steps, spread);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}
int launch_mcts_backup(int S_rt, const Tree& t, const float* pi_logits, const float* prior, const float* val, int raw,
                       hipStream_t st)
{
    if (t.vlog) {
        GOAI_DISPATCH_SIZE(S_rt, (k_mcts_backup<S, true><<<dim3(t.n_active), dim3(64), 0, st>>>(t, pi_logits, prior, val, raw)));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, (k_mcts_backup<S, false><<<dim3(t.n_active), dim3(64), 0, st>>>(t, pi_logits, prior, val, raw)));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
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
int launch_sample_keyed(int S_rt, const double* pi, uint64_t seed, const uint64_t* game_ids, const int32_t* steps, int n,
                        int32_t* actions, hipStream_t st)
{
    GOAI_DISPATCH_SIZE(S_rt, k_sample_keyed<S><<<dim3((n + 63) / 64), dim3(64), 0, st>>>(pi, seed, game_ids, steps, n, actions));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}
int launch_mcts_advance(int S_rt, const Tree& t, const int32_t* actions, int max_moves, hipStream_t st)
{
    LAUNCH_PER_GAME(k_mcts_advance, t, actions, max_moves);
}
int launch_mcts_roots(int S_rt, const Tree& t, uint8_t* states, hipStream_t st) { LAUNCH_PER_GAME(k_mcts_roots, t, states); }

int launch_mcts_export(int S_rt, const Tree& t, int game, int32_t* n_nodes, int32_t* parent, int32_t* action, int32_t* visits,
                       double* value, double* raw_sum, double* prior, uint8_t* states, hipStream_t st)
{
    GOAI_DISPATCH_SIZE(S_rt, k_mcts_export<S><<<dim3(t.ncap), dim3(64), 0, st>>>(t, game, n_nodes, parent, action, visits, value,
                                                                               raw_sum, prior, states));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_q_scan(const Tree& t, hipStream_t st)
{
    k_q_scan<<<dim3(1), dim3(1024), 0, st>>>(t);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}
int launch_q_expand(int S_rt, const Tree& t, uint8_t* children_out, uint8_t* valid_out, hipStream_t st)
{
    const long units = (long)t.n_active * (S_rt * S_rt + 1);
    const int grid = (int)(units < 262144 ? units : 262144);
    GOAI_DISPATCH_SIZE(S_rt, k_q_expand<S><<<dim3(grid), dim3(64), 0, st>>>(t, children_out, valid_out));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}
int launch_q_backup(int S_rt, const Tree& t, const double* leaf_val, const double* child_vals_padded, int raw, hipStream_t st)
{
    if (t.vlog) {
        GOAI_DISPATCH_SIZE(S_rt, (k_q_backup<S, true><<<dim3(t.n_active), dim3(64), 0, st>>>(t, leaf_val, child_vals_padded, raw)));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, (k_q_backup<S, false><<<dim3(t.n_active), dim3(64), 0, st>>>(t, leaf_val, child_vals_padded, raw)));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}
int launch_q_child_values(int S_rt, const Tree& t, double* inv_vals, hipStream_t st)
{
    LAUNCH_PER_GAME(k_q_child_values, t, inv_vals);
}
int launch_q_policy(int S_rt, const Tree& t, double temp, int from_values, double* pi, hipStream_t st)
{
    LAUNCH_PER_GAME(k_q_policy, t, temp, from_values, pi);
}

}  // namespace goai
