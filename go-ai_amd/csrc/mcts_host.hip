// mcts_host.hip -- allocation of the per-game tree arrays and the lock-step search driver:
// every step is  select (all games) -> CNN forward over the leaf batch -> expand + backup.
// (/root/reference/go_ai/search/mct.py:50-62 runs the same three phases one leaf at a time.)
#include <cmath>
#include <cstdlib>
#include <vector>

#include "ctx.h"
#include "go_board.h"
#include "goai_internal.h"

namespace goai {

template <typename T>
static int dev_alloc_t(goai_ctx* c, T** p, size_t count)
{
    GOAI_HIP_CHECK(hipMalloc((void**)p, count * sizeof(T)));
    c->allocs.push_back(*p);
    return GOAI_OK;
}

static int state_words(int S)
{
    const int P = S * S;
    return 3 * ((P + 63) / 64) + 1;
}

static int mcts_init_impl(goai_ctx* c, int n_games, int max_sims, int mode);

// A failed init leaves the context as it was (its partial allocations are freed), so it can be retried.
int mcts_init(goai_ctx* c, int n_games, int max_sims, int mode)
{
    GOAI_REQUIRE(c->tree.n_games == 0, "goai_mcts_init was already called on this context");
    const size_t mark = c->allocs.size(), host_mark = c->host_allocs.size();
    const int rc = mcts_init_impl(c, n_games, max_sims, mode);
    if (rc) {
        c->release_from(mark, host_mark);
        c->tree = Tree();
    }
    return rc;
}

static int mcts_init_impl(goai_ctx* c, int n_games, int max_sims, int mode)
{
    GOAI_REQUIRE(n_games > 0, "n_games must be positive");
    // numpy's pairwise mean is maintained incrementally for up to 128 values per node; beyond that every visited node keeps
    // its value list (Tree::vlog, (max_sims + 1)(max_sims + 2) / 2 doubles per game)
    GOAI_REQUIRE(max_sims >= 0 && max_sims <= GOAI_MAX_SIMS, "max_sims must be in [0, GOAI_MAX_SIMS = 2048]");
    GOAI_REQUIRE(mode == GOAI_SEARCH_AC || mode == GOAI_SEARCH_Q, "mode must be GOAI_SEARCH_AC or GOAI_SEARCH_Q");
    Tree& t = c->tree;
    t.max_sims = max_sims;
    t.rcap = max_sims + 2;
    // AC: one new node per step.  Q: every expansion adds up to A children (tree.py:88-98).
    t.ncap = mode == GOAI_SEARCH_AC ? max_sims + 2 : 2 + (max_sims + 1) * c->A;
    t.mode = mode;
    const size_t G = n_games, N = G * t.ncap, R = G * t.rcap, A = c->A, P = c->P;
    int rc;
#define AL(field, count) if ((rc = dev_alloc_t(c, &t.field, (count)))) return rc
    AL(st, N * state_words(c->S));
    AL(parent, N); AL(pact, N); AL(visits, N); AL(rowof, N);
    AL(accR, N * 8); AL(accT, N * 8); AL(rawsum, N);
    if (max_sims > 128) {
        const size_t M = (size_t)max_sims + 1;
        AL(vlog, G * (M * (M + 1) / 2)); AL(logof, N); AL(nlog, G);
    }
    AL(prior, R * A); AL(child, R * A); AL(crow, R * A); AL(cN, R * A); AL(cQ, R * A);
    AL(nnodes, G); AL(nrows, G); AL(leaf, G); AL(depth, G); AL(path, R);
    AL(turn, G); AL(moves, G); AL(result, G); AL(episode, G); AL(fin, 4 * G);
    AL(leaf_planes, G * 6 * P); AL(pi_logits, G * A); AL(val_logits, G);
    AL(sqrt_tab, (size_t)max_sims + 2);
    AL(counters, 4 * G);
    if (mode == GOAI_SEARCH_Q) {
        AL(nval, N);
        AL(q_cnt, G); AL(q_off, G + 1);
        AL(q_planes, G * (1 + A) * 6 * P);
        AL(q_vals, G * (1 + A));
        GOAI_HIP_CHECK(hipHostMalloc((void**)&t.q_total_host, sizeof(int32_t), hipHostMallocDefault));
        c->host_allocs.push_back(t.q_total_host);
    }
#undef AL
    std::vector<double> tab(max_sims + 2);
    for (int i = 0; i < max_sims + 2; i++) tab[i] = std::sqrt((double)i);   // IEEE correctly rounded on the host
    GOAI_HIP_CHECK(hipMemcpy(t.sqrt_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    GOAI_HIP_CHECK(hipMemset(t.counters, 0, 4 * G * sizeof(unsigned long long)));
    GOAI_HIP_CHECK(hipMemset(t.visits, 0, N * sizeof(int32_t)));
    GOAI_HIP_CHECK(hipMemset(t.episode, 0, G * sizeof(int32_t)));
    GOAI_HIP_CHECK(hipMemset(t.fin, 0, 4 * G * sizeof(long long)));
    t.n_games = n_games;
    t.n_active = n_games;
    t.steps_since_reset = -1;   // no roots yet
    return GOAI_OK;
}

#define REQUIRE_TREE(c) GOAI_REQUIRE((c)->tree.n_games > 0, "call goai_mcts_init first")
#define REQUIRE_ROOTS(c)                                                                    \
    do {                                                                                    \
        REQUIRE_TREE(c);                                                                    \
        if ((c)->tree.steps_since_reset < 0) { set_error("call goai_mcts_reset first"); return GOAI_ERR_STATE; } \
    } while (0)

int mcts_reset(goai_ctx* c, const uint8_t* roots)
{
    REQUIRE_TREE(c);
    GOAI_REQUIRE(roots != nullptr, "root_states is NULL");
    const int rc = launch_mcts_reset(c->S, c->tree, roots, c->stream);
    if (rc) return rc;
    c->tree.steps_since_reset = 0;
    c->tree.pending_select = false;
    c->tree.pending_expand = 0;
    return GOAI_OK;
}

int mcts_select(goai_ctx* c, uint8_t* leaf_states_out)
{
    REQUIRE_ROOTS(c);
    Tree& t = c->tree;
    if (t.pending_select) { set_error("goai_mcts_select called twice without goai_mcts_backup"); return GOAI_ERR_STATE; }
    if (t.steps_since_reset > t.max_sims) {
        set_error("more than max_sims + 1 search steps since the last reset/advance");
        return GOAI_ERR_STATE;
    }
    const int rc = launch_mcts_select(c->S, t, leaf_states_out, c->stream);
    if (rc) return rc;                      // the state flags move only once the launch is in the stream
    t.pending_select = true;
    t.pending_expand = t.mode == GOAI_SEARCH_Q ? 1 : 0;
    return GOAI_OK;
}

// Q mode, between select and backup: node.make_children() for every non-terminal leaf
// (mct.py:44-45).  Fills the compacted evaluation batch; optional padded outputs for callers
// that bring their own value function.
int mcts_expand(goai_ctx* c, uint8_t* children_out, uint8_t* valid_out)
{
    REQUIRE_ROOTS(c);
    Tree& t = c->tree;
    GOAI_REQUIRE(t.mode == GOAI_SEARCH_Q, "goai_mcts_expand needs a context initialised with GOAI_SEARCH_Q");
    if (!t.pending_select || t.pending_expand != 1) {
        set_error("goai_mcts_expand must follow goai_mcts_select");
        return GOAI_ERR_STATE;
    }
    int rc;
    if ((rc = launch_q_scan(t, c->stream))) return rc;
    if ((rc = launch_q_expand(c->S, t, children_out, valid_out, c->stream))) return rc;
    t.pending_expand = 2;
    return GOAI_OK;
}

int mcts_backup_q(goai_ctx* c, const double* leaf_val, const double* child_vals, int raw)
{
    REQUIRE_ROOTS(c);
    Tree& t = c->tree;
    GOAI_REQUIRE(t.mode == GOAI_SEARCH_Q, "goai_mcts_backup_q needs a context initialised with GOAI_SEARCH_Q");
    if (!t.pending_select || t.pending_expand != 2) {
        set_error("goai_mcts_backup_q must follow goai_mcts_select and goai_mcts_expand");
        return GOAI_ERR_STATE;
    }
    GOAI_REQUIRE((leaf_val != nullptr) == (child_vals != nullptr), "pass both leaf_val and child_vals, or neither");
    const int rc = launch_q_backup(c->S, t, leaf_val, child_vals, raw, c->stream);
    if (rc) return rc;
    t.pending_select = false;
    t.pending_expand = 0;
    t.steps_since_reset++;
    return GOAI_OK;
}

int mcts_backup(goai_ctx* c, const float* pi_logits, const float* prior, const float* val, int raw)
{
    REQUIRE_ROOTS(c);
    Tree& t = c->tree;
    GOAI_REQUIRE(t.mode == GOAI_SEARCH_AC, "goai_mcts_backup is the actor-critic step; use goai_mcts_backup_q");
    if (!t.pending_select) { set_error("goai_mcts_backup without a preceding goai_mcts_select"); return GOAI_ERR_STATE; }
    GOAI_REQUIRE(val != nullptr, "val_logits is NULL");
    GOAI_REQUIRE((pi_logits != nullptr) != (prior != nullptr), "pass exactly one of pi_logits / prior");
    const int rc = launch_mcts_backup(c->S, t, pi_logits, prior, val, raw, c->stream);
    if (rc) return rc;
    t.pending_select = false;
    t.steps_since_reset++;
    return GOAI_OK;
}

// n_steps x mct_step with the context's own network (mct_search = 1 + n_sims of them).
int mcts_steps(goai_ctx* c, int n_steps)
{
    REQUIRE_ROOTS(c);
    Tree& t = c->tree;
    GOAI_REQUIRE(n_steps >= 0 && t.steps_since_reset + n_steps <= t.max_sims + 1,
                 "more than max_sims + 1 search steps since the last reset/advance");
    int rc;
    if (t.mode == GOAI_SEARCH_Q) {
        GOAI_REQUIRE(c->net.loaded, "a network with loaded weights is required");
        for (int s = 0; s < n_steps; s++) {
            if ((rc = mcts_select(c, nullptr))) return rc;
            if ((rc = mcts_expand(c, nullptr, nullptr))) return rc;
            // the batch size is data dependent (number of legal moves of every leaf)
            GOAI_HIP_CHECK(hipMemcpyAsync(t.q_total_host, t.q_off + t.n_active, sizeof(int32_t), hipMemcpyDeviceToHost,
                                          c->stream));
            GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
            const int total = *t.q_total_host;
            GOAI_REQUIRE(total >= t.n_active && (size_t)total <= (size_t)t.n_active * (1 + c->A), "corrupt batch size");
            if ((rc = nn_forward_raw(c, t.q_planes, nullptr, t.q_vals, total))) return rc;
            if ((rc = mcts_backup_q(c, nullptr, nullptr, 1))) return rc;
        }
        return GOAI_OK;
    }
    GOAI_REQUIRE(c->net.loaded && c->net.kind == 0, "an actor-critic network with loaded weights is required");
    // The lock-step loop with everything between two trunk launches in ONE launch (k_post_forward: heads, backup, the
    // next step's select, first convolution -- the same device code as the separate kernels, same bits).  Needs the whole
    // batch in one forward and the matrix-core first convolution; GOAI_NO_STEP_FUSION=1 keeps the separate launches.
    const char* nf = getenv("GOAI_NO_STEP_FUSION");          // read at every call: the tests run both paths in one process
    const bool no_fusion = nf && nf[0] == '1';
    // Only while all of its workgroups (two games each, three per CU) are resident at once: then the separate kernels are
    // latency-bound chains and fusing them saves their hand-overs (C2, 1024 games: +1.2 %); with more games than that the
    // separate kernels fill the chip better than a workgroup that is half idle in its middle phase (C4 / C5: -2 %).
    const bool small_batch = t.n_active <= 6 * c->net.n_cu;
    // (k_post_forward carries the <= 128-value backup only: long searches keep the separate launches)
    c->steps_in_one_launch = 0;
    if (!no_fusion && small_batch && !t.vlog && n_steps >= 2 && !t.pending_select && t.n_active <= c->net.max_batch && c->net.w0_split) {
        float* x = nullptr;
        bool x_is_h = false;                    // x = the heads' 1x1 convolutions, formed by the trunk launch (nn_host.hip)
        if ((rc = mcts_select(c, nullptr))) return rc;
        if ((rc = nn_piece_conv0(c, t.leaf_planes, t.n_active))) return rc;
        // 9x9 with the fp16 F(3x3,3x3) trunk and its heads in the last epilogue: the whole loop below in ONE launch
        // (k_search_steps, mcts_kernels.hip: no workgroup ever waits for another).  OPT-IN, GOAI_STEPS_KERNEL=1: same bits, but
        // measured SLOWER than the launches below (bench.py C2: 29.7 against 25.5 ms per move) -- a pass of the trunk takes what
        // it takes with a start stagger (272 us), and the search's phases between two passes take 71 us instead of 22 when
        // the rest of the chip is in its trunk passes at the time (dependent loads under load: heads 6.6 -> 30.8 us,
        // backup + select 8.6 -> 17.3, first convolution 6.3 -> 22.5; tools/steps_stamps.py).
        const Net& net = c->net;
        const char* ns = getenv("GOAI_STEPS_KERNEL");
        if (c->S == 9 && net.wino3h && net.split6 && net.trunk_fused && net.heads_in_trunk && net.layers > 0 && (t.n_active & 3) == 0 &&
            t.n_active >= 4 && ns && ns[0] == '1') {
            const char* sp = getenv("GOAI_STEPS_SPREAD");          // start spread in 64-clock units (a trunk pass of 3 blocks ~ 9000)
            const int spread = sp ? atoi(sp) : 3000 * net.layers;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c->prof_enabled) {
                GOAI_HIP_CHECK(hipEventCreate(&e0));
                GOAI_HIP_CHECK(hipEventCreate(&e1));
                c->prof_events.push_back(e0);
                c->prof_events.push_back(e1);
                GOAI_HIP_CHECK(hipEventRecord(e0, c->stream));
            }
            rc = launch_search_steps(t, net.act[0], net.act[1], net.wt_split6, net.bt, 2 * net.layers, net.parts, net.wexp, net.amax0,
                                     net.hw, net.hbuf, net.w0_split, net.b0, net.shift, n_steps, spread, c->stream);
            if (e1) {
                GOAI_HIP_CHECK(hipEventRecord(e1, c->stream));
                c->prof_extra_passes += n_steps - 1;
            }
            if (rc) return rc;
            t.steps_since_reset += n_steps - 1;     // their backups and the next steps' selects ran inside
            c->steps_in_one_launch = n_steps;
            if ((rc = nn_piece_heads(c, net.hbuf, true, t.pi_logits, t.val_logits, t.n_active))) return rc;
            return mcts_backup(c, t.pi_logits, nullptr, t.val_logits, 1);
        }
        for (int s = 0; s < n_steps; s++) {
            if ((rc = nn_piece_trunk(c, t.n_active, &x, &x_is_h))) return rc;
            if (s + 1 < n_steps) {
                if ((rc = launch_post_forward(c->S, t, x, c->net.hw, c->net.w0_split, c->net.b0, c->net.shift, c->net.act[0],
                                              c->stream, x_is_h)))
                    return rc;
                t.steps_since_reset++;          // that step's backup and the next step's select are in the stream
            }
        }
        if ((rc = nn_piece_heads(c, x, x_is_h, t.pi_logits, t.val_logits, t.n_active))) return rc;
        return mcts_backup(c, t.pi_logits, nullptr, t.val_logits, 1);
    }
    for (int s = 0; s < n_steps; s++) {
        if ((rc = mcts_select(c, nullptr))) return rc;
        if ((rc = nn_forward_raw(c, t.leaf_planes, t.pi_logits, t.val_logits, t.n_active))) return rc;
        if ((rc = mcts_backup(c, t.pi_logits, nullptr, t.val_logits, 1))) return rc;
    }
    return GOAI_OK;
}

int mcts_advance(goai_ctx* c, const int32_t* actions)
{
    REQUIRE_ROOTS(c);
    GOAI_REQUIRE(actions != nullptr, "actions is NULL");
    const int rc = launch_mcts_advance(c->S, c->tree, actions, 2 * c->P, c->stream);
    if (rc) return rc;
    c->tree.steps_since_reset = 0;
    c->tree.pending_select = false;
    c->tree.pending_expand = 0;
    return GOAI_OK;
}

}  // namespace goai
