// mcts.h -- GPU-resident per-game search trees (struct-of-arrays over games x nodes).
//
// Replaces the Python object tree of /root/reference/go_ai/search/tree.py:22-52: every
// tree.Node field becomes one array indexed [game][node]; the per-action data a parent
// needs for PUCT (tree.py:170-184) is kept in rows indexed [game][node][action] so that a
// wavefront reads them coalesced, one action per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nn.h"

namespace goai {

struct Tree {
    int n_games = 0, max_sims = 0, mode = 0;
    int n_active = 0;            // games the calls work on: the first n_active slots (goai_mcts_set_active; default all)
    int ncap = 0;                // nodes per game: AC max_sims + 2 (one new node per step); Q 2 + (max_sims + 1) * A
    int rcap = 0;                // expanded nodes (rows) and path entries per game: max_sims + 2 in both modes
    int steps_since_reset = 0;
    bool pending_select = false;
    int pending_expand = 0;      // Q mode: 1 after select, 2 after expand
    // nodes: [G][ncap]
    uint64_t* st = nullptr;      // [..][SW] packed state: me[K], opp[K], inv[K], meta
    int32_t* parent = nullptr;
    int32_t* pact = nullptr;     // action that led here
    int32_t* visits = nullptr;
    int32_t* rowof = nullptr;    // row of an expanded node (set by select when the node becomes a non-terminal leaf)
    double* nval = nullptr;      // Q mode: node.val (tree.py:122-126), the critic's value of the node's state
    double* accR = nullptr;      // [..][8] lane accumulators of numpy's pairwise sum over full blocks of 8
    double* accT = nullptr;      // [..][8] pending tail (visits % 8 entries)
    // max_sims > 128 only (else NULL): every visited node keeps its tanh'ed values in order, because numpy's pairwise sum of
    // more than 128 values splits at n/2 - (n/2) % 8, a point that moves as the list grows (no incremental form).  A node
    // takes log slot s = the number of nodes visited before it in this search; at most one node is first visited per
    // step, so slot s starts at step >= s and holds at most M - s values (M = max_sims + 1 steps): slot s lives at
    // vlog[g * M (M + 1) / 2 + s M - s (s - 1) / 2 ...].
    double* vlog = nullptr;
    int32_t* logof = nullptr;    // [..] a visited node's log slot
    int32_t* nlog = nullptr;     // [G] slots handed out in this search
    double* rawsum = nullptr;    // sum of node.post_vals as backed up (raw values, not tanh): only str(node)'s "AV" reads it
    // rows: [G][rcap][A]
    double* prior = nullptr;
    int32_t* child = nullptr;    // node index of the child; -1 = valid move, no node yet; -2 = invalid move
    int32_t* crow = nullptr;     // the child's own row (rowof[child]), -1 while it is unvisited or terminal
    int32_t* cN = nullptr;       // child visit counts
    double* cQ = nullptr;        // -mean(tanh(child.post_vals)), 0 while unvisited
    // per game: [G]
    int32_t* nnodes = nullptr;
    int32_t* leaf = nullptr;
    int32_t* depth = nullptr;
    int32_t* nrows = nullptr;
    int32_t* path = nullptr;     // [G][rcap]
    int32_t* turn = nullptr;     // absolute colour to move at the root (0 black, 1 white)
    int32_t* moves = nullptr;    // moves played so far in the game
    int32_t* result = nullptr;   // 2 = running, else black's result {-1,0,1} once finished
    int32_t* episode = nullptr;  // games this slot has finished and replaced by a fresh one (refill mode)
    long long* fin = nullptr;    // [4][G] totals over the games a slot finished: count, black wins, white wins, moves
    int refill = 0;              // goai_mcts_set_refill: a finished game's slot restarts from the empty board
    // NN hand-off
    uint8_t* leaf_planes = nullptr;   // [G][6][P]
    float* pi_logits = nullptr;       // [G][A]
    float* val_logits = nullptr;      // [G]
    double* sqrt_tab = nullptr;       // sqrt(i), i <= max_sims + 1, rounded on the host
    unsigned long long* counters = nullptr;   // [4][n_games]: [0] leaf evals, [1] transitions, [2] errors, [3] states sent through the net
    // Q mode (mct.py:44-47): per step, the leaf of every game and its valid children form one
    // compacted evaluation batch: game g owns entries [q_off[g], q_off[g+1]) = leaf, children
    // in ascending action order
    int32_t* q_cnt = nullptr;         // [G]
    int32_t* q_off = nullptr;         // [G+1]
    uint8_t* q_planes = nullptr;      // [G*(1+A)][6][P]
    float* q_vals = nullptr;          // [G*(1+A)]
    int32_t* q_total_host = nullptr;  // pinned
};

// meta word of a packed state: bits 0..2 flags (go_board.h), bits 8..9 = area sign + 1
__host__ __device__ inline uint64_t pack_meta(uint32_t flags, int sign) { return (uint64_t)flags | ((uint64_t)(sign + 1) << 8); }
__host__ __device__ inline int meta_sign(uint64_t meta) { return (int)((meta >> 8) & 3) - 1; }

int launch_mcts_reset(int S, const Tree& t, const uint8_t* roots, hipStream_t st);
int launch_mcts_select(int S, const Tree& t, uint8_t* leaf_states_out, hipStream_t st);
int launch_search_steps(const Tree& t, float* x, float* y, const void* wt_split, const float* bias, int n_convs, int parts,
                        const int* wexp, float amax_first, const HeadWeights& hw, float* hbuf, const void* w0_split,
                        const float* b0, float shift, int n_steps, int spread, hipStream_t st);
int launch_post_forward(int S, const Tree& t, const float* act, const HeadWeights& hw, const void* w0_split, const float* b0,
                        float shift, float* act0, hipStream_t st, bool act_is_h = false);
int launch_mcts_backup(int S, const Tree& t, const float* pi_logits, const float* prior, const float* val,
                       int logits_are_raw, hipStream_t st);
int launch_mcts_readout(int S, const Tree& t, int32_t* visits, double* prior, int32_t* root_visits, hipStream_t st);
int launch_mcts_policy(int S, const Tree& t, double temp, double* pi, hipStream_t st);
// Q mode
int launch_q_scan(const Tree& t, hipStream_t st);
int launch_q_expand(int S, const Tree& t, uint8_t* children_out, uint8_t* valid_out, hipStream_t st);
int launch_q_backup(int S, const Tree& t, const double* leaf_val, const double* child_vals_padded, int raw, hipStream_t st);
int launch_q_child_values(int S, const Tree& t, double* inv_vals, hipStream_t st);
int launch_q_policy(int S, const Tree& t, double temp, int from_values, double* pi, hipStream_t st);
int launch_uniform_policy(int S, const Tree& t, int allow_pass, double* pi, hipStream_t st);
int launch_sample(int S, const Tree& t, const double* pi, uint64_t seed, uint64_t step, int game_offset,
                  int32_t* actions, hipStream_t st);
int launch_sample_keyed(int S, const double* pi, uint64_t seed, const uint64_t* game_ids, const int32_t* steps, int n,
                        int32_t* actions, hipStream_t st);
int launch_mcts_advance(int S, const Tree& t, const int32_t* actions, int max_moves, hipStream_t st);
int launch_mcts_roots(int S, const Tree& t, uint8_t* states, hipStream_t st);
int launch_mcts_export(int S, const Tree& t, int game, int32_t* n_nodes, int32_t* parent, int32_t* action, int32_t* visits,
                       double* value, double* raw_sum, double* prior, uint8_t* states, hipStream_t st);

}  // namespace goai
