/*
 * goai_hip.h -- C ABI of libgoai_hip.so: the MI355X (gfx950) batched MCTS leaf-expansion
 * path of huangeddie/Go-AI.
 *
 * The reference has no FFI layer: the path sits behind plain Python callables
 * (SURVEY.md section 8b).  Each entry point below names the reference callable (file:line
 * under /root/reference) whose arithmetic it replaces; the go-ai_amd/go_ai package binds them through
 * ctypes under the reference's own function names (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns GOAI_OK (0) or a negative GOAI_ERR_* code; the message is
 *     available from goai_last_error() (thread-local);
 *   - all array arguments are DEVICE pointers (e.g. torch tensor .data_ptr()) unless the
 *     name ends in _host; the caller owns them; nothing is retained after the call
 *     returns except by goai_load_weights (which copies);
 *   - work is enqueued on the context's HIP stream (default: the null stream; see
 *     goai_ctx_set_stream) and is asynchronous with respect to the host unless stated;
 *   - a context is bound to one GPU and is not thread-safe: one context per GPU per host
 *     thread, as the reference runs one environment and one policy per process
 *     (/root/reference/go_ai/utils.py:261-270);
 *   - states use the reference's GymGo layout: uint8 [n][6][S][S], planes
 *     0 black, 1 white, 2 turn, 3 invalid moves, 4 previous-move-was-pass, 5 game over
 *     (/root/reference/go_ai/data.py:15-24); actions are int32 in [0, S*S], S*S = pass.
 */
#ifndef GOAI_HIP_H
#define GOAI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOAI_OK 0
#define GOAI_ERR_ARG (-1)   /* bad argument / unsupported configuration */
#define GOAI_ERR_HIP (-2)   /* a HIP runtime call failed */
#define GOAI_ERR_STATE (-3) /* call sequence error (e.g. weights not loaded) */

typedef struct goai_ctx goai_ctx;

/* Last error message of the calling thread ("" if none). */
const char* goai_last_error(void);
/* ABI version, bumped on any signature change. */
int goai_version(void);

/* Create a context on GPU `device` for SxS boards (S in {5,7,9,13,19}). */
int goai_ctx_create(goai_ctx** out, int device, int board_size);
int goai_ctx_destroy(goai_ctx* ctx);
/* Use `hip_stream` (a hipStream_t) for all later work of this context. */
int goai_ctx_set_stream(goai_ctx* ctx, void* hip_stream);
/* Block the host until the context's stream is idle. */
int goai_ctx_sync(goai_ctx* ctx);

/* ---------------------------------------------------------------------------------------
 * Go rules (replace the un-vendored GymGo package)
 * ------------------------------------------------------------------------------------- */

/* GoGame.next_state(state, action, canonical) for n states at once
 * (/root/reference/go_ai/search/tree.py:115; gym_go.gogame.next_state).
 * states_out[i] is all-zero and status[i] = -1 where actions[i] is illegal (GymGo asserts).
 * status may be NULL. */
int goai_transition(goai_ctx* ctx, const uint8_t* states, const int32_t* actions, int canonical,
                    uint8_t* states_out, int32_t* status, int n);

/* GoGame.children(state, canonical, padded=True) for n states: children_out is
 * [n][A][6][S][S] with all-zero children at illegal actions
 * (/root/reference/go_ai/search/tree.py:92, /root/reference/go_ai/data.py:54-60);
 * valid_out [n][A] uint8 = GoGame.valid_moves (tree.py:108);
 * win_out [n][A] int8 = data.batch_win_children (data.py:41-51): winning(child) if the
 * child ended the game else 0.  valid_out / win_out may be NULL. */
int goai_children(goai_ctx* ctx, const uint8_t* states, int canonical, uint8_t* children_out,
                  uint8_t* valid_out, int8_t* win_out, int n);

/* GoGame.areas / game_ended / winning for n states
 * (/root/reference/go_ai/search/tree.py:65-69, /root/reference/go_ai/policies/baselines.py:18).
 * Areas are those of planes 0 / 1 as given; any output may be NULL. */
int goai_score(goai_ctx* ctx, const uint8_t* states, int32_t* black_area, int32_t* white_area,
               int32_t* ended, int32_t* winner, int n);

/* GoGame.get_num_liberties(state) for n states (/root/reference/go_ai/policies/baselines.py:41):
 * per colour, the number of distinct empty points adjacent to a stone of that colour (planes
 * 0 / 1 as given).  Either output may be NULL. */
int goai_liberties(goai_ctx* ctx, const uint8_t* states, int32_t* black_libs, int32_t* white_libs, int n);

/* ---------------------------------------------------------------------------------------
 * Policy / value CNN forward
 * ------------------------------------------------------------------------------------- */
#define GOAI_NET_AC 0       /* ActorCriticNet, /root/reference/go_ai/models/ac_net.py:8-64 */
#define GOAI_NET_VAL 1      /* ValueNet,       /root/reference/go_ai/models/val_net.py:8-27 */
#define GOAI_DTYPE_F32 0    /* fp32 storage, exact fp32 matrix-core arithmetic (the reference's precision) */
#define GOAI_DTYPE_BF16X3 1 /* trunk 3x3 convolutions on the bf16 matrix cores with split operands: every fp32
                             * activation and weight is split into two bf16 numbers (16 significant bits) and a
                             * product is hi*hi + lo*hi + hi*lo with fp32 accumulate; activations, residual adds,
                             * conv0 and the heads stay fp32.  Outputs agree with GOAI_DTYPE_F32 to ~1e-5, i.e.
                             * inside BASELINE.json's fp32 tolerance of 1e-3 (tests/test_gpu_nn.py) */
#define GOAI_DTYPE_BF16X6 2 /* fp32-grade results from the bf16 matrix cores: trunk convolutions as Winograd F(3x3,3x3) whose
                             * transform-domain products run as SIX bf16 MFMAs on operands split exactly into three bf16
                             * parts each (24 significant bits; the dropped cross terms are below 2^-24 of a product),
                             * fp32 accumulate, fp32 transforms.  Outputs agree with the exact-fp32 kernels at their own
                             * rounding level (tests/test_gpu_nn.py).  F(3x3,3x3) on 9x9 boards, F(2x2,3x3) on the others. */
#define GOAI_DTYPE_F16X3 3  /* the same Winograd kernels with every operand as TWO fp16 parts (h = fp16(x), l = fp16(x - h): 22
                             * significant bits and a sign) and THREE fp16 MFMAs per product, operands scaled by exact powers
                             * of two into fp16's range (weights per convolution, activations per board and convolution from
                             * a running upper bound).  Half the matrix work of GOAI_DTYPE_BF16X6 at the same measured
                             * accuracy: in both, the fp32 accumulation dominates the error (tests/test_gpu_nn.py,
                             * tools/ubench/f16x3_accuracy.hip). */
#define GOAI_DTYPE_F16 4    /* REDUCED PRECISION, opt-in (BASELINE.json north_star's second tier: 1e-2 relative error; never what
                             * bench.py's headline runs): the GOAI_DTYPE_F16X3 kernels with ONE fp16 part per operand and one
                             * MFMA per product -- operands rounded to 11 significant bits (scaled per board / convolution as
                             * above), fp32 accumulate, fp32 transforms, activations stored in fp32.  Measured 1e-3 .. 4e-3 on
                             * the 3-block and 10-block nets (tests/test_gpu_nn.py::test_f16_forward_within_reduced_tolerance). */
#define GOAI_NN_RAW 0       /* the nn.Module outputs (ac_net.forward / val_net.forward) */
#define GOAI_NN_NUMPY_SHIM 1 /* + RLNet._numpy post-processing, models/__init__.py:90-105 */

/* Allocate the network of this context: the reference trunk (conv3x3 6->128 + n_blocks
 * BasicBlocks of 128 channels, /root/reference/go_ai/models/__init__.py:14-31; the
 * reference hard-codes n_blocks = 3) with the heads of `net_kind`.  Forward calls are
 * processed in slices of at most max_batch states. */
int goai_nn_init(goai_ctx* ctx, int net_kind, int n_blocks, int max_batch, int dtype);

/* Which kernel goai_nn_init chose for the trunk's 3x3 convolutions of this context (no reference
 * counterpart: the reference leaves the choice to torch / MKL-DNN).  GOAI_DTYPE_F32 contexts run
 * Winograd in fp32 (F(3x3,3x3) on 9x9 -- its _PAIRED form when max_batch is at most 4 boards per CU --, F(2x2,3x3) on larger boards) when max_batch fills the chip
 * with its workgroups, else the direct implicit GEMM; the choice is fixed per context so that a
 * position's output never depends on the size of the batch it is evaluated in.  Returns one of
 * the GOAI_CONV_* values, or a negative error code. */
#define GOAI_CONV_DIRECT 0
#define GOAI_CONV_WINOGRAD 1
#define GOAI_CONV_BF16X3 2
#define GOAI_CONV_WINOGRAD_F3 3 /* Winograd F(3x3,3x3), fp32: 9x9 boards */
#define GOAI_CONV_WINOGRAD_F3_PAIRED 4 /* the same with two 32-channel chunks per output transform: 9x9, max_batch <= 4 boards per CU */
#define GOAI_CONV_WINOGRAD_F3_BF16X6 5 /* GOAI_DTYPE_BF16X6 contexts, 9x9: F(3x3,3x3) */
#define GOAI_CONV_WINOGRAD_BF16X6 6    /* GOAI_DTYPE_BF16X6 contexts, other boards: F(2x2,3x3) */
#define GOAI_CONV_WINOGRAD_F3_F16X3 7  /* GOAI_DTYPE_F16X3 contexts, 9x9: F(3x3,3x3) */
#define GOAI_CONV_WINOGRAD_F16X3 8     /* GOAI_DTYPE_F16X3 contexts, other boards: F(2x2,3x3) */
#define GOAI_CONV_WINOGRAD_F3_F16 9    /* GOAI_DTYPE_F16 contexts, 9x9: F(3x3,3x3), one fp16 product */
#define GOAI_CONV_WINOGRAD_F16 10      /* GOAI_DTYPE_F16 contexts, other boards: F(2x2,3x3), one fp16 product */
int goai_nn_conv_algo(goai_ctx* ctx);

/* How many of the trunk's 3x3 convolutions one kernel launch of this context runs (no reference
 * counterpart; a measurement aid like goai_nn_conv_algo): 2 * n_blocks when a kernel whose
 * workgroups own whole boards runs the whole residual trunk in one launch -- the F(3x3,3x3)
 * kernels and, since round 3, every context of a split dtype (GOAI_DTYPE_F16X3 / _BF16X6 / _F16)
 * on any board size -- else 1 (also with GOAI_NO_TRUNK_FUSION=1).  goai_profile_read counts launches. */
int goai_nn_convs_per_launch(goai_ctx* ctx);

/* Load a state_dict: `names[i]` are the reference's own keys ("main.0.weight",
 * "main.3.main.1.running_var", "act_head.4.bias", "convs.6.weight", ...; what
 * net.state_dict() / torch.load give, /root/reference/go_ai/policies/baselines.py:160-172),
 * `tensors_host[i]` HOST pointers to contiguous float32 data of `numels[i]` elements.
 * Unknown keys (game_head.*) are ignored, missing ones are an error.  BatchNorm running
 * statistics are folded into the convolutions (the reference always evaluates in eval()
 * mode, models/__init__.py:64).  Synchronous; the data is copied. */
int goai_load_weights(goai_ctx* ctx, const char* const* names, const float* const* tensors_host,
                      const int64_t* numels, int n_tensors);

/* Network forward for n states: pi_logits [n][A] float32 (NULL for the value net or when
 * only values are wanted), val_logits [n][1] float32.
 * mode GOAI_NN_RAW        = pt_actor_critic / pt_critic (ac_net.py:40-64, val_net.py:22-27);
 * mode GOAI_NN_NUMPY_SHIM = RLNet._numpy(states, 'actor_critic' | 'critic')
 *   (/root/reference/go_ai/models/__init__.py:54-113): invalid moves get
 *   np.finfo(float32).min, children that end the game get -100 * winning(child), finished
 *   states get val = 100 * winning(state). */
int goai_nn_forward(goai_ctx* ctx, const uint8_t* states, float* pi_logits, float* val_logits, int n,
                    int mode);

/* goai_nn_forward for callers that cannot allocate -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  It has NO counterpart in the reference.  Same arguments,
 * kernels and results (bit for bit) as goai_nn_forward on the same context, but the activation
 * buffers between the kernels are a file-scope device array of the library (12 MiB per GPU, held
 * until the process ends) instead of the context's hipMalloc'ed ones, and the call allocates nothing
 * (no hipMalloc / hipFree / event creation; goai_profile_* does not see it).  Limit:
 * n * S * S <= GOAI_NN_STATIC_MAX_CELLS board cells (101 boards of 9x9, 48 of 13x13, 22 of 19x19);
 * larger n is rejected with GOAI_ERR_ARG.  All contexts of one GPU share the array and it keeps the
 * last call's contents (nothing in it has to start at zero): calls that overlap on one GPU --
 * different streams or host threads -- race; calls on one stream are ordered.  (Tested: bit-equality with
 * goai_nn_forward and that the device's free memory does not change across calls; capture into a HIP graph is
 * what the entry point is for but is not itself under test.) */
#define GOAI_NN_STATIC_MAX_CELLS 8192
int goai_nn_forward_static(goai_ctx* ctx, const uint8_t* states, float* pi_logits, float* val_logits, int n,
                           int mode);

/* ---------------------------------------------------------------------------------------
 * Batched Monte-Carlo tree search: n_games independent games per context, one tree each,
 * advanced in lock step (one leaf per game per step).
 * ------------------------------------------------------------------------------------- */
#define GOAI_SEARCH_AC 0   /* mct_search(go_env, n, actor_critic=f), /root/reference/go_ai/search/mct.py:50 */
#define GOAI_SEARCH_Q 1    /* mct_search(go_env, n, critic=f): Q-value search, mct.py:27-30,43-47 */

/* Allocate tree storage for n_games games and at most max_sims searches per move
 * (0 <= max_sims <= GOAI_MAX_SIMS).  numpy's pairwise mean of a node's values (np.mean at
 * tree.py:180) is maintained incrementally up to 128 entries; contexts with max_sims > 128 also keep
 * every visited node's value list ((max_sims + 1)(max_sims + 2) / 2 doubles per game: 1.05 MB at
 * 512) and sum it with numpy's own recursion -- bit-identical for any length.
 * mode = GOAI_SEARCH_AC or GOAI_SEARCH_Q. */
#define GOAI_MAX_SIMS 2048
int goai_mcts_init(goai_ctx* ctx, int n_games, int max_sims, int mode);

/* tree.Node(go_env.canonical_state()) for every game (mct.py:52-53): root_states is
 * [n_games][6][S][S]; the side to move is taken from the turn plane. */
int goai_mcts_reset(goai_ctx* ctx, const uint8_t* root_states);

/* First half of mct_step (mct.py:20-22): find_next_node for every game -- PUCT descent
 * (tree.py:170-184, np.nanargmax: first maximum wins) creating at most one node per game
 * through GoGame.next_state (tree.py:110-117).  leaf_states_out ([n_games][6][S][S], may be
 * NULL) receives node.state of the selected leaves, i.e. what the reference passes to the
 * network at mct.py:26. */
int goai_mcts_select(goai_ctx* ctx, uint8_t* leaf_states_out);

/* Second half of mct_step (mct.py:32-42): node.backprop(val) (tree.py:141-146) and, for
 * non-terminal leaves, node.set_prior_pi(softmax(pi_logits)).  Exactly one of
 *   pi_logits [n_games][A] float32 -- logits; the prior is their fp32 softmax, or
 *   prior     [n_games][A] float32 -- the prior itself (e.g. scipy's softmax output recorded
 *                                     from the reference; used for bit-exact replay tests)
 * must be non-NULL.  val_logits is [n_games] float32.  logits_are_raw != 0 means pi_logits /
 * val_logits come straight from the network and RLNet._numpy's post-processing
 * (models/__init__.py:90-105) is applied here; 0 means they already went through it. */
int goai_mcts_backup(goai_ctx* ctx, const float* pi_logits, const float* prior, const float* val_logits,
                     int logits_are_raw);

/* GOAI_SEARCH_Q only, between goai_mcts_select and goai_mcts_backup_q: node.make_children() for
 * every non-terminal selected leaf (mct.py:44-45; tree.py:88-98): one child node per valid action
 * through GoGame.children(state, canonical=True, padded=True).  Optional outputs for callers that
 * evaluate with their own value function: children_out [n_games][A][6][S][S] (all-zero at
 * invalid actions and for terminal leaves), valid_out [n_games][A] uint8. */
int goai_mcts_expand(goai_ctx* ctx, uint8_t* children_out, uint8_t* valid_out);

/* GOAI_SEARCH_Q, second half of mct_step (mct.py:30-37,46-47): node.backprop(val) and, for
 * non-terminal leaves, tree.set_state_vals(critic, children) + node.set_prior_pi(None), i.e.
 * prior[valid] = softmax(-child values) in float64 (tree.py:14-19,152-157).
 * leaf_val [n_games] float64 = critic(leaf.state).item() (the reference keeps Python floats, so a
 * float64-valued critic such as baselines.greedy_val_func loses nothing); child_vals [n_games][A] float64, entry a =
 * critic value of the child reached by action a (ignored at invalid actions).  Both NULL: use
 * the values the context's own network produced for the batch built by goai_mcts_expand
 * (only inside goai_mcts_run / goai_mcts_steps).  vals_are_raw != 0: RLNet._numpy's override
 * for finished states (models/__init__.py:101-105) is applied here. */
int goai_mcts_backup_q(goai_ctx* ctx, const double* leaf_val, const double* child_vals, int vals_are_raw);

/* rootnode.inverted_children_values() (tree.py:128-135) -> [n_games][A] float64: -child.val,
 * 0 where the root has no child.  GOAI_SEARCH_Q only. */
int goai_mcts_child_values(goai_ctx* ctx, double* inverted_child_vals);

/* Value.__call__ (/root/reference/go_ai/policies/value.py:30-37):
 * pi [n_games][A] float64 = search.temp_norm(qs, temp, valid) (search/__init__.py:78-87) with
 * qs = root visit counts (from_values == 0, mcts > 0) or exp(-child values) (from_values != 0,
 * mcts == 0).  temp <= 0: search.greedy_pi (search/__init__.py:46-52), uniform over the best moves.
 * Bit-exact for temp == 1 and temp <= 0; other temperatures go through pow() (within 4 ulp of numpy's).
 * GOAI_SEARCH_Q only. */
int goai_mcts_policy_value(goai_ctx* ctx, double temp, int from_values, double* pi);

/* mct_search with the context's own network: 1 + n_sims times select -> forward -> backup
 * (mct.py:55-60; the first iteration "doesn't count").  In GOAI_SEARCH_Q mode every step
 * evaluates the leaf and all its valid children in one compacted batch. */
int goai_mcts_run(goai_ctx* ctx, int n_sims);

/* n_steps x mct_step (select -> forward -> backup) with the context's own network; lets a
 * caller interleave the steps of several contexts (game groups) on different streams so that
 * one group's tree kernels and kernel tails overlap the other group's convolutions. */
int goai_mcts_steps(goai_ctx* ctx, int n_steps);

/* How the last goai_mcts_steps of the context ran: k > 0 = as ONE launch of k trunk passes with the search's work between them
 * (k_search_steps: opt-in by GOAI_STEPS_KERNEL=1 -- measured slower, DESIGN.md section 5 --; 9x9, GOAI_DTYPE_F16X3, a multiple of
 * four games), 0 = as separate launches.  Same results either way, bit for bit.  No counterpart in the reference: it names the kernel that
 * bench.py's roofline is about (goai_profile_read then counts every trunk pass of such a launch). */
int goai_mcts_steps_in_one_launch(goai_ctx* ctx);

/* rootnode.get_visit_counts() (tree.py:161-168) -> visit_counts [n_games][A] int32,
 * rootnode.prior_pi -> prior [n_games][A] float64, rootnode.visits -> root_visits [n_games];
 * any output may be NULL. */
int goai_mcts_readout(goai_ctx* ctx, int32_t* visit_counts, double* prior, int32_t* root_visits);

/* ActorCritic.__call__ for mcts > 0 (/root/reference/go_ai/policies/actorcritic.py:29-42):
 * pi [n_games][A] float64 = 0.98 * normalised(counts ** (1/temp)) + 0.02 * uniform(valid).
 * Bit-exact for temp == 1.  Finished games get the uniform-valid distribution. */
int goai_mcts_policy(goai_ctx* ctx, double temp, double* pi);

/* Random policy over the root's valid moves (baselines.Random,
 * /root/reference/go_ai/policies/baselines.py:108-120); allow_pass = 0 drops the pass unless
 * nothing else is legal.  Used to build synthetic start positions. */
int goai_uniform_policy(goai_ctx* ctx, int allow_pass, double* pi);

/* GoGame.random_weighted_action(pi) per game (/root/reference/go_ai/game.py:91) from a
 * Philox4x32-10 stream keyed by (seed, game_id_offset + game, step): draws do not depend on
 * how games are sharded over GPUs.  actions is [n_games] int32. */
int goai_sample_actions(goai_ctx* ctx, const double* pi, uint64_t seed, uint64_t step, int game_id_offset,
                        int32_t* actions);

/* go_env.step(action) for every running game followed by a fresh tree (game.py:95,
 * mct.py:52-53).  A game ends after two passes or more than 2*S*S moves (game.py:74,98). */
int goai_mcts_advance(goai_ctx* ctx, const int32_t* actions);

/* Work on the first n_active <= n_games games only until further notice (default: all).  For callers that own
 * the games and give a context a different subset every move -- two policies playing each other, where each
 * policy's context searches the games in which it is to move (/root/reference/go_ai/game.py:80-89).  Every
 * array argument of the search calls then has n_active rows; goai_mcts_reset must follow. */
int goai_mcts_set_active(goai_ctx* ctx, int n_active);

/* goai_sample_actions with the Philox key given per row: row i draws from pi[i] with key
 * (seed, game_ids[i], steps[i]) -- the game's own id and move number, so a game's draws do not depend on
 * which slot, context or GPU it is played in.  Needs no search state. */
int goai_sample_actions_keyed(goai_ctx* ctx, const double* pi, uint64_t seed, const uint64_t* game_ids,
                              const int32_t* steps, int n, int32_t* actions);

/* The whole search tree of one game, for host-side views of tree.Node (/root/reference/go_ai/search/tree.py:22-52;
 * what policy(go_env, debug=True) hands to search/plot.py:39-61).  Node n < *n_nodes (all device arrays sized
 * goai_mcts_node_capacity()): parent[n] (-1 for the root, node 0), action[n] = the move that led to it,
 * visits[n], value[n] = node.val (GOAI_SEARCH_Q; NaN where none), raw_sum[n] = sum(node.post_vals),
 * prior [n][A] = node.prior_pi (a NaN row where the node has none), states [n][6][S][S] = node.state.
 * Children follow their parent; an actor-critic tree lists nodes in creation order. */
int goai_mcts_node_capacity(goai_ctx* ctx);
int goai_mcts_export_tree(goai_ctx* ctx, int game, int32_t* n_nodes, int32_t* parent, int32_t* action, int32_t* visits,
                          double* value, double* raw_sum, double* prior, uint8_t* states);

/* Slot refill for long self-play runs (play_games' loop over episodes, /root/reference/go_ai/game.py:139-145,
 * with every slot of the batch as one sequential player): when on, goai_mcts_advance replaces a game that just
 * ended by a fresh one -- go_env.reset(): the empty board, black to move -- in the same slot, so the batch
 * never carries finished games.  The slot's Philox stream moves on to game id + 2^32 * (games finished in the slot). */
int goai_mcts_set_refill(goai_ctx* ctx, int on);

/* Totals over all games finished since the last goai_mcts_reset, refilled or not (the sums play_games
 * returns and mpi_play all-reduces, game.py:147-149, utils.py:201-204): stats_host[0] games finished,
 * [1] black wins, [2] white wins, [3] moves played in them.  Synchronises the stream. */
int goai_mcts_game_stats(goai_ctx* ctx, int64_t* stats_host);

/* Current roots: canonical states [n_games][6][S][S], side to move (0 black / 1 white), moves
 * played, result (2 = running, else black's result in {-1,0,1}, game.py:113); any may be NULL. */
int goai_mcts_roots(goai_ctx* ctx, uint8_t* states, int32_t* turn, int32_t* moves, int32_t* result);

/* Host-side read of 4 counters (synchronises the stream): [0] leaf evaluations of running
 * games, [1] Go transitions made by the search, [2] internal consistency errors (must stay
 * 0), [3] GOAI_SEARCH_Q: states sent through the value network (leaves + children). */
int goai_mcts_counters(goai_ctx* ctx, int64_t* counters_host);

/* ---------------------------------------------------------------------------------------
 * Optimisation step: one train_step of the reference's networks
 *   ActorCriticNet.train_step  /root/reference/go_ai/models/ac_net.py:67-86
 *   ValueNet.train_step        /root/reference/go_ai/models/val_net.py:29-43
 *   RLNet.critic_step / actor_step  /root/reference/go_ai/models/__init__.py:115-134,167-186
 *   torch.optim.Adam(lr, weight_decay)  /root/reference/train.py:52
 * in fp32 on the GPU: train-mode forward (BatchNorm on the batch's own statistics, running
 * statistics moved with momentum 0.1), backward, Adam.  A context carries the training copy of
 * a net (raw parameters, gradients, Adam moments, BatchNorm statistics) next to, and
 * independent of, the folded inference images of goai_nn_init.
 * ------------------------------------------------------------------------------------- */
/* Allocate the training state for batches of up to max_batch positions; parameters start at
 * zero (goai_train_load fills them). */
int goai_train_init(goai_ctx* ctx, int net_kind, int n_blocks, int max_batch);

/* Parameters and BatchNorm running statistics from HOST fp32 tensors by state_dict key, the
 * table format of goai_load_weights (game_head.* and *.num_batches_tracked are skipped).
 * strict != 0: every tensor of the net must be in the table.  Adam's state is not touched. */
int goai_train_load(goai_ctx* ctx, const char* const* names, const float* const* tensors_host,
                    const int64_t* numels, int n_tensors, int strict);

/* Copy named tensors out to HOST buffers: what = 0 the values (parameters, statistics),
 * 1 the gradients the last step's optimiser consumed (before weight decay), 2 / 3 Adam's
 * first / second moments.  Synchronises the stream. */
int goai_train_export(goai_ctx* ctx, int what, const char* const* names, float* const* tensors_host,
                      const int64_t* numels, int n_tensors);

/* A fresh optimiser (train.py:52 builds one per run): moments and step count to zero. */
int goai_train_reset_optimizer(goai_ctx* ctx);

/* One train_step on device arrays: states / next_states uint8 [n][6][S][S], wins float [n] (the
 * replay's `wins`; the critic target is -wins as in ac_net.py:76), pi float [n][A] (the actor's
 * target is argmax(pi)).  The caller has already built next_states = children[actions] under
 * its random symmetries (ac_net.py:69-71, models/__init__.py:119).  The value net takes
 * states = pi = NULL.  2 <= n <= max_batch.  metrics_host[4] (HOST): critic loss, critic
 * accuracy, actor loss, actor accuracy (NaN for the value net's last two).  Every reduction
 * runs in a fixed order: the same inputs give the same bits.  Synchronises the stream.
 * Since round 3 the step replays one HIP graph per batch size n (captured at the first step of
 * that size; the arrays are copied to buffers of the context first, so they may be different
 * allocations at every call); GOAI_NO_TRAIN_GRAPH=1 at goai_train_init: plain launches, same bits. */
int goai_train_step(goai_ctx* ctx, const uint8_t* states, const uint8_t* next_states, const float* wins,
                    const float* pi, int n, float lr, float weight_decay, float* metrics_host);

/* The flat fp32 DEVICE buffer of all parameters (not the BatchNorm statistics), for
 * average_model (/root/reference/go_ai/models/__init__.py:279-282): all-reduce it in place
 * over the ranks (RCCL) and scale by 1 / world_size.  Same layout on every rank. */
int goai_train_params(goai_ctx* ctx, float** params_device, int64_t* numel);

/* Fold the trained parameters and statistics into this context's inference images (needs
 * goai_nn_init with the same net kind and n_blocks): what goai_load_weights does, from the
 * training buffers. */
int goai_train_sync_inference(goai_ctx* ctx);

/* ---------------------------------------------------------------------------------------
 * The one collective of the path: the end-of-run sums over ranks that the reference makes with
 * comm.allreduce(..., op=MPI.SUM) (/root/reference/go_ai/utils.py:201-204: seconds per game, win
 * rates, steps).  Games are sharded over GPUs and the search never communicates; this is an RCCL
 * all-reduce of a few numbers on the context's stream.  RCCL is loaded at run time (the copy
 * already in the process, e.g. PyTorch's, else the system's); the library has no link-time
 * dependency on it.
 * ------------------------------------------------------------------------------------- */
#define GOAI_COMM_ID_BYTES 128
#define GOAI_GATHER_MAX_EXTRA 64
/* ncclGetUniqueId: rank 0 calls this and hands the 128 bytes to the other ranks by any means. */
int goai_comm_unique_id(void* id_host);
/* ncclCommInitRank on the context's GPU: a communicator of n_ranks ranks (one process per GPU). */
int goai_comm_init(goai_ctx* ctx, const void* id_host, int n_ranks, int rank, void** rccl_comm_out);
int goai_comm_destroy(goai_ctx* ctx, void* rccl_comm);
/* totals_host[8] (HOST): the sums over all ranks of goai_mcts_game_stats ([0] games finished,
 * [1] black wins, [2] white wins, [3] moves played) and goai_mcts_counters ([4] leaf evaluations,
 * [5] transitions, [6] errors, [7] states through the value net) of this context's search.
 * extra_inout_host[n_extra] (HOST doubles, n_extra <= GOAI_GATHER_MAX_EXTRA, may be NULL/0): the
 * caller's own additive figures (seconds, first-player wins ...), summed over ranks in place.
 * rccl_comm: an ncclComm_t made by goai_comm_init (or by the same RCCL library); NULL = one rank,
 * the local figures.  Every rank of the communicator must call it.  Synchronises the stream. */
int goai_gather_stats(goai_ctx* ctx, void* rccl_comm, double* extra_inout_host, int n_extra, int64_t* totals_host);

/* ---------------------------------------------------------------------------------------
 * Measurement aid (no counterpart in the reference, which only logs seconds per game,
 * /root/reference/go_ai/utils.py:196-207): when enabled, every launch of the dominant kernel
 * (the 3x3 convolution) is bracketed by HIP events on the context's stream.
 * goai_profile_read synchronises, returns the summed duration in ms and the launch count, and
 * clears the record.
 * ------------------------------------------------------------------------------------- */
int goai_profile_enable(goai_ctx* ctx, int enable);
int goai_profile_read(goai_ctx* ctx, double* conv_ms_total, int64_t* conv_launches);

#ifdef __cplusplus
}
#endif
#endif /* GOAI_HIP_H */
