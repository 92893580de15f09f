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

/* ---------------------------------------------------------------------------------------
 * Policy / value CNN forward
 * ------------------------------------------------------------------------------------- */
#define GOAI_NET_AC 0       /* ActorCriticNet, /root/reference/go_ai/models/ac_net.py:8-64 */
#define GOAI_NET_VAL 1      /* ValueNet,       /root/reference/go_ai/models/val_net.py:8-27 */
#define GOAI_DTYPE_F32 0    /* fp32 storage, fp32 matrix-core accumulate (the reference's precision) */
#define GOAI_NN_RAW 0       /* the nn.Module outputs (ac_net.forward / val_net.forward) */
#define GOAI_NN_NUMPY_SHIM 1 /* + RLNet._numpy post-processing, models/__init__.py:90-105 */

/* Allocate the network of this context: the reference trunk (conv3x3 6->128 + n_blocks
 * BasicBlocks of 128 channels, /root/reference/go_ai/models/__init__.py:14-31; the
 * reference hard-codes n_blocks = 3) with the heads of `net_kind`.  Forward calls are
 * processed in slices of at most max_batch states. */
int goai_nn_init(goai_ctx* ctx, int net_kind, int n_blocks, int max_batch, int dtype);

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

#ifdef __cplusplus
}
#endif
#endif /* GOAI_HIP_H */
