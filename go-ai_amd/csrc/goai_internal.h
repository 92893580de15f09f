// goai_internal.h -- shared host-side declarations of libgoai_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/goai_hip.h"

namespace goai {

void set_error(const std::string& msg);

#define GOAI_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::goai::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));             \
            return GOAI_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define GOAI_REQUIRE(cond, msg)                                                               \
    do {                                                                                      \
        if (!(cond)) { ::goai::set_error(msg); return GOAI_ERR_ARG; }                         \
    } while (0)

inline bool size_supported(int s) { return s == 5 || s == 7 || s == 9 || s == 13 || s == 19; }

// Dispatch a templated launcher on the board size.
#define GOAI_DISPATCH_SIZE(S_RT, ...)                                                         \
    switch (S_RT) {                                                                           \
        case 5: { constexpr int S = 5; __VA_ARGS__; } break;                                  \
        case 7: { constexpr int S = 7; __VA_ARGS__; } break;                                  \
        case 9: { constexpr int S = 9; __VA_ARGS__; } break;                                  \
        case 13: { constexpr int S = 13; __VA_ARGS__; } break;                                \
        case 19: { constexpr int S = 19; __VA_ARGS__; } break;                                \
        default: ::goai::set_error("unsupported board size"); return GOAI_ERR_ARG;            \
    }

// ---- Go rule kernels (go_kernels.hip) ----
int launch_transition(int S, const uint8_t* states, const int32_t* actions, int canonical, uint8_t* out,
                      int32_t* status, int n, hipStream_t st);
int launch_children(int S, const uint8_t* states, int canonical, uint8_t* children, uint8_t* valid,
                    int8_t* win, int n, hipStream_t st);
int launch_score(int S, const uint8_t* states, int32_t* black_area, int32_t* white_area, int32_t* ended,
                 int32_t* winner, int n, hipStream_t st);

int launch_liberties(int S, const uint8_t* states, int32_t* black_libs, int32_t* white_libs, int n, hipStream_t st);

// ---- CNN (nn_host.hip) ----
int nn_init(goai_ctx* c, int kind, int layers, int max_batch, int dtype);
int nn_load_weights(goai_ctx* c, const char* const* names, const float* const* tensors, const int64_t* numels, int n);
int nn_forward_raw(goai_ctx* c, const uint8_t* states, float* pi, float* val, int n);
int nn_piece_conv0(goai_ctx* c, const uint8_t* states, int n);
int nn_piece_trunk(goai_ctx* c, int n, float** out, bool* out_is_h);
int nn_piece_heads(goai_ctx* c, const float* x, bool x_is_h, float* pi, float* val, int n);
int nn_static_setup(int device);   // once per device, with that device current (goai_ctx_create)
int nn_forward_static(goai_ctx* c, const uint8_t* states, float* pi, float* val, int n);

// ---- optimisation step (train_host.hip) ----
int train_init(goai_ctx* c, int kind, int layers, int max_batch);
int train_load(goai_ctx* c, const char* const* names, const float* const* tensors, const int64_t* numels, int n, int strict);
int train_reset_optimizer(goai_ctx* c);
int train_export(goai_ctx* c, int what, const char* const* names, float* const* tensors, const int64_t* numels, int n);
int train_step(goai_ctx* c, const uint8_t* states, const uint8_t* next_states, const float* wins, const float* pi, int n, float lr,
               float weight_decay, float* metrics_host);
int train_sync_inference(goai_ctx* c);

// ---- result gather over ranks (gather.hip) ----
int comm_unique_id(void* id128_host);
int comm_init(goai_ctx* c, const void* id128_host, int n_ranks, int rank, void** comm_out);
int comm_destroy(goai_ctx* c, void* comm);
int gather_stats(goai_ctx* c, void* comm, double* extra_inout_host, int n_extra, int64_t* totals_host);

// ---- search (mcts_host.hip) ----
int mcts_init(goai_ctx* c, int n_games, int max_sims, int mode);
int mcts_reset(goai_ctx* c, const uint8_t* roots);
int mcts_select(goai_ctx* c, uint8_t* leaf_states_out);
int mcts_backup(goai_ctx* c, const float* pi_logits, const float* prior, const float* val, int raw);
int mcts_expand(goai_ctx* c, uint8_t* children_out, uint8_t* valid_out);
int mcts_backup_q(goai_ctx* c, const double* leaf_val, const double* child_vals, int raw);
int mcts_steps(goai_ctx* c, int n_steps);
int mcts_advance(goai_ctx* c, const int32_t* actions);

}  // namespace goai
