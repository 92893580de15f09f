// capi.hip -- extern "C" surface of libgoai_hip.so (declared in include/goai_hip.h).
#include <new>
#include <string>

#include "ctx.h"
#include <vector>

#include "goai_internal.h"

namespace goai {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace goai

using namespace goai;

#define CTX_ENTER(ctx)                                                   \
    GOAI_REQUIRE((ctx) != nullptr, "ctx is NULL");                       \
    GOAI_HIP_CHECK(hipSetDevice((ctx)->device))

extern "C" {

const char* goai_last_error(void) { return g_last_error.c_str(); }
int goai_version(void) { return 4; }

int goai_ctx_create(goai_ctx** out, int device, int board_size)
{
    GOAI_REQUIRE(out != nullptr, "out is NULL");
    GOAI_REQUIRE(size_supported(board_size), "board_size must be one of 5, 7, 9, 13, 19");
    int ndev = 0;
    GOAI_HIP_CHECK(hipGetDeviceCount(&ndev));
    GOAI_REQUIRE(device >= 0 && device < ndev, "no such GPU");
    GOAI_HIP_CHECK(hipSetDevice(device));
    {
        const int rc = nn_static_setup(device);   // device address of goai_nn_forward_static's buffers, once per GPU
        if (rc) return rc;
    }
    goai_ctx* c = new (std::nothrow) goai_ctx();
    GOAI_REQUIRE(c != nullptr, "out of host memory");
    c->device = device;
    c->S = board_size;
    c->P = board_size * board_size;
    c->A = c->P + 1;
    c->stream = nullptr;
    *out = c;
    return GOAI_OK;
}

int goai_ctx_destroy(goai_ctx* ctx)
{
    if (!ctx) return GOAI_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->release();
    delete ctx;
    return GOAI_OK;
}

int goai_ctx_set_stream(goai_ctx* ctx, void* hip_stream)
{
    GOAI_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return GOAI_OK;
}

int goai_ctx_sync(goai_ctx* ctx)
{
    CTX_ENTER(ctx);
    GOAI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GOAI_OK;
}

int goai_transition(goai_ctx* ctx, const uint8_t* states, const int32_t* actions, int canonical,
                    uint8_t* states_out, int32_t* status, int n)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(states && actions && states_out, "NULL array argument");
    return launch_transition(ctx->S, states, actions, canonical, states_out, status, n, ctx->stream);
}

int goai_children(goai_ctx* ctx, const uint8_t* states, int canonical, uint8_t* children_out, uint8_t* valid_out,
                  int8_t* win_out, int n)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(states && children_out, "NULL array argument");
    return launch_children(ctx->S, states, canonical, children_out, valid_out, win_out, n, ctx->stream);
}

int goai_score(goai_ctx* ctx, const uint8_t* states, int32_t* black_area, int32_t* white_area, int32_t* ended,
               int32_t* winner, int n)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(states != nullptr, "NULL array argument");
    return launch_score(ctx->S, states, black_area, white_area, ended, winner, n, ctx->stream);
}

int goai_liberties(goai_ctx* ctx, const uint8_t* states, int32_t* black_libs, int32_t* white_libs, int n)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(states != nullptr, "NULL array argument");
    return launch_liberties(ctx->S, states, black_libs, white_libs, n, ctx->stream);
}

int goai_nn_init(goai_ctx* ctx, int net_kind, int n_blocks, int max_batch, int dtype)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(dtype == GOAI_DTYPE_F32 || dtype == GOAI_DTYPE_BF16X3 || dtype == GOAI_DTYPE_BF16X6 || dtype == GOAI_DTYPE_F16X3 ||
                     dtype == GOAI_DTYPE_F16,
                 "dtype must be GOAI_DTYPE_F32, GOAI_DTYPE_BF16X3, GOAI_DTYPE_BF16X6, GOAI_DTYPE_F16X3 or GOAI_DTYPE_F16");
    return nn_init(ctx, net_kind, n_blocks, max_batch, dtype);
}

int goai_load_weights(goai_ctx* ctx, const char* const* names, const float* const* tensors_host,
                      const int64_t* numels, int n_tensors)
{
    CTX_ENTER(ctx);
    return nn_load_weights(ctx, names, tensors_host, numels, n_tensors);
}

int goai_nn_conv_algo(goai_ctx* ctx)
{
    GOAI_REQUIRE(ctx, "ctx is NULL");
    GOAI_REQUIRE(ctx->net.kind >= 0, "call goai_nn_init first");
    if (ctx->net.dtype == GOAI_DTYPE_BF16X3) return GOAI_CONV_BF16X3;
    if (ctx->net.split6 && ctx->net.parts == 1) return ctx->net.wino_mode == 3 ? GOAI_CONV_WINOGRAD_F3_F16 : GOAI_CONV_WINOGRAD_F16;
    if (ctx->net.split6 && ctx->net.parts == 2) return ctx->net.wino_mode == 3 ? GOAI_CONV_WINOGRAD_F3_F16X3 : GOAI_CONV_WINOGRAD_F16X3;
    if (ctx->net.split6) return ctx->net.wino_mode == 3 ? GOAI_CONV_WINOGRAD_F3_BF16X6 : GOAI_CONV_WINOGRAD_BF16X6;
    if (ctx->net.wino_mode == 3) return ctx->net.wino3_paired ? GOAI_CONV_WINOGRAD_F3_PAIRED : GOAI_CONV_WINOGRAD_F3;
    return ctx->net.wino_mode ? GOAI_CONV_WINOGRAD : GOAI_CONV_DIRECT;
}

int goai_mcts_steps_in_one_launch(goai_ctx* ctx)
{
    GOAI_REQUIRE(ctx, "ctx is NULL");
    return ctx->steps_in_one_launch;
}

int goai_nn_convs_per_launch(goai_ctx* ctx)
{
    GOAI_REQUIRE(ctx, "ctx is NULL");
    GOAI_REQUIRE(ctx->net.kind >= 0, "call goai_nn_init first");
    return (ctx->net.trunk_fused && ctx->net.layers > 0) ? 2 * ctx->net.layers : 1;
}

int goai_nn_forward(goai_ctx* ctx, const uint8_t* states, float* pi_logits, float* val_logits, int n, int mode)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    GOAI_REQUIRE(mode == GOAI_NN_RAW || mode == GOAI_NN_NUMPY_SHIM, "unknown mode");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(states && val_logits, "NULL array argument");
    GOAI_REQUIRE(ctx->net.kind == 0 || pi_logits == nullptr, "the value net has no policy head");
    int rc = nn_forward_raw(ctx, states, pi_logits, val_logits, n);
    if (rc) return rc;
    if (mode == GOAI_NN_NUMPY_SHIM) rc = launch_numpy_shim(ctx->S, states, pi_logits, val_logits, n, ctx->stream);
    return rc;
}

int goai_nn_forward_static(goai_ctx* ctx, const uint8_t* states, float* pi_logits, float* val_logits, int n, int mode)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    GOAI_REQUIRE(mode == GOAI_NN_RAW || mode == GOAI_NN_NUMPY_SHIM, "unknown mode");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(states && val_logits, "NULL array argument");
    GOAI_REQUIRE(ctx->net.kind >= 0, "call goai_nn_init first");
    GOAI_REQUIRE(ctx->net.kind == 0 || pi_logits == nullptr, "the value net has no policy head");
    int rc = nn_forward_static(ctx, states, pi_logits, val_logits, n);
    if (rc) return rc;
    if (mode == GOAI_NN_NUMPY_SHIM) rc = launch_numpy_shim(ctx->S, states, pi_logits, val_logits, n, ctx->stream);
    return rc;
}

int goai_mcts_init(goai_ctx* ctx, int n_games, int max_sims, int mode)
{
    CTX_ENTER(ctx);
    return mcts_init(ctx, n_games, max_sims, mode);
}

int goai_mcts_reset(goai_ctx* ctx, const uint8_t* root_states)
{
    CTX_ENTER(ctx);
    return mcts_reset(ctx, root_states);
}

int goai_mcts_select(goai_ctx* ctx, uint8_t* leaf_states_out)
{
    CTX_ENTER(ctx);
    return mcts_select(ctx, leaf_states_out);
}

int goai_mcts_backup(goai_ctx* ctx, const float* pi_logits, const float* prior, const float* val_logits,
                     int logits_are_raw)
{
    CTX_ENTER(ctx);
    return mcts_backup(ctx, pi_logits, prior, val_logits, logits_are_raw);
}

int goai_mcts_expand(goai_ctx* ctx, uint8_t* children_out, uint8_t* valid_out)
{
    CTX_ENTER(ctx);
    return mcts_expand(ctx, children_out, valid_out);
}

int goai_mcts_backup_q(goai_ctx* ctx, const double* leaf_val, const double* child_vals, int vals_are_raw)
{
    CTX_ENTER(ctx);
    return mcts_backup_q(ctx, leaf_val, child_vals, vals_are_raw);
}

int goai_mcts_child_values(goai_ctx* ctx, double* inverted_child_vals)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    GOAI_REQUIRE(ctx->tree.mode == GOAI_SEARCH_Q, "needs a context initialised with GOAI_SEARCH_Q");
    GOAI_REQUIRE(inverted_child_vals != nullptr, "NULL output");
    return launch_q_child_values(ctx->S, ctx->tree, inverted_child_vals, ctx->stream);
}

int goai_mcts_policy_value(goai_ctx* ctx, double temp, int from_values, double* pi)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    GOAI_REQUIRE(ctx->tree.mode == GOAI_SEARCH_Q, "needs a context initialised with GOAI_SEARCH_Q");
    GOAI_REQUIRE(pi != nullptr, "pi is NULL");
    return launch_q_policy(ctx->S, ctx->tree, temp, from_values, pi, ctx->stream);
}

int goai_mcts_run(goai_ctx* ctx, int n_sims)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n_sims >= 0, "n_sims < 0");
    return mcts_steps(ctx, n_sims + 1);
}

int goai_mcts_steps(goai_ctx* ctx, int n_steps)
{
    CTX_ENTER(ctx);
    return mcts_steps(ctx, n_steps);
}

int goai_mcts_readout(goai_ctx* ctx, int32_t* visit_counts, double* prior, int32_t* root_visits)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    return launch_mcts_readout(ctx->S, ctx->tree, visit_counts, prior, root_visits, ctx->stream);
}

int goai_mcts_policy(goai_ctx* ctx, double temp, double* pi)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    GOAI_REQUIRE(pi != nullptr && temp > 0.0, "pi is NULL or temp <= 0");
    return launch_mcts_policy(ctx->S, ctx->tree, temp, pi, ctx->stream);
}

int goai_uniform_policy(goai_ctx* ctx, int allow_pass, double* pi)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    GOAI_REQUIRE(pi != nullptr, "pi is NULL");
    return launch_uniform_policy(ctx->S, ctx->tree, allow_pass, pi, ctx->stream);
}

int goai_sample_actions(goai_ctx* ctx, const double* pi, uint64_t seed, uint64_t step, int game_id_offset,
                        int32_t* actions)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.n_games > 0, "call goai_mcts_init first");
    GOAI_REQUIRE(pi && actions, "NULL array argument");
    return launch_sample(ctx->S, ctx->tree, pi, seed, step, game_id_offset, actions, ctx->stream);
}

int goai_mcts_advance(goai_ctx* ctx, const int32_t* actions)
{
    CTX_ENTER(ctx);
    return mcts_advance(ctx, actions);
}

int goai_mcts_set_active(goai_ctx* ctx, int n_active)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.n_games > 0, "call goai_mcts_init first");
    GOAI_REQUIRE(n_active >= 1 && n_active <= ctx->tree.n_games, "n_active must be in [1, n_games]");
    ctx->tree.n_active = n_active;
    ctx->tree.steps_since_reset = -1;         // the roots must be set again
    ctx->tree.pending_select = false;         // a half-done step of the old set is dropped with it
    ctx->tree.pending_expand = 0;
    return GOAI_OK;
}

int goai_sample_actions_keyed(goai_ctx* ctx, const double* pi, uint64_t seed, const uint64_t* game_ids,
                              const int32_t* steps, int n, int32_t* actions)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return GOAI_OK;
    GOAI_REQUIRE(pi && game_ids && steps && actions, "NULL array argument");
    return launch_sample_keyed(ctx->S, pi, seed, game_ids, steps, n, actions, ctx->stream);
}

int goai_mcts_node_capacity(goai_ctx* ctx)
{
    GOAI_REQUIRE(ctx != nullptr && ctx->tree.n_games > 0, "call goai_mcts_init first");
    return ctx->tree.ncap;
}

int goai_mcts_export_tree(goai_ctx* ctx, int game, int32_t* n_nodes, int32_t* parent, int32_t* action, int32_t* visits,
                          double* value, double* raw_sum, double* prior, uint8_t* states)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    GOAI_REQUIRE(game >= 0 && game < ctx->tree.n_games, "no such game");
    GOAI_REQUIRE(n_nodes && parent && action && visits && value && raw_sum && prior && states, "NULL output");
    return launch_mcts_export(ctx->S, ctx->tree, game, n_nodes, parent, action, visits, value, raw_sum, prior, states, ctx->stream);
}

int goai_mcts_set_refill(goai_ctx* ctx, int on)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.n_games > 0, "call goai_mcts_init first");
    ctx->tree.refill = on != 0;
    return GOAI_OK;
}

int goai_mcts_game_stats(goai_ctx* ctx, int64_t* stats_host)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.steps_since_reset >= 0 && stats_host, "call goai_mcts_init and goai_mcts_reset first / NULL output");
    GOAI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const size_t G = (size_t)ctx->tree.n_games, active = (size_t)ctx->tree.n_active;
    std::vector<long long> slots(4 * G);
    GOAI_HIP_CHECK(hipMemcpy(slots.data(), ctx->tree.fin, slots.size() * sizeof(long long), hipMemcpyDeviceToHost));
    // totals "since the last goai_mcts_reset" of the ACTIVE slots: the reset clears exactly those, so a slot that
    // goai_mcts_set_active left outside the set does not leak what it finished earlier
    for (int k = 0; k < 4; k++) {
        int64_t sum = 0;
        for (size_t g = 0; g < active; g++) sum += slots[k * G + g];
        stats_host[k] = sum;
    }
    return GOAI_OK;
}

int goai_mcts_roots(goai_ctx* ctx, uint8_t* states, int32_t* turn, int32_t* moves, int32_t* result)
{
    CTX_ENTER(ctx);
    const goai::Tree& t = ctx->tree;
    GOAI_REQUIRE(t.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    const size_t bytes = (size_t)t.n_active * sizeof(int32_t);
    if (turn) GOAI_HIP_CHECK(hipMemcpyAsync(turn, t.turn, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (moves) GOAI_HIP_CHECK(hipMemcpyAsync(moves, t.moves, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (result) GOAI_HIP_CHECK(hipMemcpyAsync(result, t.result, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (states) return launch_mcts_roots(ctx->S, t, states, ctx->stream);
    return GOAI_OK;
}

int goai_mcts_counters(goai_ctx* ctx, int64_t* counters_host)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->tree.n_games > 0 && counters_host, "call goai_mcts_init first / NULL output");
    GOAI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // one slot per counter and game on the device (a single address would serialise n_games atomics
    // per kernel at the memory side): summed here
    const size_t G = (size_t)ctx->tree.n_games;
    std::vector<int64_t> slots(4 * G);
    GOAI_HIP_CHECK(hipMemcpy(slots.data(), ctx->tree.counters, slots.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) {
        int64_t sum = 0;
        for (size_t g = 0; g < G; g++) sum += slots[k * G + g];
        counters_host[k] = sum;
    }
    return GOAI_OK;
}

int goai_train_init(goai_ctx* ctx, int net_kind, int n_blocks, int max_batch)
{
    CTX_ENTER(ctx);
    return train_init(ctx, net_kind, n_blocks, max_batch);
}

int goai_train_load(goai_ctx* ctx, const char* const* names, const float* const* tensors_host, const int64_t* numels,
                    int n_tensors, int strict)
{
    CTX_ENTER(ctx);
    return train_load(ctx, names, tensors_host, numels, n_tensors, strict);
}

int goai_train_export(goai_ctx* ctx, int what, const char* const* names, float* const* tensors_host, const int64_t* numels,
                      int n_tensors)
{
    CTX_ENTER(ctx);
    return train_export(ctx, what, names, tensors_host, numels, n_tensors);
}

int goai_train_reset_optimizer(goai_ctx* ctx)
{
    CTX_ENTER(ctx);
    return train_reset_optimizer(ctx);
}

int goai_train_step(goai_ctx* ctx, const uint8_t* states, const uint8_t* next_states, const float* wins, const float* pi,
                    int n, float lr, float weight_decay, float* metrics_host)
{
    CTX_ENTER(ctx);
    return train_step(ctx, states, next_states, wins, pi, n, lr, weight_decay, metrics_host);
}

int goai_train_params(goai_ctx* ctx, float** params_device, int64_t* numel)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(ctx->train.ready, "call goai_train_init first");
    GOAI_REQUIRE(params_device && numel, "NULL output");
    *params_device = ctx->train.params;
    *numel = (int64_t)ctx->train.n_params;
    return GOAI_OK;
}

int goai_train_sync_inference(goai_ctx* ctx)
{
    CTX_ENTER(ctx);
    return train_sync_inference(ctx);
}

int goai_comm_unique_id(void* id_host) { return comm_unique_id(id_host); }

int goai_comm_init(goai_ctx* ctx, const void* id_host, int n_ranks, int rank, void** rccl_comm_out)
{
    CTX_ENTER(ctx);
    return comm_init(ctx, id_host, n_ranks, rank, rccl_comm_out);
}

int goai_comm_destroy(goai_ctx* ctx, void* rccl_comm)
{
    CTX_ENTER(ctx);
    return comm_destroy(ctx, rccl_comm);
}

int goai_gather_stats(goai_ctx* ctx, void* rccl_comm, double* extra_inout_host, int n_extra, int64_t* totals_host)
{
    CTX_ENTER(ctx);
    return gather_stats(ctx, rccl_comm, extra_inout_host, n_extra, totals_host);
}

int goai_profile_enable(goai_ctx* ctx, int enable)
{
    CTX_ENTER(ctx);
    ctx->prof_enabled = enable != 0;
    return GOAI_OK;
}

int goai_profile_read(goai_ctx* ctx, double* conv_ms_total, int64_t* conv_launches)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(conv_ms_total && conv_launches, "NULL output");
    GOAI_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    for (size_t i = 0; i + 1 < ctx->prof_events.size(); i += 2) {
        float ms = 0.0f;
        GOAI_HIP_CHECK(hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
        total += ms;
    }
    *conv_ms_total = total;
    // (a k_search_steps launch counts as the trunk passes it holds: conv_ms_total / conv_launches stays "per trunk pass")
    *conv_launches = (int64_t)(ctx->prof_events.size() / 2) + ctx->prof_extra_passes;
    ctx->prof_extra_passes = 0;
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    return GOAI_OK;
}

}  // extern "C"
