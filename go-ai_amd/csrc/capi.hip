// capi.hip -- extern "C" surface of libgoai_hip.so (declared in include/goai_hip.h).
#include <new>
#include <string>

#include "ctx.h"
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
int goai_version(void) { return 1; }

int goai_ctx_create(goai_ctx** out, int device, int board_size)
{
    GOAI_REQUIRE(out != nullptr, "out is NULL");
    GOAI_REQUIRE(size_supported(board_size), "board_size must be one of 5, 7, 9, 13, 19");
    int ndev = 0;
    GOAI_HIP_CHECK(hipGetDeviceCount(&ndev));
    GOAI_REQUIRE(device >= 0 && device < ndev, "no such GPU");
    GOAI_HIP_CHECK(hipSetDevice(device));
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

int goai_nn_init(goai_ctx* ctx, int net_kind, int n_blocks, int max_batch, int dtype)
{
    CTX_ENTER(ctx);
    GOAI_REQUIRE(dtype == GOAI_DTYPE_F32, "only GOAI_DTYPE_F32 is implemented");
    return nn_init(ctx, net_kind, n_blocks, max_batch);
}

int goai_load_weights(goai_ctx* ctx, const char* const* names, const float* const* tensors_host,
                      const int64_t* numels, int n_tensors)
{
    CTX_ENTER(ctx);
    return nn_load_weights(ctx, names, tensors_host, numels, n_tensors);
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

}  // extern "C"
