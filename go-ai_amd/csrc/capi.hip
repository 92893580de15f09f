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

}  // extern "C"
