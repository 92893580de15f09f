// gather.hip -- the one collective of the path: the end-of-run sums over ranks
//   /root/reference/go_ai/utils.py:201-204   comm.allreduce(..., op=MPI.SUM) of seconds per game, win rates, steps
// as an RCCL all-reduce on the context's stream.  The search itself never communicates (games are sharded).
//
// RCCL is resolved at run time (dlopen of librccl.so.1: the copy the process already has, e.g. PyTorch's, else the
// system's), so libgoai_hip.so has no link-time dependency on it and single-GPU users never load it.  A communicator
// handed to goai_gather_stats must come from that same library -- goai_comm_init makes one.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "ctx.h"
#include "goai_internal.h"

namespace goai {

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : { "librccl.so.1", "librccl.so" }) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process, if any
            if (r.lib) break;
        }
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.lib) { r.error = std::string("RCCL not found: ") + dlerror(); return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString)
            r.error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
    });
    return r;
}

#define GOAI_RCCL_CHECK(expr)                                                                 \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) {                                                              \
            set_error(std::string(#expr) + ": " + rccl().GetErrorString(_r));                 \
            return GOAI_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)
#define REQUIRE_RCCL()                                                                        \
    do {                                                                                      \
        if (!rccl().error.empty()) { set_error(rccl().error); return GOAI_ERR_STATE; }        \
    } while (0)
}  // namespace

int comm_unique_id(void* id128_host)
{
    GOAI_REQUIRE(id128_host != nullptr, "id is NULL");
    REQUIRE_RCCL();
    static_assert(sizeof(ncclUniqueId) == GOAI_COMM_ID_BYTES, "ncclUniqueId size");
    GOAI_RCCL_CHECK(rccl().GetUniqueId((ncclUniqueId*)id128_host));
    return GOAI_OK;
}

int comm_init(goai_ctx* c, const void* id128_host, int n_ranks, int rank, void** comm_out)
{
    GOAI_REQUIRE(id128_host && comm_out, "NULL argument");
    GOAI_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank must be in [0, n_ranks)");
    REQUIRE_RCCL();
    ncclUniqueId id;
    memcpy(&id, id128_host, sizeof(id));
    ncclComm_t comm = nullptr;
    GOAI_RCCL_CHECK(rccl().CommInitRank(&comm, n_ranks, id, rank));
    *comm_out = comm;
    return GOAI_OK;
}

int comm_destroy(goai_ctx* c, void* comm)
{
    if (!comm) return GOAI_OK;
    REQUIRE_RCCL();
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    GOAI_RCCL_CHECK(rccl().CommDestroy((ncclComm_t)comm));
    return GOAI_OK;
}

// totals_host[8]: sums over the communicator's ranks of this context's goai_mcts_game_stats [0..3] and
// goai_mcts_counters [4..7]; extra_inout_host[n_extra]: the caller's own additive figures (seconds, first-player
// wins ...), summed in place.  comm == NULL: one rank, the local figures.
int gather_stats(goai_ctx* c, void* comm, double* extra_inout_host, int n_extra, int64_t* totals_host)
{
    GOAI_REQUIRE(totals_host != nullptr, "totals is NULL");
    GOAI_REQUIRE(n_extra >= 0 && n_extra <= GOAI_GATHER_MAX_EXTRA && (n_extra == 0 || extra_inout_host), "0 <= n_extra <= GOAI_GATHER_MAX_EXTRA");
    Tree& t = c->tree;
    GOAI_REQUIRE(t.n_games > 0 && t.steps_since_reset >= 0, "call goai_mcts_init and goai_mcts_reset first");
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    const size_t G = (size_t)t.n_games;
    std::vector<long long> slots(4 * G);
    const void* src[2] = { t.fin, t.counters };
    for (int h = 0; h < 2; h++) {
        GOAI_HIP_CHECK(hipMemcpy(slots.data(), src[h], slots.size() * sizeof(long long), hipMemcpyDeviceToHost));
        // finished-game totals: the active slots only (goai_mcts_reset clears those; a slot outside the active set
        // still holds what it finished before goai_mcts_set_active shrank the set); counters: every slot
        const size_t upto = h == 0 ? (size_t)t.n_active : G;
        for (int k = 0; k < 4; k++) {
            int64_t sum = 0;
            for (size_t g = 0; g < upto; g++) sum += slots[k * G + g];
            totals_host[4 * h + k] = sum;
        }
    }
    if (!comm) return GOAI_OK;
    REQUIRE_RCCL();
    if (!c->gather_buf) {
        GOAI_HIP_CHECK(hipMalloc(&c->gather_buf, 8 * sizeof(int64_t) + GOAI_GATHER_MAX_EXTRA * sizeof(double)));
        c->allocs.push_back(c->gather_buf);
    }
    int64_t* d_tot = (int64_t*)c->gather_buf;
    double* d_extra = (double*)(d_tot + 8);
    GOAI_HIP_CHECK(hipMemcpyAsync(d_tot, totals_host, 8 * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    if (n_extra) GOAI_HIP_CHECK(hipMemcpyAsync(d_extra, extra_inout_host, n_extra * sizeof(double), hipMemcpyHostToDevice, c->stream));
    GOAI_RCCL_CHECK(rccl().AllReduce(d_tot, d_tot, 8, ncclInt64, ncclSum, (ncclComm_t)comm, c->stream));
    if (n_extra) GOAI_RCCL_CHECK(rccl().AllReduce(d_extra, d_extra, n_extra, ncclFloat64, ncclSum, (ncclComm_t)comm, c->stream));
    GOAI_HIP_CHECK(hipMemcpyAsync(totals_host, d_tot, 8 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    if (n_extra) GOAI_HIP_CHECK(hipMemcpyAsync(extra_inout_host, d_extra, n_extra * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GOAI_OK;
}

}  // namespace goai
