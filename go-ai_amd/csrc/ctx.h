// ctx.h -- the opaque goai_ctx: one GPU, one board size, one stream, plus the device
// buffers of the network and of the per-game search trees.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "mcts.h"
#include "nn.h"
#include "train.h"

struct goai_ctx {
    int device = 0;
    int S = 0, P = 0, A = 0;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;   // everything hipMalloc'ed on behalf of this context
    std::vector<void*> host_allocs;   // pinned host memory
    goai::Net net;
    // optional per-launch timing of the dominant kernel (conv3x3) with HIP events on ctx->stream
    bool prof_enabled = false;
    std::vector<hipEvent_t> prof_events;   // start/stop pairs
    int64_t prof_extra_passes = 0;         // trunk passes inside the timed launches beyond one per launch (k_search_steps)
    int steps_in_one_launch = 0;           // the last goai_mcts_steps: trunk passes per k_search_steps launch, 0 = separate launches
    goai::Tree tree;
    goai::Train train;
    void* gather_buf = nullptr;   // device staging of goai_gather_stats (in allocs)

    // undo the allocations made since the marks (a failed *_init)
    void release_from(size_t mark, size_t host_mark)
    {
        while (allocs.size() > mark) { (void)hipFree(allocs.back()); allocs.pop_back(); }
        while (host_allocs.size() > host_mark) { (void)hipHostFree(host_allocs.back()); host_allocs.pop_back(); }
        (void)hipGetLastError();   // the failure was reported by the init call; do not leave it for the next launch check
    }

    void release()
    {
        for (void* p : allocs) (void)hipFree(p);
        for (void* p : host_allocs) (void)hipHostFree(p);
        host_allocs.clear();
        for (auto& kv : train.graphs) (void)hipGraphExecDestroy(kv.second);
        train.graphs.clear();
        if (train.gstream) (void)hipStreamDestroy(train.gstream);
        train.gstream = nullptr;
        for (hipEvent_t e : prof_events) (void)hipEventDestroy(e);
        prof_events.clear();
        allocs.clear();
    }
};
