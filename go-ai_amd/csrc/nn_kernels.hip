// nn_kernels.hip -- the policy/value CNN forward over a batch of leaf states.
//
// Replaces the torch forward the reference runs on batch 1 per leaf
// (/root/reference/go_ai/models/__init__.py:14-31,234-251 trunk + BasicBlock,
//  /root/reference/go_ai/models/ac_net.py:15-31,40-45, /root/reference/go_ai/models/val_net.py:12-24).
// BatchNorm is folded into the preceding convolution on the host (eval() is forced by the
// reference, models/__init__.py:64).
//
// Data layout: activations are channel-last fp32, act[row][128] with row = board*P + cell,
// so a 3x3 convolution is an implicit GEMM over flat rows (conv_kernels.hip).  This file
// holds the small kernels around it: the first convolution, the heads, the numpy shim.
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"
#include "nn_groups.h"

namespace goai {

// --------------------------------------------------------------------------------------
// conv0: 6 -> 128, 3x3, pad 1, + folded BN + ReLU, reading the uint8 state planes.
// fp32 VALU version (GOAI_CONV0=valu; the default is k_conv0_mfma below, 29 -> 12 us).
// One 256-thread workgroup per board; thread = (output channel, cell parity); the 54 weights
// of the thread's channel stay in registers, the board sits in LDS as [cell][8] floats with
// a zero border (padding is applied AFTER the AC net's -0.5 input shift, so it is 0).
// --------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_conv0(const uint8_t* __restrict__ states, const float* __restrict__ w0,
                                               const float* __restrict__ b0, float shift, float* __restrict__ out,
                                               int nboards)
{
    using G = Geom<S>;
    __shared__ __attribute__((aligned(16))) float xin[G::NPAD * 8];
    const int tid = threadIdx.x, cout = tid & 127, half = tid >> 7;
    float w[54];
#pragma unroll
    for (int k = 0; k < 54; k++) w[k] = w0[k * 128 + cout];
    const float bias = b0[cout];
    for (int b = blockIdx.x; b < nboards; b += gridDim.x) {
        const uint8_t* st = states + (size_t)b * 6 * G::P;
        __syncthreads();
        for (int i = tid; i < G::NPAD * 8; i += 256) xin[i] = 0.0f;
        __syncthreads();
        for (int i = tid; i < 6 * G::P; i += 256) {
            int ci = i / G::P, p = i % G::P;
            xin[cell_q<S>(p) * 8 + ci] = (float)st[i] - shift;
        }
        __syncthreads();
        for (int p = half; p < G::P; p += 2) {
            const int q = cell_q<S>(p);
            float acc = bias;
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int qq = q + (t / 3 - 1) * G::PS + (t % 3 - 1);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&xin[qq * 8]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&xin[qq * 8 + 4]);
                acc = fmaf(lo[0], w[0 * 9 + t], acc);
                acc = fmaf(lo[1], w[1 * 9 + t], acc);
                acc = fmaf(lo[2], w[2 * 9 + t], acc);
                acc = fmaf(lo[3], w[3 * 9 + t], acc);
                acc = fmaf(hi[0], w[4 * 9 + t], acc);
                acc = fmaf(hi[1], w[5 * 9 + t], acc);
            }
            out[((size_t)b * G::P + p) * 128 + cout] = fmaxf(acc, 0.0f);
        }
    }
}

template <int S, int NBT>
__global__ __launch_bounds__(256) void k_conv0_mfma(const uint8_t* __restrict__ states, const uint4* __restrict__ w0s,
                                                    const float* __restrict__ b0, float shift,
                                                    float* __restrict__ out, int nboards)
{
    __shared__ Conv0Lds<S, NBT> lds;
    conv0_group<S, NBT>(lds, states, w0s, b0, shift, out, nboards, blockIdx.x * NBT);
}

template <int S, int HBT>
__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ act, HeadWeights hw, int is_ac,
                                               float* __restrict__ pi_out, float* __restrict__ val_out, int nboards, bool act_is_h)
{
    __shared__ HeadsLds<S, HBT> lds;
    heads_groups<S, HBT>(lds, act, hw, is_ac, pi_out, val_out, nboards, blockIdx.x * HBT, gridDim.x * HBT, act_is_h);
}

// --------------------------------------------------------------------------------------
// RLNet._numpy post-processing (/root/reference/go_ai/models/__init__.py:90-105):
//   pi_logits += F32_LOWEST * invalid ; pi_logits += -100 * winning(child) for children that
//   ended the game ; val = 100 * winning(state) for ended states.  One wavefront per state.
// The additions are done as numpy does them: in float64, rounded to float32 on store.
// --------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(64) void k_numpy_shim(const uint8_t* __restrict__ states, float* __restrict__ pi,
                                                   float* __restrict__ val, int n)
{
    using G = Geom<S>;
    __shared__ BoardLds<S> L;
    board_init_border<S>(L);
    const int lane = lane_id();
    const double LOWEST = -3.4028234663852886e38;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const uint8_t* in = states + (size_t)i * 6 * G::P;
        uint64_t me[G::K], opp[G::K], inv[G::K];
        uint32_t flags;
        const int player = planes_load<S>(in, me, opp, inv, flags);
        if (pi) {
            float* row = pi + (size_t)i * G::A;
#pragma unroll
            for (int k = 0; k < G::K; k++) {
                int idx = lane + 64 * k;
                if (idx < G::P && ((inv[k] >> lane) & 1ull)) row[idx] = (float)((double)row[idx] + LOWEST);
            }
        }
        const bool done = flags & FLAG_DONE;
        if (done || (flags & FLAG_PASS)) {
            board_load<S>(L, me, opp);
            board_analyze<S>(L);
            int a_me, a_opp;
            board_areas<S>(L, a_me, a_opp);
            const int sign_mover = (a_me > a_opp) - (a_me < a_opp);
            if (done && val && lane == 0) val[i] = (float)(100 * (player ? -sign_mover : sign_mover));
            // pass child: same board seen by the opponent, game over -> winning(child) = -sign_mover
            if (pi && lane == 0) {
                float* row = pi + (size_t)i * G::A;
                row[G::P] = (float)((double)row[G::P] + (double)(100 * sign_mover));
            }
            if (done && pi) {
                // children of a finished game are all "ended" (GymGo copies the DONE plane):
                // every legal move gets -100 * winning(child) on the child's own board.
                for (int a = 0; a < G::P; a++) {
                    uint64_t nme[G::K], nopp[G::K], ninv[G::K];
                    uint32_t nflags;
                    int sgn;
                    if (board_next_state<S>(L, me, opp, inv, flags, a, nme, nopp, ninv, nflags, sgn) == 0 && lane == 0) {
                        float* row = pi + (size_t)i * G::A;
                        row[a] = (float)((double)row[a] + (double)(-100 * sgn));
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
    }
}

// ---- launchers ----------------------------------------------------------------------------
int launch_conv0(int S_rt, const uint8_t* states, const float* w0, const void* w0_split, const float* b0, float shift,
                 float* out, int nboards, hipStream_t st)
{
    if (w0_split) {
        // eight boards per workgroup when there are workgroups enough (nn_groups.h)
        const char* nb = getenv("GOAI_CONV0_NB");
        const bool wide = nb ? nb[0] == '8' : (S_rt >= 13 && nboards >= 4096);     // (9x9, 32,768 boards: no difference)
        const int per = wide ? 8 : NB0;
        const int grid = (nboards + per - 1) / per;
        if (wide) {
            GOAI_DISPATCH_SIZE(S_rt, (k_conv0_mfma<S, 8><<<dim3(grid), dim3(256), 0, st>>>(states, (const uint4*)w0_split, b0, shift, out, nboards)));
        } else {
            GOAI_DISPATCH_SIZE(S_rt, (k_conv0_mfma<S, NB0><<<dim3(grid), dim3(256), 0, st>>>(states, (const uint4*)w0_split, b0, shift, out, nboards)));
        }
        GOAI_HIP_CHECK(hipGetLastError());
        return GOAI_OK;
    }
    int grid = nboards < 4096 ? nboards : 4096;
    GOAI_DISPATCH_SIZE(S_rt, k_conv0<S><<<dim3(grid), dim3(256), 0, st>>>(states, w0, b0, shift, out, nboards));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_heads(int S_rt, const float* act, const HeadWeights& hw, int is_ac, float* pi, float* val, int nboards,
                 hipStream_t st, bool act_is_h)
{
    // eight boards per workgroup where the Linear weights are large and there are workgroups enough (nn_groups.h)
    const char* hb = getenv("GOAI_HEADS_HB");
    const bool wide = hb ? hb[0] == '8' : (S_rt >= 13 && nboards >= 2048);
    const int per = wide ? 8 : HB;
    int groups = (nboards + per - 1) / per;
    int grid = groups < 8192 ? groups : 8192;
    if (wide) {
        GOAI_DISPATCH_SIZE(S_rt, (k_heads<S, 8><<<dim3(grid), dim3(256), 0, st>>>(act, hw, is_ac, pi, val, nboards, act_is_h)));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, (k_heads<S, HB><<<dim3(grid), dim3(256), 0, st>>>(act, hw, is_ac, pi, val, nboards, act_is_h)));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_numpy_shim(int S_rt, const uint8_t* states, float* pi, float* val, int n, hipStream_t st)
{
    int grid = n < 65536 ? n : 65536;
    GOAI_DISPATCH_SIZE(S_rt, k_numpy_shim<S><<<dim3(grid), dim3(64), 0, st>>>(states, pi, val, n));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
