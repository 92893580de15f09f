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

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// --------------------------------------------------------------------------------------
// conv0 on the bf16 matrix cores, exactly: the inputs are 0 / 1 (value net) or -0.5 / +0.5
// (actor-critic net) and 0 in the padding, all exact in bf16, and every fp32 weight is the sum
// of three bf16 numbers (w1 = bf16(w), w2 = bf16(w - w1), w3 = bf16(w - w1 - w2): 3 x 8
// significant bits), so x * w = x*w1 + x*w2 + x*w3 with every product exact in the fp32
// accumulator -- the result differs from the fmaf chain of k_conv0 only by the order of the
// fp32 additions.  K is laid out as tap * 8 + channel (6 real channels, 12 taps: 3 k-steps of
// 32), so a lane's 8 consecutive k are ONE neighbour cell's channels = one 16-byte LDS read.
// Products are formed transposed (A = weights, B = cells): a lane ends with 4 consecutive
// output channels of one cell -> 16-byte stores.  Workgroup = 4 waves x 32 output channels,
// NB0 boards; the 72 weight registers of a wave are loaded once.
// --------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int NB0 = 2;      // boards per workgroup (measured at 1024 boards of 9x9: 4 -> 13.4 us, 2 -> 11.7 us, 1 -> 13.2 us)

template <int S>
__global__ __launch_bounds__(256) void k_conv0_mfma(const uint8_t* __restrict__ states, const uint4* __restrict__ w0s,
                                                    const float* __restrict__ b0, float shift,
                                                    float* __restrict__ out, int nboards)
{
    using G = Geom<S>;
    constexpr int CELLS = NB0 * G::P, TILES = (CELLS + 15) / 16;
    __shared__ uint4 xin[NB0 * G::NPAD + 1];                       // [board][padded cell] 8 bf16; last = zero cell
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * NB0;

    // weights: w0s[split][kstep][cout][kgroup] = 8 bf16
    bf16x8 wa[2][3][3];                                            // [cout tile of the wave][kstep][split]
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
#pragma unroll
            for (int sp = 0; sp < 3; sp++)
                wa[ct][ks][sp] = __builtin_bit_cast(bf16x8, w0s[((sp * 3 + ks) * 128 + wave * 32 + ct * 16 + li) * 4 + lq]);
    f32x4 bias[2];
#pragma unroll
    for (int ct = 0; ct < 2; ct++) bias[ct] = *reinterpret_cast<const f32x4*>(&b0[wave * 32 + ct * 16 + 4 * lq]);

    for (int i = tid; i < NB0 * G::NPAD + 1; i += 256) xin[i] = uint4{ 0u, 0u, 0u, 0u };
    __syncthreads();
    // planes -> bf16 (x - shift): one thread per (board, cell), 6 channels packed into 16 bytes
    const uint32_t one = __builtin_bit_cast(uint16_t, (__bf16)(1.0f - shift)), zer = __builtin_bit_cast(uint16_t, (__bf16)(0.0f - shift));
    for (int i = tid; i < CELLS; i += 256) {
        const int bl = i / G::P, p = i - bl * G::P;
        if (b_first + bl >= nboards) break;
        const uint8_t* st = states + ((size_t)(b_first + bl) * 6) * G::P + p;
        uint32_t h[6];
#pragma unroll
        for (int ci = 0; ci < 6; ci++) h[ci] = st[ci * G::P] ? one : zer;
        xin[bl * G::NPAD + cell_q<S>(p)] = uint4{ h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), 0u };
    }
    __syncthreads();

    for (int t = 0; t < TILES; t++) {
        // this lane's cell (B operand column li) and the 3 neighbour cells it feeds: tap = ks*4 + lq
        const int cell = t * 16 + li;
        const bool on = cell < CELLS;
        const int bl = on ? cell / G::P : 0, p = on ? cell - bl * G::P : 0;
        const int q = bl * G::NPAD + cell_q<S>(p);
        f32x4 acc[2] = { bias[0], bias[1] };
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            const int tap = ks * 4 + lq;
            const int qq = (on && tap < 9) ? q + (tap / 3 - 1) * G::PS + (tap % 3 - 1) : NB0 * G::NPAD;
            const bf16x8 xb = __builtin_bit_cast(bf16x8, xin[qq]);
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int sp = 2; sp >= 0; sp--)                    // small terms first
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][ks][sp], xb, acc[ct], 0, 0, 0);
        }
        // C/D layout: col = lane & 15 = cell, row = 4 * (lane >> 4) + reg = output channel within the tile
        if (on && b_first + bl < nboards) {
            float* dst = out + ((size_t)(b_first + bl) * G::P + p) * 128 + wave * 32 + 4 * lq;
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                f32x4 v = acc[ct];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                *reinterpret_cast<f32x4*>(dst + ct * 16) = v;
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// Heads: one 256-thread workgroup per HB = 2 boards, so that every Linear weight is read once
// per HB boards instead of once per board (the weights, 136 KB at 9x9, come from L2).  HB = 2
// measured 16 us at 1024 boards of 9x9 against 45 us for HB = 4: the kernel is latency-bound
// and more workgroups in flight hide it better than fewer weight reads.
//   1x1 convs (+ folded BN, ReLU): NC = 3 channels for the actor-critic net (act 0, act 1,
//   crit) or 1 for the value net; then Linear(2P -> A) and Linear(P -> 256) ReLU Linear(256 -> 1).
// 1x1 convs: a work item is (board, cell, 16-channel segment); the 8 segments of a cell sit in
// 8 adjacent lanes (one 512-byte row of the activation tensor per 8 lanes) and are combined
// by three xor-shuffles.
// --------------------------------------------------------------------------------------
constexpr int HB = 2;

template <int S>
__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ act, HeadWeights hw, int is_ac,
                                               float* __restrict__ pi_out, float* __restrict__ val_out, int nboards)
{
    using G = Geom<S>;
    constexpr int JG = (256 / G::A) > 0 ? (256 / G::A) : 1;       // input slices of the actor Linear
    constexpr int JLEN = (2 * G::P + JG - 1) / JG;
    __shared__ float h[HB][3 * G::P];  // [board][channel][cell]: act0, act1, crit (value net: crit only)
    __shared__ float part[JG][HB][G::A];
    __shared__ float z[HB][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = tid & 7;
    const int c_first = is_ac ? 0 : 2;
    f32x4 w[3][4];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int i = 0; i < 4; i++)
            w[c][i] = (c >= c_first) ? *reinterpret_cast<const f32x4*>(&hw.conv_w[c * 128 + seg * 16 + i * 4])
                                     : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    const float cb[3] = { hw.conv_b[0], hw.conv_b[1], hw.conv_b[2] };

    for (int b0 = blockIdx.x * HB; b0 < nboards; b0 += gridDim.x * HB) {
        const int nb = min(HB, nboards - b0);
        __syncthreads();
        // ---- 1x1 convolutions -> h ----
        const int items = nb * G::P * 8;
        for (int it0 = 0; it0 < items; it0 += 256) {
            const int it = it0 + tid;                 // items is a multiple of 8: a cell's 8 lanes are all in or all out
            const bool on = it < items;
            const int cellidx = on ? (it >> 3) : 0;   // = board * P + cell
            const float* src = act + ((size_t)b0 * G::P + cellidx) * 128 + seg * 16;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
            if (on) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(src + i * 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        s0 = fmaf(x[e], w[0][i][e], s0);
                        s1 = fmaf(x[e], w[1][i][e], s1);
                        s2 = fmaf(x[e], w[2][i][e], s2);
                    }
                }
            }
#pragma unroll
            for (int off = 4; off >= 1; off >>= 1) {
                s0 += __shfl_xor(s0, off);
                s1 += __shfl_xor(s1, off);
                s2 += __shfl_xor(s2, off);
            }
            if (on && seg == 0) {
                const int b = cellidx / G::P, p = cellidx - b * G::P;
                if (is_ac) {
                    h[b][p] = fmaxf(s0 + cb[0], 0.0f);
                    h[b][G::P + p] = fmaxf(s1 + cb[1], 0.0f);
                }
                h[b][2 * G::P + p] = fmaxf(s2 + cb[2], 0.0f);
            }
        }
        __syncthreads();
        // ---- actor: Linear(2P -> A): a thread owns (output a, slice jg of the 2P inputs) for all
        // boards of the group, so that every thread has work and the weight loads of a slice
        // are independent of one another; the JG partial sums are combined through LDS ----
        if (is_ac) {
            for (int o = tid; o < JG * G::A; o += 256) {
                const int jg = o / G::A, a = o - jg * G::A;
                const int j0 = jg * JLEN, j1 = min(2 * G::P, j0 + JLEN);
                float sum[HB];
#pragma unroll
                for (int b = 0; b < HB; b++) sum[b] = 0.0f;
#pragma unroll 6
                for (int j = j0; j < j1; j++) {
                    const float wv = hw.act_lin_wt[(size_t)j * G::A + a];
#pragma unroll
                    for (int b = 0; b < HB; b++) sum[b] = fmaf(wv, h[b][j], sum[b]);
                }
#pragma unroll
                for (int b = 0; b < HB; b++) part[jg][b][a] = sum[b];
            }
            __syncthreads();
            for (int o = tid; o < nb * G::A; o += 256) {
                const int b = o / G::A, a = o - b * G::A;
                float sum = hw.act_lin_b[a];
#pragma unroll
                for (int jg = 0; jg < JG; jg++) sum += part[jg][b][a];
                pi_out[(size_t)(b0 + b) * G::A + a] = sum;
            }
        }
        // ---- critic: Linear(P -> 256) ReLU Linear(256 -> 1) ----
        {
            float sum[HB];
            const float bias = hw.crit_b1[tid];
#pragma unroll
            for (int b = 0; b < HB; b++) sum[b] = bias;
#pragma unroll 9
            for (int p = 0; p < G::P; p++) {
                const float wv = hw.crit_w1t[p * 256 + tid];
#pragma unroll
                for (int b = 0; b < HB; b++) sum[b] = fmaf(wv, h[b][2 * G::P + p], sum[b]);
            }
            const float w2 = hw.crit_w2[tid];
#pragma unroll
            for (int b = 0; b < HB; b++) {
                float v = fmaxf(sum[b], 0.0f) * w2;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0) z[b][wave] = v;
            }
        }
        __syncthreads();
        if (tid < nb) val_out[b0 + tid] = ((z[tid][0] + z[tid][1]) + (z[tid][2] + z[tid][3])) + hw.crit_b2[0];
    }
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
        const int grid = (nboards + NB0 - 1) / NB0;
        GOAI_DISPATCH_SIZE(S_rt, k_conv0_mfma<S><<<dim3(grid), dim3(256), 0, st>>>(
                                     states, (const uint4*)w0_split, b0, shift, out, nboards));
        GOAI_HIP_CHECK(hipGetLastError());
        return GOAI_OK;
    }
    int grid = nboards < 4096 ? nboards : 4096;
    GOAI_DISPATCH_SIZE(S_rt, k_conv0<S><<<dim3(grid), dim3(256), 0, st>>>(states, w0, b0, shift, out, nboards));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_heads(int S_rt, const float* act, const HeadWeights& hw, int is_ac, float* pi, float* val, int nboards,
                 hipStream_t st)
{
    int groups = (nboards + HB - 1) / HB;
    int grid = groups < 8192 ? groups : 8192;
    GOAI_DISPATCH_SIZE(S_rt, k_heads<S><<<dim3(grid), dim3(256), 0, st>>>(act, hw, is_ac, pi, val, nboards));
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
