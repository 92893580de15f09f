// nn_groups.h -- the two small network kernels around the trunk as workgroup-level device functions, so that a
// kernel which does more than one of them for its boards (k_post_forward, mcts_kernels.hip) runs the very same code
// as k_conv0_mfma / k_heads (nn_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "go_board.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// (large batches: k_conv0_mfma<S, 8> -- a workgroup's weights are 55 KB from L2, once per NBT boards)

template <int S, int NBT = NB0>
struct Conv0Lds {
    uint4 xin[NBT * Geom<S>::NPAD + 1];                            // [board][padded cell] 8 bf16; last = zero cell
};

// boards b_first .. b_first + NBT - 1 by one 256-thread workgroup (all of its threads must call this); tid_in >= 0: by the 256
// threads that pass 0..255 here -- one half of k_search_steps' 512-thread workgroup, whose other half works on another `lds`
// (the barriers inside are the whole workgroup's: both halves make the same calls)
template <int S, int NBT = NB0>
__device__ __forceinline__ void conv0_group(Conv0Lds<S, NBT>& lds, const uint8_t* __restrict__ states, const uint4* __restrict__ w0s,
                                            const float* __restrict__ b0, float shift, float* __restrict__ out, int nboards,
                                            int b_first, int tid_in = -1)
{
    using G = Geom<S>;
    constexpr int CELLS = NBT * G::P, TILES = (CELLS + 15) / 16;
    uint4* xin = lds.xin;
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;

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

    for (int i = tid; i < NBT * G::NPAD + 1; i += 256) xin[i] = uint4{ 0u, 0u, 0u, 0u };
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
            const int qq = (on && tap < 9) ? q + (tap / 3 - 1) * G::PS + (tap % 3 - 1) : NBT * G::NPAD;
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
// (large batches of large boards: k_heads<S, 8> -- at 19x19 the two Linear layers' weights are 1.05 MB per workgroup, and 2048
// workgroups of two boards re-read them from L2 for 270 us per 4096 boards; same sums per board in the same order, same bits)

template <int S, int HBT = HB>
struct HeadsLds {
    static constexpr int JG = (256 / Geom<S>::A) > 0 ? (256 / Geom<S>::A) : 1;       // input slices of the actor Linear
    float h[HBT][3 * Geom<S>::P];  // [board][channel][cell]: act0, act1, crit (value net: crit only)
    float part[JG][HBT][Geom<S>::A];
    float z[HBT][4];
};

// groups of HB boards starting at b_begin, then every b_stride boards, by one 256-thread workgroup (all threads call it;
// tid_in as for conv0_group).
// act_is_h: `act` holds the 1x1 convolutions already, [board][3][P] (the split F(3x3,3x3) trunk forms them in its last
// epilogue), and they are only copied into LDS.
template <int S, int HBT = HB>
__device__ __forceinline__ void heads_groups(HeadsLds<S, HBT>& lds, const float* __restrict__ act, const HeadWeights& hw, int is_ac,
                                             float* __restrict__ pi_out, float* __restrict__ val_out, int nboards, int b_begin,
                                             int b_stride, bool act_is_h = false, int tid_in = -1)
{
    using G = Geom<S>;
    constexpr int JG = HeadsLds<S, HBT>::JG;
    constexpr int JLEN = (2 * G::P + JG - 1) / JG;
    auto& h = lds.h;
    auto& part = lds.part;
    auto& z = lds.z;
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    for (int b0 = b_begin; b0 < nboards; b0 += b_stride) {
        const int nb = min(HBT, nboards - b0);
        __syncthreads();
        // ---- 1x1 convolutions -> h ----
        // All of a thread's activation loads (64 B per item, up to NLD items in flight) are issued before the first
        // multiply: issued inside the loop they were one L2 round trip per iteration (6 of them at 9x9).
        const int items = act_is_h ? 0 : nb * G::P * 8;
        if (act_is_h)
            for (int i = tid; i < nb * 3 * G::P; i += 256) (&h[0][0])[i] = act[(size_t)b0 * 3 * G::P + i];
        constexpr int NIT = (HBT * G::P * 8 + 255) / 256, NLD = NIT < 6 ? NIT : 6;
        for (int it0 = 0; it0 < items; it0 += 256 * NLD) {
            f32x4 xs[NLD][4];
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const int it = it0 + u * 256 + tid;       // items is a multiple of 8: a cell's 8 lanes are all in or all out
                const float* src = act + ((size_t)b0 * G::P + (it < items ? (it >> 3) : 0)) * 128 + seg * 16;
#pragma unroll
                for (int i = 0; i < 4; i++) xs[u][i] = *reinterpret_cast<const f32x4*>(src + i * 4);
            }
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const int it = it0 + u * 256 + tid;
                const bool on = it < items;
                const int cellidx = on ? (it >> 3) : 0;   // = board * P + cell
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        s0 = fmaf(xs[u][i][e], w[0][i][e], s0);
                        s1 = fmaf(xs[u][i][e], w[1][i][e], s1);
                        s2 = fmaf(xs[u][i][e], w[2][i][e], s2);
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
        }
        __syncthreads();
        // ---- actor: Linear(2P -> A): a thread owns (output a, slice jg of the 2P inputs) for all
        // boards of the group, so that every thread has work and the weight loads of a slice
        // are independent of one another; the JG partial sums are combined through LDS ----
        if (is_ac) {
            for (int o = tid; o < JG * G::A; o += 256) {
                const int jg = o / G::A, a = o - jg * G::A;
                const int j0 = jg * JLEN, j1 = min(2 * G::P, j0 + JLEN);
                float sum[HBT];
#pragma unroll
                for (int b = 0; b < HBT; b++) sum[b] = 0.0f;
#pragma unroll 18
                for (int j = j0; j < j1; j++) {
                    const float wv = hw.act_lin_wt[(size_t)j * G::A + a];
#pragma unroll
                    for (int b = 0; b < HBT; b++) sum[b] = fmaf(wv, h[b][j], sum[b]);
                }
#pragma unroll
                for (int b = 0; b < HBT; b++) part[jg][b][a] = sum[b];
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
            float sum[HBT];
            const float bias = hw.crit_b1[tid];
#pragma unroll
            for (int b = 0; b < HBT; b++) sum[b] = bias;
#pragma unroll 27
            for (int p = 0; p < G::P; p++) {
                const float wv = hw.crit_w1t[p * 256 + tid];
#pragma unroll
                for (int b = 0; b < HBT; b++) sum[b] = fmaf(wv, h[b][2 * G::P + p], sum[b]);
            }
            const float w2 = hw.crit_w2[tid];
#pragma unroll
            for (int b = 0; b < HBT; b++) {
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

}  // namespace goai
