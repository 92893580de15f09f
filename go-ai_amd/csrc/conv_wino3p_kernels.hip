// conv_wino3p_kernels.hip -- the F(3x3,3x3) trunk of conv_wino3_kernels.hip for contexts whose batches
// always fit one round of the chip at 4 boards per workgroup (max_batch <= 4 * CUs; the bench's
// 1024 games on 256 CUs), with TWO 32-channel chunks per output transform.
//
// Why: on a SIMD, v_mfma_f32_16x16x4_f32 and ordinary vector instructions do not overlap
// (tools/ubench/mfma_valu*.hip), so the output transform's v_fma (Y += At[i][p] At[j][q] M, once
// per transform position AND chunk in conv_wino3_kernels.hip) costs its full issue time.  Here M is
// accumulated over two chunks (K = 64) inside the MFMA accumulator first: 50 output transforms per
// convolution instead of 100.  That needs both chunks' images and Bt d B rows in LDS at once:
//   [zero cell 128 B][image chunk A 41,472 B] [zero cell][image chunk B]   (un-swizzled: cell * 128 B)
//   V[2 chunks][5 q][48 rows][32 ch]  single-buffered (a double buffer buys nothing when matrix and
//                                     vector work cannot overlap)
//   Tab[36 wtiles][5 patch rows] -> five 16-bit cell addresses (chunk A; chunk B = + 41,600 as the
//                                     ds_read immediate offset); lanes add their channel slot
// = 147 KB.  Everything else (whole boards per workgroup, weights L2 -> registers one half-stage ahead,
// transposed products, the whole residual trunk in one launch) is as in conv_wino3_kernels.hip.
// The summation order differs from that kernel (64 channels per accumulator instead of 32), so a
// context uses one or the other for its whole life (nn_host.hip).
#include <type_traits>
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S>
struct Wino3P {
    static_assert(S == 9, "row mapping: 8 + 1 wtiles per board, 4 boards");
    static constexpr int T = S / 3, T2 = T * T, P = S * S;
    static constexpr int NBW = 4;                                  // whole boards per workgroup
    static constexpr int WT = NBW * T2;                            // real wtiles per workgroup
    static constexpr int MT = (WT + 15) / 16, ROWS = MT * 16;      // MFMA row tiles
    static constexpr int IMG_CH_U4 = 8 + NBW * P * 8;              // 16-byte units per chunk region: zero cell + cells
    static constexpr int V_CH_U4 = 5 * ROWS * 8;                   // [5 q][ROWS][32 ch] of one chunk
    static constexpr int TAB_U4 = WT * 5;                          // [wtile][a] -> 8 x u16 (5 used)
    static constexpr int NI = (NBW * P * 8 + 511) / 512;
    static constexpr int LDS_U4 = 2 * IMG_CH_U4 + 2 * V_CH_U4 + TAB_U4;
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(IMG_CH_U4 * 16 < 65536, "cell addresses and the chunk offset are 16-bit");
};

#ifdef GOAI_CONV_STAMPS
// diagnostic build only (tools/wino3p_stamps.py): s_memrealtime (100 MHz) stamps of the SECOND convolution of
// a launch, 32 per workgroup: 0 layer top, 1 image staged, 2 + 2r transform of row r done (r = 5 cp + p),
// 3 + 2r its stages done, 22 epilogue done; 23 / 24 = s_memtime (shader clock) at stamps 0 / 22
__device__ unsigned long long g_wino3p_stamps[32 * 1024];
extern "C" int goai_debug_wino3p_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino3p_stamps), (size_t)n * sizeof(unsigned long long));
}
#define W3P_STAMP(i) if (threadIdx.x == 0 && k == 1) { g_wino3p_stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0 || (i) == 22) g_wino3p_stamps[blockIdx.x * 32 + 23 + ((i) == 22)] = __builtin_amdgcn_s_memtime(); }
#else
#define W3P_STAMP(i)
#endif

__device__ __forceinline__ f32x4 p_fma4(float c, f32x4 a, f32x4 b)
{
    return f32x4{ fmaf(c, a[0], b[0]), fmaf(c, a[1], b[1]), fmaf(c, a[2], b[2]), fmaf(c, a[3], b[3]) };
}
__device__ __forceinline__ int p_swz(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

// Same launch contract as k_conv3x3_wino3 (n_convs convolutions in a row on the same boards).
template <int S>
__global__ __launch_bounds__(512) void k_conv3x3_wino3p(float* x, float* y, const float* resid0,
                                                        const float* __restrict__ wt_all,
                                                        const float* __restrict__ bias_all,
                                                        const uint4* __restrict__ tab_g, int nboards, int n_convs)
{
    using W = Wino3P<S>;
    constexpr int P = W::P, MT = W::MT, ROWS = W::ROWS;
    __shared__ f32x4 lds[W::LDS_U4];
    f32x4* Img = lds;                                  // two chunk regions
    f32x4* Vs = lds + 2 * W::IMG_CH_U4;
    uint4* Tab = reinterpret_cast<uint4*>(lds + 2 * W::IMG_CH_U4 + 2 * W::V_CH_U4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * W::NBW;
    const float* in = x;
    const float* resid = resid0;
    float* out = y;

    for (int i = tid; i < 2 * W::V_CH_U4; i += 512) Vs[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };   // rows >= WT stay zero
    if (tid < 16) Img[(tid >> 3) * W::IMG_CH_U4 + (tid & 7)] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };   // the two zero cells
    for (int e = tid; e < W::TAB_U4; e += 512) Tab[e] = tab_g[e];

    // input-transform job of this thread: WT * 8 (wtile, channel slot) items spread evenly over the
    // eight waves, lanes 0..LPG-1 of every 16 take one; each item is done for both chunks
    constexpr int LPG = (W::WT * 8 + 31) / 32;
    static_assert(LPG <= 16, "too many wtiles for one transform pass");
    const int item = (tid >> 4) * LPG + (tid & 15);
    const bool t_on = (tid & 15) < LPG && item < W::WT * 8;
    const int ts = item & 7, ttw = t_on ? item >> 3 : 0;
    // MFMA row of a wtile: wtiles 0..7 of board b -> row 8 b + wtile (two 16-row tiles, v_mfma_f32_16x16x4_f32);
    // wtile 8 of board b -> row 32 + b (v_mfma_f32_4x4x1_16b_f32, below).  Every board is treated alike.
    const int tvrow = (ttw % W::T2 < 8) ? (ttw / W::T2) * 8 + ttw % W::T2 : 32 + ttw / W::T2;

    f32x4 ireg[W::NI];
    auto load_img_to = [&](int chunk, f32x4* reg) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            const int cr = idx >> 3, slot = idx & 7;
            const int board = b_first + cr / P;
            f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (idx < W::NBW * P * 8 && board < nboards)
                v = *reinterpret_cast<const f32x4*>(&in[((long)b_first * P + cr) * 128 + chunk * 32 + slot * 4]);
            reg[i] = v;
        }
    };
    auto store_img_from = [&](int region, const f32x4* reg) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            if (idx < W::NBW * P * 8) Img[region * W::IMG_CH_U4 + 8 + idx] = reg[i];
        }
    };
    auto load_img = [&](int chunk) { load_img_to(chunk, ireg); };
    auto store_img = [&](int region) { store_img_from(region, ireg); };
    const char* img_lane = reinterpret_cast<const char*>(Img) + ts * 16;
    const uint4* tabrow = Tab + ttw * 5;
    // V[c][q][tw][ts] = (Bt d B)[p][q], q = 0..4, for row p (compile-time after unrolling), chunk region c
    auto transform = [&](int p, int c) {
        if (!t_on) return;
        const char* base = img_lane + c * (W::IMG_CH_U4 * 16);
        auto patch_row = [&](int a, f32x4* d) {
            const uint4 t = tabrow[a];
            d[0] = *reinterpret_cast<const f32x4*>(base + (t.x & 0xffffu));
            d[1] = *reinterpret_cast<const f32x4*>(base + (t.x >> 16));
            d[2] = *reinterpret_cast<const f32x4*>(base + (t.y & 0xffffu));
            d[3] = *reinterpret_cast<const f32x4*>(base + (t.y >> 16));
            d[4] = *reinterpret_cast<const f32x4*>(base + (t.z & 0xffffu));
        };
        f32x4 r[5], d1[5], d2[5], d3[5];
        patch_row(1, d1);
        if (p != 3) patch_row(2, d2);
        patch_row(3, d3);
        if (p == 0) {
            f32x4 d0[5];
            patch_row(0, d0);
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = p_fma4(-2.0f, d2[b], p_fma4(2.0f, d0[b], d3[b] - d1[b]));
        } else if (p == 1) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = p_fma4(-2.0f, d1[b], d3[b] - d2[b]);
        } else if (p == 2) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = p_fma4(-3.0f, d2[b], p_fma4(2.0f, d1[b], d3[b]));
        } else if (p == 3) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = d3[b] - d1[b];
        } else {
            f32x4 d4[5];
            patch_row(4, d4);
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = p_fma4(-2.0f, d3[b], p_fma4(2.0f, d1[b], d4[b] - d2[b]));
        }
        f32x4* V = Vs + c * W::V_CH_U4;
        V[p_swz(0 * ROWS + tvrow, ts)] = p_fma4(-2.0f, r[2], p_fma4(2.0f, r[0], r[3] - r[1]));
        V[p_swz(1 * ROWS + tvrow, ts)] = p_fma4(-2.0f, r[1], r[3] - r[2]);
        V[p_swz(2 * ROWS + tvrow, ts)] = p_fma4(-3.0f, r[2], p_fma4(2.0f, r[1], r[3]));
        V[p_swz(3 * ROWS + tvrow, ts)] = r[3] - r[1];
        V[p_swz(4 * ROWS + tvrow, ts)] = p_fma4(-2.0f, r[3], p_fma4(2.0f, r[1], r[4] - r[2]));
    };

    // weight fragments of a half-stage: U[stage = (chunk * 5 + p) * 5 + q] in the lanes' own order (nn_host.hip):
    // lane (li, lq) gets channel cout = 16 wave + li, inputs k = 16 kk + 4 lq + jj
    const float* wsrc = wt_all + wave * 512 + lane * 4;   // [stage][wave][kk][lane][4] (nn_host.hip): whole lines per load
    f32x4 bcur[2], bnxt[2];
    auto load_b = [&](int stage, f32x4* b) {
        const float* src = wsrc + (size_t)stage * 128 * 32;
        b[0] = *reinterpret_cast<const f32x4*>(src);
        b[1] = *reinterpret_cast<const f32x4*>(src + 256);
    };

    f32x4 Y[3][3][MT];                               // [i][j]: output (3ty + i, 3tx + j) of the wtile
    const int col = wave * 16 + 4 * lq;

#pragma unroll 1
    for (int k = 0; k < n_convs; k++) {
    const float* bias = bias_all + k * 128;
    load_b(0, bcur);
    // k = 0: the zero fills and the table are complete; k > 0: every wave's MFMAs of the previous
    // convolution have read V and its stores to `in` are visible to this workgroup
    __syncthreads();
    W3P_STAMP(0);
    {
        f32x4 ireg1[W::NI];                          // both chunks in flight at once: the accumulators are not live yet
        load_img(0);
        load_img_to(1, ireg1);
        store_img(0);
        store_img_from(1, ireg1);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int t = 0; t < MT; t++) Y[i][j][t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    W3P_STAMP(1);

#pragma unroll 1
    for (int cp = 0; cp < 2; cp++) {
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const bool stage_next = (p == 4 && cp == 0);       // chunks 2, 3 replace 0, 1 behind the last transform
            if (stage_next) load_img(2);
            transform(p, 0);
            transform(p, 1);
            __syncthreads();                                   // V complete; (p == 4) the image is dead
            W3P_STAMP(2 + 2 * (5 * cp + p));
            if (stage_next) {
                store_img(0);
                load_img(3);
            }
#pragma unroll 1
            for (int q = 0; q < 5; q++) {
                f32x4 m[MT];
#pragma unroll
                for (int t = 0; t < MT; t++) m[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
                f32x4 m2b = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    // the next half-stage's weights: (q, c = 1), else (q + 1, 0), else the next row / chunk pair
                    int nxt;
                    if (c == 0) nxt = ((2 * cp + 1) * 5 + p) * 5 + q;
                    else if (q < 4) nxt = ((2 * cp) * 5 + p) * 5 + q + 1;
                    else if (p < 4) nxt = ((2 * cp) * 5 + p + 1) * 5;
                    else nxt = (2 * (cp + 1)) * 25;
                    if (nxt < 100) load_b(nxt, bnxt);
                    const f32x4* V = Vs + c * W::V_CH_U4;
#pragma unroll
                    for (int kk = 0; kk < 2; kk++) {
                        f32x4 a[MT];
#pragma unroll
                        for (int t = 0; t < 2; t++) a[t] = V[p_swz(q * ROWS + t * 16 + li, kk * 4 + lq)];
                        a[2] = V[p_swz(q * ROWS + 32 + (li & 3), kk * 4 + lq)];
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
#pragma unroll
                            for (int t = 0; t < 2; t++)
                                m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[kk][jj], a[t][jj], m[t], 0, 0, 0);
                            // rows 32..35 (4 wtiles x this wave's 16 output channels): sixteen 4x4 blocks,
                            // block = lane / 4 = (k slice lq, channel group li / 4); the lane's weight register
                            // IS the block's A element (channel li, k = 16 kk + 4 lq + jj), B = V[32 + li % 4].
                            // m[2] = partial sums over this lane's k slice; the four slices meet in the epilogue.
                            // (two accumulators, even / odd jj: back-to-back dependent 4x4 MFMAs would need wait states)
                            if (jj & 1) m2b = __builtin_amdgcn_mfma_f32_4x4x1f32(bcur[kk][jj], a[2][jj], m2b, 0, 0, 0);
                            else m[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(bcur[kk][jj], a[2][jj], m[2], 0, 0, 0);
                        }
                    }
                    bcur[0] = bnxt[0];
                    bcur[1] = bnxt[1];
                }
                m[2] += m2b;
                // output transform: Y[i][j] += At[i][p] * At[j][q] * M; p is unrolled (row factors at
                // compile time, zeros skipped), q is a run-time loop (column factors as floats)
                const float cj0 = (q < 4) ? 1.0f : 0.0f;
                const float cj1 = (q == 1) ? 1.0f : (q == 2 ? -1.0f : (q == 3 ? 2.0f : 0.0f));
                const float cj2 = (q == 0) ? 0.0f : (q == 3 ? 4.0f : 1.0f);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int ci = (i == 0) ? (p < 4 ? 1 : 0)
                                 : (i == 1) ? (p == 1 ? 1 : (p == 2 ? -1 : (p == 3 ? 2 : 0)))
                                            : (p == 0 ? 0 : (p == 3 ? 4 : 1));
                    if (ci != 0) {
                        const float f0 = (float)ci * cj0, f1 = (float)ci * cj1, f2 = (float)ci * cj2;
#pragma unroll
                        for (int t = 0; t < MT; t++)
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                Y[i][0][t][e] = fmaf(f0, m[t][e], Y[i][0][t][e]);
                                Y[i][1][t][e] = fmaf(f1, m[t][e], Y[i][1][t][e]);
                                Y[i][2][t][e] = fmaf(f2, m[t][e], Y[i][2][t][e]);
                            }
                    }
                }
            }
            __syncthreads();                                   // every wave has read this row of V
            W3P_STAMP(3 + 2 * (5 * cp + p));
            if (stage_next) {
                store_img(1);
                __syncthreads();
            }
        }
    }

    // epilogue.  The products are formed transposed (A operand = weights, B operand = V), so in the
    // C/D layout of a 16x16 tile (col = lane & 15, row = 4 * (lane >> 4) + reg) a lane holds FOUR
    // CONSECUTIVE output channels of ONE wtile: one 16-byte load / store per output cell.  In the 4x4
    // blocks a lane holds channels 4 (li / 4) .. + 3 of wtile row 32 + li % 4 for its k slice: the four
    // slices (lanes l, l ^ 16, l ^ 32, l ^ 48) are summed in a fixed order, lanes 0..15 store.
    // `out` may alias `resid` (the trunk's in-place convolutions), so loads and stores stay in program
    // order: load -> add -> store per cell is 27 dependent round trips (12 us of a 110 us layer at any
    // grid size, tools/wino3p_stamps.py).  The nine residual loads of a row tile are issued together,
    // and the residual / no-residual choice is made once, not per cell.
    auto epilogue = [&](auto has_resid) {
        constexpr bool R = decltype(has_resid)::value;
        auto tile = [&](const f32x4 (&yv)[3][3], const f32x4 bv, long base, bool on) {
            // base = element offset of output (0, 0) of the wtile, this lane's channels
            if (!on) return;
            f32x4 rv[3][3];
            if (R) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) rv[i][j] = *reinterpret_cast<const f32x4*>(&resid[base + (i * S + j) * 128]);
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    f32x4 v = yv[i][j] + bv;
                    if (R) v += rv[i][j];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                    *reinterpret_cast<f32x4*>(&out[base + (i * S + j) * 128]) = v;
                }
        };
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[col]);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int w = t * 16 + li;
            const int bl = w >> 3, wt = w & 7;
            const long board = b_first + bl;
            const int ty = wt / W::T, tx = wt - ty * W::T;
            f32x4 yv[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) yv[i][j] = Y[i][j][t];
            tile(yv, bv, (board * P + 3 * ty * S + 3 * tx) * 128 + col, board < nboards);
        }
        const int col2 = wave * 16 + (li & 12);
        const f32x4 bv2 = *reinterpret_cast<const f32x4*>(&bias[col2]);
        const long board2 = b_first + (li & 3);
        f32x4 yv[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                f32x4 v = Y[i][j][2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] += __shfl_xor(v[e], 16);
                    v[e] += __shfl_xor(v[e], 32);
                }
                yv[i][j] = v;
            }
        tile(yv, bv2, (board2 * P + 6 * S + 6) * 128 + col2, lq == 0 && board2 < nboards);   // wtile 8 = (ty, tx) = (2, 2)
    };
    if (resid) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    W3P_STAMP(22);
    // next convolution of the launch: swap the buffers, step the weights
    if (k & 1) { in = x; out = y; resid = resid0; } else { in = y; out = x; resid = x; }
    wsrc += (size_t)25 * 128 * 128;
    }
}

int wino3p_supported(int S_rt, int max_batch, int n_cu) { return S_rt == 9 && (max_batch + 3) / 4 <= n_cu; }

// Host side of the input-transform address table: out[(tw * 5 + a) * 8 + b], b < 5 = byte address of
// cell (3ty - 1 + a, 3tx - 1 + b) of wtile tw's patch inside a chunk region (the zero cell at 0, cell c
// at 128 + 128 c); returns the number of uint16 entries.
size_t wino3p_build_table(int S_rt, uint16_t* out)
{
    const int S = S_rt, T = S / 3, T2 = T * T, P = S * S, WT = 4 * T2;
    if (out)
        for (int tw = 0; tw < WT; tw++)
            for (int a = 0; a < 5; a++)
                for (int b = 0; b < 8; b++) {
                    const int bl = tw / T2, rem = tw % T2, ty = rem / T, tx = rem % T;
                    const int yy = 3 * ty - 1 + a, xx = 3 * tx - 1 + b;
                    const bool ok = b < 5 && yy >= 0 && yy < S && xx >= 0 && xx < S;
                    out[((size_t)tw * 5 + a) * 8 + b] = (uint16_t)(ok ? 128 + (bl * P + yy * S + xx) * 128 : 0);
                }
    return (size_t)WT * 5 * 8;
}

int launch_conv3x3_wino3p(int S_rt, float* x, float* y, const float* resid0, const float* wt_wino3, const float* bias,
                          const void* tab, int nboards, int n_cu, int n_convs, hipStream_t st)
{
    if (S_rt != 9 || (nboards + 3) / 4 > n_cu) {
        set_error("the paired F(3x3,3x3) convolution runs 9x9 batches of at most 4 boards per CU");
        return GOAI_ERR_ARG;
    }
    if (n_convs < 1 || (n_convs > 1 && ((n_convs & 1) || resid0))) {
        set_error("a multi-convolution launch is a whole residual trunk: an even count, no outer residual");
        return GOAI_ERR_ARG;
    }
    k_conv3x3_wino3p<9><<<dim3((nboards + 3) / 4), dim3(512), 0, st>>>(x, y, resid0, wt_wino3, bias,
                                                                       reinterpret_cast<const uint4*>(tab), nboards, n_convs);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
