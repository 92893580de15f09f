// conv_wino3_kernels.hip -- the trunk's conv3x3 128 -> 128 (+ folded BatchNorm bias, optional
// residual, ReLU; /root/reference/go_ai/models/__init__.py:234-251) in fp32 by Winograd's
// minimal filtering F(3x3, 3x3) on the fp32 matrix cores, for boards whose side is a multiple
// of 3 (9x9: three tiles of three, no tile padding at all).
//
//   Y = At [ (G g Gt) . (Bt d B) ] A,   interpolation points 0, 1, -1, 2, inf  (Lavin & Gray 2016)
//   At = (1 1  1 1 0)   Bt = (2 -1 -2  1 0)   G = ( 1/2    0    0 )
//        (0 1 -1 2 0)        (0 -2 -1  1 0)       (-1/2 -1/2 -1/2 )
//        (0 1  1 4 1)        (0  2 -3  1 0)       (-1/6  1/6 -1/6 )
//                            (0 -1  0  1 0)       ( 1/6  1/3  2/3 )
//                            (0  2 -1 -2 1)       (  0    0    1  )
// A 3x3 block of outputs ("wtile") needs 25 multiplications per (cin, cout) instead of 81: 3.24x
// fewer MFMA operations than the direct implicit GEMM, 1.78x fewer than F(2x2,3x3) on a 9x9 board
// (whose 2x2 tiles cover 10x10).  fp32 throughout; G g Gt is folded into the weights on the host in
// fp64.  The larger transform constants cost accuracy: ~2e-6 relative per layer (F(2x2,3x3): ~1e-6),
// far inside the 1e-3 bound (tests/test_gpu_nn.py).
//
// Structure: as conv_wino_kernels.hip (whole boards per workgroup, one workgroup per CU, weights
// L2 -> registers, double-buffered V row, table-driven input transform), with
//   - NBW = 4 boards = 36 wtiles in MT = 3 MFMA row tiles (48 rows) per 512-thread workgroup,
//   - 5 x 5 transform positions: 4 chunks x 5 rows p x 5 columns q = 100 stages of 24 MFMAs per wave,
//   - nine output accumulator sets + one M in flight per wave (120 registers).
#include <type_traits>
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S, int NBW_>
struct Wino3 {
    static_assert(S % 3 == 0, "F(3x3,3x3) tiles must cover the board exactly");
    static constexpr int T = S / 3, T2 = T * T, P = S * S;
    static constexpr int NBW = NBW_;                               // whole boards per workgroup (4 or 5 at 9x9)
    static constexpr int WT = NBW * T2;                            // real wtiles per workgroup
    static constexpr int MT = (WT + 15) / 16, ROWS = MT * 16;      // MFMA row tiles
    static constexpr int IMG_U4 = NBW * P * 8 + 8;                 // 16-byte units (+ the zero cell)
    static constexpr int ZERO_U4 = NBW * P * 8;
    static constexpr int V_U4 = 5 * ROWS * 8;                      // one buffer: [5 q][ROWS][32 ch]
    static constexpr int TAB_U4 = WT * 8 * 5;                      // [wtile][slot][a] -> 8 x u16 (5 used): final LDS byte addresses
    static constexpr int NI = (NBW * P * 8 + 511) / 512;
    static constexpr int LDS_U4 = IMG_U4 + 2 * V_U4 + TAB_U4;
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(IMG_U4 * 16 < 65536, "input-transform addresses are 16-bit");
};

#ifdef GOAI_CONV_STAMPS
__device__ unsigned long long g_wino3_stamps[8 * 4096];   // diagnostic build only (tools/wino_stamps.py)
extern "C" int goai_debug_wino3_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino3_stamps), (size_t)n * sizeof(unsigned long long));
}
#define W3_STAMP(i) if (threadIdx.x == 0) { g_wino3_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 1 || (i) == 4) g_wino3_stamps[blockIdx.x * 8 + 6 + ((i) == 4)] = __builtin_amdgcn_s_memtime(); }
#else
#define W3_STAMP(i)
#endif

// Diagnostic builds only (tools/wino3_ablate.sh): bit mask of loop parts that are LEFT OUT, to see what
// each costs.  1: output transform, 2: input transform, 4: A-fragment LDS reads, 8: weight loads,
// 16: row barrier, 32: the MFMAs.  Any non-zero value computes wrong results.
#ifndef W3_DIAG
#define W3_DIAG 0
#endif

// c * a + b per element as ONE fused operation (the file is compiled with -ffp-contract=off, which would
// leave the transforms' 2x, 3x and 4x as separate multiplies; vector instructions cost MFMA time here)
__device__ __forceinline__ f32x4 fma4(float c, f32x4 a, f32x4 b)
{
    return f32x4{ fmaf(c, a[0], b[0]), fmaf(c, a[1], b[1]), fmaf(c, a[2], b[2]), fmaf(c, a[3], b[3]) };
}

__device__ __forceinline__ int w3swz(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

// One launch runs `n_convs` convolutions in a row on the same boards (a workgroup owns whole boards
// and all 128 output channels, so consecutive layers need no exchange between workgroups):
//   conv k reads  (k even ? x : y),  writes (k even ? y : x),  adds the residual (k even ? resid0 : x)
// i.e. n_convs = 1 is a single convolution x -> y (+ resid0, may be NULL), and an even n_convs with
// resid0 = NULL is the whole residual trunk of n_convs / 2 BasicBlocks (models/__init__.py:234-251),
// ending in x; the odd convolutions add and overwrite x in place (each lane reads the 16 bytes it is
// about to write).  Layer k's weights / bias are at wt + k * 25 * 128 * 128 / bias + k * 128.
// x and y are written and re-read inside the launch: no __restrict__ on them.
template <int S, int NBW>
__global__ __launch_bounds__(512) void k_conv3x3_wino3(float* x, float* y, const float* resid0,
                                                       const float* __restrict__ wt_all,
                                                       const float* __restrict__ bias_all,
                                                       const uint4* __restrict__ tab_g, int nboards, int n_convs)
{
    using W = Wino3<S, NBW>;
    constexpr int P = W::P, MT = W::MT, ROWS = W::ROWS;
    __shared__ f32x4 lds[W::LDS_U4];
    f32x4* Img = lds;
    f32x4* Vs = lds + W::IMG_U4;
    uint4* Tab = reinterpret_cast<uint4*>(lds + W::IMG_U4 + 2 * W::V_U4);

    W3_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * W::NBW;
    const float* in = x;
    const float* resid = resid0;
    float* out = y;

    for (int i = tid; i < 2 * W::V_U4; i += 512) Vs[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };   // rows >= WT stay zero
    if (tid < 8) Img[W::ZERO_U4 + tid] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    // Tab[tw][slot][a] = the five LDS byte addresses (u16) of row a of wtile tw's 5x5 input patch for
    // channel slot `slot`, slot swizzle applied, off-board cells redirected to the zero cell: built once
    // on the host (wino3_build_table), so a patch read costs one unpack and no address arithmetic
    for (int e = tid; e < W::TAB_U4; e += 512) Tab[e] = tab_g[e];

    // input-transform job of this thread: WT * 8 (wtile, channel slot) items spread evenly over the
    // eight waves, lanes 0..LPG-1 of every 16 take one
    constexpr int LPG = (W::WT * 8 + 31) / 32;
    static_assert(LPG <= 16, "too many wtiles for one transform pass");
    const int item = (tid >> 4) * LPG + (tid & 15);
    const bool t_on = (tid & 15) < LPG && item < W::WT * 8;
    const int ts = item & 7, ttw = t_on ? item >> 3 : 0;

    f32x4 ireg[W::NI];
    auto load_img = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            const int cr = idx >> 3, slot = idx & 7;
            const int board = b_first + cr / P;
            f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (idx < W::NBW * P * 8 && board < nboards)
                v = *reinterpret_cast<const f32x4*>(&in[((long)b_first * P + cr) * 128 + chunk * 32 + slot * 4]);
            ireg[i] = v;
        }
    };
    auto store_img = [&]() {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            if (idx < W::NBW * P * 8) Img[w3swz(idx >> 3, idx & 7)] = ireg[i];
        }
    };
    auto img_at = [&](uint32_t byte_addr) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Img) + byte_addr);
    };
    // one row of the patch (a), the five cells b = 0..4
    const uint4* tabrow = Tab + (ttw * 8 + ts) * 5;
    auto patch_row = [&](int a, f32x4* d) {
        const uint4 t = tabrow[a];
        d[0] = img_at(t.x & 0xffffu);
        d[1] = img_at(t.x >> 16);
        d[2] = img_at(t.y & 0xffffu);
        d[3] = img_at(t.y >> 16);
        d[4] = img_at(t.z & 0xffffu);
    };
    // V[q][tw][ts] = (Bt d B)[p][q], q = 0..4, for row p (compile-time after unrolling)
    auto transform = [&](int p, int vbuf) {
        if (!t_on) return;
        f32x4 r[5], d1[5], d2[5], d3[5];
        patch_row(1, d1);
        if (p != 3) patch_row(2, d2);
        patch_row(3, d3);
        if (p == 0) {
            f32x4 d0[5];
            patch_row(0, d0);
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = fma4(-2.0f, d2[b], fma4(2.0f, d0[b], d3[b] - d1[b]));
        } else if (p == 1) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = fma4(-2.0f, d1[b], d3[b] - d2[b]);
        } else if (p == 2) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = fma4(-3.0f, d2[b], fma4(2.0f, d1[b], d3[b]));
        } else if (p == 3) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = d3[b] - d1[b];
        } else {
            f32x4 d4[5];
            patch_row(4, d4);
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = fma4(-2.0f, d3[b], fma4(2.0f, d1[b], d4[b] - d2[b]));
        }
        f32x4* V = Vs + vbuf * W::V_U4;
        V[w3swz(0 * ROWS + ttw, ts)] = fma4(-2.0f, r[2], fma4(2.0f, r[0], r[3] - r[1]));
        V[w3swz(1 * ROWS + ttw, ts)] = fma4(-2.0f, r[1], r[3] - r[2]);
        V[w3swz(2 * ROWS + ttw, ts)] = fma4(-3.0f, r[2], fma4(2.0f, r[1], r[3]));
        V[w3swz(3 * ROWS + ttw, ts)] = r[3] - r[1];
        V[w3swz(4 * ROWS + ttw, ts)] = fma4(-2.0f, r[3], fma4(2.0f, r[1], r[4] - r[2]));
    };

    // weight fragments of a stage: U[stage] in the lanes' own order (nn_host.hip): lane (li, lq) gets channel
    // cout = 16 wave + li, inputs k = kk*16 + lq*4 + jj
    const float* wsrc = wt_all + wave * 512 + lane * 4;   // [stage][wave][kk][lane][4] (nn_host.hip): whole lines per load
    f32x4 bcur[2], bnxt[2];
    auto load_b = [&](int stage, f32x4* b) {
        const float* src = wsrc + (size_t)stage * 128 * 32;
        b[0] = *reinterpret_cast<const f32x4*>(src);
        b[1] = *reinterpret_cast<const f32x4*>(src + 256);
    };

    f32x4 Y[3][3][MT];                               // [i][j]: output (3ty + i, 3tx + j) of the wtile
    const int tq = (wave < 4) ? 0 : 2;               // waves w and w + 4 share a SIMD: transform in different stages
    const int col = wave * 16 + 4 * lq;

#pragma unroll 1
    for (int k = 0; k < n_convs; k++) {
    const float* bias = bias_all + k * 128;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int t = 0; t < MT; t++) Y[i][j][t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    load_b(0, bcur);
    // k = 0: the zero fills and the table are complete; k > 0: every wave's transforms of the previous
    // convolution have ended and its stores to `in` are visible to this workgroup
    __syncthreads();
    load_img(0);
    store_img();
    __syncthreads();
    transform(0, 0);
    __syncthreads();

    W3_STAMP(1);
    f32x4 afix[MT];
    if (W3_DIAG & 4) {
#pragma unroll
        for (int t = 0; t < MT; t++) afix[t] = Vs[w3swz(t * 16 + li, lq)];
    }
    if (W3_DIAG & 8) { bnxt[0] = bcur[0]; bnxt[1] = bcur[1]; }
#pragma unroll 1
    for (int chunk = 0; chunk < 4; chunk++) {
        if (chunk == 1) { W3_STAMP(2); }
        if (chunk == 3) { W3_STAMP(3); }
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const int rowi = chunk * 5 + p;          // V buffer of this row = rowi & 1
            const bool more = p < 4 || chunk + 1 < 4;
            if (p == 3 && chunk + 1 < 4) load_img(chunk + 1);
            if (p == 4 && chunk + 1 < 4) {
                store_img();                         // every transform of this chunk ended before the last barrier
                __syncthreads();
            }
            const f32x4* V = Vs + (rowi & 1) * W::V_U4;
#pragma unroll 1
            for (int q = 0; q < 5; q++) {
                if (more && q == tq && !(W3_DIAG & 2)) transform(p == 4 ? 0 : p + 1, (rowi + 1) & 1);
                const int s = rowi * 5 + q;
                if (s + 1 < 100 && !(W3_DIAG & 8)) load_b(s + 1, bnxt);

                f32x4 m[MT];
#pragma unroll
                for (int t = 0; t < MT; t++) m[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    f32x4 a[MT];
#pragma unroll
                    for (int t = 0; t < MT; t++) a[t] = (W3_DIAG & 4) ? afix[t] : V[w3swz(q * ROWS + t * 16 + li, kk * 4 + lq)];
                    if (W3_DIAG & 32) {
#pragma unroll
                        for (int t = 0; t < MT; t++) m[t] += a[t] * bcur[kk];
                        continue;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; jj++)
#pragma unroll
                        for (int t = 0; t < MT; t++)
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[kk][jj], a[t][jj], m[t], 0, 0, 0);
                }
                if (W3_DIAG & 1) {
#pragma unroll
                    for (int t = 0; t < MT; t++) Y[0][0][t] += m[t];
                    bcur[0] = bnxt[0];
                    bcur[1] = bnxt[1];
                    continue;
                }
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
                bcur[0] = bnxt[0];
                bcur[1] = bnxt[1];
            }
            if (!(W3_DIAG & 16)) __syncthreads();
        }
    }

    W3_STAMP(4);
    // epilogue.  The products are formed transposed (A operand = weights, B operand = V), so in the
    // C/D layout of a 16x16 tile (col = lane & 15, row = 4 * (lane >> 4) + reg) a lane holds FOUR
    // CONSECUTIVE output channels of ONE wtile: one 16-byte load / store per output cell.
    // `out` may alias `resid` (the in-place convolutions of a trunk launch), so loads and stores stay in
    // program order: load -> add -> store per cell would be 27 dependent round trips.  The residual /
    // no-residual choice is made once and a row tile's nine residual loads are issued together.
    auto epilogue = [&](auto has_resid) {
        constexpr bool R = decltype(has_resid)::value;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[col]);
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int w = t * 16 + li;
            const int bl = w / W::T2, rem = w - bl * W::T2;
            const long board = b_first + bl;
            if (w >= W::WT || board >= nboards) continue;
            const int ty = rem / W::T, tx = rem - ty * W::T;
            const long base = (board * P + 3 * ty * S + 3 * tx) * 128 + col;
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
                    f32x4 v = Y[i][j][t] + bv;
                    if (R) v += rv[i][j];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                    *reinterpret_cast<f32x4*>(&out[base + (i * S + j) * 128]) = v;
                }
        }
    };
    if (resid) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    W3_STAMP(5);
    // next convolution of the launch: swap the buffers, step the weights
    if (k & 1) { in = x; out = y; resid = resid0; } else { in = y; out = x; resid = x; }
    wsrc += (size_t)25 * 128 * 128;
    }
}

int wino3_supported(int S_rt) { return S_rt == 9; }
int wino3_boards_per_wg(int S_rt) { return S_rt == 9 ? 4 : 1; }

template <int NBW>
static int launch_wino3_9(float* x, float* y, const float* resid0, const float* wt_wino3, const float* bias,
                          const void* tab, int nboards, int n_convs, hipStream_t st)
{
    const int grid = (nboards + NBW - 1) / NBW;
    const uint4* tg = reinterpret_cast<const uint4*>(tab);
    k_conv3x3_wino3<9, NBW><<<dim3(grid), dim3(512), 0, st>>>(x, y, resid0, wt_wino3, bias, tg, nboards, n_convs);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

// A workgroup costs the same whether 36 or 45 of its 48 tile rows are real: 4 boards per workgroup
// while that fits one round of the chip, else 5 (20 % fewer workgroups).  A board's result does not
// depend on the choice (every tile row is computed independently, in the same order).
// Host side of the input-transform address table for `nbw` boards per workgroup (4 or 5):
// out[((tw * 8 + slot) * 5 + a) * 8 + b], b < 5; returns the number of uint16 entries.
size_t wino3_build_table(int S_rt, int nbw, uint16_t* out)
{
    const int S = S_rt, T = S / 3, T2 = T * T, P = S * S, WT = nbw * T2, zero_u4 = nbw * P * 8;
    if (out)
        for (int tw = 0; tw < WT; tw++)
            for (int slot = 0; slot < 8; slot++)
                for (int a = 0; a < 5; a++)
                    for (int b = 0; b < 8; b++) {
                        const int bl = tw / T2, rem = tw % T2, ty = rem / T, tx = rem % T;
                        const int y = 3 * ty - 1 + a, x = 3 * tx - 1 + b;
                        const bool ok = b < 5 && y >= 0 && y < S && x >= 0 && x < S;
                        const int cell = bl * P + y * S + x;
                        const int u4 = ok ? cell * 8 + (slot ^ ((cell >> 1) & 7)) : zero_u4 + slot;
                        out[(((size_t)tw * 8 + slot) * 5 + a) * 8 + b] = (uint16_t)(u4 * 16);
                    }
    return (size_t)WT * 8 * 5 * 8;
}

// n_convs = 1: y = relu(conv(x) + bias (+ resid0)); n_convs even, resid0 = NULL: the residual trunk
// in one launch, result in x (see the kernel's header)
int launch_conv3x3_wino3(int S_rt, float* x, float* y, const float* resid0, const float* wt_wino3, const float* bias,
                         const void* const* tabs, int nboards, int n_cu, int n_convs, hipStream_t st)
{
    if (S_rt != 9) {
        set_error("the F(3x3,3x3) convolution is built for 9x9 boards");
        return GOAI_ERR_ARG;
    }
    if (n_convs < 1 || (n_convs > 1 && ((n_convs & 1) || resid0))) {
        set_error("a multi-convolution launch is a whole residual trunk: an even count, no outer residual");
        return GOAI_ERR_ARG;
    }
    if ((nboards + 3) / 4 <= n_cu) return launch_wino3_9<4>(x, y, resid0, wt_wino3, bias, tabs[0], nboards, n_convs, st);
    return launch_wino3_9<5>(x, y, resid0, wt_wino3, bias, tabs[1], nboards, n_convs, st);
}

}  // namespace goai
