// conv_wino_kernels.hip -- the trunk's conv3x3 128 -> 128 (+ folded BatchNorm bias, optional
// residual, ReLU; /root/reference/go_ai/models/__init__.py:234-251) in fp32 by Winograd's
// minimal filtering F(2x2, 3x3) on the fp32 matrix cores.
//
//   Y = At [ (G g Gt) . (Bt d B) ] A      (Lavin & Gray 2016; "." = elementwise, summed over cin)
//
// A 2x2 block of outputs ("wtile") needs 16 multiplications per (cin, cout) instead of 36:
// 2.25x fewer MFMA operations than the direct implicit GEMM of conv_kernels.hip (1.82x after
// the 10x10-for-9x9 tile padding).  All arithmetic is fp32 (v_mfma_f32_16x16x4_f32); the
// transforms only add and subtract, G g Gt is folded into the weights on the host in fp64.
// Results differ from the direct kernel by ordinary fp32 re-association (~1e-6).
//
// Work decomposition.  A workgroup (8 waves, one per CU) owns NBW whole boards = NBW * T*T
// wtiles (T = ceil(S/2)), at most ROWS = 112 = 7 MFMA row tiles, and all 128 output channels:
// 9x9 -> 4 boards (100 wtiles), 13x13 -> 2 (98), 19x19 -> 1 (100), 7x7 -> 7 (112), 5x5 -> 12 (108);
// 1024 boards of 9x9 are exactly one workgroup per CU.  Wave w owns the 16 output channels
// 16w..16w+15 for all 7 row tiles.  For every position xi = (p, q) of the 4x4 transform domain
// there is an independent GEMM
//   M_xi[wtile][cout] = sum_cin V_xi[wtile][cin] * U_xi[cin][cout]
// and the four outputs of a wtile are signed sums of the M_xi, so a wave keeps FIVE accumulator
// sets: the four outputs and a temporary for the M_xi in flight (140 registers).
//
// LDS:
//   IMG  the workgroup's boards, one 32-channel chunk, compact: [NBW * P cells][32 ch] fp32 plus
//        one all-zero cell that every off-board tap of the input transform is redirected to;
//   V    Bt d B for ONE row p of the transform domain: [4 q][ROWS wtiles][32 ch], double-buffered:
//        row p + 1 is transformed (by all threads, once per workgroup) while row p is multiplied,
//        so there is one barrier per row of 4 stages (224 MFMAs per wave), not one per stage.
// The weights U_xi never touch LDS: a wave's 16 output channels x 32 inputs of a stage are 2 KB
// that no other wave of the workgroup needs, so each lane loads its two B fragments straight
// from global memory (L2-resident: 1 MB per convolution), one stage ahead.
// Loop: 4 channel chunks x 4 rows p x 4 columns q = 64 stages of 56 MFMAs per wave.
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S>
struct Wino {
    static constexpr int T = (S + 1) / 2, T2 = T * T, P = S * S;
    static constexpr int MT = 7, ROWS = MT * 16;                   // wtile rows per workgroup
    static constexpr int NBW = ROWS / T2;                          // whole boards per workgroup
    static constexpr int WT = NBW * T2;                            // real wtiles per workgroup
    static constexpr int VP = (WT + 7) & ~7;                       // row pitch of V per q
    static constexpr int IMG_U4 = NBW * P * 8 + 8;                 // in 16-byte units (+ the zero cell)
    static constexpr int ZERO_U4 = NBW * P * 8;
    static constexpr int V_U4 = 4 * VP * 8;                        // one buffer
    static constexpr int SLACK_U4 = (ROWS - VP) * 8;               // rows WT..ROWS-1 of the last q read past it
    static constexpr int TAB_U4 = WT * 2;                          // 16 x u16 LDS addresses per wtile
    static constexpr int NI = (NBW * P * 8 + 511) / 512;           // staged 16-byte pieces per thread per chunk
    // 9x9 (100 wtiles), 13x13 (98) and 19x19 (100) are six full 16-row MFMA tiles plus 4 / 2 / 4 wtiles:
    // that tail runs on v_mfma_f32_4x4x1_16b_f32 blocks (see the loop) instead of a seventh, three-quarters
    // empty tile.  The tail holds the LAST TPB wtiles of every board, so all boards are treated alike:
    // MFMA row of (board b, wtile w) = w < MAIN ? b * MAIN + w : 96 + b * TPB + (w - MAIN).
    static constexpr int TAIL = WT - 96;
    static constexpr bool USE4 = TAIL > 0 && TAIL <= 4 && TAIL % NBW == 0;
    static constexpr int TPB = USE4 ? TAIL / NBW : 0, MAIN = T2 - TPB;
    static constexpr int LDS_U4 = IMG_U4 + 2 * V_U4 + SLACK_U4 + TAB_U4;
    static_assert(NBW >= 1, "board too large for one workgroup");
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(IMG_U4 * 16 < 65536, "input-transform addresses are 16-bit");
};

#ifdef GOAI_CONV_STAMPS
__device__ unsigned long long g_wino_stamps[8 * 4096];   // diagnostic build only (tools/wino_stamps.py)
extern "C" int goai_debug_wino_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_stamps), (size_t)n * sizeof(unsigned long long));
}
#define W_STAMP(i) if (threadIdx.x == 0) { g_wino_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 1 || (i) == 4) g_wino_stamps[blockIdx.x * 8 + 6 + ((i) == 4)] = __builtin_amdgcn_s_memtime(); }
#else
#define W_STAMP(i)
#endif

__device__ __forceinline__ int wswz(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

template <int S, bool RESID>
__global__ __launch_bounds__(512) void k_conv3x3_wino(const float* __restrict__ in, const float* __restrict__ wt,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ resid, float* __restrict__ out,
                                                      int nboards)
{
    using W = Wino<S>;
    constexpr int P = W::P, MT = W::MT;
    __shared__ f32x4 lds[W::LDS_U4];
    f32x4* Img = lds;
    f32x4* Vs = lds + W::IMG_U4;
    uint16_t* Tab = reinterpret_cast<uint16_t*>(lds + W::IMG_U4 + 2 * W::V_U4 + W::SLACK_U4);

    W_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * W::NBW;

    // V rows WT..ROWS-1 of the MFMA row tiles are padding: whatever finite values they hold only
    // reach accumulator rows that are never stored (for q < 3 they alias the next q's first rows)
    for (int i = tid; i < 2 * W::V_U4 + W::SLACK_U4; i += 512) Vs[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    if (tid < 8) Img[W::ZERO_U4 + tid] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    // Tab[tw][a][b] = byte address in Img (slot 0, before the slot swizzle) of cell (2ty - 1 + a,
    // 2tx - 1 + b) of wtile tw's 4x4 input patch, or of the zero cell when that is off the board
    for (int e = tid; e < W::WT * 16; e += 512) {
        const int tw = e >> 4, a = (e >> 2) & 3, b = e & 3;
        const int bl = tw / W::T2, rem = tw - bl * W::T2;
        const int ty = rem / W::T, tx = rem - ty * W::T;
        const int y = 2 * ty - 1 + a, x = 2 * tx - 1 + b;
        const bool ok = y >= 0 && y < S && x >= 0 && x < S;
        Tab[e] = (uint16_t)((ok ? (bl * P + y * S + x) * 8 : W::ZERO_U4) * 16);
    }

    // input-transform jobs of this thread: items (wtile tw, 16-byte channel slot ts), two passes
    const int ts = tid & 7;

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
            if (idx < W::NBW * P * 8) Img[wswz(idx >> 3, idx & 7)] = ireg[i];
        }
    };
    // V[q][tw][ts] = (Bt d B)[p][q] for q = 0..3: rows of Bt are (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    auto img_at = [&](uint32_t byte_addr) -> f32x4 {
        // slot swizzle of wswz() on a byte address: cell = addr >> 7, zero cell: every slot is zero
        const uint32_t cell = byte_addr >> 7;
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Img) + byte_addr +
                                               ((ts ^ ((cell >> 1) & 7)) << 4));
    };
    auto transform = [&](int p, int vbuf) {
        const int a1 = (p == 0) ? 0 : 1, a2 = (p == 3) ? 3 : 2;
        const float s1 = (p == 2) ? -1.0f : 1.0f, s2 = (p == 0 || p == 3) ? -1.0f : 1.0f;
        f32x4* V = Vs + vbuf * W::V_U4;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int tw = (tid >> 3) + 64 * j;
            if (tw >= W::WT) continue;
            int vr = tw;                             // MFMA row of this wtile
            if constexpr (W::USE4) {
                const int bl = tw / W::T2, wt_ = tw - bl * W::T2;
                vr = wt_ < W::MAIN ? bl * W::MAIN + wt_ : 96 + bl * W::TPB + (wt_ - W::MAIN);
            }
            const uint2 t1 = *reinterpret_cast<const uint2*>(&Tab[tw * 16 + a1 * 4]);
            const uint2 t2 = *reinterpret_cast<const uint2*>(&Tab[tw * 16 + a2 * 4]);
            f32x4 r[4];
            r[0] = s1 * img_at(t1.x & 0xffffu) + s2 * img_at(t2.x & 0xffffu);
            r[1] = s1 * img_at(t1.x >> 16) + s2 * img_at(t2.x >> 16);
            r[2] = s1 * img_at(t1.y & 0xffffu) + s2 * img_at(t2.y & 0xffffu);
            r[3] = s1 * img_at(t1.y >> 16) + s2 * img_at(t2.y >> 16);
            V[wswz(0 * W::VP + vr, ts)] = r[0] - r[2];
            V[wswz(1 * W::VP + vr, ts)] = r[1] + r[2];
            V[wswz(2 * W::VP + vr, ts)] = r[2] - r[1];
            V[wswz(3 * W::VP + vr, ts)] = r[1] - r[3];
        }
    };

    // weight fragments of a stage: U[stage] in the lanes' own order (nn_host.hip): lane (li, lq) gets channel
    // cout = 16 wave + li, inputs k = kk*16 + lq*4 + jj
    const float* wsrc = wt + wave * 512 + lane * 4;       // [stage][wave][kk][lane][4] (nn_host.hip): whole lines per load
    f32x4 bcur[2], bnxt[2];
    auto load_b = [&](int stage, f32x4* b) {
        const float* src = wsrc + (size_t)stage * 128 * 32;
        b[0] = *reinterpret_cast<const f32x4*>(src);
        b[1] = *reinterpret_cast<const f32x4*>(src + 256);
    };

    f32x4 Y[2][2][MT];                               // [i][j]: output (2ty + i, 2tx + j) of the wtile
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int t = 0; t < MT; t++) Y[i][j][t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    load_img(0);
    load_b(0, bcur);
    __syncthreads();                                 // the zero fills are complete
    store_img();
    __syncthreads();
    transform(0, 0);
    __syncthreads();

    W_STAMP(1);
#pragma unroll 1
    for (int chunk = 0; chunk < 4; chunk++) {
        if (chunk == 1) { W_STAMP(2); }
        if (chunk == 3) { W_STAMP(3); }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            // The NEXT row is transformed into the other V buffer while this one is multiplied.  The two
            // waves that share a SIMD (w and w + 4) do it in different stages of the row, so that the
            // matrix pipe always has the other wave's MFMAs to run meanwhile.
            const bool more = p < 3 || chunk + 1 < 4;
            if (p == 2 && chunk + 1 < 4) load_img(chunk + 1);
            if (p == 3 && chunk + 1 < 4) {
                store_img();                         // every transform of this chunk ended before the last barrier
                __syncthreads();
            }
            const int tq = (wave < 4) ? 0 : 2;
            const f32x4* V = Vs + (p & 1) * W::V_U4;
#pragma unroll 1
            for (int q = 0; q < 4; q++) {
                if (more && q == tq) transform((p + 1) & 3, (p + 1) & 1);
                const int s = chunk * 16 + p * 4 + q;
                if (s + 1 < 64) load_b(s + 1, bnxt);

                f32x4 m[MT];
#pragma unroll
                for (int t = 0; t < MT; t++) m[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
                f32x4 m6b = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    f32x4 a[MT];
#pragma unroll
                    for (int t = 0; t < MT; t++)
                        a[t] = V[wswz(q * W::VP + ((W::USE4 && t == 6) ? 96 + (li & 3) : t * 16 + li), kk * 4 + lq)];
#pragma unroll
                    for (int jj = 0; jj < 4; jj++)
#pragma unroll
                        for (int t = 0; t < MT; t++) {
                            if (W::USE4 && t == 6) {
                                // rows 96..99: sixteen 4x4 blocks, block = lane / 4 = (k slice lq, channel group
                                // li / 4); the lane's weight register IS the block's A element (channel li,
                                // k = 16 kk + 4 lq + jj), B = V[96 + li % 4].  Partial sums over the lane's k slice
                                // (two accumulators: dependent 4x4 MFMAs need wait states); the output transform
                                // is linear, the four slices meet in the epilogue.
                                if (jj & 1) m6b = __builtin_amdgcn_mfma_f32_4x4x1f32(bcur[kk][jj], a[t][jj], m6b, 0, 0, 0);
                                else m[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(bcur[kk][jj], a[t][jj], m[t], 0, 0, 0);
                            } else {
                                m[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[kk][jj], a[t][jj], m[t], 0, 0, 0);
                            }
                        }
                }
                if (W::USE4) m[6] += m6b;
                // output transform: Y[i][j] += At[i][p] * At[j][q] * M,  At = (1 1 1 0 ; 0 1 -1 -1);
                // p is unrolled (compile-time row factor), q is a run-time loop (column factor as +-1.0 / 0)
                const float cj0 = (q < 3) ? 1.0f : 0.0f;
                const float cj1 = (q == 0) ? 0.0f : (q == 1 ? 1.0f : -1.0f);
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int ci = (i == 0) ? (p < 3 ? 1 : 0) : (p == 0 ? 0 : (p == 1 ? 1 : -1));
                    if (ci != 0) {
                        const float f0 = ci > 0 ? cj0 : -cj0, f1 = ci > 0 ? cj1 : -cj1;
#pragma unroll
                        for (int t = 0; t < MT; t++)
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                Y[i][0][t][e] = fmaf(f0, m[t][e], Y[i][0][t][e]);
                                Y[i][1][t][e] = fmaf(f1, m[t][e], Y[i][1][t][e]);
                            }
                    }
                }
                bcur[0] = bnxt[0];
                bcur[1] = bnxt[1];
            }
            __syncthreads();
        }
    }

    W_STAMP(4);
    // epilogue.  The products are formed transposed (A operand = weights, B operand = V), so in the
    // C/D layout of a 16x16 tile (col = lane & 15, row = 4 * (lane >> 4) + reg) a lane holds FOUR
    // CONSECUTIVE output channels of ONE wtile: one 16-byte load / store per output cell.
    auto finish = [&](const f32x4 yv, const f32x4 bv, int y, int x, long board, int c0) {
        if (y < S && x < S) {
            const long off = (board * P + y * S + x) * 128 + c0;
            f32x4 v = yv + bv;
            if (RESID) v += *reinterpret_cast<const f32x4*>(&resid[off]);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
            *reinterpret_cast<f32x4*>(&out[off]) = v;
        }
    };
    const int col = wave * 16 + 4 * lq;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[col]);
#pragma unroll
    for (int t = 0; t < (W::USE4 ? 6 : MT); t++) {
        const int w = t * 16 + li;
        int bl, rem;
        if constexpr (W::USE4) { bl = w / W::MAIN; rem = w - bl * W::MAIN; }
        else { bl = w / W::T2; rem = w - bl * W::T2; }
        const long board = b_first + bl;
        if (w >= W::WT || board >= nboards) continue;
        const int ty = rem / W::T, tx = rem - ty * W::T;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) finish(Y[i][j][t], bv, 2 * ty + i, 2 * tx + j, board, col);
    }
    if constexpr (W::USE4) {
        // the 4x4 blocks: a lane holds channels 4 (li / 4) .. + 3 of tail row li % 4 for its k slice; the four
        // slices (lanes l, l ^ 16, l ^ 32, l ^ 48) are summed in a fixed order, lanes 0..15 store
        const int col2 = wave * 16 + (li & 12), k4 = li & 3;
        const f32x4 bv2 = *reinterpret_cast<const f32x4*>(&bias[col2]);
        const int bl = k4 / W::TPB, rem = W::MAIN + k4 - bl * W::TPB;
        const long board = b_first + bl;
        const int ty = rem / W::T, tx = rem - ty * W::T;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x4 v = Y[i][j][6];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] += __shfl_xor(v[e], 16);
                    v[e] += __shfl_xor(v[e], 32);
                }
                if (lq == 0 && k4 < W::TAIL && board < nboards) finish(v, bv2, 2 * ty + i, 2 * tx + j, board, col2);
            }
    }
    W_STAMP(5);
}

template <int S>
static int launch_wino_size(const float* in, const float* wt_wino, const float* bias, const float* resid, float* out,
                            int nboards, hipStream_t st)
{
    const int grid = (nboards + Wino<S>::NBW - 1) / Wino<S>::NBW;
    if (resid) k_conv3x3_wino<S, true><<<dim3(grid), dim3(512), 0, st>>>(in, wt_wino, bias, resid, out, nboards);
    else k_conv3x3_wino<S, false><<<dim3(grid), dim3(512), 0, st>>>(in, wt_wino, bias, resid, out, nboards);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int wino_boards_per_wg(int S_rt)
{
    switch (S_rt) {
        case 5: return Wino<5>::NBW;
        case 7: return Wino<7>::NBW;
        case 9: return Wino<9>::NBW;
        case 13: return Wino<13>::NBW;
        case 19: return Wino<19>::NBW;
        default: return 1;
    }
}

int launch_conv3x3_wino(int S_rt, const float* in, const float* wt_wino, const float* bias, const float* resid,
                        float* out, int nboards, hipStream_t st)
{
    GOAI_DISPATCH_SIZE(S_rt, return launch_wino_size<S>(in, wt_wino, bias, resid, out, nboards, st));
    return GOAI_OK;
}

}  // namespace goai
