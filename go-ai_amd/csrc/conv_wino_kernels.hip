// conv_wino_kernels.hip -- the trunk's conv3x3 128 -> 128 (+ folded BatchNorm bias, optional
// residual, ReLU; /root/reference/go_ai/models/__init__.py:234-251) in fp32 by Winograd's
// minimal filtering F(2x2, 3x3) on the fp32 matrix cores, for boards up to 9x9.
//
//   Y = At [ (G g Gt) . (Bt d B) ] A      (Lavin & Gray 2016; "." = elementwise, summed over cin)
//
// A 2x2 block of outputs ("wtile") needs 16 multiplications per (cin, cout) instead of 36:
// 2.25x fewer MFMA operations than the direct implicit GEMM of conv_kernels.hip (1.82x after
// the 10x10-for-9x9 tile padding).  All arithmetic is fp32 (v_mfma_f32_16x16x4_f32); the
// transforms only add and subtract, G g Gt is folded into the weights on the host in fp64.
// Results differ from the direct kernel by ordinary fp32 re-association (~1e-6).
//
// STATUS (round 1): correct (tests/test_gpu_nn.py passes with GOAI_CONV_ALGO=wino) but NOT faster
// than the direct kernel yet: 208 us against 206 us per launch at 1024 boards of 9x9.  400
// workgroups on 256 CUs run as two rounds, and a stage (32 MFMAs per wave, then an fp32 VALU
// output transform that waits for them) is too short to hide its barrier and LDS latencies;
// without barriers, weight staging and input transform the loop alone takes 167 us.  Opt-in
// through GOAI_CONV_ALGO=wino; the direct kernel stays the default.
//
// Work decomposition.  wtiles are numbered flat, w = board * T*T + ty * T + tx (T = ceil(S/2)).
// A workgroup (8 waves, one per CU) owns WT = 64 consecutive wtiles x all 128 output channels;
// wave (wr, wc) owns 2 MFMA row tiles (32 wtiles) x 32 output channels.  For every position
// xi = (p, q) of the 4x4 transform domain there is an independent GEMM
//   M_xi[wtile][cout] = sum_cin V_xi[wtile][cin] * U_xi[cin][cout]
// and the four outputs of a wtile are signed sums of the M_xi, so a wave keeps FIVE accumulator
// sets: the four outputs and a temporary for the M_xi in flight.
//
// LDS (one workgroup per CU):
//   IMG  the zero-padded input boards the workgroup's wtiles touch, one 32-channel chunk:
//        [NB boards][(2T+2)^2 cells][32 ch] fp32; only real cells are ever rewritten, so the
//        zero border written once at start is the convolution's padding;
//   V    Bt d B for ONE row p of the transform domain: [4 q][WT wtiles][32 ch], computed once per
//        workgroup by all threads (not per wave) from IMG -- the MFMA stages then read their A
//        fragments with plain conflict-free ds_read_b128, exactly like the direct kernel;
//   B    U_xi^T for the stage, [128 out][32 in] = 16 KiB, double-buffered through registers.
// Loop: 4 channel chunks x 4 rows p x 4 columns q = 64 stages of 32 MFMAs per wave.
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int S>
struct Wino {
    static constexpr int T = (S + 1) / 2, T2 = T * T, PG = 2 * T + 2, CELLS = PG * PG;
    static constexpr int RT = 2, WT = 2 * RT * 16;                 // wtiles per workgroup
    static constexpr int NB = (WT - 1 + T2 - 1) / T2 + 1;          // boards a workgroup can touch
    static constexpr int IMG_U4 = NB * CELLS * 8;                  // in 16-byte units
    static constexpr int V_U4 = 4 * WT * 8;
    static constexpr int B_U4 = 128 * 8;
    static constexpr int NI = (NB * S * S * 8 + 511) / 512;        // staged 16-byte pieces per thread per chunk
    static constexpr int LDS_BYTES = (IMG_U4 + V_U4 + 2 * B_U4) * 16;
};

__device__ __forceinline__ int wswz(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

template <int S, bool RESID>
__global__ __launch_bounds__(512, 2) void k_conv3x3_wino(const float* __restrict__ in, const float* __restrict__ wt,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ resid, float* __restrict__ out,
                                                         int nboards)
{
    using W = Wino<S>;
    constexpr int P = S * S, RT = W::RT, WT = W::WT, PG = W::PG;
    __shared__ f32x4 lds[W::IMG_U4 + W::V_U4 + 2 * W::B_U4];
    f32x4* Img = lds;
    f32x4* Vs = lds + W::IMG_U4;
    f32x4* Bs = Vs + W::V_U4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 15, lq = lane >> 4;
    const long w0 = (long)blockIdx.x * WT;
    const long total_w = (long)nboards * W::T2;
    const int b_first = (int)(w0 / W::T2);

    for (int i = tid; i < W::IMG_U4; i += 512) Img[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    // this thread's job in the input transform: wtile tw, 16-byte channel slot ts
    const int tw = tid >> 3, ts = tid & 7;
    int tbase;                                       // Img index (16-byte units) of d[0][0], slot ts
    {
        long w = w0 + tw;
        if (w >= total_w) w = w0;                    // past the last board: any finite data, never stored
        const int bl = (int)(w / W::T2) - b_first, rem = (int)(w % W::T2);
        const int ty = rem / W::T, tx = rem - ty * W::T;
        tbase = ((bl * W::CELLS + 2 * ty * PG + 2 * tx) << 3) + ts;
    }

    f32x4 ireg[W::NI];
    auto load_img = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            const int cr = idx >> 3, slot = idx & 7;
            const int bl = cr / P, cell = cr - bl * P;
            const int board = b_first + bl;
            f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (idx < W::NB * P * 8 && board < nboards)
                v = *reinterpret_cast<const f32x4*>(&in[((long)board * P + cell) * 128 + chunk * 32 + slot * 4]);
            ireg[i] = v;
        }
    };
    auto store_img = [&]() {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            const int cr = idx >> 3, slot = idx & 7;
            const int bl = cr / P, cell = cr - bl * P;
            const int y = cell / S, x = cell - y * S;
            if (idx < W::NB * P * 8) Img[((bl * W::CELLS + (y + 1) * PG + (x + 1)) << 3) + slot] = ireg[i];
        }
    };
    f32x4 breg[2];
    auto load_b = [&](int stage) {
        const float* src = wt + (size_t)stage * 128 * 32;
#pragma unroll
        for (int i = 0; i < 2; i++) breg[i] = *reinterpret_cast<const f32x4*>(&src[(tid + i * 512) * 4]);
    };
    auto store_b = [&](int buf) {
        f32x4* dst = Bs + buf * W::B_U4;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = tid + i * 512;
            dst[wswz(idx >> 3, idx & 7)] = breg[i];
        }
    };
    // V[q][tw][ts] = (Bt d B)[p][q] for q = 0..3: rows of Bt are (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    auto transform = [&](int p) {
        const int a1 = (p == 0) ? 0 : 1, a2 = (p == 3) ? 3 : 2;
        const float s1 = (p == 2) ? -1.0f : 1.0f, s2 = (p == 0 || p == 3) ? -1.0f : 1.0f;
        f32x4 r[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const f32x4 d1 = Img[tbase + ((a1 * PG + b) << 3)], d2 = Img[tbase + ((a2 * PG + b) << 3)];
            r[b] = s1 * d1 + s2 * d2;
        }
        Vs[wswz(0 * WT + tw, ts)] = r[0] - r[2];
        Vs[wswz(1 * WT + tw, ts)] = r[1] + r[2];
        Vs[wswz(2 * WT + tw, ts)] = r[2] - r[1];
        Vs[wswz(3 * WT + tw, ts)] = r[1] - r[3];
    };

    int brow[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) brow[tn] = wc * 32 + tn * 16 + li;

    f32x4 Y[2][2][RT][2];                            // [i][j]: output (2ty + i, 2tx + j) of the wtile
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) Y[i][j][t][tn] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    load_img(0);
    load_b(0);
    __syncthreads();                                 // the zero fill is complete
    store_img();
    store_b(0);
    __syncthreads();

    for (int chunk = 0; chunk < 4; chunk++) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            transform(p);                            // the previous stage's barrier freed V
            __syncthreads();
#pragma unroll 1
            for (int q = 0; q < 4; q++) {
                const int s = chunk * 16 + p * 4 + q;
                const int buf = s & 1;
                if (s + 1 < 64) load_b(s + 1);
                const bool last_of_chunk = (p == 3 && q == 3);
                if (last_of_chunk && chunk + 1 < 4) load_img(chunk + 1);

                const f32x4* Bb = Bs + buf * W::B_U4;
                f32x4 b[2][2], a[RT][2];
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) b[kk][tn] = Bb[wswz(brow[tn], kk * 4 + lq)];
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int kk = 0; kk < 2; kk++) a[t][kk] = Vs[wswz(q * WT + (wr * RT + t) * 16 + li, kk * 4 + lq)];

                f32x4 m[RT][2];
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++) m[t][tn] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++)
#pragma unroll
                        for (int t = 0; t < RT; t++)
#pragma unroll
                            for (int tn = 0; tn < 2; tn++)
                                m[t][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][kk][jj], b[kk][tn][jj], m[t][tn], 0, 0, 0);

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
                        for (int t = 0; t < RT; t++)
#pragma unroll
                            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    Y[i][0][t][tn][e] = fmaf(f0, m[t][tn][e], Y[i][0][t][tn][e]);
                                    Y[i][1][t][tn][e] = fmaf(f1, m[t][tn][e], Y[i][1][t][tn][e]);
                                }
                    }
                }

                if (s + 1 < 64) store_b(buf ^ 1);     // buf^1 was last read in stage s-1
                if (last_of_chunk && chunk + 1 < 4) store_img();   // the transforms of this chunk are done
                __syncthreads();
            }
        }
    }

    // epilogue.  C/D layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg.
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long w = w0 + (wr * RT + t) * 16 + 4 * lq + r;
            if (w >= total_w) continue;
            const long board = w / W::T2;
            const int rem = (int)(w - board * W::T2);
            const int ty = rem / W::T, tx = rem - ty * W::T;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int y = 2 * ty + i, x = 2 * tx + j;
                    if (y < S && x < S) {
                        const long rowoff = (board * P + y * S + x) * 128;
#pragma unroll
                        for (int tn = 0; tn < 2; tn++) {
                            const int col = wc * 32 + tn * 16 + li;
                            float v = Y[i][j][t][tn][r] + bias[col];
                            if (RESID) v += resid[rowoff + col];
                            out[rowoff + col] = fmaxf(v, 0.0f);
                        }
                    }
                }
        }
}

int launch_conv3x3_wino(int S_rt, const float* in, const float* wt_wino, const float* bias, const float* resid,
                        float* out, int nboards, hipStream_t st)
{
#define GOAI_WINO_CASE(SZ)                                                                                     \
    case SZ: {                                                                                                 \
        const long total = (long)nboards * Wino<SZ>::T2;                                                       \
        const int grid = (int)((total + Wino<SZ>::WT - 1) / Wino<SZ>::WT);                                     \
        if (resid) k_conv3x3_wino<SZ, true><<<dim3(grid), dim3(512), 0, st>>>(in, wt_wino, bias, resid, out, nboards); \
        else k_conv3x3_wino<SZ, false><<<dim3(grid), dim3(512), 0, st>>>(in, wt_wino, bias, resid, out, nboards);     \
    } break
    switch (S_rt) {
        GOAI_WINO_CASE(5);
        GOAI_WINO_CASE(7);
        GOAI_WINO_CASE(9);
        default: set_error("the Winograd convolution supports boards up to 9x9"); return GOAI_ERR_ARG;
    }
#undef GOAI_WINO_CASE
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
