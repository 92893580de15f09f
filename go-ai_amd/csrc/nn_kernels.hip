// nn_kernels.hip -- the policy/value CNN forward over a batch of leaf states.
//
// Replaces the torch forward the reference runs on batch 1 per leaf
// (/root/reference/go_ai/models/__init__.py:14-31,234-251 trunk + BasicBlock,
//  /root/reference/go_ai/models/ac_net.py:15-31,40-45, /root/reference/go_ai/models/val_net.py:12-24).
// BatchNorm is folded into the preceding convolution on the host (eval() is forced by the
// reference, models/__init__.py:64).
//
// Data layout: activations are channel-last fp32, act[row][128] with row = board*P + cell,
// so a 3x3 convolution is an implicit GEMM  out[M = B*P][128] = sum over 9 taps of
// shift_tap(act)[M][128] x W_tap[128][128]  on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32 fma chains).
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------
// conv0: 6 -> 128, 3x3, pad 1, + folded BN + ReLU, reading the uint8 state planes.
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
// conv3x3 128 -> 128 (+ folded BN bias, optional residual, ReLU) as an implicit GEMM on
// v_mfma_f32_32x32x2_f32.
//
// Workgroup = 4 waves = a 128-row x 128-channel output tile over FLAT rows (tiles may span
// boards); each wave owns 64x64 = 2x2 MFMA tiles (64 accumulator registers).
// K = 9 taps x 128 input channels is walked as 4 channel chunks x 9 taps ("stages"):
//   - the A image (tile rows + a halo of S+1 rows either side, 32 channels) is staged in LDS
//     once per chunk and re-read with a per-tap row shift; rows whose tap falls off the
//     board read a dedicated zero row, so the inner loop has no branches;
//   - the B image (W_tap^T, [128 out][32 in], 16 KiB) is double-buffered: the next stage's
//     weights are fetched into registers while the MFMAs of this stage run.
// LDS rows are padded 32 -> 36 floats: ds_read_b128 of 16 consecutive rows is conflict-free.
// --------------------------------------------------------------------------------------
constexpr int CONV_BM = 128;
constexpr int CONV_CK = 32;
constexpr int CONV_LD = 36;

template <int S>
struct ConvCfg {
    static constexpr int HALO = S + 1;
    static constexpr int ROWS_A = CONV_BM + 2 * HALO;          // data rows; row ROWS_A is the zero row
    static constexpr int A_FLOATS = (ROWS_A + 1) * CONV_LD;
    static constexpr int B_FLOATS = 128 * CONV_LD;
    static constexpr int NA = (ROWS_A * 8 + 255) / 256;        // float4 per thread per A chunk
};

template <int S, bool RESID>
__global__ __launch_bounds__(256, 2) void k_conv3x3(const float* __restrict__ in, const float* __restrict__ wt,
                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                    float* __restrict__ out, int M)
{
    using G = Geom<S>;
    using C = ConvCfg<S>;
    __shared__ __attribute__((aligned(16))) float lds[C::A_FLOATS + 2 * C::B_FLOATS];
    float* As = lds;
    float* Bs = lds + C::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * CONV_BM;
    const int li = lane & 31, lh = lane >> 5;

    // zero row
    if (tid < CONV_LD) As[C::ROWS_A * CONV_LD + tid] = 0.0f;

    // per-lane geometry of the two 32-row tiles this wave owns
    int rloc[2], rr[2], cc[2];
    bool mvalid[2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        rloc[tm] = wr * 64 + tm * 32 + li;
        int m = m0 + rloc[tm];
        mvalid[tm] = m < M;
        int p = m % G::P;
        rr[tm] = p / S;
        cc[tm] = p % S;
    }
    int boff[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) boff[tn] = (wc * 64 + tn * 32 + li) * CONV_LD + lh * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    f32x4 breg[4];
    f32x4 areg[C::NA];

    auto load_a = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < C::NA; i++) {
            int idx = tid + i * 256;
            int row = idx >> 3, piece = idx & 7;
            long grow = (long)m0 - C::HALO + row;
            f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (row < C::ROWS_A && grow >= 0 && grow < M)
                v = *reinterpret_cast<const f32x4*>(&in[grow * 128 + chunk * CONV_CK + piece * 4]);
            areg[i] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < C::NA; i++) {
            int idx = tid + i * 256;
            int row = idx >> 3, piece = idx & 7;
            if (row < C::ROWS_A) *reinterpret_cast<f32x4*>(&As[row * CONV_LD + piece * 4]) = areg[i];
        }
    };
    auto load_b = [&](int stage) {
        const float* src = wt + (size_t)stage * 128 * CONV_CK;
#pragma unroll
        for (int i = 0; i < 4; i++) breg[i] = *reinterpret_cast<const f32x4*>(&src[(tid + i * 256) * 4]);
    };
    auto store_b = [&](int buf) {
        float* dst = Bs + buf * C::B_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(&dst[(idx >> 3) * CONV_LD + (idx & 7) * 4]) = breg[i];
        }
    };

    load_a(0);
    load_b(0);
    store_a();
    store_b(0);
    __syncthreads();

    constexpr int NSTAGE = 4 * 9;
    for (int s = 0; s < NSTAGE; s++) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const int buf = s & 1;
        if (s + 1 < NSTAGE) load_b(s + 1);
        const bool next_chunk = (tap == 8) && (chunk + 1 < 4);
        if (next_chunk) load_a(chunk + 1);

        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        int aoff[2];
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            bool ok = mvalid[tm] && (unsigned)(rr[tm] + dy) < (unsigned)S && (unsigned)(cc[tm] + dx) < (unsigned)S;
            int srow = ok ? (rloc[tm] + C::HALO + dy * S + dx) : C::ROWS_A;
            aoff[tm] = srow * CONV_LD + lh * 4;
        }
        const float* Bb = Bs + buf * C::B_FLOATS;
#pragma unroll
        for (int kk = 0; kk < CONV_CK / 8; kk++) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int tm = 0; tm < 2; tm++) a[tm] = *reinterpret_cast<const f32x4*>(&As[aoff[tm] + kk * 8]);
#pragma unroll
            for (int tn = 0; tn < 2; tn++) b[tn] = *reinterpret_cast<const f32x4*>(&Bb[boff[tn] + kk * 8]);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int tm = 0; tm < 2; tm++)
#pragma unroll
                    for (int tn = 0; tn < 2; tn++)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
        }
        if (s + 1 < NSTAGE) store_b(buf ^ 1);      // buf^1 was last read in stage s-1
        if (next_chunk) {
            __syncthreads();                        // everyone is done with this chunk's A image
            store_a();
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int col = wc * 64 + tn * 32 + li;
        const float bv = bias[col];
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            const long row0 = (long)m0 + wr * 64 + tm * 32 + 4 * lh;
            if (m0 + CONV_BM <= M) {
                // full tile: batch the 16 residual loads, then the 16 stores (no per-element waits)
                float rv[16];
                if (RESID) {
#pragma unroll
                    for (int r = 0; r < 16; r++) rv[r] = resid[(row0 + (r & 3) + 8 * (r >> 2)) * 128 + col];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[tm][tn][r] + bv;
                    if (RESID) v += rv[r];
                    out[(row0 + (r & 3) + 8 * (r >> 2)) * 128 + col] = fmaxf(v, 0.0f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const long row = row0 + (r & 3) + 8 * (r >> 2);
                    if (row < M) {
                        float v = acc[tm][tn][r] + bv;
                        if (RESID) v += resid[row * 128 + col];
                        out[row * 128 + col] = fmaxf(v, 0.0f);
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// Heads: one 256-thread workgroup per board.
//   1x1 convs (+ folded BN, ReLU): NC = 3 channels for the actor-critic net (act 0, act 1,
//   crit) or 1 for the value net; then Linear(2P -> A) and Linear(P -> 256) ReLU Linear(256 -> 1).
// --------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ act, HeadWeights hw, int is_ac,
                                               float* __restrict__ pi_out, float* __restrict__ val_out, int nboards)
{
    using G = Geom<S>;
    __shared__ float h[3 * G::P];      // [channel][cell]: act0, act1, crit (value net: crit only at h[2P..])
    __shared__ float z[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c_first = is_ac ? 0 : 2;
    for (int b = blockIdx.x; b < nboards; b += gridDim.x) {
        __syncthreads();
        for (int p = wave; p < G::P; p += 4) {
            const float2 x = *reinterpret_cast<const float2*>(&act[((size_t)b * G::P + p) * 128 + 2 * lane]);
            for (int c = c_first; c < 3; c++) {
                const float2 w = *reinterpret_cast<const float2*>(&hw.conv_w[c * 128 + 2 * lane]);
                float sum = fmaf(x.y, w.y, x.x * w.x);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
                if (lane == 0) h[c * G::P + p] = fmaxf(sum + hw.conv_b[c], 0.0f);
            }
        }
        __syncthreads();
        if (is_ac) {
            for (int a = tid; a < G::A; a += 256) {
                float sum = hw.act_lin_b[a];
                for (int j = 0; j < 2 * G::P; j++) sum = fmaf(hw.act_lin_wt[(size_t)j * G::A + a], h[j], sum);
                pi_out[(size_t)b * G::A + a] = sum;
            }
        }
        {
            float sum = hw.crit_b1[tid];
            for (int p = 0; p < G::P; p++) sum = fmaf(hw.crit_w1t[p * 256 + tid], h[2 * G::P + p], sum);
            float v = fmaxf(sum, 0.0f) * hw.crit_w2[tid];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) z[wave] = v;
        }
        __syncthreads();
        if (tid == 0) val_out[b] = ((z[0] + z[1]) + (z[2] + z[3])) + hw.crit_b2[0];
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
int launch_conv0(int S_rt, const uint8_t* states, const float* w0, const float* b0, float shift, float* out,
                 int nboards, hipStream_t st)
{
    int grid = nboards < 4096 ? nboards : 4096;
    GOAI_DISPATCH_SIZE(S_rt, k_conv0<S><<<dim3(grid), dim3(256), 0, st>>>(states, w0, b0, shift, out, nboards));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_conv3x3(int S_rt, const float* in, const float* wt, const float* bias, const float* resid, float* out,
                   int M, hipStream_t st)
{
    int grid = (M + CONV_BM - 1) / CONV_BM;
    if (resid) {
        GOAI_DISPATCH_SIZE(S_rt, k_conv3x3<S, true><<<dim3(grid), dim3(256), 0, st>>>(in, wt, bias, resid, out, M));
    } else {
        GOAI_DISPATCH_SIZE(S_rt, k_conv3x3<S, false><<<dim3(grid), dim3(256), 0, st>>>(in, wt, bias, resid, out, M));
    }
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_heads(int S_rt, const float* act, const HeadWeights& hw, int is_ac, float* pi, float* val, int nboards,
                 hipStream_t st)
{
    int grid = nboards < 8192 ? nboards : 8192;
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
