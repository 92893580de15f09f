// train_kernels.hip -- one optimisation step of the policy/value CNN on the GPU: train-mode forward (BatchNorm
// on the batch's own statistics), backward, Adam.  What it replaces:
//   ActorCriticNet.train_step  /root/reference/go_ai/models/ac_net.py:67-86
//   ValueNet.train_step        /root/reference/go_ai/models/val_net.py:29-43
//   critic_step / actor_step   /root/reference/go_ai/models/__init__.py:115-134,167-186
//   torch.optim.Adam           /root/reference/train.py:52 (lr 1e-3, weight_decay 1e-4)
// Everything is fp32 like the reference; every reduction runs in a fixed order (partials per row slice, then a
// sequential sum), so a step is reproducible bit for bit.  Activations are channel-last [rows = boards * P][128].
// The 128 -> 128 convolutions (forward and data gradient) are the direct implicit GEMM of conv_kernels.hip with
// weight images rebuilt from the raw parameters every step (k_tr_relayout); the weight gradient is its own
// MFMA kernel (k_tr_wgrad).  Batches are small (the reference trains on 32 positions), so most kernels here are
// latency-bound and written for clarity.
#include "go_board.h"
#include "goai_internal.h"
#include "train.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.1f;

// ---- first convolution: 6 -> 128 from the uint8 planes (ac_net.py:41: the actor-critic net shifts them by -0.5) ----
template <int S>
__global__ __launch_bounds__(256) void k_tr_conv0_fwd(const uint8_t* __restrict__ states, const float* __restrict__ w,
                                                      const float* __restrict__ b, float shift, float* __restrict__ z, int M)
{
    constexpr int P = S * S;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * 128) return;
    const int row = (int)(i >> 7), co = (int)(i & 127);
    const int board = row / P, p = row - board * P, y = p / S, x = p - y * S;
    const uint8_t* st = states + (size_t)board * 6 * P;
    float acc = b[co];
    for (int ci = 0; ci < 6; ci++)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if ((unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S)
                acc = fmaf((float)st[ci * P + yy * S + xx] - shift, w[co * 54 + ci * 9 + tap], acc);
        }
    z[i] = acc;
}

// dW0[co][ci][tap] += sum_rows dz[row][co] * x(row, ci, tap), db0[co] += sum_rows dz[row][co]: one workgroup per
// output channel, 55 running sums per thread over its rows, combined through LDS in a fixed tree.
template <int S>
__global__ __launch_bounds__(256) void k_tr_conv0_bwd(const uint8_t* __restrict__ states, const float* __restrict__ dz,
                                                      float shift, float* __restrict__ dw, float* __restrict__ db, int M)
{
    constexpr int P = S * S;
    __shared__ float red[256][56];
    const int co = blockIdx.x, t = threadIdx.x;
    float acc[55];
#pragma unroll
    for (int k = 0; k < 55; k++) acc[k] = 0.0f;
    for (int row = t; row < M; row += 256) {
        const float d = dz[(size_t)row * 128 + co];
        const int board = row / P, p = row - board * P, y = p / S, x = p - y * S;
        const uint8_t* st = states + (size_t)board * 6 * P;
#pragma unroll
        for (int ci = 0; ci < 6; ci++)
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                const bool ok = (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
                const float xv = ok ? (float)st[ci * P + yy * S + xx] - shift : 0.0f;
                acc[ci * 9 + tap] = fmaf(d, xv, acc[ci * 9 + tap]);
            }
        acc[54] += d;
    }
#pragma unroll
    for (int k = 0; k < 55; k++) red[t][k] = acc[k];
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (t < half)
            for (int k = 0; k < 55; k++) red[t][k] += red[t + half][k];
        __syncthreads();
    }
    if (t < 54) dw[co * 54 + t] += red[0][t];
    if (t == 54) db[co] += red[0][54];
}

// raw conv weights [co][ci][3][3] -> the direct kernel's LDS images [chunk][tap][128 out][32 in]: forward image
// (out = co, in = ci) and data-gradient image (out = ci, in = co, taps mirrored)
__global__ __launch_bounds__(256) void k_tr_relayout(const float* __restrict__ w, float* __restrict__ img_f,
                                                     float* __restrict__ img_d)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 128 * 128 * 9) return;
    const int tap = i % 9, ci = (i / 9) % 128, co = i / (9 * 128);
    const float v = w[i];
    img_f[(((size_t)(ci >> 5) * 9 + tap) * 128 + co) * 32 + (ci & 31)] = v;
    img_d[(((size_t)(co >> 5) * 9 + (8 - tap)) * 128 + ci) * 32 + (co & 31)] = v;
}

// ---- column reductions over [M][128] in a fixed order ----------------------------------------------------------------
// grid = slices; block = 2 x 128 threads (two row phases per channel); partial[slice][k][128], k < NOUT
enum { TR_SUM = 0, TR_SQDEV = 1, TR_BNGRAD = 2, TR_WEIGHTED = 3 };

// TR_WEIGHTED: partial[.][k][c] = sum_rows x[row][c] * a[row * wc + k], k < wc <= 2 (the head convolution's weight gradient)
template <int MODE>
__global__ __launch_bounds__(256) void k_tr_colreduce(const float* __restrict__ x, const float* __restrict__ a,
                                                      const float* __restrict__ z, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, float* __restrict__ partial, int M,
                                                      int rows_per_slice, int wc)
{
    __shared__ float red[2][2][128];
    const int c = threadIdx.x & 127, h = threadIdx.x >> 7;
    const int r0 = blockIdx.x * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
    float s0 = 0.0f, s1 = 0.0f;
    const float mu = (MODE == TR_SQDEV || MODE == TR_BNGRAD) ? mean[c] : 0.0f, is = (MODE == TR_BNGRAD) ? invstd[c] : 0.0f;
    for (int r = r0 + h; r < r1; r += 2) {
        const size_t i = (size_t)r * 128 + c;
        if (MODE == TR_SUM) s0 += x[i];
        else if (MODE == TR_WEIGHTED) {
            s0 = fmaf(x[i], a[r * wc], s0);
            if (wc > 1) s1 = fmaf(x[i], a[r * wc + 1], s1);
        } else if (MODE == TR_SQDEV) { const float d = x[i] - mu; s0 = fmaf(d, d, s0); }
        else {
            const float dm = a[i] > 0.0f ? x[i] : 0.0f;          // ReLU backward fused in
            s0 += dm;
            s1 = fmaf(dm, (z[i] - mu) * is, s1);
        }
    }
    red[0][h][c] = s0;
    red[1][h][c] = s1;
    __syncthreads();
    if (h == 0) {
        partial[((size_t)blockIdx.x * 2 + 0) * 128 + c] = red[0][0][c] + red[0][1][c];
        partial[((size_t)blockIdx.x * 2 + 1) * 128 + c] = red[1][0][c] + red[1][1][c];
    }
}

// finalisers (one block of 128 threads): sum the slices in order
__global__ __launch_bounds__(128) void k_tr_mean_final(const float* __restrict__ partial, int slices, int M, float* __restrict__ mean)
{
    const int c = threadIdx.x;
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < slices; k++) s += partial[((size_t)k * 2) * 128 + c];
    mean[c] = s / (float)M;
}
// biased variance -> invstd for the normalisation; running statistics move by momentum 0.1 with the unbiased variance
__global__ __launch_bounds__(128) void k_tr_var_final(const float* __restrict__ partial, int slices, int M,
                                                      const float* __restrict__ mean, float* __restrict__ invstd,
                                                      float* __restrict__ run_mean, float* __restrict__ run_var)
{
    const int c = threadIdx.x;
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < slices; k++) s += partial[((size_t)k * 2) * 128 + c];
    const float var = s / (float)M;
    invstd[c] = 1.0f / sqrtf(var + BN_EPS);
    run_mean[c] = (1.0f - BN_MOMENTUM) * run_mean[c] + BN_MOMENTUM * mean[c];
    run_var[c] = (1.0f - BN_MOMENTUM) * run_var[c] + BN_MOMENTUM * (s / (float)(M > 1 ? M - 1 : 1));
}
// sums[0][c] = sum dm, sums[1][c] = sum dm * xhat; dgamma += sums[1], dbeta += sums[0]
__global__ __launch_bounds__(128) void k_tr_bngrad_final(const float* __restrict__ partial, int slices, float* __restrict__ sums,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    const int c = threadIdx.x;
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 8
    for (int k = 0; k < slices; k++) {
        s0 += partial[((size_t)k * 2) * 128 + c];
        s1 += partial[((size_t)k * 2 + 1) * 128 + c];
    }
    sums[c] = s0;
    sums[128 + c] = s1;
    dgamma[c] += s1;
    dbeta[c] += s0;
}
__global__ __launch_bounds__(256) void k_tr_headw_final(const float* __restrict__ partial, int slices, float* __restrict__ dw, int wc)
{
    const int c = threadIdx.x & 127, k = threadIdx.x >> 7;
    if (k >= wc) return;
    float s = 0.0f;
#pragma unroll 8
    for (int j = 0; j < slices; j++) s += partial[((size_t)j * 2 + k) * 128 + c];
    dw[k * 128 + c] += s;
}
__global__ __launch_bounds__(128) void k_tr_sum_final(const float* __restrict__ partial, int slices, float* __restrict__ dst)
{
    const int c = threadIdx.x;
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < slices; k++) s += partial[((size_t)k * 2) * 128 + c];
    dst[c] += s;
}

// a = relu(gamma * (z - mean) * invstd + beta (+ resid))
__global__ __launch_bounds__(256) void k_tr_bn_fwd(const float* __restrict__ z, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ resid,
                                                   float* __restrict__ a, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i & 127);
    float v = fmaf(gamma[c], (z[i] - mean[c]) * invstd[c], beta[c]);
    if (resid) v += resid[i];
    a[i] = fmaxf(v, 0.0f);
}

// dz = gamma * invstd * (dm - s0 / M - xhat * s1 / M), dm = d where a > 0; dres (optional) = dm: the residual branch's share
__global__ __launch_bounds__(256) void k_tr_bn_bwd(const float* __restrict__ d, const float* __restrict__ a,
                                                   const float* __restrict__ z, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                   const float* __restrict__ sums, float* __restrict__ dz,
                                                   float* __restrict__ dres, long n, float inv_m)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i & 127);
    const float dm = a[i] > 0.0f ? d[i] : 0.0f;
    const float xh = (z[i] - mean[c]) * invstd[c];
    dz[i] = gamma[c] * invstd[c] * (dm - sums[c] * inv_m - xh * sums[128 + c] * inv_m);
    if (dres) dres[i] = dm;
}

// ---- weight gradient of a 128 -> 128 3x3 convolution -----------------------------------------------------------------
// dW[co][ci][tap] = sum_rows dz[row][co] * a_in[row + off(tap)][ci] over the rows whose tap stays on the board.
// Workgroup (tap, row slice): 4 waves, wave w owns output channels 32w .. 32w+31 (two 16-row MFMA tiles) x all 128
// input channels (eight 16-column tiles): 16 accumulators; K = the slice's rows, four per v_mfma_f32_16x16x4_f32.
// partial[slice][tap][co][ci]; k_tr_wgrad_reduce sums the slices in order into the raw [co][ci][3][3] gradient.
template <int S>
__global__ __launch_bounds__(256) void k_tr_wgrad(const float* __restrict__ a_in, const float* __restrict__ dz,
                                                  float* __restrict__ partial, int M, int rows_per_slice)
{
    constexpr int P = S * S;
    const int tap = blockIdx.x, slice = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1, off = dy * S + dx;
    const int r0 = slice * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
    f32x4 acc[2][8];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 8; u++) acc[t][u] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    for (int r = r0; r < r1; r += 4) {
        const int row = r + lq;
        bool ok = row < r1;
        const int p = row % P, y = p / S, x = p - y * S;
        const bool src_ok = ok && (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S;
        float av[2], bv[8];
#pragma unroll
        for (int t = 0; t < 2; t++) av[t] = ok ? dz[(size_t)row * 128 + (wave * 2 + t) * 16 + li] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) bv[u] = src_ok ? a_in[(size_t)(row + off) * 128 + u * 16 + li] : 0.0f;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int u = 0; u < 8; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[u], acc[t][u], 0, 0, 0);
    }
    // C/D layout: col = lane & 15 (input channel), row = 4 * (lane >> 4) + reg (output channel)
    float* dst = partial + ((size_t)slice * 9 + tap) * 128 * 128;
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                dst[(size_t)((wave * 2 + t) * 16 + 4 * lq + e) * 128 + u * 16 + li] = acc[t][u][e];
}

__global__ __launch_bounds__(256) void k_tr_wgrad_reduce(const float* __restrict__ partial, int slices, float* __restrict__ dw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;          // (tap, co, ci)
    if (i >= 9 * 128 * 128) return;
    const int ci = i & 127, co = (i >> 7) & 127, tap = i >> 14;
    float s = 0.0f;
#pragma unroll 8
    for (int k = 0; k < slices; k++) s += partial[(size_t)k * 9 * 128 * 128 + i];
    dw[((size_t)co * 128 + ci) * 9 + tap] += s;
}

// ---- heads ------------------------------------------------------------------------------------------------------------
// 1x1 convolution 128 -> C (C = 2 actor, 1 critic): hz[row][c]
__global__ __launch_bounds__(256) void k_tr_head_conv_fwd(const float* __restrict__ a, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ hz, int M, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * C) return;
    const int row = i / C, c = i - row * C;
    float acc = b[c];
    for (int k = 0; k < 128; k++) acc = fmaf(a[(size_t)row * 128 + k], w[c * 128 + k], acc);
    hz[i] = acc;
}

// block-wide sum of one float per thread, fixed tree (1024 threads)
__device__ __forceinline__ float block_sum_1024(float v, float* red)
{
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int half = 512; half >= 1; half >>= 1) {
        if (t < half) red[t] += red[t + half];
        __syncthreads();
    }
    const float s = red[0];
    __syncthreads();
    return s;
}

// train-mode BatchNorm + ReLU over the C head channels (one workgroup), flattened like nn.Flatten of [B, C, S, S]:
// f[b][c * P + p]; saves mean / invstd, moves the running statistics
__global__ __launch_bounds__(1024) void k_tr_head_bn_fwd(const float* __restrict__ hz, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ mean,
                                                         float* __restrict__ invstd, float* __restrict__ run_mean,
                                                         float* __restrict__ run_var, float* __restrict__ f, int M, int C, int P)
{
    __shared__ float red[1024];
    for (int c = 0; c < C; c++) {
        float s = 0.0f;
        for (int r = threadIdx.x; r < M; r += 1024) s += hz[r * C + c];
        const float mu = block_sum_1024(s, red) / (float)M;
        float q = 0.0f;
        for (int r = threadIdx.x; r < M; r += 1024) { const float d = hz[r * C + c] - mu; q = fmaf(d, d, q); }
        const float ss = block_sum_1024(q, red);
        const float is = 1.0f / sqrtf(ss / (float)M + BN_EPS);
        if (threadIdx.x == 0) {
            mean[c] = mu;
            invstd[c] = is;
            run_mean[c] = (1.0f - BN_MOMENTUM) * run_mean[c] + BN_MOMENTUM * mu;
            run_var[c] = (1.0f - BN_MOMENTUM) * run_var[c] + BN_MOMENTUM * (ss / (float)(M > 1 ? M - 1 : 1));
        }
        for (int r = threadIdx.x; r < M; r += 1024) {
            const int b = r / P, p = r - b * P;
            f[(size_t)b * C * P + c * P + p] = fmaxf(fmaf(gamma[c], (hz[r * C + c] - mu) * is, beta[c]), 0.0f);
        }
    }
}

// backward of the above: df [B][C*P] -> dhz [M][C]; dgamma / dbeta accumulate
__global__ __launch_bounds__(1024) void k_tr_head_bn_bwd(const float* __restrict__ df, const float* __restrict__ f,
                                                         const float* __restrict__ hz, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ dbias, float* __restrict__ dhz, int M, int C, int P)
{
    __shared__ float red[1024];
    for (int c = 0; c < C; c++) {
        const float mu = mean[c], is = invstd[c];
        float s0 = 0.0f, s1 = 0.0f;
        for (int r = threadIdx.x; r < M; r += 1024) {
            const int b = r / P, p = r - b * P;
            const size_t j = (size_t)b * C * P + c * P + p;
            const float dm = f[j] > 0.0f ? df[j] : 0.0f;
            s0 += dm;
            s1 = fmaf(dm, (hz[r * C + c] - mu) * is, s1);
        }
        s0 = block_sum_1024(s0, red);
        s1 = block_sum_1024(s1, red);
        if (threadIdx.x == 0) { dgamma[c] += s1; dbeta[c] += s0; }
        const float inv_m = 1.0f / (float)M;
        float sb = 0.0f;
        for (int r = threadIdx.x; r < M; r += 1024) {
            const int b = r / P, p = r - b * P;
            const size_t j = (size_t)b * C * P + c * P + p;
            const float dm = f[j] > 0.0f ? df[j] : 0.0f;
            const float xh = (hz[r * C + c] - mu) * is;
            const float v = gamma[c] * is * (dm - s0 * inv_m - xh * s1 * inv_m);
            dhz[r * C + c] = v;
            sb += v;
        }
        sb = block_sum_1024(sb, red);        // the 1x1 convolution's bias gradient (zero in exact arithmetic)
        if (threadIdx.x == 0) dbias[c] += sb;
    }
}

// dA[row][k] = sum_c dhz[row][c] * w[c][k]: the gradient that enters the trunk from this head
__global__ __launch_bounds__(256) void k_tr_head_conv_bwd_x(const float* __restrict__ dhz, const float* __restrict__ w,
                                                            float* __restrict__ dA, int M, int C)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * 128) return;
    const int row = (int)(i >> 7), k = (int)(i & 127);
    float s = 0.0f;
    for (int c = 0; c < C; c++) s = fmaf(dhz[row * C + c], w[c * 128 + k], s);
    dA[i] = s;
}

// out[b][o] = bias[o] + sum_j f[b][j] * W[o][j] (+ ReLU)
__global__ __launch_bounds__(256) void k_tr_linear_fwd(const float* __restrict__ f, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int O,
                                                       int J, int relu)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * O) return;
    const int b = i / O, o = i - b * O;
    float acc = bias[o];
    for (int j = 0; j < J; j++) acc = fmaf(f[(size_t)b * J + j], W[(size_t)o * J + j], acc);
    out[i] = relu ? fmaxf(acc, 0.0f) : acc;
}
// dW[o][j] += sum_b dout[b][o] * f[b][j]; db[o] += sum_b dout[b][o]
__global__ __launch_bounds__(256) void k_tr_linear_bwd_w(const float* __restrict__ dout, const float* __restrict__ f,
                                                         float* __restrict__ dW, float* __restrict__ db, int B, int O, int J)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= O * (J + 1)) return;
    const int o = i / (J + 1), j = i - o * (J + 1);
    float s = 0.0f;
    if (j < J) {
        for (int b = 0; b < B; b++) s = fmaf(dout[b * O + o], f[(size_t)b * J + j], s);
        dW[(size_t)o * J + j] += s;
    } else {
        for (int b = 0; b < B; b++) s += dout[b * O + o];
        db[o] += s;
    }
}
// df[b][j] = sum_o dout[b][o] * W[o][j], masked by the ReLU of the layer's input when `act` is given (act > 0)
__global__ __launch_bounds__(256) void k_tr_linear_bwd_x(const float* __restrict__ dout, const float* __restrict__ W,
                                                         const float* __restrict__ act, float* __restrict__ df, int B, int O, int J)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * J) return;
    const int b = i / J, j = i - b * J;
    float s = 0.0f;
    for (int o = 0; o < O; o++) s = fmaf(dout[b * O + o], W[(size_t)o * J + j], s);
    df[i] = (act && !(act[i] > 0.0f)) ? 0.0f : s;
}

// ---- losses (one workgroup) -------------------------------------------------------------------------------------------
// critic_step (models/__init__.py:126-133): vals = tanh(val_logits); loss = mse(vals, target); acc = mean(sign(vals) == target)
// dval = weight * d loss / d val_logits.  metrics[0] = loss, metrics[1] = accuracy.
__global__ __launch_bounds__(256) void k_tr_critic_loss(const float* __restrict__ val, const float* __restrict__ wins,
                                                        float* __restrict__ dval, float* __restrict__ metrics, int B, float weight)
{
    __shared__ float sl[256], sa[256];
    float l = 0.0f, a = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float v = tanhf(val[b]), t = -wins[b], e = v - t;   // ac_net.py:76 / val_net.py:38: the target is -wins
        l = fmaf(e, e, l);
        const float sg = (v > 0.0f) - (v < 0.0f);
        a += (sg == t) ? 1.0f : 0.0f;
        dval[b] = weight * 2.0f * e / (float)B * (1.0f - v * v);
    }
    sl[threadIdx.x] = l;
    sa[threadIdx.x] = a;
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (threadIdx.x < half) { sl[threadIdx.x] += sl[threadIdx.x + half]; sa[threadIdx.x] += sa[threadIdx.x + half]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { metrics[0] = sl[0] / (float)B; metrics[1] = sa[0] / (float)B; }
}

// actor_step (models/__init__.py:175-184): cross_entropy(pi_logits, greedy), accuracy of argmax(pi_logits);
// dlogits = (softmax - onehot) / B.  One thread per sample.  metrics[2] = loss, metrics[3] = accuracy.
__global__ __launch_bounds__(256) void k_tr_actor_loss(const float* __restrict__ logits, const float* __restrict__ pi,
                                                       float* __restrict__ dlogits, float* __restrict__ metrics, int B, int A)
{
    __shared__ float sl[256], sa[256];
    float l = 0.0f, acc = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* x = logits + (size_t)b * A;
        float m = x[0];
        int arg = 0;
        for (int a = 1; a < A; a++) if (x[a] > m) { m = x[a]; arg = a; }
        float s = 0.0f;
        for (int a = 0; a < A; a++) s += expf(x[a] - m);
        const float lse = m + logf(s);
        int g = 0;                                   // greedy = argmax(pi), first maximum
        for (int a = 1; a < A; a++) if (pi[(size_t)b * A + a] > pi[(size_t)b * A + g]) g = a;
        l += lse - x[g];
        acc += (arg == g) ? 1.0f : 0.0f;
        for (int a = 0; a < A; a++) dlogits[(size_t)b * A + a] = (expf(x[a] - lse) - (a == g ? 1.0f : 0.0f)) / (float)B;
    }
    sl[threadIdx.x] = l;
    sa[threadIdx.x] = acc;
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (threadIdx.x < half) { sl[threadIdx.x] += sl[threadIdx.x + half]; sa[threadIdx.x] += sa[threadIdx.x + half]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { metrics[2] = sl[0] / (float)B; metrics[3] = sa[0] / (float)B; }
}

// torch.optim.Adam with L2 weight decay folded into the gradient (betas 0.9 / 0.999, eps 1e-8); bc1 / bc2 are the
// bias corrections 1 - beta^t of this step.  The step's scalars come from device memory (scal: lr, weight decay, bc1,
// sqrt(bc2)), so that the launch is the same from step to step (the step as a HIP graph, train_host.hip).
__global__ __launch_bounds__(256) void k_tr_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, long n, const float* __restrict__ scal)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = scal[0], wd = scal[1], bc1 = scal[2], sqrt_bc2 = scal[3];
    const float gi = fmaf(wd, p[i], g[i]);
    const float mi = 0.9f * m[i] + 0.1f * gi;
    const float vi = 0.999f * v[i] + 0.001f * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrt_bc2 + 1e-8f);
}

// dst[0..n) = v (the step's zero_grad and its NaN metrics as a launch: memset / host-memcpy nodes of a captured stream
// gave garbage gradients from the second replay on, on this ROCm)
__global__ __launch_bounds__(256) void k_tr_fill(float* __restrict__ dst, long n, float v)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v;
}

// ---- launchers --------------------------------------------------------------------------------------------------------
#define TR_CHECK() do { GOAI_HIP_CHECK(hipGetLastError()); return GOAI_OK; } while (0)
static inline int nblk(long n) { return (int)((n + 255) / 256); }

int tr_conv0_fwd(int S_rt, const uint8_t* states, const float* w, const float* b, float shift, float* z, int M, hipStream_t st)
{
    GOAI_DISPATCH_SIZE(S_rt, k_tr_conv0_fwd<S><<<nblk((long)M * 128), 256, 0, st>>>(states, w, b, shift, z, M));
    TR_CHECK();
}
int tr_conv0_bwd(int S_rt, const uint8_t* states, const float* dz, float shift, float* dw, float* db, int M, hipStream_t st)
{
    GOAI_DISPATCH_SIZE(S_rt, k_tr_conv0_bwd<S><<<128, 256, 0, st>>>(states, dz, shift, dw, db, M));
    TR_CHECK();
}
int tr_relayout(const float* w, float* img_f, float* img_d, hipStream_t st)
{
    k_tr_relayout<<<nblk(128 * 128 * 9), 256, 0, st>>>(w, img_f, img_d);
    TR_CHECK();
}
static inline int tr_slices(int M, int& rows_per_slice)
{
    rows_per_slice = (M + TR_MAX_SLICES - 1) / TR_MAX_SLICES;
    if (rows_per_slice < 32) rows_per_slice = 32;
    rows_per_slice = (rows_per_slice + 3) & ~3;
    return (M + rows_per_slice - 1) / rows_per_slice;
}
int tr_bn_stats(const float* z, float* partial, float* mean, float* invstd, float* run_mean, float* run_var, int M, hipStream_t st)
{
    int rps;
    const int sl = tr_slices(M, rps);
    k_tr_colreduce<TR_SUM><<<sl, 256, 0, st>>>(z, nullptr, nullptr, nullptr, nullptr, partial, M, rps, 0);
    k_tr_mean_final<<<1, 128, 0, st>>>(partial, sl, M, mean);
    k_tr_colreduce<TR_SQDEV><<<sl, 256, 0, st>>>(z, nullptr, nullptr, mean, nullptr, partial, M, rps, 0);
    k_tr_var_final<<<1, 128, 0, st>>>(partial, sl, M, mean, invstd, run_mean, run_var);
    TR_CHECK();
}
int tr_bn_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* resid,
              float* a, int M, hipStream_t st)
{
    k_tr_bn_fwd<<<nblk((long)M * 128), 256, 0, st>>>(z, mean, invstd, gamma, beta, resid, a, (long)M * 128);
    TR_CHECK();
}
int tr_bn_bwd(const float* d, const float* a, const float* z, const float* mean, const float* invstd, const float* gamma,
              float* partial, float* sums, float* dgamma, float* dbeta, float* dz, float* dres, int M, hipStream_t st)
{
    int rps;
    const int sl = tr_slices(M, rps);
    k_tr_colreduce<TR_BNGRAD><<<sl, 256, 0, st>>>(d, a, z, mean, invstd, partial, M, rps, 0);
    k_tr_bngrad_final<<<1, 128, 0, st>>>(partial, sl, sums, dgamma, dbeta);
    k_tr_bn_bwd<<<nblk((long)M * 128), 256, 0, st>>>(d, a, z, mean, invstd, gamma, sums, dz, dres, (long)M * 128, 1.0f / (float)M);
    TR_CHECK();
}
int tr_colsum_add(const float* x, float* partial, float* dst, int M, hipStream_t st)
{
    int rps;
    const int sl = tr_slices(M, rps);
    k_tr_colreduce<TR_SUM><<<sl, 256, 0, st>>>(x, nullptr, nullptr, nullptr, nullptr, partial, M, rps, 0);
    k_tr_sum_final<<<1, 128, 0, st>>>(partial, sl, dst);
    TR_CHECK();
}
int tr_wgrad(int S_rt, const float* a_in, const float* dz, float* partial, float* dw, int M, hipStream_t st)
{
    int rps;
    const int sl = tr_slices(M, rps);
    GOAI_DISPATCH_SIZE(S_rt, k_tr_wgrad<S><<<dim3(9, sl), 256, 0, st>>>(a_in, dz, partial, M, rps));
    k_tr_wgrad_reduce<<<nblk(9 * 128 * 128), 256, 0, st>>>(partial, sl, dw);
    TR_CHECK();
}
int tr_head_fwd(const float* a, const float* w, const float* b, const float* gamma, const float* beta, float* hz, float* mean,
                float* invstd, float* run_mean, float* run_var, float* f, int M, int C, int P, hipStream_t st)
{
    k_tr_head_conv_fwd<<<nblk((long)M * C), 256, 0, st>>>(a, w, b, hz, M, C);
    k_tr_head_bn_fwd<<<1, 1024, 0, st>>>(hz, gamma, beta, mean, invstd, run_mean, run_var, f, M, C, P);
    TR_CHECK();
}
int tr_head_bwd(const float* df, const float* f, const float* hz, const float* mean, const float* invstd, const float* gamma,
                const float* a, const float* w, float* partial, float* dgamma, float* dbeta, float* dw, float* db, float* dhz,
                float* dA, int M, int C, int P, hipStream_t st)
{
    int rps;
    const int sl = tr_slices(M, rps);
    k_tr_head_bn_bwd<<<1, 1024, 0, st>>>(df, f, hz, mean, invstd, gamma, dgamma, dbeta, db, dhz, M, C, P);
    k_tr_colreduce<TR_WEIGHTED><<<sl, 256, 0, st>>>(a, dhz, nullptr, nullptr, nullptr, partial, M, rps, C);
    k_tr_headw_final<<<1, 256, 0, st>>>(partial, sl, dw, C);
    k_tr_head_conv_bwd_x<<<nblk((long)M * 128), 256, 0, st>>>(dhz, w, dA, M, C);
    TR_CHECK();
}
int tr_linear_fwd(const float* f, const float* W, const float* bias, float* out, int B, int O, int J, int relu, hipStream_t st)
{
    k_tr_linear_fwd<<<nblk((long)B * O), 256, 0, st>>>(f, W, bias, out, B, O, J, relu);
    TR_CHECK();
}
int tr_linear_bwd(const float* dout, const float* f, const float* W, const float* act_mask, float* dW, float* db, float* df, int B,
                  int O, int J, hipStream_t st)
{
    k_tr_linear_bwd_w<<<nblk((long)O * (J + 1)), 256, 0, st>>>(dout, f, dW, db, B, O, J);
    if (df) k_tr_linear_bwd_x<<<nblk((long)B * J), 256, 0, st>>>(dout, W, act_mask, df, B, O, J);
    TR_CHECK();
}
int tr_critic_loss(const float* val, const float* wins, float* dval, float* metrics, int B, float weight, hipStream_t st)
{
    k_tr_critic_loss<<<1, 256, 0, st>>>(val, wins, dval, metrics, B, weight);
    TR_CHECK();
}
int tr_actor_loss(const float* logits, const float* pi, float* dlogits, float* metrics, int B, int A, hipStream_t st)
{
    k_tr_actor_loss<<<1, 256, 0, st>>>(logits, pi, dlogits, metrics, B, A);
    TR_CHECK();
}
int tr_fill(float* dst, long n, float v, hipStream_t st)
{
    k_tr_fill<<<nblk(n), 256, 0, st>>>(dst, n, v);
    TR_CHECK();
}
int tr_adam(float* p, const float* g, float* m, float* v, long n, const float* scal, hipStream_t st)
{
    k_tr_adam<<<nblk(n), 256, 0, st>>>(p, g, m, v, n, scal);
    TR_CHECK();
}

}  // namespace goai
