// f16x3_accuracy.hip -- is a dot product of K = 1152 fp32 values formed from TWO fp16 parts per operand (three fp16
// MFMAs: a0 w0 + a0 w1 + a1 w0, fp32 accumulate) as accurate as the THREE bf16 parts / six bf16 MFMAs of
// conv_wino3s_kernels.hip, and how do both compare with a plain fp32 fma chain?  Reference: double.
// One wave per 16x16 output tile; operands in the MFMA 16x16x32 layouts (A: row = lane & 15, k = 8 (lane >> 4) ..+7;
// B: col = lane & 15, same k; D: col = lane & 15, row = 4 (lane >> 4) + reg).
//   hipcc -O2 --offload-arch=gfx950 -o f16x3_accuracy f16x3_accuracy.hip && ./f16x3_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 1152, M = 16, N = 16, TILES = 64;

__device__ inline unsigned pack_hi(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }

// mode 0: fp32 fma chain; 1: bf16 x 6; 2: fp16 x 3 (a scaled by sa, w by sw: powers of two); 3: fp16 x 3 with round-toward-zero parts
__global__ void k_dot(const float* A, const float* W, float* D, int mode, float sa, float sw)
{
    const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4, tile = blockIdx.x;
    const float* a = A + (size_t)tile * M * K;
    const float* w = W + (size_t)tile * N * K;
    f32x4 acc = { 0, 0, 0, 0 };
    if (mode == 0) {
        for (int r = 0; r < 4; r++) {
            const int row = 4 * lq + r;
            float s = 0.0f;
            for (int k = 0; k < K; k++) s = fmaf(a[row * K + k], w[li * K + k], s);
            acc[r] = s;
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            float av[8], wv[8];
            for (int e = 0; e < 8; e++) {
                av[e] = a[li * K + k0 + 8 * lq + e];
                wv[e] = w[li * K + k0 + 8 * lq + e];
            }
            if (mode == 1) {
                bf16x8 ap[3], wp[3];
                for (int e = 0; e < 8; e++) {
                    float x = av[e];
                    for (int part = 0; part < 3; part++) {
                        const unsigned h = __float_as_uint(x) & 0xffff0000u;
                        ap[part][e] = __builtin_bit_cast(__bf16, (unsigned short)(h >> 16));
                        x -= __uint_as_float(h);
                    }
                    x = wv[e];
                    for (int part = 0; part < 3; part++) {
                        const unsigned h = __float_as_uint(x) & 0xffff0000u;
                        wp[part][e] = __builtin_bit_cast(__bf16, (unsigned short)(h >> 16));
                        x -= __uint_as_float(h);
                    }
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[2], wp[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], wp[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], wp[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1], wp[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], wp[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0], wp[0], acc, 0, 0, 0);
            } else {
                f16x8 ap[2], wp[2];
                for (int e = 0; e < 8; e++) {
                    const float x = av[e] * sa, y = wv[e] * sw;
                    if (mode == 2) {
                        ap[0][e] = (_Float16)x;
                        ap[1][e] = (_Float16)(x - (float)ap[0][e]);
                        wp[0][e] = (_Float16)y;
                        wp[1][e] = (_Float16)(y - (float)wp[0][e]);
                    } else {
                        ap[0][e] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, (_Float16)x)));
                        // truncation: step towards zero when the rounded value is larger in magnitude
                        if (fabsf((float)ap[0][e]) > fabsf(x)) ap[0][e] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, ap[0][e]) - 1));
                        ap[1][e] = (_Float16)(x - (float)ap[0][e]);
                        wp[0][e] = (_Float16)y;
                        if (fabsf((float)wp[0][e]) > fabsf(y)) wp[0][e] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, wp[0][e]) - 1));
                        wp[1][e] = (_Float16)(y - (float)wp[0][e]);
                    }
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap[1], wp[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap[0], wp[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap[0], wp[0], acc, 0, 0, 0);
            }
        }
        if (mode >= 2)
            for (int r = 0; r < 4; r++) acc[r] *= 1.0f / (sa * sw);
    }
    // D[row][col]: mode 0 computed (row = 4 lq + r, col = li); MFMA D = A(16 x K) * B(K x 16): row index comes from A's lane & 15
    // -> D(row = 4 lq + r, col = li) = sum_k A[row][k] W[col][k] in both cases
    for (int r = 0; r < 4; r++) D[(size_t)tile * M * N + (4 * lq + r) * N + li] = acc[r];
}

int main()
{
    std::mt19937_64 rng(12345);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    for (int dist = 0; dist < 3; dist++) {
        std::vector<float> A((size_t)TILES * M * K), W((size_t)TILES * N * K);
        float amax = 0.0f, wmax = 0.0f;
        for (auto& v : A) {
            float x = nd(rng);
            if (dist == 0) x = x > 0 ? x : 0.0f;                        // post-ReLU like
            if (dist == 1) x = x * std::exp(2.0f * nd(rng));            // wide dynamic range
            if (dist == 2) x = (x > 0 ? x : 0.0f) * 37.0f + ((rng() % 97) == 0 ? 3000.0f : 0.0f);   // outliers far above the bulk
            v = x;
            amax = std::fmax(amax, std::fabs(x));
        }
        for (auto& v : W) {
            v = 0.04f * nd(rng) * (dist == 1 ? std::exp(1.5f * nd(rng)) : 1.0f);
            wmax = std::fmax(wmax, std::fabs(v));
        }
        int ea, ew;
        std::frexp(amax, &ea);                     // amax in [2^(ea-1), 2^ea)
        std::frexp(wmax, &ew);
        const float sa = std::ldexp(1.0f, 15 - ea), sw = std::ldexp(1.0f, 15 - ew);      // largest magnitude in [2^14, 2^15)
        float *dA, *dW, *dD;
        hipMalloc(&dA, A.size() * 4);
        hipMalloc(&dW, W.size() * 4);
        hipMalloc(&dD, (size_t)TILES * M * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> ref((size_t)TILES * M * N), mag((size_t)TILES * M * N);
        for (int t = 0; t < TILES; t++)
            for (int r = 0; r < M; r++)
                for (int c = 0; c < N; c++) {
                    double s = 0, m = 0;
                    for (int k = 0; k < K; k++) {
                        const double p = (double)A[((size_t)t * M + r) * K + k] * (double)W[((size_t)t * N + c) * K + k];
                        s += p;
                        m += p * p;
                    }
                    ref[((size_t)t * M + r) * N + c] = s;
                    mag[((size_t)t * M + r) * N + c] = std::sqrt(m);       // the scale rounding errors of the sum live on
                }
        const char* names[4] = { "fp32 fma chain", "bf16 x 6 MFMA ", "fp16 x 3 MFMA ", "fp16 x 3 (rtz)" };
        // a second fp16 run with the scale 2^6 lower (the kernel's bound on transformed values is 36 x the raw maximum)
        for (int mode = 0; mode < 5; mode++) {
            const int md = mode == 4 ? 2 : mode;
            const float sa_m = mode == 4 ? sa / 64.0f : sa;
            k_dot<<<TILES, 64>>>(dA, dW, dD, md, sa_m, sw);
            std::vector<float> D((size_t)TILES * M * N);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, sq = 0;
            for (size_t i = 0; i < D.size(); i++) {
                const double e = std::fabs((double)D[i] - ref[i]) / mag[i];
                worst = std::fmax(worst, e);
                sq += e * e;
            }
            printf("dist %d  %s%s  error / |terms|_2: rms %.3e  max %.3e\n", dist, mode == 4 ? "fp16 x 3, a / 64" : names[md], mode == 4 ? "" : "  ",
                   std::sqrt(sq / D.size()), worst);
        }
        hipFree(dA); hipFree(dW); hipFree(dD);
    }
    return 0;
}
