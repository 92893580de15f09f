// detmath.h -- bit-reproducible exp / tanh / softmax pieces for the search kernels.
//
// The reference takes softmax(pi_logits) in fp32 (/root/reference/go_ai/search/mct.py:41),
// softmax(-child values) in fp64 (/root/reference/go_ai/search/tree.py:152-157) and
// np.tanh of every backed-up value in fp64 (tree.py:180) from numpy/scipy SIMD routines
// whose last bits depend on the host CPU.  The search kernels instead use the functions
// below, built only from IEEE-754 +, -, *, /, fma and round-to-nearest-even (this file is
// compiled with -ffp-contract=off), so that a CPU restatement of the same operation
// sequence (the test oracle) matches the GPU bit for bit.  Accuracy versus numpy:
// <= 2 ulp (exp32), <= 3 ulp (tanh64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace goai {

__device__ __forceinline__ float det_exp32(float x)
{
    if (!(x >= -104.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float n = rintf(x * 0x1.715476p+0f);
    float r = fmaf(n, -0x1.62e4p-1f, x);
    r = fmaf(n, -0x1.7f7d1cp-20f, r);
    float p = 0x1.a01a02p-13f;
    p = fmaf(p, r, 0x1.6c16c2p-10f);
    p = fmaf(p, r, 0x1.111112p-7f);
    p = fmaf(p, r, 0x1.555556p-5f);
    p = fmaf(p, r, 0x1.555556p-3f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int ni = (int)n;
    if (ni >= -126) return p * __uint_as_float((uint32_t)(ni + 127) << 23);
    return (p * __uint_as_float((uint32_t)(ni + 100 + 127) << 23)) * 0x1p-100f;
}

__device__ __forceinline__ double det_exp64(double x)
{
    if (!(x >= -746.0)) return 0.0;
    if (x > 709.0) x = 709.0;
    const double n = rint(x * 0x1.71547652b82fep+0);
    double r = fma(n, -0x1.62e42fee00000p-1, x);
    r = fma(n, -0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = fma(p, r, 0x1.1eed8eff8d898p-29);
    p = fma(p, r, 0x1.ae64567f544e4p-26);
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma(p, r, 0x1.71de3a556c734p-19);
    p = fma(p, r, 0x1.a01a01a01a01ap-16);
    p = fma(p, r, 0x1.a01a01a01a01ap-13);
    p = fma(p, r, 0x1.6c16c16c16c17p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ni = (int)n;
    if (ni >= -1022) return p * __longlong_as_double((long long)((uint64_t)(ni + 1023) << 52));
    return (p * __longlong_as_double((long long)((uint64_t)(ni + 1000 + 1023) << 52))) * 0x1p-1000;
}

__device__ __forceinline__ double det_expm1_small(double y)   // 0 <= y < 1
{
    double p = 0x1.71b8ef6dcf572p-66;
    p = fma(p, y, 0x1.e542ba4020225p-62);
    p = fma(p, y, 0x1.2f49b46814157p-57);
    p = fma(p, y, 0x1.6827863b97d97p-53);
    p = fma(p, y, 0x1.952c77030ad4ap-49);
    p = fma(p, y, 0x1.ae7f3e733b81fp-45);
    p = fma(p, y, 0x1.ae7f3e733b81fp-41);
    p = fma(p, y, 0x1.93974a8c07c9dp-37);
    p = fma(p, y, 0x1.6124613a86d09p-33);
    p = fma(p, y, 0x1.1eed8eff8d898p-29);
    p = fma(p, y, 0x1.ae64567f544e4p-26);
    p = fma(p, y, 0x1.27e4fb7789f5cp-22);
    p = fma(p, y, 0x1.71de3a556c734p-19);
    p = fma(p, y, 0x1.a01a01a01a01ap-16);
    p = fma(p, y, 0x1.a01a01a01a01ap-13);
    p = fma(p, y, 0x1.6c16c16c16c17p-10);
    p = fma(p, y, 0x1.1111111111111p-7);
    p = fma(p, y, 0x1.5555555555555p-5);
    p = fma(p, y, 0x1.5555555555555p-3);
    p = fma(p, y, 0.5);
    p = fma(p, y, 1.0);
    return y * p;
}

__device__ __forceinline__ double det_tanh64(double x)
{
    const double a = fabs(x);
    double r;
    if (a != a) return x;
    if (a < 0x1p-28) return x;
    if (a >= 20.0) r = 1.0;
    else if (a < 0.5) { const double t = det_expm1_small(2.0 * a); r = t / (t + 2.0); }
    else { const double e = det_exp64(2.0 * a); r = 1.0 - 2.0 / (e + 1.0); }
    return x < 0 ? -r : r;
}

// Wave-wide reductions with a fixed association order (xor butterfly 32,16,...,1): every
// lane ends with the same bits, equal to a halving tree over the 64 lane values.
__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// numpy's pairwise float64 summation (what np.sum / np.mean do on a contiguous array).
__device__ inline double np_pairwise_sum(const double* a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

// Philox4x32-10 (Salmon et al. 2011): counter-based generator, one independent stream
// per (seed, game id, step) so that sampled actions do not depend on how games are
// sharded over GPUs.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int round = 0; round < 10; round++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace goai
