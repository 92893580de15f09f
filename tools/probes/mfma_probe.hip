// Micro-probe: cycles per f32 MFMA as a function of the accumulator interleave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, unsigned long long* cyc, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, int nacc, int waves_per_simd)
{
    float* out; unsigned long long* cyc;
    int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;
    hipMalloc(&out, blocks * threads * sizeof(float)); hipMalloc(&cyc, blocks * 8);
    kern<<<blocks, threads>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); kern<<<blocks, threads>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double n = (double)iters * 8 * nacc;
    printf("%s nacc=%d waves/simd=%d: %.1f shader cycles per MFMA per wave; wall %.3f ms -> %.1f cyc/MFMA/SIMD @2.4GHz-equivalent\n", name, nacc, waves_per_simd,
           h[7] / n, ms, ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run("16x16x4", k16<1>, 1, 1); run("16x16x4", k16<2>, 2, 1); run("16x16x4", k16<4>, 4, 1); run("16x16x4", k16<14>, 14, 1);
    run("16x16x4", k16<2>, 2, 2); run("16x16x4", k16<14>, 14, 3);
    run("32x32x2", k32<1>, 1, 1); run("32x32x2", k32<2>, 2, 1); run("32x32x2", k32<4>, 4, 1); run("32x32x2", k32<4>, 4, 2);
    return 0;
}
