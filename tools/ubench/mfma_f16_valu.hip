// Microbenchmark 3: do v_mfma_f32_16x16x32_f16 and vector instructions of the same SIMD overlap?
// Per stage and wave: NM MFMAs (CHAIN = 3: chains of three on one accumulator, as the f16x3 trunk kernel issues them;
// CHAIN = 1: each on its own accumulator of a ring of six), then NV v_pk_fma_f32 that do not depend on them.  Two waves
// per SIMD (512 threads per workgroup, one workgroup per CU).  ns per stage: MFMAs alone, vector work alone, both.
//   hipcc -O2 --offload-arch=gfx950 -o mfma_f16_valu mfma_f16_valu.hip && ./mfma_f16_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int CHAIN>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed)
{
    f32x4 m[6];
    f32x2 z[8];
    for (int i = 0; i < 6; i++) m[i] = f32x4{ 0, 0, 0, 0 };
    for (int i = 0; i < 8; i++) z[i] = f32x2{ (float)i, 1.0f };
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    const float f = seed * 1e-3f;
    const f32x2 f2 = { f, f };
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NM; n++) {
            const int acc = CHAIN == 3 ? (n / 3) % 6 : n % 6;
            m[acc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, m[acc], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int n = 0; n < NV; n++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8]) : "v"(f2), "v"(f2));
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += z[i][0] + z[i][1];
    for (int i = 0; i < 6; i++) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int CHAIN>
static float run(float* out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NV, CHAIN>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    return best * 1e6f / iters;
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    printf("per stage and wave: 18 MFMAs 16x16x32 f16, 54 v_pk_fma_f32; two waves per SIMD; ns per stage (2.4 GHz: 1 ns = 2.4 cycles)\n");
    printf("chains of 3:   MFMA only %.1f   vector only %.1f   both %.1f\n", run<18, 0, 3>(out), run<0, 54, 3>(out), run<18, 54, 3>(out));
    printf("independent:   MFMA only %.1f   vector only %.1f   both %.1f\n", run<18, 0, 1>(out), run<0, 54, 1>(out), run<18, 54, 1>(out));
    printf("36 MFMAs:      chains %.1f   independent %.1f\n", run<36, 0, 3>(out), run<36, 0, 1>(out));
    return 0;
}
