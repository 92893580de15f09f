// Microbenchmark 2: which vector instructions overlap with fp32 MFMAs of the same SIMD?
// Per stage and wave: 24 x v_mfma_f32_16x16x4_f32 (three chains), then NV instructions of one kind that do
// NOT depend on the MFMA results.  Two waves per SIMD.  If a kind overlapped with the partner wave's
// MFMAs the stage would stay at the MFMA-only time.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int KIND, bool MFMA>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed)
{
    f32x4 m[3];
    float y[16];
    f32x2 z[8];
    int u[16];
    const float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int i = 0; i < 3; i++) m[i] = f32x4{ 0, 0, 0, 0 };
    for (int i = 0; i < 16; i++) { y[i] = (float)i; u[i] = i + threadIdx.x; }
    for (int i = 0; i < 8; i++) z[i] = f32x2{ (float)i, 1.0f };
    const float f = seed * 1e-3f;
    const f32x2 f2 = { f, f };
    __shared__ float lds[4096];
    lds[threadIdx.x] = a;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (MFMA) {
#pragma unroll
            for (int n = 0; n < 24; n++) {
                m[n % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[n % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int n = 0; n < NV; n++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y[n % 16]) : "v"(f), "v"(a));
            if (KIND == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(y[n % 16]) : "v"(f));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8]) : "v"(f2), "v"(f2));
            if (KIND == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[n % 16]) : "v"(n));
            if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(u[n % 16]) : "v"(u[(n + 1) % 16]));
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(z[n % 8]) : "v"(f2));
            if (KIND == 6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(y[n % 16]) : "v"(f));
            if (KIND == 7) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[n % 16]) : "v"(n));
            if (KIND == 8) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(z[n % 8]) : "v"(f2));
            if (KIND == 9) asm volatile("v_fma_f16 %0, %1, %2, %0" : "+v"(y[n % 16]) : "v"(f), "v"(a));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += y[i] + (float)u[i];
    for (int i = 0; i < 8; i++) s += z[i][0] + z[i][1];
    for (int i = 0; i < 3; i++) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x * 7) & 4095];
}

template <int NV, int KIND, bool MFMA>
static float run(float* out)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, KIND, MFMA>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    return best * 1e6f / iters;
}

template <int KIND>
static void kind(const char* name, float* out, float base)
{
    const float alone = run<96, KIND, false>(out), with = run<96, KIND, true>(out);
    printf("%-14s 96/stage/wave: alone %6.1f ns  with 24 MFMA %6.1f ns  (MFMA only %6.1f)  added %6.1f = %.2f of alone\n", name, alone,
           with, base, with - base, (with - base) / alone);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const float base = run<0, 0, true>(out);
    kind<0>("v_fma_f32", out, base);
    kind<1>("v_add_f32", out, base);
    kind<6>("v_mul_f32", out, base);
    kind<2>("v_pk_fma_f32", out, base);
    kind<5>("v_pk_add_f32", out, base);
    kind<3>("v_add_u32", out, base);
    kind<7>("v_xor_b32", out, base);
    kind<4>("v_mov_b32", out, base);
    kind<8>("v_fma_f64", out, base);
    kind<9>("v_fma_f16", out, base);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
