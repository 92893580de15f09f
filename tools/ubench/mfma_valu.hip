// Microbenchmark: how do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and v_fma_f32 of the SAME and of the
// PARTNER wave share a SIMD?  Each wave loops over "stages" of 24 MFMAs (three independent accumulator
// chains) and NV independent v_fma, either clumped after the MFMAs or spread between them.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu mfma_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>   // MODE 0: clump, 1: spread evenly, 2: VALU only (no MFMA), 3: clump, partner wave half a stage out of phase
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed, int skew)
{
    f32x4 m[3];
    float y[48];
    const float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int i = 0; i < 3; i++) m[i] = f32x4{ 0, 0, 0, 0 };
    for (int i = 0; i < 48; i++) y[i] = (float)i;
    const float f = seed * 1e-3f;
    if (MODE == 3 && (threadIdx.x >> 6) >= 4) {
        // partner wave starts `skew` MFMAs late
#pragma unroll 1
        for (int n = 0; n < skew; n++) m[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[0], 0, 0, 0);
    }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int n = 0; n < 24; n++) {
                m[n % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[n % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int n = 0; n < NV; n++) {
                y[n % 48] = fmaf(f, m[n % 3][n & 3], y[n % 48]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 1) {
            constexpr int PER = NV / 24;
            f32x4 mp[3] = { m[0], m[1], m[2] };
#pragma unroll
            for (int n = 0; n < 24; n++) {
                m[n % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[n % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < PER; v++) {
                    const int idx = n * PER + v;
                    y[idx % 48] = fmaf(f, mp[idx % 3][idx & 3], y[idx % 48]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NV; n++) {
                y[n % 48] = fmaf(f, a, y[n % 48]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 48; i++) s += y[i];
    for (int i = 0; i < 3; i++) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int MODE>
static void run(const char* name, int threads, float* out, int skew = 0)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, MODE>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f, skew);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-10s NV=%3d waves/SIMD=%d skew=%2d  %.1f ns/stage\n", name, NV, threads / 256, skew, best * 1e6f / iters);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    for (int threads : { 256, 512 }) {
        run<0, 0>("mfma-only", threads, out);
        run<48, 2>("valu-only", threads, out);
        run<96, 2>("valu-only", threads, out);
        run<144, 2>("valu-only", threads, out);
        run<48, 0>("clump", threads, out);
        run<96, 0>("clump", threads, out);
        run<144, 0>("clump", threads, out);
        run<48, 1>("spread", threads, out);
        run<96, 1>("spread", threads, out);
        run<144, 1>("spread", threads, out);
        if (threads == 512)
            for (int skew : { 0, 6, 12, 18, 24, 36 }) {
                run<48, 3>("clump-skew", threads, out, skew);
                run<96, 3>("clump-skew", threads, out, skew);
                run<144, 3>("clump-skew", threads, out, skew);
            }
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
