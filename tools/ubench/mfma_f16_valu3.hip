// Microbenchmark 5 (round 3): can one wave's vector work run beside ANOTHER wave's v_mfma_f32_16x16x32_f16 on the same SIMD?
// 512 threads = two waves per SIMD.  (a) roles split: half the waves issue only MFMAs, the other half only vector
// instructions (split by wave >> 2 or by wave & 1: whichever pairs the two kinds on one SIMD shows the interaction);
// (b) every wave does 21 MFMAs then its vector work (the stage of conv_wino2s_kernels.hip), with and without a raised
// priority (s_setprio) for half the waves, and with half the waves starting with the vector work (phase-shifted).
//   hipcc -O2 --offload-arch=gfx950 -o mfma_f16_valu3 mfma_f16_valu3.hip && ./mfma_f16_valu3
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// MODE 0: roles by wave >> 2; 1: roles by wave & 1; 2: all waves MFMAs then vector; 3: as 2, s_setprio 1 for waves >> 2 == 1;
// 4: as 2, waves >> 2 == 1 begin with the vector work; 5: as 3 with prio by wave & 1; 6: as 4 by wave & 1
template <int NM, int NF, bool PK, int MODE>
__global__ void k(float* out, int iters, float seed)
{
    f32x4 m[7];
    f32x2 z[8];
    for (int i = 0; i < 7; i++) m[i] = f32x4{ 0, 0, 0, 0 };
    for (int i = 0; i < 8; i++) z[i] = f32x2{ (float)i, 1.0f };
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    const float f = seed * 1e-3f;
    const f32x2 f2 = { f, f };
    const int wave = threadIdx.x >> 6;
    const int half = (MODE == 1 || MODE == 5 || MODE == 6) ? (wave & 1) : (wave >> 2);
    auto mfmas = [&]() {
#pragma unroll
        for (int n = 0; n < NM; n++) m[(n / 3) % 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, m[(n / 3) % 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto valu = [&]() {
#pragma unroll
        for (int n = 0; n < NF; n++) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8]) : "v"(f2), "v"(f2));
            else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8][0]) : "v"(f), "v"(f));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8][1]) : "v"(f), "v"(f));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if ((MODE == 3 || MODE == 5) && half) __builtin_amdgcn_s_setprio(1);
    if (MODE <= 1) {
        if (half == 0) {
#pragma unroll 1
            for (int it = 0; it < iters; it++) mfmas();
        } else {
#pragma unroll 1
            for (int it = 0; it < iters; it++) valu();
        }
    } else {
        if ((MODE == 4 || MODE == 6) && half) valu();
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
            mfmas();
            valu();
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += z[i][0] + z[i][1];
    for (int i = 0; i < 7; i++) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NF, bool PK, int MODE>
static float run(float* out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NF, PK, MODE>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
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
    printf("two waves per SIMD; ns per iteration; MFMA = 21 x v_mfma_f32_16x16x32_f16 per wave, vector = 32 v_pk_fma_f32 or 64 v_fma_f32 per wave\n");
    printf("(a) roles split, half the waves each: MFMA waves alone %.1f / %.1f (by wave>>2 / wave&1)\n", run<21, 0, true, 0>(out), run<21, 0, true, 1>(out));
    printf("    packed waves alone %.1f / %.1f   plain waves alone %.1f / %.1f\n", run<0, 32, true, 0>(out), run<0, 32, true, 1>(out),
           run<0, 32, false, 0>(out), run<0, 32, false, 1>(out));
    printf("    MFMA waves + packed waves %.1f / %.1f   MFMA waves + plain waves %.1f / %.1f\n", run<21, 32, true, 0>(out), run<21, 32, true, 1>(out),
           run<21, 32, false, 0>(out), run<21, 32, false, 1>(out));
    printf("    MFMA waves + 2x packed %.1f / %.1f   + 2x plain %.1f / %.1f\n", run<21, 64, true, 0>(out), run<21, 64, true, 1>(out),
           run<21, 64, false, 0>(out), run<21, 64, false, 1>(out));
    printf("(b) every wave MFMAs then vector work: packed %.1f  plain %.1f   (MFMAs only %.1f, packed only %.1f, plain only %.1f)\n",
           run<21, 32, true, 2>(out), run<21, 32, false, 2>(out), run<21, 0, true, 2>(out), run<0, 32, true, 2>(out), run<0, 32, false, 2>(out));
    printf("    half the waves at s_setprio 1 (wave>>2 / wave&1): packed %.1f / %.1f  plain %.1f / %.1f\n", run<21, 32, true, 3>(out), run<21, 32, true, 5>(out),
           run<21, 32, false, 3>(out), run<21, 32, false, 5>(out));
    printf("    half the waves phase-shifted (wave>>2 / wave&1): packed %.1f / %.1f  plain %.1f / %.1f\n", run<21, 32, true, 4>(out), run<21, 32, true, 6>(out),
           run<21, 32, false, 4>(out), run<21, 32, false, 6>(out));
    return 0;
}
