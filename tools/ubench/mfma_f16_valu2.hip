// Microbenchmark 4 (round 3): v_mfma_f32_16x16x32_f16 beside PLAIN vector instructions.
// mfma_f16_valu.hip (round 2) measured 18 MFMAs followed by 54 v_pk_fma_f32 per stage and wave and found their times to
// add up exactly.  MI355X_MICROARCH.md says an MFMA 16x16x32 holds the SIMD's vector issue for 8 of its 16 cycles and that
// PACKED fp32 instructions are an anti-lever beside MFMAs.  This one measures the same arithmetic (108 fmas per lane,
// stage and wave) as v_pk_fma_f32 (54) or v_fma_f32 (108), clumped behind the MFMAs or spread between them (PER per MFMA),
// two waves per SIMD (512 threads, one workgroup per CU) or one (256 threads).
//   hipcc -O2 --offload-arch=gfx950 -o mfma_f16_valu2 mfma_f16_valu2.hip && ./mfma_f16_valu2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// NM MFMAs per stage; NF fma "units" per stage (a unit = 2 fmas per lane = one v_pk_fma_f32 or two v_fma_f32);
// PK: packed or plain; SPREAD: 0 = all vector work behind the MFMAs, else that many units after every MFMA
template <int NM, int NF, bool PK, int SPREAD>
__global__ void k(float* out, int iters, float seed)
{
    f32x4 m[6];
    f32x2 z[8];
    for (int i = 0; i < 6; i++) m[i] = f32x4{ 0, 0, 0, 0 };
    for (int i = 0; i < 8; i++) z[i] = f32x2{ (float)i, 1.0f };
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    const float f = seed * 1e-3f;
    const f32x2 f2 = { f, f };
    auto unit = [&](int n) {
        if (PK) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8]) : "v"(f2), "v"(f2));
        } else {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8][0]) : "v"(f), "v"(f));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[n % 8][1]) : "v"(f), "v"(f));
        }
    };
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        int done = 0;
#pragma unroll
        for (int n = 0; n < NM; n++) {
            const int acc = (n / 3) % 6;
            m[acc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, m[acc], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (SPREAD) {
#pragma unroll
                for (int s = 0; s < SPREAD; s++)
                    if (done < NF) unit(done++);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (; done < NF; done++) unit(done);
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += z[i][0] + z[i][1];
    for (int i = 0; i < 6; i++) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NF, bool PK, int SPREAD>
static float run(float* out, int threads)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NF, PK, SPREAD>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
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
    for (int threads : { 512, 256 }) {
        printf("%d waves per SIMD; per stage and wave 18 MFMAs 16x16x32 f16 and 108 fmas per lane; ns per stage\n", threads / 256);
        printf("  MFMA only %.1f\n", run<18, 0, true, 0>(out, threads));
        printf("  v_pk_fma_f32 x54:   alone %.1f   behind the MFMAs %.1f   3 per MFMA %.1f\n", run<0, 54, true, 0>(out, threads),
               run<18, 54, true, 0>(out, threads), run<18, 54, true, 3>(out, threads));
        printf("  v_fma_f32 x108:     alone %.1f   behind the MFMAs %.1f   6 per MFMA %.1f\n", run<0, 54, false, 0>(out, threads),
               run<18, 54, false, 0>(out, threads), run<18, 54, false, 3>(out, threads));
        printf("  v_fma_f32 x36 (2 per MFMA): alone %.1f  spread %.1f ;  x72 (4 per MFMA): alone %.1f  spread %.1f\n",
               run<0, 18, false, 0>(out, threads), run<18, 18, false, 1>(out, threads), run<0, 36, false, 0>(out, threads),
               run<18, 36, false, 2>(out, threads));
        printf("  v_pk_fma_f32 x18 (1 per MFMA): alone %.1f  spread %.1f ;  x36 (2 per MFMA): alone %.1f  spread %.1f\n",
               run<0, 18, true, 0>(out, threads), run<18, 18, true, 1>(out, threads), run<0, 36, true, 0>(out, threads),
               run<18, 36, true, 2>(out, threads));
    }
    return 0;
}
