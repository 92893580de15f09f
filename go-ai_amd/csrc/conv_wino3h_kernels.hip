// conv_wino3h_kernels.hip -- round 3's F(3x3,3x3) trunk kernel for 9x9 boards on the fp16 matrix cores
// (GOAI_DTYPE_F16X3: two fp16 parts per operand, three MFMAs per product tile; GOAI_DTYPE_F16: one part, one MFMA --
// the reduced-precision tier).  Same arithmetic, operand split, per-board power-of-two scales, LDS hand-over between the
// convolutions of a launch and heads-in-the-last-epilogue as conv_wino3s_kernels.hip (whose NP = 2 form it replaces and
// whose header explains all of that); bit-identical results (tests/test_gpu_nn.py).  What is new is WHERE the input
// transform runs.
//
// The fact behind it (tools/ubench/mfma_f16_valu2.hip, MI355X): a v_mfma_f32_16x16x32_f16 holds a SIMD's vector issue
// for 8 of its 16 cycles.  Two plain (not packed) vector instructions per MFMA issue in the other 8 for free; a third
// and fourth cost their full 4 cycles each, and a v_pk_fma_f32 beside an MFMA costs 13.  In conv_wino3s_kernels.hip
// the input transform of a row (Bt d B of the 5x5 patches, ~150 vector instructions and 20 LDS reads per lane, a chain of
// dependent LDS round trips) runs between two barriers with the matrix cores idle: ten phases of 1.5-1.9 us per convolution,
// 19 of its 52 us.  Here
//   * the image regions are ZERO-BORDERED (rows of ten cells: one zero cell + nine; a zero row above every board), so a
//     patch is one base address + immediates: no address table, no table read in front of the patch reads;
//   * the row half of the transform of the NEXT row (patch reads and r[b] = sum_a Bt[p][a] d[a][b]) is issued in the gaps
//     between the MFMAs of the CURRENT row's stages, as plain v_fma_f32 / v_sub_f32, one patch column per stage: its LDS
//     latency and its vector issue disappear under the MFMAs; only the column half + fp16 split + V writes (no LDS read in
//     front of them) stay between the barriers;
//   * a 16-lane LDS read group is one wtile x both chunk regions (the regions are an odd number of cells apart, i.e.
//     always in different bank halves), so the 36 wtiles are 32 "main" ones = the first pass of the eight waves (V rows
//     0..31) + the 4 "tail" ones (wtile 8 of every board, V rows 32..35) = a second pass of one wave.
// LDS: 2 x 52,608 B image regions + V 47,360 B (NP = 2) + 32 B = 152,608 B.
#include "conv_wino3h_body.h"

namespace goai {

// Same launch contract as k_conv3x3_wino3s (wino3h_trunk, conv_wino3h_body.h).
template <int NP>
__global__ __launch_bounds__(512) void k_conv3x3_wino3h(float* x, float* y, const float* resid0, const uint4* wt_all,
                                                        const float* __restrict__ bias_all, int nboards, int n_convs,
                                                        const int* __restrict__ wexp_all, float* amax_io, float amax_first,
                                                        const float* __restrict__ head_w, const float* __restrict__ head_b,
                                                        float* __restrict__ head_out)
{
    // workgroups of a several-convolution launch start in sixteen groups 0.37 us apart, so that their memory phases do not all
    // fall into the same microseconds: a convolution's residual reads and output stores are 85 MB over the chip, 10 us at the
    // full HBM rate if every workgroup is in its epilogue at once (conv_wino3s_kernels.hip; profiles/r02_wino3s_ablation.txt).
    // bench.py C2 launch, round 3 (profiles/r03_wino3h_notes.txt sections 9, 12): with the weight loads ONE half-stage ahead
    // no groups 289 us, round 2's 4 x 3.9 us 277, 16 x 1.5 us 273.9; with them TWO ahead (W3H_WAHEAD) the exposed weight
    // latency no longer multiplies the contention: no groups 262.5 us, 16 x 1.5 us 270, 16 x 0.5 us 258.9, 16 x 0.37 us 258.2
#ifndef W3H_START_GROUPS
#define W3H_START_GROUPS 16
#define W3H_START_SLEEP 12
#endif
#ifndef W3H_BIG_SLEEP
#define W3H_BIG_SLEEP 0
#endif
    // (that is for a launch of one workgroup per CU, the lock-step search's.  With more workgroups than CUs -- the Q search's
    // 8192 -- no workgroup is delayed: while the weight loads ran one half-stage ahead, round 2's four groups 3.9 us apart for
    // every workgroup were worth 2 % there (W3H_BIG_SLEEP = 127); with them two ahead they cost 8 %: bench.py C3 0.196 of peak
    // with the delays, 0.213 without)
    if (n_convs > 1 && W3H_START_GROUPS > 1) {
        if (gridDim.x <= 256) for (int i = (blockIdx.x >> 3) % W3H_START_GROUPS; i > 0; i--) __builtin_amdgcn_s_sleep(W3H_START_SLEEP);
        else if (W3H_BIG_SLEEP > 0) for (int i = (blockIdx.x >> 3) % 4; i > 0; i--) __builtin_amdgcn_s_sleep(W3H_BIG_SLEEP);
    }
#define W3H_WG blockIdx.x
#include "conv_wino3h_body.inc"
#undef W3H_WG
}

// wt_split: [n_convs][100 stages][8 waves][parts][64 lanes][8 fp16] (nn_host.hip), parts = 2 (f16x3) or 1 (f16); wexp[n_convs]:
// the weights' scale exponents; the bound of the first convolution's input: amax_first > 0 or amax_io[nboards]; amax_io (may
// be NULL) receives the output's bounds.  Any number of boards (4 per workgroup).  head_out: see launch_conv3x3_wino3s.
int launch_conv3x3_wino3h(int S_rt, float* x, float* y, const float* resid0, const void* wt_split, const float* bias, int nboards,
                          int n_convs, hipStream_t st, int parts, const int* wexp, float* amax_io, float amax_first,
                          const float* head_w, const float* head_b, float* head_out)
{
    if (S_rt != 9 || nboards < 1 || (size_t)nboards * 81 * 512 >= ((size_t)1 << 32)) {
        set_error("the fp16 F(3x3,3x3) convolution runs 9x9 boards, fewer than 103,563 per launch");
        return GOAI_ERR_ARG;
    }
    if ((parts != 1 && parts != 2) || !wexp || !(amax_first > 0.0f || amax_io)) {
        set_error("the fp16 F(3x3,3x3) convolution takes one or two fp16 parts, its weight exponents and a bound of its input");
        return GOAI_ERR_ARG;
    }
    if (n_convs < 1 || (n_convs > 1 && ((n_convs & 1) || resid0))) {
        set_error("a multi-convolution launch is a whole residual trunk: an even count, no outer residual");
        return GOAI_ERR_ARG;
    }
    const dim3 grid((nboards + 3) / 4), block(512);
    const uint4* w = reinterpret_cast<const uint4*>(wt_split);
    if (parts == 2)
        k_conv3x3_wino3h<2><<<grid, block, 0, st>>>(x, y, resid0, w, bias, nboards, n_convs, wexp, amax_io, amax_first, head_w, head_b, head_out);
    else
        k_conv3x3_wino3h<1><<<grid, block, 0, st>>>(x, y, resid0, w, bias, nboards, n_convs, wexp, amax_io, amax_first, head_w, head_b, head_out);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
