// conv_wino3s_kernels.hip -- the F(3x3,3x3) trunk of conv_wino3p_kernels.hip with its 25 transform-domain
// GEMMs on the BF16 matrix cores, every fp32 operand SPLIT INTO THREE bf16 parts (GOAI_DTYPE_BF16X6).
//
// Why: the fp32 MFMA runs on the vector ALUs' own multipliers (same peak, 157 TF, and no overlap with
// vector instructions on a SIMD: tools/ubench/mfma_valu*.hip), so in conv_wino3p_kernels.hip the
// Winograd transforms, which are vector work, add their full issue time to the products'.  The bf16
// MFMA is separate hardware (2.5 PF dense, 16 times the fp32 rate) that runs beside vector
// instructions.  An fp32 number is EXACTLY the sum of three bf16 numbers (3 x 8 significant bits:
// h = top 16 bits of x, m = top 16 bits of x - h, l = x - h - m), so with U = u0 + u1 + u2 (folded
// weights, split once on the host) and V = v0 + v1 + v2 (transformed inputs, split by the threads that
// compute them)
//       U V  =  u0 v0 + (u0 v1 + u1 v0) + (u0 v2 + u1 v1 + u2 v0)  +  terms below 2^-24 |U V|
// every bf16 product is exact in the fp32 accumulator and the three dropped terms are below the
// rounding of a single fp32 product: fp32-grade results from 6 bf16 MFMAs of K = 32 per 16x16 tile
// (96 matrix-pipe cycles) where the fp32 path issues 8 MFMAs of K = 4 (256 cycles on the vector ALUs).
//
// Structure = conv_wino3p_kernels.hip (whole boards per workgroup, two 32-channel chunks per output
// transform, the whole residual trunk in one launch, fp32 input / output transforms on the vector ALUs):
//   [zero cell 128 B][image chunk A 41,472 B] [zero cell][image chunk B]          fp32, cell * 128 B
//   V[2 chunks][5 q][3 parts][36 rows + zero row][32 ch] bf16                       71 KB, single-buffered
//   Tab[36 wtiles][5 patch rows] -> five 16-bit cell addresses
// = 157 KB.  The 36 wtile rows of 4 boards are two full 16-column MFMA tiles and a third whose other
// 12 columns read the zero row (the bf16 pipe has cycles to spare; no 4x4 blocks here).
// Weights: [conv][100 stages][8 waves][3 parts][64 lanes][8 bf16] -- a lane's A operand (its output
// channel, its 8 input channels) as one 16-byte load per part.
//
// NP = 2 (GOAI_DTYPE_F16X3): the same kernel with every operand as TWO fp16 parts and three fp16 MFMAs per tile step,
//       U V  =  u0 v0 + (u0 v1 + u1 v0)  +  terms below 2^-22 |U V|
// (h = fp16(x), l = fp16(x - h): 11 + 11 significant bits and the sign of l).  Measured against fp64 it is as close
// as the six bf16 products and as an fp32 fma chain (tools/ubench/f16x3_accuracy.hip: the fp32 accumulation dominates
// all three), for half the matrix work, two thirds of the LDS and weight traffic and a cheaper split.  fp16 has five
// exponent bits, so both operands are SCALED BY POWERS OF TWO (exact) into its range:
//   weights: per convolution, largest |U| in [2^14, 2^15) -- on the host, exponent in wexp_all[conv];
//   inputs:  per BOARD and convolution, from an upper bound `amax` of the board's input activations (all >= 0):
//            |Bt d B| <= 36 amax, so amax in [2^E, 2^(E+1)) is scaled by 2^(9 - E): transformed values below 36864.
//            The bound of a convolution's input is the maximum its producer stored: the previous epilogue of this
//            launch (kept in LDS), the previous launch (amax_io, per board) or, for the first convolution of the
//            trunk, a bound on conv0's output computed from its weights (amax_first: the input planes are 0 / 1).
//            Transformed values are scaled as they are split, the accumulators are scaled back (v_ldexp_f32) in the
//            epilogue, which also leaves the next convolution's first two chunks (input channels 0..63 = the output
//            channels of waves 0..3) in the LDS image regions: only the launch's first convolution stages them from memory.
// Parts that fall below fp16's normal range (2^-14) lose bits, at an absolute level of 2^-25 of a scaled value range
// of 2^15: nothing an fp32 sum can see.  The scale of a board depends on that board alone, so its results do not
// depend on what shares its workgroup or its batch.
#include <type_traits>
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int S, int NP>
struct Wino3S {
    static_assert(S == 9, "row mapping: 8 + 1 wtiles per board, 4 boards");
    static constexpr int T = S / 3, T2 = T * T, P = S * S;
    static constexpr int NBW = 4;                                  // whole boards per workgroup
    static constexpr int WT = NBW * T2;                            // real wtiles per workgroup
    static constexpr int MT = (WT + 15) / 16;                      // MFMA column tiles of 16 wtile rows (the last: 4 real)
    static constexpr int VROWS = WT + 1;                           // the wtile rows + one zero row
    static constexpr int IMG_CH_U4 = 16 + NBW * P * 8;             // 16-byte units per chunk region: two zero cells + cells
    static constexpr int V_PART_U4 = VROWS * 4;                    // one (chunk, q, part): [VROWS][32 ch] bf16 = 64 B per row
    static constexpr int V_U4 = 2 * 5 * NP * V_PART_U4;            // [2 chunks][5 q][NP parts]
    static constexpr int TAB_U4 = WT * 5;                          // [wtile][a] -> 8 x u16 (5 used)
    static constexpr int NI = (NBW * P * 8 + 511) / 512;
    static constexpr int AMAX_U4 = 2;                              // [2][4 boards] activation bounds as fp32 bits (NP == 2)
    static constexpr int LDS_U4 = 2 * IMG_CH_U4 + V_U4 + TAB_U4 + AMAX_U4;
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(IMG_CH_U4 * 16 < 65536, "cell addresses and the chunk offset are 16-bit");
};

// Diagnostic builds only (tools/wino3s_stamps.sh -DW3S_DIAG=<mask>, tools/wino3s_ablate_run.sh): leave a part of
// the loop out to see what it costs: 1 weight loads (after the first), 2 output transform (the compiler then drops
// the MFMAs too), 4 V fragment reads (after the first), 8 input transform and split, 16 the MFMAs.  Results are
// wrong, timings are the point.  32 / 128: every patch read of the input transform from a fixed address pattern
// (two addresses per wave / two whole cells per 16-lane LDS group), 64: V writes at lane * 8 bytes.
#ifndef W3S_DIAG
#define W3S_DIAG 0
#endif
#ifdef GOAI_CONV_STAMPS
// diagnostic build only (tools/wino3p_stamps.py): s_memrealtime (100 MHz) stamps of the SECOND convolution of
// a launch, 32 per workgroup: 0 layer top, 1 image staged, 2 + 2r transform of row r done (r = 5 cp + p),
// 3 + 2r its stages done, 22 epilogue done; 23 / 24 = s_memtime (shader clock) at stamps 0 / 22
__device__ unsigned long long g_wino3s_stamps[32 * 1024];
extern "C" int goai_debug_wino3s_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino3s_stamps), (size_t)n * sizeof(unsigned long long));
}
#define W3S_STAMP(i) if (threadIdx.x == 0 && k == 1) { g_wino3s_stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0 || (i) == 22) g_wino3s_stamps[blockIdx.x * 32 + 23 + ((i) == 22)] = __builtin_amdgcn_s_memtime(); }
#else
#define W3S_STAMP(i)
#endif

__device__ __forceinline__ f32x4 s_fma4(float c, f32x4 a, f32x4 b)
{
    return f32x4{ fmaf(c, a[0], b[0]), fmaf(c, a[1], b[1]), fmaf(c, a[2], b[2]), fmaf(c, a[3], b[3]) };
}
// A V row is 64 B = four 16-byte quarters (8 channels each).  An MFMA fragment read (ds_read_b128, lane (li, lq) = row li,
// quarter lq) is served in groups of 16 lanes = rows {0-3, 12-15} at one quarter + rows {4-11} at the next: with the
// quarters in place these touch 32 of the 64 banks twice.  Quarter q of row r therefore sits at position q ^ w3s_vswz(r),
// which spreads every such group over all 64 banks (SQ_LDS_BANK_CONFLICT: 44 % of the LDS cycles before).
__device__ __forceinline__ int w3s_vswz(int row) { return (0 - (row >> 2)) & 3; }
// two fp32 bit patterns whose low halves are dropped -> one word of two bf16 (a in the low half)
__device__ __forceinline__ unsigned s_pack(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }


// Same launch contract as k_conv3x3_wino3 (n_convs convolutions in a row on the same boards).  NP == 2 only:
// wexp_all[conv] = the weights' scale exponent; amax_first > 0: bound of the first convolution's input, else
// amax_io[board]; amax_io != NULL: receives the bound of the last convolution's output, per board.
template <int S, int NP>
__global__ __launch_bounds__(512) void k_conv3x3_wino3s(float* x, float* y, const float* resid0,
                                                        const uint4* wt_all,
                                                        const float* __restrict__ bias_all,
                                                        const uint4* __restrict__ tab_g, int nboards, int n_convs,
                                                        const int* __restrict__ wexp_all, float* amax_io, float amax_first,
                                                        const float* __restrict__ head_w, const float* __restrict__ head_b,
                                                        float* __restrict__ head_out)
{
    using W = Wino3S<S, NP>;
    constexpr int P = W::P, MT = W::MT;
    __shared__ f32x4 lds[W::LDS_U4];
    f32x4* Img = lds;                                  // two chunk regions
    f32x4* Vs = lds + 2 * W::IMG_CH_U4;
    uint4* Tab = reinterpret_cast<uint4*>(lds + 2 * W::IMG_CH_U4 + W::V_U4);
    unsigned* Amax = reinterpret_cast<unsigned*>(lds + 2 * W::IMG_CH_U4 + W::V_U4 + W::TAB_U4);      // [2][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * W::NBW;
    const float* in = x;
    const float* resid = resid0;
    float* out = y;

    // All workgroups reach their memory phases (epilogue stores, residual and image loads: 85 MB per convolution) at the same
    // microsecond and wait for one another's bytes.  Starting them in W3S_GROUPS groups, W3S_SKEW x 64 cycles apart, spreads
    // those phases (measured, six-convolution launch of 1024 boards: 336 us -> 328 us with 4 groups 3.9 us apart, the 12 us
    // the last group starts late included; profiles/r02_wino3s_ablation.txt).  Only launches of several convolutions.
#ifndef W3S_SKEW
#define W3S_SKEW 127
#define W3S_GROUPS 4
#endif
#ifndef W3S_REPS
#define W3S_REPS 1
#endif
    if (n_convs > 1)
        for (int i = ((blockIdx.x >> 3) % W3S_GROUPS) * W3S_REPS; i > 0; i--) __builtin_amdgcn_s_sleep(W3S_SKEW);
    for (int i = tid; i < W::V_U4; i += 512) Vs[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };   // the zero rows stay zero
    if (tid < 32) Img[(tid >> 4) * W::IMG_CH_U4 + (tid & 15)] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };   // the zero cells of both regions
    for (int e = tid; e < W::TAB_U4; e += 512) Tab[e] = tab_g[e];
    if (NP == 2 && tid < 4) {
        const bool on = b_first + tid < nboards;
        Amax[tid] = __float_as_uint(amax_first > 0.0f ? amax_first : (on ? amax_io[b_first + tid] : 0.0f));
    }

    // Input-transform jobs: one (wtile, chunk region, 4-channel slot) item per lane and pass.  The lanes are laid
    // out for the LDS: a ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
    // and the same + 32, MI355X_MICROARCH.md, LDS), one cycle each when the 16 lanes touch 64 different banks.  A
    // group here = the 8 slots of TWO CONSECUTIVE wtiles: a patch cell is 128 B = half of the banks, and the same patch
    // position of consecutive wtiles is an odd number of cells apart (3 or 21, also across a board edge), i.e. in the
    // other half (off-board positions go to one of two zero cells, by the parity the cell would have had).
    // 36 wtiles x 2 regions = 36 such groups G (region G / 18, wtiles 2 (G % 18) and + 1) = nine waves of four: wave w
    // takes G = 4 w + g in a first pass with all of its lanes at work, wave 0 the last four groups in a second pass
    // (with one item per lane for both regions, 288 of 512 lanes worked and every wave ran the instruction stream twice).
    const int l32 = lane & 31;
    const int hgj = l32 < 4 ? l32 : l32 < 12 ? 16 + (l32 - 4) : l32 < 16 ? 4 + (l32 - 12) : l32 < 20 ? 24 + (l32 - 16)
                  : l32 < 28 ? 8 + (l32 - 20) : 28 + (l32 - 28);          // 16 * (group within the half) + position in the group
    const int t_g = (lane >> 5) * 2 + (hgj >> 4), t_w = (hgj >> 3) & 1, ts = hgj & 7;     // group of the wave, wtile of the group, slot
    // an item as LDS byte offsets: its patch base (region + slot), its row of the address table, its V row (region part 0, q 0)
    struct TItem { unsigned img, tab, v; int board; };
    auto t_item = [&](int G) {
        const int tc = G >= 18 ? 1 : 0, ttw = 2 * (G - 18 * tc) + t_w;
        // V row of a wtile: wtiles 0..7 of board b -> row 8 b + wtile (two full column tiles);
        // wtile 8 of board b -> row 32 + b (the third tile; its columns 4..15 read the zero row).
        const int tvrow = (ttw % W::T2 < 8) ? (ttw / W::T2) * 8 + ttw % W::T2 : 32 + ttw / W::T2;
        return TItem{ (unsigned)(tc * (W::IMG_CH_U4 * 16) + ts * 16), (unsigned)((2 * W::IMG_CH_U4 + W::V_U4 + ttw * 5) * 16),
                      (unsigned)((2 * W::IMG_CH_U4 + tc * (5 * NP * W::V_PART_U4)) * 16 + tvrow * 64 +
                                 (((ts >> 1) ^ w3s_vswz(tvrow)) * 16) + (ts & 1) * 8), ttw / W::T2 };
    };
    const TItem it0 = t_item(4 * wave + t_g), it1 = t_item(32 + t_g);

    f32x4 ireg[W::NI];
    auto load_img_to = [&](int chunk, f32x4* reg) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            const int cr = idx >> 3, slot = idx & 7;
            const int board = b_first + cr / P;
            f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (idx < W::NBW * P * 8 && board < nboards)
                v = *reinterpret_cast<const f32x4*>(&in[((long)b_first * P + cr) * 128 + chunk * 32 + slot * 4]);
            reg[i] = v;
        }
    };
    const unsigned* amax_in = Amax;                    // this convolution's input bounds: Amax + 4 (k & 1)
    auto store_img_from = [&](int region, const f32x4* reg) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            if (idx < W::NBW * P * 8) Img[region * W::IMG_CH_U4 + 16 + idx] = reg[i];
        }
    };
    auto load_img = [&](int chunk) { load_img_to(chunk, ireg); };
    auto store_img = [&](int region) { store_img_from(region, ireg); };
    // V[region][q][tw][ts] = (Bt d B)[p][q], q = 0..4, for row p (compile-time after unrolling) of item `it`
    auto transform = [&](int p, const TItem& it) {
        const char* base = reinterpret_cast<const char*>(lds) + it.img;
        const uint4* tabrow = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds) + it.tab);
        float t_scale = 1.0f;                          // NP == 2: the scale of the item's board
        if constexpr (NP == 2) t_scale = __uint_as_float((unsigned)(127 + f16_scale_exp(amax_in[it.board], 36)) << 23);
        auto patch_row = [&](int a, f32x4* d) {
            uint4 t = tabrow[a];
            if (W3S_DIAG & 32) { t.x = (lane & 8) << 4 | (lane & 8) << 20; t.y = t.x; t.z = t.x; }          // two addresses per wave
            if (W3S_DIAG & 128) { t.x = (256u + t_w * 128u) * 0x10001u; t.y = t.x; t.z = t.x; }       // two whole cells per 16-lane group
            d[0] = *reinterpret_cast<const f32x4*>(base + (t.x & 0xffffu));
            d[1] = *reinterpret_cast<const f32x4*>(base + (t.x >> 16));
            d[2] = *reinterpret_cast<const f32x4*>(base + (t.y & 0xffffu));
            d[3] = *reinterpret_cast<const f32x4*>(base + (t.y >> 16));
            d[4] = *reinterpret_cast<const f32x4*>(base + (t.z & 0xffffu));
        };
        f32x4 r[5], d1[5], d2[5], d3[5];
        patch_row(1, d1);
        if (p != 3) patch_row(2, d2);
        patch_row(3, d3);
        if (p == 0) {
            f32x4 d0[5];
            patch_row(0, d0);
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = s_fma4(-2.0f, d2[b], s_fma4(2.0f, d0[b], d3[b] - d1[b]));
        } else if (p == 1) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = s_fma4(-2.0f, d1[b], d3[b] - d2[b]);
        } else if (p == 2) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = s_fma4(-3.0f, d2[b], s_fma4(2.0f, d1[b], d3[b]));
        } else if (p == 3) {
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = d3[b] - d1[b];
        } else {
            f32x4 d4[5];
            patch_row(4, d4);
#pragma unroll
            for (int b = 0; b < 5; b++) r[b] = s_fma4(-2.0f, d3[b], s_fma4(2.0f, d1[b], d4[b] - d2[b]));
        }
        // each fp32 value as three bf16 parts (exact: top 16 bits, top 16 bits of the rest, the rest), 4 channels
        // = 8 bytes per part at [row tvrow][channel 4 ts]
        uint2* V = reinterpret_cast<uint2*>(reinterpret_cast<char*>(lds) + it.v);
        auto put = [&](int q, const f32x4 v) {
            if constexpr (NP == 2) {
                // two fp16 parts, round to nearest: h = fp16(v), l = fp16(v - h) (the difference is exact in fp32)
                // (vs - h by v_fma_mix_f32, which reads h's halves in place: one instruction per value instead of a
                // conversion back and a subtraction -- the vector ALU's issue is what this kernel is bound by)
                const f32x4 vs = v * t_scale;
                const f16x4 h = __builtin_convertvector(vs, f16x4);
                const uint2 hp = __builtin_bit_cast(uint2, h);
                f32x4 rest;
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest[0]) : "v"(hp.x), "v"(vs[0]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest[1]) : "v"(hp.x), "v"(vs[1]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest[2]) : "v"(hp.y), "v"(vs[2]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest[3]) : "v"(hp.y), "v"(vs[3]));
                const f16x4 l = __builtin_convertvector(rest, f16x4);
                uint2* dst = V + q * (2 * W::V_PART_U4 * 2);
                dst[0] = hp;
                dst[W::V_PART_U4 * 2] = __builtin_bit_cast(uint2, l);
            } else {
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                h[e] = __float_as_uint(v[e]) & 0xffff0000u;
                const float r1 = v[e] - __uint_as_float(h[e]);
                m[e] = __float_as_uint(r1) & 0xffff0000u;
                l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));      // <= 8 significant bits: its low half is zero
            }
            uint2* dst = (W3S_DIAG & 64) ? reinterpret_cast<uint2*>(Vs) + lane + q * (3 * W::V_PART_U4 * 2) : V + q * (3 * W::V_PART_U4 * 2);
            dst[0] = uint2{ s_pack(h[0], h[1]), s_pack(h[2], h[3]) };
            dst[W::V_PART_U4 * 2] = uint2{ s_pack(m[0], m[1]), s_pack(m[2], m[3]) };
            dst[(NP - 1) * W::V_PART_U4 * 2] = uint2{ s_pack(l[0], l[1]), s_pack(l[2], l[3]) };
            }
        };
        put(0, s_fma4(-2.0f, r[2], s_fma4(2.0f, r[0], r[3] - r[1])));
        put(1, s_fma4(-2.0f, r[1], r[3] - r[2]));
        put(2, s_fma4(-3.0f, r[2], s_fma4(2.0f, r[1], r[3])));
        put(3, r[3] - r[1]);
        put(4, s_fma4(-2.0f, r[3], s_fma4(2.0f, r[1], r[4] - r[2])));
    };

    // weight fragments of a half-stage: the three bf16 parts of U[stage = (chunk * 5 + p) * 5 + q] in the lanes' own
    // order (nn_host.hip): lane (li, lq) gets output channel 16 wave + li, input channels 8 lq .. 8 lq + 7 of the chunk
    // [stage][wave][part][lane] x 16 B: whole lines per load, as buffer loads: descriptor = this convolution's image,
    // scalar offset = the stage, vector offset = the lane's place in the stage -- no address arithmetic in the loop
    const uint4* wconv = wt_all;
    const unsigned wlane = (wave * (64 * NP) + lane) * 16;        // bytes
    constexpr unsigned W_STAGE_BYTES = 8 * 64 * NP * 16, W_CONV_BYTES = 100 * W_STAGE_BYTES;
    auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wconv), 0, W_CONV_BYTES, 0x00020000);
    u32x4 wf[2][NP];                                       // two half-stages of weights, used alternately
    auto load_b = [&](int stage, u32x4* b) {
#pragma unroll
        for (int r = 0; r < NP; r++)
            b[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024 * r, stage * W_STAGE_BYTES, 0));
        // the optimiser otherwise sinks a load whose value is first used in the next loop iteration down to the
        // loop's end, i.e. right in front of its use (wt_all is deliberately not __restrict__: loads from a
        // noalias pointer may cross this barrier)
        asm volatile("" ::: "memory");
    };

    f32x4 Y[3][3][MT];                               // [i][j]: output (3ty + i, 3tx + j) of the wtile
    const int col = wave * 16 + 4 * lq;
    // V row of this lane's column in each of the three column tiles (the B operand's "column" = a wtile row)
    int vrow[MT];
    vrow[0] = li;
    vrow[1] = 16 + li;
    vrow[2] = li < 4 ? 32 + li : W::WT;              // the zero row
    unsigned vaddr[MT];                              // LDS byte address of the lane's fragment of (chunk 0, q 0, part 0) per column tile
#pragma unroll
    for (int t = 0; t < MT; t++) {
        vaddr[t] = (unsigned)((2 * W::IMG_CH_U4 + (lq ^ w3s_vswz(vrow[t])) + vrow[t] * 4) * 16);
        asm volatile("" : "+v"(vaddr[t]));           // (a register, not a constant to be folded into each read's address again)
    }

#pragma unroll 1
    for (int k = 0; k < n_convs; k++) {
    const float* bias = bias_all + k * 128;
    load_b(0, wf[0]);
    if constexpr (NP == 2) amax_in = Amax + 4 * (k & 1);
    // k = 0: the zero fills and the table are complete; k > 0: every wave's MFMAs of the previous
    // convolution have read V and its stores to `in` are visible to this workgroup
    __syncthreads();
    W3S_STAMP(0);
    if (NP == 2 && tid < 4) Amax[4 * ((k + 1) & 1) + tid] = 0u;      // collects this convolution's output bounds (epilogue)
    if (NP != 2 || k == 0) {                         // (NP == 2, k > 0: the previous epilogue left chunks 0 and 1 in LDS)
        f32x4 ireg1[W::NI];                          // both chunks in flight at once: the accumulators are not live yet
        load_img(0);
        load_img_to(1, ireg1);
        store_img(0);
        store_img_from(1, ireg1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int t = 0; t < MT; t++) Y[i][j][t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    W3S_STAMP(1);

#pragma unroll 1
    for (int cp = 0; cp < 2; cp++) {
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const bool stage_next = (p == 4 && cp == 0);       // chunks 2, 3 replace 0, 1 behind the last transform
            if (stage_next) load_img(2);
            if (!(W3S_DIAG & 8) || (p == 0 && cp == 0)) {
                // one body for both passes (wave 0's second pass then runs code that is already in the instruction cache)
#pragma unroll 1
                for (int pass = 0; pass < (wave == 0 ? 2 : 1); pass++) {
                    TItem it = it0;
                    if (pass) it = it1;
                    transform(p, it);
                }
            }
            __syncthreads();                                   // V complete; (p == 4) the image is dead
            W3S_STAMP(2 + 2 * (5 * cp + p));
            if (stage_next) {
                store_img(0);
                load_img(3);
            }
            // The row's five stages as 30 tile steps (q, chunk c, column tile t), fully unrolled and software-
            // pipelined: the V fragments of step n + AH are read from LDS while the MFMAs of step n run, and the weights of
            // half-stage h + 1 = (q, c) + 1 are requested at the top of half-stage h (two register sets,
            // alternating: no copies, so nothing waits for a load before its data is needed).
            constexpr int AH = 2, RING = AH + 1;               // (two steps ahead: 309.5 -> 305.6 us once the registers were there;
                                                               //  with 50 spilled registers it had been 2 % slower than one step)
            // vq[t] = LDS byte address of the lane's fragment of (chunk 0, stage q, part 0, column tile t); everything
            // else is a compile-time offset that fits the instruction's 16-bit field
            unsigned vq[MT];
            auto v_read = [&](int dq, int c, int t, uint4* v) {
                const char* V = reinterpret_cast<const char*>(lds) + vq[t];
#pragma unroll
                for (int r = 0; r < NP; r++) v[r] = *reinterpret_cast<const uint4*>(V + (((c * 5 + dq) * NP + r) * W::V_PART_U4) * 16);
            };
            // step nn = 3 c + t of the stage; nn >= 6: of the next stage (after the row's last: a harmless read of chunk 1)
            auto v_read_step = [&](int nn, uint4* v) {
                if (nn >= 2 * MT) v_read(1, (nn - 2 * MT) / MT, (nn - 2 * MT) % MT, v);
                else v_read(0, nn / MT, nn % MT, v);
            };
            uint4 vf[RING][NP];                                     // (kept as plain words: a bf16 vector across a branch is re-packed element-wise)
#pragma unroll
            for (int t = 0; t < MT; t++) vq[t] = vaddr[t];
#pragma unroll
            for (int a = 0; a < AH; a++) v_read_step(a, vf[a]);
            // one stage (q = the column of the row's five); JM = which output columns j its factors At[j][q] reach
            // (q = 0: only j = 0, q = 4: only j = 2 -- those two stages are peeled so that their zero factors cost nothing:
            // the vector ALU's issue is what a stage waits for)
            auto stage = [&](const int q, auto jmask) {
                constexpr int JM = decltype(jmask)::value;
                // the four corner stages (p, q in {0, 4}) reach one output each, with factor 1: their MFMAs accumulate
                // straight into it
                const bool corner = (JM == 1 || JM == 4) && (p == 0 || p == 4);
                const int ci_c = (p == 0) ? 0 : 2, cj_c = (JM == 1) ? 0 : 2;
                f32x4 m[MT];
#pragma unroll
                for (int t = 0; t < MT; t++) m[t] = corner ? Y[ci_c][cj_c][t] : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int h = c;                                       // half-stage of this stage; its weights sit in wf[c]
                    {
                        // the next half-stage's weights: (q, c = 1), else (q + 1, 0), else the next row / chunk pair
                        int nxt;
                        if (c == 0) nxt = ((2 * cp + 1) * 5 + p) * 5 + q;
                        else if (q < 4) nxt = ((2 * cp) * 5 + p) * 5 + q + 1;
                        else if (p < 4) nxt = ((2 * cp) * 5 + p + 1) * 5;
                        else nxt = (2 * (cp + 1)) * 25;
                        if (!(W3S_DIAG & 1)) load_b(nxt < 100 ? nxt : 99, wf[(h + 1) & 1]);     // unconditional: the wait counts stay exact
                    }
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        const int n = h * MT + t;                          // tile step of the stage; its fragments sit in vf[n % RING]
                        if (W3S_DIAG & 4) {
#pragma unroll
                            for (int r = 0; r < NP; r++) vf[(n + AH) % RING][r] = vf[n % RING][r];
                        }
                        else v_read_step(n + AH, vf[(n + AH) % RING]);
                        // the reads go out before the step's first MFMA, not (as the scheduler would have it, to shorten live
                        // ranges) before its last
                        if constexpr (NP == 2) __builtin_amdgcn_sched_barrier(0);
                        if (W3S_DIAG & 16) { m[t][0] += __uint_as_float(vf[n % RING][0].x ^ wf[h & 1][0].x); continue; }
                        if constexpr (NP == 2) {
                            const f16x8 u0 = __builtin_bit_cast(f16x8, wf[h & 1][0]), u1 = __builtin_bit_cast(f16x8, wf[h & 1][1]);
                            const f16x8 v0 = __builtin_bit_cast(f16x8, vf[n % RING][0]), v1 = __builtin_bit_cast(f16x8, vf[n % RING][1]);
                            // the three products of weight up to 2^-11, smallest first
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u1, v0, m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v1, m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v0, m[t], 0, 0, 0);
                        } else {
                            const bf16x8 u0 = __builtin_bit_cast(bf16x8, wf[h & 1][0]), u1 = __builtin_bit_cast(bf16x8, wf[h & 1][1]),
                                         u2 = __builtin_bit_cast(bf16x8, wf[h & 1][NP - 1]);
                            const bf16x8 v[3] = { __builtin_bit_cast(bf16x8, vf[n % RING][0]), __builtin_bit_cast(bf16x8, vf[n % RING][1]),
                                                  __builtin_bit_cast(bf16x8, vf[n % RING][NP - 1]) };
                            // the six products of weight up to 2^-16, smallest first
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u2, v[0], m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u0, v[2], m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u1, v[1], m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u1, v[0], m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u0, v[1], m[t], 0, 0, 0);
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u0, v[0], m[t], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);                 // keep the pipeline as written (the scheduler
                    }                                                      // would hoist every read of the row to its top)
                }
                // output transform: Y[i][j] += At[i][p] * At[j][q] * M; p is unrolled (row factors at
                // compile time, zeros skipped), q is a run-time loop (column factors as floats)
                const float cj0 = (q < 4) ? 1.0f : 0.0f;
                const float cj1 = (q == 1) ? 1.0f : (q == 2 ? -1.0f : (q == 3 ? 2.0f : 0.0f));
                const float cj2 = (q == 0) ? 0.0f : (q == 3 ? 4.0f : 1.0f);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int ci = (i == 0) ? (p < 4 ? 1 : 0)
                                 : (i == 1) ? (p == 1 ? 1 : (p == 2 ? -1 : (p == 3 ? 2 : 0)))
                                            : (p == 0 ? 0 : (p == 3 ? 4 : 1));
                    if (corner) {
                        if (i == ci_c) {
#pragma unroll
                            for (int t = 0; t < MT; t++) Y[ci_c][cj_c][t] = m[t];
                        }
                    } else if (ci != 0 && !(W3S_DIAG & 2)) {
                        const float f0 = (float)ci * cj0, f1 = (float)ci * cj1, f2 = (float)ci * cj2;
#pragma unroll
                        for (int t = 0; t < MT; t++)
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                if (JM & 1) Y[i][0][t][e] = fmaf(f0, m[t][e], Y[i][0][t][e]);
                                if (JM & 2) Y[i][1][t][e] = fmaf(f1, m[t][e], Y[i][1][t][e]);
                                if (JM & 4) Y[i][2][t][e] = fmaf(f2, m[t][e], Y[i][2][t][e]);
                            }
                    }
                }
#pragma unroll
                for (int t = 0; t < MT; t++) vq[t] += NP * W::V_PART_U4 * 16;
                __builtin_amdgcn_sched_barrier(0);
            };
            stage(0, std::integral_constant<int, 1>{});
#pragma unroll 1
            for (int q = 1; q < 4; q++) stage(q, std::integral_constant<int, 7>{});
            stage(4, std::integral_constant<int, 4>{});
            __syncthreads();                                   // every wave has read this row of V
            W3S_STAMP(3 + 2 * (5 * cp + p));
            if (stage_next) {
                store_img(1);
                __syncthreads();
            }
        }
    }

    // epilogue.  The products are formed transposed (A operand = weights, B operand = V), so in the
    // C/D layout of a 16x16 tile (col = lane & 15, row = 4 * (lane >> 4) + reg) a lane holds FOUR
    // CONSECUTIVE output channels of ONE wtile: one 16-byte load / store per output cell.  Columns 0..3
    // of the third tile are wtile 8 of the four boards.
    // `out` may alias `resid` (the trunk's in-place convolutions), so loads and stores stay in program
    // order; the nine residual loads of a tile are issued together, and the residual / no-residual
    // choice is made once, not per cell (conv_wino3p_kernels.hip).
    // NP == 2: the accumulators are scaled back by 2^-(input scale of the tile's board + weight scale) first, and
    // the largest value a lane stores per tile goes, reduced over the lanes that share the board, into Amax.
    // Where a cell goes is decided once per convolution, not per cell (EM, below): every variant is straight-line code.
    auto epilogue = [&](auto has_resid, auto mode) {
        constexpr bool R = decltype(has_resid)::value;
        constexpr int EM = decltype(mode)::value;        // 0: memory; 1: memory + LDS; 2: LDS only; 3: the heads' 1x1 convolutions
        const int wexp = NP == 2 ? wexp_all[k] : 0;
        unsigned* amax_out = Amax + 4 * ((k + 1) & 1);
        // NP == 2: waves 0..3 hold input channels 0..63 of the next convolution of the launch = its chunks 0 and 1: they
        // also leave their cells in the image regions (dead since the last transform), so the next convolution starts
        // without a trip through memory.  lcell = the wtile's cell (0, 0) within the workgroup's boards.
        constexpr bool to_lds = EM == 1 || EM == 2;
        // ... and when this is the first convolution of a residual block (even k), whose output only the next convolution
        // reads (the block's second convolution adds the block's INPUT), those cells need not go to memory at all
        // head_out != NULL: the launch's last convolution feeds the heads' three 1x1 convolutions (+ folded BN, ReLU;
        // nn_groups.h) and nothing else, so they are formed here and only their 3 values per cell leave the workgroup:
        // a lane's four channels times the three weight rows go to LDS (the image and V regions are dead), one slot per
        // (wave, lq), and are summed over the 32 slots in a fixed order below.
        constexpr bool to_heads = EM == 3, to_mem = EM < 2;
        f32x4 hwv[3];
        if (to_heads) {
#pragma unroll
            for (int c = 0; c < 3; c++) hwv[c] = *reinterpret_cast<const f32x4*>(&head_w[c * 128 + col]);
        }
        float* hp = reinterpret_cast<float*>(lds) + (size_t)((wave * 4 + lq) * (W::NBW * P)) * 3;
        f32x4* img_next = Img + (wave >> 1) * W::IMG_CH_U4 + 16 + (wave & 1) * 4 + lq;
        auto tile = [&](const f32x4 (&yv)[3][3], const f32x4 bv, long base, bool on, int board_l, int group_lanes, int lcell) {
            // base = element offset of output (0, 0) of the wtile, this lane's channels
            float mx = 0.0f;
            if (on) {
            const int back = NP == 2 ? -(f16_scale_exp(amax_in[board_l], 36) + wexp) : 0;
            f32x4 rv[3][3];
            if (R) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) rv[i][j] = *reinterpret_cast<const f32x4*>(&resid[base + (i * S + j) * 128]);
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    f32x4 v = yv[i][j];
                    if constexpr (NP == 2) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = __builtin_ldexpf(v[e], back);
                    }
                    v += bv;
                    if (R) v += rv[i][j];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                    if (to_mem) *reinterpret_cast<f32x4*>(&out[base + (i * S + j) * 128]) = v;
                    if constexpr (NP == 2) {
                        mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                        if (to_lds) img_next[(lcell + i * S + j) * 8] = v;
                    }
                    if (to_heads) {
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            hp[(lcell + i * S + j) * 3 + c] = fmaf(v[3], hwv[c][3], fmaf(v[2], hwv[c][2], fmaf(v[1], hwv[c][1], v[0] * hwv[c][0])));
                    }
                }
            }
            if constexpr (NP == 2) {
                // lanes of one board: `group_lanes` (8 or 1) consecutive li, all four lq; every lane of the wave takes part
                for (int d = 1; d < group_lanes; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (on && lq == 0 && (li & (group_lanes - 1)) == 0) atomicMax(&amax_out[board_l], __float_as_uint(mx));
            }
        };
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[col]);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int w = t * 16 + li;
            const int bl = w >> 3, wt = w & 7;
            const long board = b_first + bl;
            const int ty = wt / W::T, tx = wt - ty * W::T;
            f32x4 yv[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) yv[i][j] = Y[i][j][t];
            tile(yv, bv, (board * P + 3 * ty * S + 3 * tx) * 128 + col, board < nboards, bl, 8, bl * P + 3 * ty * S + 3 * tx);
        }
        const long board2 = b_first + li;
        f32x4 yv[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) yv[i][j] = Y[i][j][2];
        tile(yv, bv, (board2 * P + 6 * S + 6) * 128 + col, li < 4 && board2 < nboards, li & 3, 1, (li & 3) * P + 6 * S + 6);   // wtile 8 = (ty, tx) = (2, 2)
    };
    {
        const bool lds_next = NP == 2 && wave < 4 && k + 1 < n_convs;
        const int em = (head_out != nullptr && k == n_convs - 1) ? 3 : !lds_next ? 0 : (k & 1) ? 1 : 2;
        auto with_mode = [&](auto has_resid) {
            if (em == 0) epilogue(has_resid, std::integral_constant<int, 0>{});
            else if (em == 1) epilogue(has_resid, std::integral_constant<int, 1>{});
            else if (em == 2) epilogue(has_resid, std::integral_constant<int, 2>{});
            else epilogue(has_resid, std::integral_constant<int, 3>{});
        };
        if (resid) with_mode(std::true_type{});
        else with_mode(std::false_type{});
    }
    if (head_out != nullptr && k == n_convs - 1) {
        __syncthreads();
        const float* hs = reinterpret_cast<const float*>(lds);
        for (int o = tid; o < W::NBW * P * 3; o += 512) {
            const int cell = o / 3, c = o - 3 * cell, board = cell / P;
            float sum = 0.0f;
#pragma unroll 8
            for (int sl = 0; sl < 32; sl++) sum += hs[(size_t)sl * (W::NBW * P * 3) + o];
            if (b_first + board < nboards)
                head_out[((size_t)(b_first + board) * 3 + c) * P + (cell - board * P)] = fmaxf(sum + head_b[c], 0.0f);
        }
    }
    W3S_STAMP(22);
    if (NP == 2 && k == n_convs - 1 && amax_io) {
        __syncthreads();
        if (tid < 4 && b_first + tid < nboards) amax_io[b_first + tid] = __uint_as_float(Amax[4 * ((k + 1) & 1) + tid]);
    }
    // next convolution of the launch: swap the buffers, step the weights
    if (k & 1) { in = x; out = y; resid = resid0; } else { in = y; out = x; resid = x; }
    wconv += (size_t)100 * 8 * 64 * NP;
    wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wconv), 0, W_CONV_BYTES, 0x00020000);
    }
}

// Host side of the input-transform address table of this kernel: out[(tw * 5 + a) * 8 + b], b < 5 = byte address of
// cell (3ty - 1 + a, 3tx - 1 + b) of wtile tw's patch inside a chunk region: cell c at 256 + 128 c; an off-board
// position goes to the zero cell at 0 or 128, whichever has the bank half (parity) the cell would have had, so that
// consecutive wtiles always read different halves.  Returns the number of uint16 entries.
size_t wino3s_build_table(int S_rt, uint16_t* out)
{
    const int S = S_rt, T = S / 3, T2 = T * T, P = S * S, WT = 4 * T2;
    if (out)
        for (int tw = 0; tw < WT; tw++)
            for (int a = 0; a < 5; a++)
                for (int b = 0; b < 8; b++) {
                    const int bl = tw / T2, rem = tw % T2, ty = rem / T, tx = rem % T;
                    const int yy = 3 * ty - 1 + a, xx = 3 * tx - 1 + b;
                    const bool ok = b < 5 && yy >= 0 && yy < S && xx >= 0 && xx < S;
                    const int c = bl * P + yy * S + xx;
                    out[((size_t)tw * 5 + a) * 8 + b] = (uint16_t)(ok ? 256 + c * 128 : 128 * (((c % 2) + 2) % 2));
                }
    return (size_t)WT * 5 * 8;
}

// wt_split: [n_convs][100 stages][8 waves][parts][64 lanes][8 x 16 bit] (nn_host.hip); tab: wino3s_build_table.
// Any number of boards (4 per workgroup, as many rounds of workgroups as it takes).
// parts = 3: bf16; parts = 2: fp16, which also takes wexp[n_convs] (the weights' scale exponents) and the bound of the
// first convolution's input, amax_first > 0 or amax_io[nboards]; amax_io (may be NULL) receives the output's bounds.
// head_out != NULL: instead of the last convolution's output, the launch stores head_out[board][3][P] = ReLU(the three
// 1x1 head convolutions head_w[3][128], head_b[3] of it) -- what launch_heads takes with act_is_h.
int launch_conv3x3_wino3s(int S_rt, float* x, float* y, const float* resid0, const void* wt_split, const float* bias,
                          const void* tab, int nboards, int n_convs, hipStream_t st, int parts, const int* wexp,
                          float* amax_io, float amax_first, const float* head_w, const float* head_b, float* head_out)
{
    if (S_rt != 9 || nboards < 1) {
        set_error("the split F(3x3,3x3) convolution runs 9x9 boards");
        return GOAI_ERR_ARG;
    }
    if (parts != 3 && !(parts == 2 && wexp && (amax_first > 0.0f || amax_io))) {
        set_error("the fp16 F(3x3,3x3) convolution needs its weight exponents and a bound of its input");
        return GOAI_ERR_ARG;
    }
    if (n_convs < 1 || (n_convs > 1 && ((n_convs & 1) || resid0))) {
        set_error("a multi-convolution launch is a whole residual trunk: an even count, no outer residual");
        return GOAI_ERR_ARG;
    }
    if (parts == 2)
        k_conv3x3_wino3s<9, 2><<<dim3((nboards + 3) / 4), dim3(512), 0, st>>>(x, y, resid0, reinterpret_cast<const uint4*>(wt_split), bias,
                                                                              reinterpret_cast<const uint4*>(tab), nboards, n_convs,
                                                                              wexp, amax_io, amax_first, head_w, head_b, head_out);
    else
        k_conv3x3_wino3s<9, 3><<<dim3((nboards + 3) / 4), dim3(512), 0, st>>>(x, y, resid0, reinterpret_cast<const uint4*>(wt_split), bias,
                                                                              reinterpret_cast<const uint4*>(tab), nboards, n_convs,
                                                                              nullptr, nullptr, 0.0f, head_w, head_b, head_out);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
