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
#include <type_traits>
#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NP>
struct Wino3H {
    static constexpr int S = 9, T = 3, T2 = 9, P = 81;
    static constexpr int NBW = 4;                                  // whole boards per workgroup
    static constexpr int WT = NBW * T2;                            // 36 wtiles
    static constexpr int MT = 3;                                   // MFMA column tiles of 16 wtile rows (the last: 4 real)
    static constexpr int VROWS = WT + 1;                           // the wtile rows + one zero row
    static constexpr int ROWC = 10;                                // cells per image row: [zero, x = 0..8]
    static constexpr int REG_CELLS = (1 + NBW * 10) * ROWC + 1;    // zero row, then per board 9 rows + a zero row; + the cell "x = 9" of the last row
    static constexpr int REG_U4 = REG_CELLS * 8;                   // 16-byte units per chunk region (a cell = 32 channels x 4 B)
    static constexpr int V_PART_U4 = VROWS * 4;                    // one (chunk, q, part): [VROWS][32 ch] fp16 = 64 B per row
    static constexpr int V_U4 = 2 * 5 * NP * V_PART_U4;
    static constexpr int AMAX_U4 = 2;                              // [2][4 boards] activation bounds as fp32 bits
    static constexpr int LDS_U4 = 2 * REG_U4 + V_U4 + AMAX_U4;
    static constexpr int NI = (NBW * P * 8 + 511) / 512;
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(REG_CELLS % 2 == 1, "the two regions must lie in different bank halves");
    static_assert(32 * NBW * P * 3 * 4 <= LDS_U4 * 16, "the heads' partial sums reuse the whole array");
    // cell (y, x) of board b, y and x in -1..9 (the border cells are zero)
    static constexpr int cell(int b, int y, int x) { return (1 + 10 * b + y) * ROWC + x + 1; }
};

#ifdef GOAI_CONV_STAMPS
// diagnostic build only (tools/wino3h_stamps.py): s_memrealtime (100 MHz) stamps of the SECOND convolution of a launch,
// 32 per workgroup: 0 layer top, 1 image staged, 2 + 2r V of row r complete (r = 5 cp + p), 3 + 2r its stages done,
// 22 epilogue done; 23 / 24 = s_memtime (shader clock) at stamps 0 / 22
__device__ unsigned long long g_wino3h_stamps[32 * 1024];
extern "C" int goai_debug_wino3h_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino3h_stamps), (size_t)n * sizeof(unsigned long long));
}
#ifndef W3H_STAMP_K
#define W3H_STAMP_K 1     // the convolution of the launch that is stamped
#endif
#define W3H_STAMP(i) if (threadIdx.x == 0 && k == W3H_STAMP_K) { g_wino3h_stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0 || (i) == 22) g_wino3h_stamps[blockIdx.x * 32 + 23 + ((i) == 22)] = __builtin_amdgcn_s_memtime(); }
#else
#define W3H_STAMP(i)
#endif

__device__ __forceinline__ f32x4 h_fma4(float c, f32x4 a, f32x4 b)
{
    return f32x4{ fmaf(c, a[0], b[0]), fmaf(c, a[1], b[1]), fmaf(c, a[2], b[2]), fmaf(c, a[3], b[3]) };
}
// quarter q of V row r sits at position q ^ w3h_vswz(r) (conv_wino3s_kernels.hip: fragment reads then touch all 64 banks)
__device__ __forceinline__ int w3h_vswz(int row) { return (0 - (row >> 2)) & 3; }

// Same launch contract as k_conv3x3_wino3s (n_convs convolutions in a row on the same boards): wexp_all[conv] = the
// weights' scale exponent; amax_first > 0: bound of the first convolution's input, else amax_io[board]; amax_io != NULL:
// receives the bound of the last convolution's output, per board; head_out != NULL: the last convolution feeds the
// heads' three 1x1 convolutions and only those leave the workgroup.
template <int NP>
__global__ __launch_bounds__(512) void k_conv3x3_wino3h(float* x, float* y, const float* resid0, const uint4* wt_all,
                                                        const float* __restrict__ bias_all, int nboards, int n_convs,
                                                        const int* __restrict__ wexp_all, float* amax_io, float amax_first,
                                                        const float* __restrict__ head_w, const float* __restrict__ head_b,
                                                        float* __restrict__ head_out)
{
    using W = Wino3H<NP>;
    constexpr int S = W::S, P = W::P, MT = W::MT;
    __shared__ f32x4 lds[W::LDS_U4];
    f32x4* Img = lds;                                  // two chunk regions
    unsigned* Amax = reinterpret_cast<unsigned*>(lds + 2 * W::REG_U4 + W::V_U4);      // [2][4]
    constexpr unsigned V_BASE = 2 * W::REG_U4 * 16, V_PART = W::V_PART_U4 * 16, REG_BYTES = W::REG_U4 * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * W::NBW;
    const float* in = x;
    const float* resid = resid0;
    float* out = y;

    // workgroups of a several-convolution launch start in sixteen groups 1.5 us apart, so that their memory phases do not all
    // fall into the same microseconds: a convolution's residual reads and output stores are 85 MB over the chip, 10 us at the
    // full HBM rate if every workgroup is in its epilogue at once (conv_wino3s_kernels.hip; profiles/r02_wino3s_ablation.txt;
    // round 3, bench.py C2 launch: no groups 289 us, 2 x 3.9 us 284, 4 x 3.9 us -- round 2's -- 277, 8 x 2.9 us 274.3,
    // 16 x 1.5 us 273.9: profiles/r03_wino3h_notes.txt section 9)
#ifndef W3H_START_GROUPS
#define W3H_START_GROUPS 16
#define W3H_START_SLEEP 48
#endif
    // (that is for a launch of one workgroup per CU, the lock-step search's.  With more workgroups than CUs -- the Q search's
    // 8192 -- every workgroup keeps round 2's four groups 3.9 us apart, which re-spreads the phases round after round: bench.py C3
    // 0.207 of peak against 0.203 with sixteen groups or with the first round delayed only)
    if (n_convs > 1 && W3H_START_GROUPS > 1) {
        if (gridDim.x <= 256) for (int i = (blockIdx.x >> 3) % W3H_START_GROUPS; i > 0; i--) __builtin_amdgcn_s_sleep(W3H_START_SLEEP);
        else for (int i = (blockIdx.x >> 3) % 4; i > 0; i--) __builtin_amdgcn_s_sleep(127);
    }
    // everything zero: the borders of the image regions and the zero row of V stay zero for the whole launch
    for (int i = tid; i < 2 * W::REG_U4 + W::V_U4; i += 512) lds[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    if (tid < 4) {
        const bool on = b_first + tid < nboards;
        Amax[tid] = __float_as_uint(amax_first > 0.0f ? amax_first : (on ? amax_io[b_first + tid] : 0.0f));
    }

    // Input-transform items: (wtile, chunk region, 4-channel slot), one per lane and pass.  A ds_read_b128 is served in
    // four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32); a group here = one wtile x
    // two regions x eight slots: a cell is 128 B = half of the banks and the regions are an odd number of cells apart.
    const int l32 = lane & 31;
    const int hgj = l32 < 4 ? l32 : l32 < 12 ? 16 + (l32 - 4) : l32 < 16 ? 4 + (l32 - 12) : l32 < 20 ? 24 + (l32 - 16)
                  : l32 < 28 ? 8 + (l32 - 20) : 28 + (l32 - 28);          // 16 * (group within the half) + position in the group
    const int t_g = (lane >> 5) * 2 + (hgj >> 4), t_rg = (hgj >> 3) & 1, ts = hgj & 7;     // group of the wave, region, slot
    struct TItem { unsigned img, v; int board; };      // LDS byte addresses: patch cell (-1, -1), V row (q 0, part 0)
    auto t_item = [&](int bl, int rem, int vrow) {
        const int ty = rem / 3, tx = rem - 3 * ty;
        return TItem{ (unsigned)(t_rg * REG_BYTES + W::cell(bl, 3 * ty - 1, 3 * tx - 1) * 128 + ts * 16),
                      (unsigned)(V_BASE + t_rg * (5 * NP * V_PART) + vrow * 64 + (((ts >> 1) ^ w3h_vswz(vrow)) * 16) + (ts & 1) * 8), bl };
    };
    // main pass: wtile rem < 8 of board bl -> V row 8 bl + rem = 4 wave + t_g; tail pass (one wave): wtile 8 of board t_g -> V row 32 + t_g
    const int mrow = 4 * wave + t_g;
    const TItem it_main = t_item(mrow >> 3, mrow & 7, mrow), it_tail = t_item(t_g, 8, 32 + t_g);
#ifndef W3H_TAIL_WAVE
#define W3H_TAIL_WAVE 7
#endif
    constexpr int TAIL_WAVE = W3H_TAIL_WAVE;
    const bool tail_wave = wave == TAIL_WAVE;

    f32x4 ireg[W::NI];
    // the workgroup's boards of the current input as a buffer (as conv_wino2s_kernels.hip): piece idx = tid + 512 i of chunk c at byte
    // (idx / 8) * 512 + c * 128 + (idx % 8) * 16 = one vector offset per thread + a scalar offset; pieces of boards that do not
    // exist lie beyond the descriptor's range and read as zeros
    const int nb_boards = nboards - b_first < W::NBW ? nboards - b_first : W::NBW;
    auto irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)b_first * P * 128, 0, nb_boards * P * 512, 0x00020000);
    auto load_img_to = [&](int chunk, f32x4* reg) {
        const unsigned voff = (unsigned)((tid >> 3) * 512 + (tid & 7) * 16);
#pragma unroll
        for (int i = 0; i < W::NI; i++)
            reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, voff, chunk * 128 + i * 32768, 0));
    };
    auto store_img_from = [&](int region, const f32x4* reg) {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tid + i * 512;
            const int cr = idx >> 3, slot = idx & 7;
            const int bl = cr / P, rem = cr - bl * P, yy = rem / S, xx = rem - yy * S;
            if (idx < W::NBW * P * 8) Img[region * W::REG_U4 + W::cell(bl, yy, xx) * 8 + slot] = reg[i];
        }
    };
    auto load_img = [&](int chunk) { load_img_to(chunk, ireg); };
    auto store_img = [&](int region) { store_img_from(region, ireg); };
    const unsigned* amax_in = Amax;                    // this convolution's input bounds: Amax + 4 (k & 1)

    // ---- the input transform V[p][q] = (Bt d B)[p][q] of one item, in two halves ----------------------------------
    // row half: r[b] = sum_a Bt[p][a] d[a][b].  Patch row a, column b of an item = the 16 bytes at img + (10 a + b) * 128.
    auto patch = [&](unsigned img, int a, int b) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + img + (a * W::ROWC + b) * 128);
    };
    // the rows of the patch that row p of Bt reads (its nonzero entries), as a bit mask over a = 0..4
    auto rows_of = [](int p) { return p == 0 ? 0xf : p == 1 ? 0xe : p == 2 ? 0xe : p == 3 ? 0xa : 0x1e; };
    // r[b] from the column's cells d[a] (element e only: the pipelined form computes one element per MFMA gap)
    auto row_elem = [](int p, const f32x4* d, int e) {
        return p == 0 ? fmaf(-2.0f, d[2][e], fmaf(2.0f, d[0][e], d[3][e] - d[1][e]))
             : p == 1 ? fmaf(-2.0f, d[1][e], d[3][e] - d[2][e])
             : p == 2 ? fmaf(-3.0f, d[2][e], fmaf(2.0f, d[1][e], d[3][e]))
             : p == 3 ? d[3][e] - d[1][e]
                      : fmaf(-2.0f, d[3][e], fmaf(2.0f, d[1][e], d[4][e] - d[2][e]));
    };
    auto row_half = [&](int p, unsigned img, f32x4* r) {           // all five columns at once (the rows that are not pipelined)
#pragma unroll
        for (int b = 0; b < 5; b++) {
            f32x4 d[5];
#pragma unroll
            for (int a = 0; a < 5; a++)
                if ((rows_of(p) >> a) & 1) d[a] = patch(img, a, b);
#pragma unroll
            for (int e = 0; e < 4; e++) r[b][e] = row_elem(p, d, e);
        }
    };
    // column half + split into fp16 parts (scaled by the board's power of two) + V writes
    auto col_half = [&](const TItem& it, const f32x4* r) {
        const float t_scale = __uint_as_float((unsigned)(127 + f16_scale_exp(amax_in[it.board], 36)) << 23);
        uint2* V = reinterpret_cast<uint2*>(reinterpret_cast<char*>(lds) + it.v);
        auto put = [&](int q, const f32x4 v) {
            // h = fp16(v), l = fp16(v - h), round to nearest (v - h by v_fma_mix_f32, which reads h's halves in place)
            const f32x4 vs = v * t_scale;
            const f16x4 h = __builtin_convertvector(vs, f16x4);
            const uint2 hp = __builtin_bit_cast(uint2, h);
            uint2* dst = V + q * (NP * W::V_PART_U4 * 2);
            dst[0] = hp;
            if constexpr (NP == 2) {
                f32x4 rest;
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest[0]) : "v"(hp.x), "v"(vs[0]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest[1]) : "v"(hp.x), "v"(vs[1]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rest[2]) : "v"(hp.y), "v"(vs[2]));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rest[3]) : "v"(hp.y), "v"(vs[3]));
                const f16x4 l = __builtin_convertvector(rest, f16x4);
                dst[W::V_PART_U4 * 2] = __builtin_bit_cast(uint2, l);
            }
        };
        put(0, h_fma4(-2.0f, r[2], h_fma4(2.0f, r[0], r[3] - r[1])));
        put(1, h_fma4(-2.0f, r[1], r[3] - r[2]));
        put(2, h_fma4(-3.0f, r[2], h_fma4(2.0f, r[1], r[3])));
        put(3, r[3] - r[1]);
        put(4, h_fma4(-2.0f, r[3], h_fma4(2.0f, r[1], r[4] - r[2])));
    };
    // V of row p between two barriers.  have_main: the main item's row half is already in rm (formed under the previous
    // row's MFMAs).  One loop body for both passes (the tail wave's second pass runs code that is already in the cache).
    f32x4 rm[5];
    auto finish_row = [&](int p, bool have_main) {
#pragma unroll 1
        for (int pass = 0; pass < (tail_wave ? 2 : 1); pass++) {
            TItem it = it_main;
            if (pass) it = it_tail;
            if (pass || !have_main) row_half(p, it.img, rm);
            col_half(it, rm);
        }
    };

    // weight fragments of a half-stage (nn_host.hip): [stage = (chunk * 5 + p) * 5 + q][wave][part][lane] x 16 B: lane (li, lq)
    // gets output channel 16 wave + li, input channels 8 lq .. 8 lq + 7 of the chunk; buffer loads: descriptor = this
    // convolution's image, scalar offset = the stage, vector offset = the lane's place in the stage
    const uint4* wconv = wt_all;
    const unsigned wlane = (wave * (64 * NP) + lane) * 16;        // bytes
    constexpr unsigned W_STAGE_BYTES = 8 * 64 * NP * 16, W_CONV_BYTES = 100 * W_STAGE_BYTES;
    auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wconv), 0, W_CONV_BYTES, 0x00020000);
    u32x4 wf[2][NP];                                       // two half-stages of weights, used alternately
#ifndef W3H_AH
#define W3H_AH 2        // tile steps the V fragment reads run ahead of their MFMAs
#endif
#ifndef W3H_DIAG
#define W3H_DIAG 0      // diagnostic builds only (wrong results): 1 no output transform, 2 every weight load reads stage 0 (L2-hot, same
#endif                  // instructions), 4 no weight loads after the first, 8 no residual loads, 16 no global stores in the epilogue
    auto load_b = [&](int stage, u32x4* b) {
        if ((W3H_DIAG & 4) && stage > 0) return;
        if (W3H_DIAG & 2) stage = 0;
#pragma unroll
        for (int r = 0; r < NP; r++)
            b[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024 * r, stage * W_STAGE_BYTES, 0));
        asm volatile("" ::: "memory");                     // (keeps the load where it is written: conv_wino3s_kernels.hip)
    };

    f32x4 Y[3][3][MT];                               // [i][j]: output (3ty + i, 3tx + j) of the wtile
    const int col = wave * 16 + 4 * lq;
    // V row of this lane's column in each of the three column tiles; the third tile's columns 4..15 read the zero row
    int vrow[MT];
    vrow[0] = li;
    vrow[1] = 16 + li;
    vrow[2] = li < 4 ? 32 + li : W::WT;
    unsigned vaddr[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        vaddr[t] = V_BASE + (unsigned)(((lq ^ w3h_vswz(vrow[t])) + vrow[t] * 4) * 16);
        asm volatile("" : "+v"(vaddr[t]));
    }

#pragma unroll 1
    for (int k = 0; k < n_convs; k++) {
    const float* bias = bias_all + k * 128;
    load_b(0, wf[0]);
    amax_in = Amax + 4 * (k & 1);
    // k = 0: the zero fills are complete; k > 0: every wave's MFMAs of the previous convolution have read V, and its
    // epilogue's cells are in the image regions
    __syncthreads();
    W3H_STAMP(0);
    if (tid < 4) Amax[4 * ((k + 1) & 1) + tid] = 0u;      // collects this convolution's output bounds (epilogue)
    if (k == 0) {                                    // (k > 0: the previous epilogue left chunks 0 and 1 in LDS)
        f32x4 ireg1[W::NI];
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
    W3H_STAMP(1);

#pragma unroll 1
    for (int cp = 0; cp < 2; cp++) {
        finish_row(0, false);                              // the chunk pair's first row: nothing to form it under
#pragma unroll
        for (int p = 0; p < 5; p++) {
            const bool stage_next = (p == 4 && cp == 0);       // chunks 2, 3 replace 0, 1 during the pair's last row
            __syncthreads();                                   // V of row p complete; (p == 4) the image is dead
            W3H_STAMP(2 + 2 * (5 * cp + p));
            if (stage_next) {
                store_img(0);
                load_img(3);
            }
            constexpr int AH = W3H_AH, RING = AH + 1;
            unsigned vq[MT];
            auto v_read = [&](int dq, int c, int t, uint4* v) {
                const char* V = reinterpret_cast<const char*>(lds) + vq[t];
#pragma unroll
                for (int r = 0; r < NP; r++) v[r] = *reinterpret_cast<const uint4*>(V + (((c * 5 + dq) * NP + r) * W::V_PART_U4) * 16);
            };
            auto v_read_step = [&](int nn, uint4* v) {
                if (nn >= 2 * MT) v_read(1, (nn - 2 * MT) / MT, (nn - 2 * MT) % MT, v);
                else v_read(0, nn / MT, nn % MT, v);
            };
            uint4 vf[RING][NP];
#pragma unroll
            for (int t = 0; t < MT; t++) vq[t] = vaddr[t];
#pragma unroll
            for (int a = 0; a < AH; a++) v_read_step(a, vf[a]);
            // Output transform (round 3, second form): Y[i][j] += At[i][p] * At[j][q] * M of stage q - 1 as PLAIN vector
            // instructions in the shadow of stage q's MFMAs (tools/ubench/mfma_f16_valu3.hip: two plain instructions per MFMA
            // issue for free, beside the wave's own MFMAs and beside the other wave's; a packed fp32 instruction never
            // overlaps an MFMA, and behind the MFMAs -- the round-2 form -- the transform cost as much as the products).
            // Every factor is a compile-time constant (the row's five stages are unrolled; zero factors drop out), the
            // instructions are asm so that they stay in their gaps and stay unpacked; fma(f, m, y) rounds as before:
            // bit-identical results.  The products of two stages are alive at a time (mb).
            auto ci_of = [](int pp, int i) { return i == 0 ? (pp < 4 ? 1 : 0) : i == 1 ? (pp == 1 ? 1 : (pp == 2 ? -1 : (pp == 3 ? 2 : 0))) : (pp == 0 ? 0 : (pp == 3 ? 4 : 1)); };
            auto is_corner = [](int pp, int qq) { return (pp == 0 || pp == 4) && (qq == 0 || qq == 4); };
            // the k-th of the rows (columns) i of At with At[i][pp] != 0, and how many there are
            auto nz_at = [](int pp, int k) { return pp == 0 ? 0 : (pp == 4 ? 2 : k); };
            auto nnz_of = [](int pp) { return (pp == 0 || pp == 4) ? 1 : 3; };
            auto n_ops = [&](int pp, int qq) { return (qq < 0 || is_corner(pp, qq)) ? 0 : nnz_of(pp) * nnz_of(qq) * MT * 4; };
            // operation number n (tile-major) of stage (p, qq)'s transform; jonly >= 0: only if it goes into output column
            // jonly, jskip >= 0: unless it goes into column jskip
            // (decoded from n, one instruction form for every factor with the factor in a scalar register: a scan of all
            // operations per gap, or a choice of instruction by factor, multiplies the code in front of the unroller,
            // which then gives up -- and Y lands in scratch memory)
            auto tf_op = [&](const int qq, f32x4 (&m)[MT], const int n, const int jonly, const int jskip) {
                if ((W3H_DIAG & 1) && (n & 7)) return;
                const int nj = nnz_of(qq), per_tile = nnz_of(p) * nj * 4;
                const int t = n / per_tile, r = n % per_tile, pair = r >> 2, e = r & 3;
                const int i = nz_at(p, pair / nj), j = nz_at(qq, pair % nj);
                if ((jonly >= 0 && j != jonly) || j == jskip) return;
                const float f = (float)(ci_of(p, i) * ci_of(qq, j));
                asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(Y[i][j][t][e]) : "v"(m[t][e]), "s"(f));
            };
            f32x4 mb[2][MT];
            // one stage (q = the column of the row's five); qpend: the stage whose transform runs in this one's gaps (-1: none)
            auto stage = [&](const int q, const int qpend) {
                const bool corner = is_corner(p, q);
                const int ci_c = (p == 0) ? 0 : 2, cj_c = (q == 0) ? 0 : 2;
                f32x4 (&m)[MT] = mb[q & 1];
                f32x4 (&mp)[MT] = mb[(q + 1) & 1];
                constexpr int NG = 2 * MT * (NP == 2 ? 3 : 1);               // MFMAs = gaps of a stage
                const int np = n_ops(p, qpend), per_gap = (np + NG - 1) / NG;
                // a corner stage accumulates in Y itself: what the pending transform adds to that column goes first
                const int jskip = (corner && np > 0) ? cj_c : -1;
                if (jskip >= 0) {
#pragma unroll
                    for (int n = 0; n < np; n++) tf_op(qpend, mp, n, jskip, -1);
                    asm volatile("s_nop 7");
                }
#pragma unroll
                for (int t = 0; t < MT; t++) m[t] = corner ? Y[ci_c][cj_c][t] : f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
                int gapn = 0;
                auto gap = [&]() {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < per_gap; k++)
                        if (gapn * per_gap + k < np) tf_op(qpend, mp, gapn * per_gap + k, -1, jskip);
                    gapn++;
                    __builtin_amdgcn_sched_barrier(0);
                };
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int h = c;
                    {
                        int nxt;
                        if (c == 0) nxt = ((2 * cp + 1) * 5 + p) * 5 + q;
                        else if (q < 4) nxt = ((2 * cp) * 5 + p) * 5 + q + 1;
                        else if (p < 4) nxt = ((2 * cp) * 5 + p + 1) * 5;
                        else nxt = (2 * (cp + 1)) * 25;
                        load_b(nxt < 100 ? nxt : 99, wf[(h + 1) & 1]);
                    }
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        const int n = h * MT + t;
                        v_read_step(n + AH, vf[(n + AH) % RING]);
                        __builtin_amdgcn_sched_barrier(0);
                        const f16x8 u0 = __builtin_bit_cast(f16x8, wf[h & 1][0]);
                        const f16x8 v0 = __builtin_bit_cast(f16x8, vf[n % RING][0]);
                        if constexpr (NP == 2) {
                            const f16x8 u1 = __builtin_bit_cast(f16x8, wf[h & 1][1]);
                            const f16x8 v1 = __builtin_bit_cast(f16x8, vf[n % RING][1]);
                            // the three products of weight up to 2^-11, smallest first
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u1, v0, m[t], 0, 0, 0);
                            gap();
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v1, m[t], 0, 0, 0);
                            gap();
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v0, m[t], 0, 0, 0);
                            gap();
                        } else {
                            m[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v0, m[t], 0, 0, 0);
                            gap();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (corner) {
#pragma unroll
                    for (int t = 0; t < MT; t++) Y[ci_c][cj_c][t] = m[t];
                }
#pragma unroll
                for (int t = 0; t < MT; t++) vq[t] += NP * W::V_PART_U4 * 16;
                __builtin_amdgcn_sched_barrier(0);
            };
            stage(0, -1);
            stage(1, 0);
            stage(2, 1);
            stage(3, 2);
            stage(4, 3);
            // the last stage's transform: tile-major, so the last tile's products (the stage's last MFMAs) are read
            // behind the 2/3 of the operations that do not need them
#pragma unroll
            for (int n = 0; n < n_ops(p, 4); n++) tf_op(4, mb[0], n, -1, -1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                   // every wave has read this row of V (and, p < 4, the image for row p + 1's row half)
            W3H_STAMP(3 + 2 * (5 * cp + p));
            if (stage_next) {
                store_img(1);
                __syncthreads();
            } else if (p < 4) {
                if (p == 3 && cp == 0) load_img(2);            // in flight under the column half
                finish_row(p + 1, false);
            }
        }
    }

    // epilogue (conv_wino3s_kernels.hip): a lane holds FOUR CONSECUTIVE output channels of ONE wtile per column tile;
    // scale back, bias, residual, ReLU, the board's new activation bound; destinations by EM: 0 memory; 1 memory + LDS
    // (the next convolution's chunks 0, 1: waves 0..3); 2 LDS only; 3 the heads' 1x1 convolutions
    auto epilogue = [&](auto has_resid, auto mode) {
        constexpr bool R = decltype(has_resid)::value;
        constexpr int EM = decltype(mode)::value;
        const int wexp = wexp_all[k];
        unsigned* amax_out = Amax + 4 * ((k + 1) & 1);
        constexpr bool to_lds = EM == 1 || EM == 2;
        constexpr bool to_heads = EM == 3, to_mem = EM < 2;
        f32x4 hwv[3];
        if (to_heads) {
#pragma unroll
            for (int c = 0; c < 3; c++) hwv[c] = *reinterpret_cast<const f32x4*>(&head_w[c * 128 + col]);
        }
        float* hp = reinterpret_cast<float*>(lds) + (size_t)((wave * 4 + lq) * (W::NBW * P)) * 3;
        f32x4* img_next = Img + (wave >> 1) * W::REG_U4 + (wave & 1) * 4 + lq;
        auto tile = [&](const f32x4 (&yv)[3][3], const f32x4 bv, long base, bool on, int board_l, int group_lanes, int lcell, int icell) {
            // base = element offset of output (0, 0) of the wtile, this lane's channels; lcell = its cell within the
            // workgroup's boards (row-major), icell = the same cell in the bordered image regions
            float mx = 0.0f;
            if (on) {
            const int back = -(f16_scale_exp(amax_in[board_l], 36) + wexp);
            f32x4 rv[3][3];
            if (R) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) rv[i][j] = (W3H_DIAG & 8) ? bv : *reinterpret_cast<const f32x4*>(&resid[base + (i * S + j) * 128]);
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    f32x4 v = yv[i][j];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = __builtin_ldexpf(v[e], back);
                    v += bv;
                    if (R) v += rv[i][j];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
                    if (to_mem && (!(W3H_DIAG & 16) || v[0] == 12345.0f)) *reinterpret_cast<f32x4*>(&out[base + (i * S + j) * 128]) = v;
                    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                    if (to_lds) img_next[(icell + i * W::ROWC + j) * 8] = v;
                    if (to_heads) {
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            hp[(lcell + i * S + j) * 3 + c] = fmaf(v[3], hwv[c][3], fmaf(v[2], hwv[c][2], fmaf(v[1], hwv[c][1], v[0] * hwv[c][0])));
                    }
                }
            }
            // lanes of one board: `group_lanes` (8 or 1) consecutive li, all four lq; every lane of the wave takes part
            for (int d = 1; d < group_lanes; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (on && lq == 0 && (li & (group_lanes - 1)) == 0) atomicMax(&amax_out[board_l], __float_as_uint(mx));
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
            tile(yv, bv, (board * P + 3 * ty * S + 3 * tx) * 128 + col, board < nboards, bl, 8, bl * P + 3 * ty * S + 3 * tx,
                 W::cell(bl, 3 * ty, 3 * tx));
        }
        const long board2 = b_first + li;
        f32x4 yv[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) yv[i][j] = Y[i][j][2];
        tile(yv, bv, (board2 * P + 6 * S + 6) * 128 + col, li < 4 && board2 < nboards, li & 3, 1, (li & 3) * P + 6 * S + 6,
             W::cell(li & 3, 6, 6));   // wtile 8 = (ty, tx) = (2, 2)
    };
    {
        const bool lds_next = wave < 4 && k + 1 < n_convs;
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
    W3H_STAMP(22);
    if (k == n_convs - 1 && amax_io) {
        __syncthreads();
        if (tid < 4 && b_first + tid < nboards) amax_io[b_first + tid] = __uint_as_float(Amax[4 * ((k + 1) & 1) + tid]);
    }
    // next convolution of the launch: swap the buffers, step the weights
    if (k & 1) { in = x; out = y; resid = resid0; } else { in = y; out = x; resid = x; }
    irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)b_first * P * 128, 0, nb_boards * P * 512, 0x00020000);
    wconv += (size_t)100 * 8 * 64 * NP;
    wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wconv), 0, W_CONV_BYTES, 0x00020000);
    }
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
