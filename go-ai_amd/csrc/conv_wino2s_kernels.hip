// conv_wino2s_kernels.hip -- the trunk's conv3x3 128 -> 128 (+ folded BatchNorm bias, optional residual,
// ReLU; /root/reference/go_ai/models/__init__.py:234-251) by Winograd F(2x2,3x3) with its 16
// transform-domain GEMMs on the BF16 matrix cores, every fp32 operand split EXACTLY into three bf16
// parts (GOAI_DTYPE_BF16X6 on boards other than 9x9; the 9x9 form is conv_wino3s_kernels.hip, which
// explains the arithmetic: U V = u0 v0 + (u0 v1 + u1 v0) + (u0 v2 + u1 v1 + u2 v0) + terms below 2^-24).
//
// Decomposition = conv_wino_kernels.hip: a workgroup (8 waves) owns NBW whole boards = NBW * T*T wtiles of
// 2x2 outputs (13x13: 2 boards = 98 wtiles, 19x19: 1 board = 100), at most 112 = seven 16-column MFMA tiles,
// and all 128 output channels; wave w owns output channels 16w .. 16w+15.  Loop: 4 channel chunks x 4
// transform rows p; per row one input-transform phase (all threads; Bt d B in fp32 on the vector ALUs, then
// the three-way split) and 4 stages q of 7 tiles x 6 bf16 MFMAs (K = 32) per wave, each followed by the
// fp32 output transform of its products.
// LDS: IMG  one 32-channel chunk of the boards, fp32, ZERO-BORDERED since round 3 (as conv_wino3h_kernels.hip): rows of
//           S + 2 cells = [x = -1][x = 0 .. S-1][x = S] (x = S + 1 is the next row's x = -1), a zero row above the first board
//           and two below every board (a wtile's 4x4 patch spans y, x = -1 .. S + 1): a patch is one base address +
//           immediates -- no address table, no table read in front of the patch reads, no per-read address arithmetic
//           (59 KB at 13x13 and 19x19); a cell's two 64-byte halves in two regions (Wino2S::HALF_B: conflict-free patch reads)
//      V    [4 q][3 parts][112 rows][32 ch] bf16 = 86 KB, single-buffered
// Weights: [64 stages][8 waves][3 parts][64 lanes][8 bf16] per convolution (nn_host.hip), L2 -> registers one
// stage ahead into alternating register sets; V fragments are read one tile ahead.
//
// NP = 2 (GOAI_DTYPE_F16X3): two fp16 parts per operand, three fp16 MFMAs per tile step, operands scaled by powers of
// two -- conv_wino3s_kernels.hip explains it; here |Bt d B| <= 4 amax and a launch is one convolution, so the boards'
// bounds travel between launches in amax_io[board] (read at the start, overwritten with the output's at the end, by
// the one workgroup that owns the board).
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
struct Wino2S {
    static constexpr int T = (S + 1) / 2, T2 = T * T, P = S * S;
    static constexpr int MT = 7, ROWS = MT * 16;                   // wtile rows per workgroup
    static constexpr int NBW = ROWS / T2;                          // whole boards per workgroup
    static constexpr int WT = NBW * T2;                            // real wtiles per workgroup
    static constexpr int ROWC = S + 2;                             // cells per image row
    static constexpr int IMG_CELLS = (1 + NBW * (S + 2)) * ROWC + 1;
    // a cell's 32 channels x 4 B as two 64-byte halves (channel slots 0-3 | 4-7) in two regions HALF_B = 64 (mod 256) bytes
    // apart: the lanes of a ds_read_b128 group ({0-3, 12-15, 20-27}, ...: halves of four consecutive wtiles, whose
    // cells lie 2 apart) then touch all 64 banks once -- with whole 128-byte cells every patch read is a 2-way conflict
    static constexpr int HALF_B = (IMG_CELLS * 64 + 255) / 256 * 256 + 64;
    static constexpr int IMG_U4 = (HALF_B + IMG_CELLS * 64) / 16;  // in 16-byte units
    static constexpr int slot_at(int cell, int slot) { return ((slot >> 2) * HALF_B + cell * 64 + (slot & 3) * 16) / 16; }
    // cell (y, x) of board b, y and x in -1 .. S + 1 (everything off the board is zero)
    static constexpr int cell(int b, int y, int x) { return (1 + b * (S + 2) + y) * ROWC + x + 1; }
    static constexpr int V_PART_U4 = ROWS * 4;                     // one (q, part): [ROWS][32 ch] bf16 = 64 B per row
    static constexpr int V_U4 = 4 * NP * V_PART_U4;
    static constexpr int NI = (NBW * P * 8 + 511) / 512;           // staged 16-byte pieces per thread per chunk
    static constexpr int AMAX_U4 = 8;                              // [2][16]: input / output bounds of the boards as fp32 bits (NP == 2)
    // HEADS: the heads' 1x1 convolutions of the output, [32 slots = (wave, lq)][NBW * P cells][3] fp32, over the dead image and V
    static constexpr int HP_U4 = (32 * NBW * P * 3 * 4 + 15) / 16;
    // (the bounds, which the epilogue still reads and collects, stay clear of it: they are the block's last 128 bytes)
    // epilogue: the area through which a tile's cells change lanes, per wave [2 tiles][4 cells][4 quads][17] x 16 B over
    // the dead image and V -- or, with HEADS, [2 cells][4 quads][17] x 16 B behind the heads' area
    static constexpr int XCH_WAVE_U4 = 2 * 4 * 72, XCH_HEADS_WAVE_U4 = 2 * 72;
    static constexpr int WORK_U4 = IMG_U4 + V_U4 + AMAX_U4;
    static constexpr int LDS_U4 = WORK_U4 > HP_U4 + 8 * XCH_HEADS_WAVE_U4 + AMAX_U4 ? WORK_U4 : HP_U4 + 8 * XCH_HEADS_WAVE_U4 + AMAX_U4;
    static_assert(8 * XCH_WAVE_U4 + AMAX_U4 <= WORK_U4, "exchange area");
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
    static_assert(NBW <= 16, "bounds table");
    static_assert(NBW >= 1, "board too large for one workgroup");
    static_assert(LDS_U4 * 16 <= 160 * 1024, "LDS budget");
};

#ifdef GOAI_CONV_STAMPS
// diagnostic build only (tools/wino2s_stamps.py): s_memrealtime (100 MHz) stamps, 40 per workgroup: 0 top, 1 first image
// staged, 2 + 2r V of row r complete (r = 4 chunk + p), 3 + 2r its four stages done, 34 epilogue done; 35 / 36 = s_memtime at 0 / 34; 37 / 38 epilogue: behind its first barrier / wave 0's last store issued
__device__ unsigned long long g_wino2s_stamps[40 * 8192];
extern "C" int goai_debug_wino2s_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino2s_stamps), (size_t)n * sizeof(unsigned long long));
}
#define W2S_STAMP(i) if (k == (n_convs > 1) && threadIdx.x == 0 && blockIdx.x < 8192) { g_wino2s_stamps[blockIdx.x * 40 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0 || (i) == 34) g_wino2s_stamps[blockIdx.x * 40 + 35 + ((i) == 34)] = __builtin_amdgcn_s_memtime(); }
#else
#define W2S_STAMP(i)
#endif
#ifndef W2S_DIAG
#define W2S_DIAG 0      // diagnostic builds only (wrong results): 1 no output transform, 2 no patch reads, 4 no residual loads,
#endif                  // 8 one image only, 16 one weight set only, 32 no stores, 64 every workgroup's residual / output at the first boards' place

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned w2s_v4u __attribute__((vector_size(16)));
// y -= m as two v_pk_fma_f32 by n1 = (-1, -1), a value the optimiser cannot see through: it turns a plain vector
// subtraction (and a multiply-add by a visible -1.0) into four v_sub_f32 -- twice the issue slots.  fma(m, -1, y) = y - m exactly.
__device__ __forceinline__ void w2s_sub4(f32x4& y, const f32x4 m, const f32x2 n1)
{
    const f32x2 lo = __builtin_elementwise_fma(__builtin_shufflevector(m, m, 0, 1), n1, __builtin_shufflevector(y, y, 0, 1));
    const f32x2 hi = __builtin_elementwise_fma(__builtin_shufflevector(m, m, 2, 3), n1, __builtin_shufflevector(y, y, 2, 3));
    y = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// The 16-byte quarter q (8 channels) of V row r sits at position q ^ w2s_vswz(r) of the row's 64 bytes (round 3, as
// conv_wino3h_kernels.hip): the 16 lanes that the LDS serves together for a ds_read_b128 ({0-3, 12-15, 20-27}, ...: rows
// li = 0-3 and 12-15 of one quarter, 4-11 of the next) then touch all 64 banks; unswizzled, every fragment read was a
// 2-way conflict -- 8 waves x 2 KB per tile step at half rate is more than the 96 cycles of the step's MFMAs.
__device__ __forceinline__ int w2s_vswz(int row) { return (0 - (row >> 2)) & 3; }

// two fp32 bit patterns whose low halves are dropped -> one word of two bf16 (a in the low half)
__device__ __forceinline__ unsigned w2s_pack(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// One launch runs n_convs convolutions in a row on the same boards (a workgroup owns whole boards; round 3: the whole
// residual trunk, as conv_wino3h_kernels.hip does on 9x9): convolution 0 is y = relu(conv(x) + bias (+ resid0)), then
// x = relu(conv(y) + bias + x), y = relu(conv(x) + bias), ... : what a workgroup wrote it reads back itself (L2), the
// boards' bounds stay in LDS, the zero borders are filled once per convolution.  wt_all / bias_all / wexp_all: the
// convolutions' weight images, biases and (NP <= 2) weight scale exponents, consecutive.
// NP <= 2: amax_first > 0: bound of every board's input, else amax_io[board]; amax_io (may be NULL) receives the bounds
// of the last output, per board.
// head_out != NULL (the last convolution must have a residual): instead of the last output, head_out[board][3][P] =
// ReLU(the heads' three 1x1 convolutions head_w[3][128], head_b[3] of it), formed as in conv_wino3s_kernels.hip: the
// output's 128 channels never go to memory.
template <int S, int NP>
__global__ __launch_bounds__(512) void k_conv3x3_wino2s(float* x, float* y, const float* resid0, const uint4* wt_all,
                                                        const float* __restrict__ bias_all, int nboards, int n_convs,
                                                        const int* __restrict__ wexp_all, float* amax_io, float amax_first,
                                                        const float* __restrict__ head_w, const float* __restrict__ head_b,
                                                        float* head_out)
{
    using W = Wino2S<S, NP>;
    constexpr int P = W::P, MT = W::MT;
    __shared__ f32x4 lds[W::LDS_U4];
    f32x4* Img = lds;
    f32x4* Vs = lds + W::IMG_U4;
    unsigned* amax_in = reinterpret_cast<unsigned*>(lds + W::LDS_U4 - W::AMAX_U4);
    unsigned* amax_out = amax_in + 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b_first = blockIdx.x * W::NBW;
#ifdef W2S_SLEEP_ALL
    // experiment: EVERY workgroup starts 0..3 x W2S_SLEEP_ALL sleep units late (conv_wino3h_kernels.hip's multi-round form)
    for (int i = (blockIdx.x >> 3) % 4; i > 0; i--) __builtin_amdgcn_s_sleep(W2S_SLEEP_ALL);
#endif
#ifdef W2S_STAGGER
    // experiment: the first round's workgroups start W2S_STAGGER phases apart, so that the epilogues (the only HBM-heavy
    // part) of the workgroups resident at any time do not coincide
    if (blockIdx.x < 256)
        for (int i = 0; i < (int)(blockIdx.x % W2S_STAGGER) * (20 / (W2S_STAGGER - 1) + 1); i++) __builtin_amdgcn_s_sleep(127);
#endif

    const float* in = x;                              // of the current convolution (not __restrict__: the workgroup reads
    float* out = y;                                   // back what it wrote)
    const float* resid = resid0;
    if (NP <= 2 && tid < 16) {
        const bool on = tid < W::NBW && b_first + tid < nboards;
        amax_out[tid] = __float_as_uint(!on ? 0.0f : amax_first > 0.0f ? amax_first : amax_io[b_first + tid]);
    }
    f32x2 neg1 = { -1.0f, -1.0f };                    // (w2s_sub4)
    asm volatile("" : "+v"(neg1));
    const int ts = tid & 7;                           // 4-channel slot of this thread's input-transform items
    // the thread's number as the image staging and the epilogue see it: renewed (opaquely) at every convolution, or the
    // optimiser computes their ~60 offsets, which do not depend on the convolution, once in front of the loop over the
    // convolutions and spills them around the stages
    int tidv = tid;
    f32x4 ireg[W::NI];
    // the workgroup's boards of the current input as a buffer: piece idx = tid + 512 i (cell idx / 8, 16-byte slot idx % 8) of chunk c sits
    // at byte (idx / 8) * 512 + c * 128 + (idx % 8) * 16 -- one vector offset per thread, the rest a scalar offset; pieces beyond the
    // boards that exist lie beyond the descriptor's range and read as zeros (no branches, no 64-bit addresses)
    const int nb_boards = nboards - b_first < W::NBW ? nboards - b_first : W::NBW;
    auto irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)b_first * P * 128, 0, nb_boards * P * 512, 0x00020000);
    auto load_img = [&](int chunk) {
        const unsigned voff = (unsigned)((tidv >> 3) * 512 + (tidv & 7) * 16);
#pragma unroll
        for (int i = 0; i < W::NI; i++)
            ireg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, voff, chunk * 128 + i * 32768, 0));
    };
    auto store_img = [&]() {
#pragma unroll
        for (int i = 0; i < W::NI; i++) {
            const int idx = tidv + i * 512;
            if (idx < W::NBW * P * 8) {
                const int cr = idx >> 3, bl = cr / P, rem = cr - bl * P, yy = rem / S, xx = rem - yy * S;
                f32x4* dst = &Img[W::slot_at(W::cell(bl, yy, xx), idx & 7)];
                if constexpr (NP <= 2)
                    *dst = ireg[i] * __uint_as_float((unsigned)(127 + f16_scale_exp(amax_in[bl], 4)) << 23);
                else
                    *dst = ireg[i];
            }
        }
    };
    // patch cell (a, b) of the item whose cell (-1, -1) (+ its slot) is at byte address `base`
    auto img_at = [&](uint32_t base, int a, int b) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Img) + base + (a * W::ROWC + b) * 64);
    };
    // this thread's two items (wtiles tid / 8 and tid / 8 + 64, slot ts): byte address of patch cell (-1, -1)
    uint32_t tbase[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int tw = (tid >> 3) + 64 * j;
        const int bl = tw / W::T2, rem = tw - bl * W::T2, ty = rem / W::T, tx = rem - ty * W::T;
        tbase[j] = (uint32_t)(W::slot_at(W::cell(tw < W::WT ? bl : 0, 2 * ty - 1, 2 * tx - 1), ts) * 16);
    }
    // V[q][part][tw][ts] = the three bf16 parts of (Bt d B)[p][q], q = 0..3; rows of Bt are
    // (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    auto transform = [&](int p) {
        const int a1 = (p == 0) ? 0 : 1, a2 = (p == 3) ? 3 : 2;
        const float s1 = (p == 2) ? -1.0f : 1.0f, s2 = (p == 0 || p == 3) ? -1.0f : 1.0f;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int tw = (tid >> 3) + 64 * j;
            if (tw >= W::WT) continue;
            f32x4 r[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (W2S_DIAG & 2) r[b] = f32x4{ (float)tbase[j], (float)b, (float)p, s1 };
                else if (s1 > 0.0f && s2 > 0.0f) r[b] = img_at(tbase[j], a1, b) + img_at(tbase[j], a2, b);
                else if (s1 > 0.0f) { r[b] = img_at(tbase[j], a1, b); w2s_sub4(r[b], img_at(tbase[j], a2, b), neg1); }
                else { r[b] = img_at(tbase[j], a2, b); w2s_sub4(r[b], img_at(tbase[j], a1, b), neg1); }
            }
            uint2* V = reinterpret_cast<uint2*>(Vs) + tw * 8 + (((ts >> 1) ^ w2s_vswz(tw)) << 1) + (ts & 1);
            auto put = [&](int q, const f32x4 v) {
                if constexpr (NP == 1) {
                    V[q * (W::V_PART_U4 * 2)] = __builtin_bit_cast(uint2, __builtin_convertvector(v, f16x4));     // GOAI_DTYPE_F16: one part
                } else if constexpr (NP == 2) {
                    // two fp16 parts, round to nearest: h = fp16(v), l = fp16(v - h) (the difference is exact in fp32)
                    // (l by v_fma_mixlo / mixhi_f16: reads h's halves in place, subtracts in fp32, writes a half register)
                    const uint2 hp = __builtin_bit_cast(uint2, __builtin_convertvector(v, f16x4));
                    uint2 lp;
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp.x) : "v"(hp.x), "v"(v[0]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp.x) : "v"(hp.x), "v"(v[1]));
                    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp.y) : "v"(hp.y), "v"(v[2]));
                    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp.y) : "v"(hp.y), "v"(v[3]));
                    uint2* dst = V + q * (2 * W::V_PART_U4 * 2);
                    dst[0] = hp;
                    dst[W::V_PART_U4 * 2] = lp;
                } else {
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    h[e] = __float_as_uint(v[e]) & 0xffff0000u;
                    const float r1 = v[e] - __uint_as_float(h[e]);
                    m[e] = __float_as_uint(r1) & 0xffff0000u;
                    l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));      // <= 8 significant bits: its low half is zero
                }
                uint2* dst = V + q * (3 * W::V_PART_U4 * 2);
                dst[0] = uint2{ w2s_pack(h[0], h[1]), w2s_pack(h[2], h[3]) };
                dst[W::V_PART_U4 * 2] = uint2{ w2s_pack(m[0], m[1]), w2s_pack(m[2], m[3]) };
                dst[(NP - 1) * W::V_PART_U4 * 2] = uint2{ w2s_pack(l[0], l[1]), w2s_pack(l[2], l[3]) };
                }
            };
            f32x4 c0 = r[0], c2 = r[2], c3 = r[1];
            w2s_sub4(c0, r[2], neg1);
            w2s_sub4(c2, r[1], neg1);
            w2s_sub4(c3, r[3], neg1);
            put(0, c0);
            put(1, r[1] + r[2]);
            put(2, c2);
            put(3, c3);
        }
    };

    // weights of a stage: the three bf16 parts of U[stage = (chunk * 4 + p) * 4 + q] in the lanes' own order:
    // lane (li, lq) gets output channel 16 wave + li, input channels 8 lq .. 8 lq + 7 of the chunk
    // [stage][wave][part][lane] x 16 B as buffer loads: descriptor = the convolution's image, scalar offset = the stage,
    // vector offset = the lane's place in the stage -- no address arithmetic in the loop
    constexpr unsigned W_STAGE_BYTES = 8 * 64 * NP * 16;
    const unsigned wlane = (wave * (64 * NP) + lane) * 16;
    const uint4* wt = wt_all;
    auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wt), 0, 64 * W_STAGE_BYTES, 0x00020000);
#ifndef W2S_WAHEAD
#define W2S_WAHEAD 2    // stages the weight loads run ahead of their MFMAs (2: a ring of three buffers, as conv_wino3h_body.inc; 1 -> 2: C4 launch 5.93 -> 5.86 ms, C5 19.63 -> 19.51)
#endif
    // (W2S_WAHEAD == 2: a chunk is a sequence of 16 stages -- position 4 p + q -- plus two idle positions, 18 = 0 mod 3, so that the
    // ring's phase is the same in every chunk; the stage at position pos uses buffer pos % 3 and requests position pos + 2)
    constexpr int WRING = W2S_WAHEAD + 1;
    u32x4 wf[WRING][NP];
    auto load_b = [&](int stage, u32x4* b) {
#pragma unroll
        for (int r = 0; r < NP; r++)
            b[r] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024 * r, stage * W_STAGE_BYTES, 0));
        // keep the load where it is written (the optimiser would sink it to the loop's end, in front of its use;
        // wt is deliberately not __restrict__: loads from a noalias pointer may cross this fence)
        asm volatile("" ::: "memory");
    };
    auto v_read = [&](int q, int t, uint4* v) {
        const uint4* V = reinterpret_cast<const uint4*>(Vs) + (q * NP) * W::V_PART_U4 + (t * 16 + li) * 4 + (lq ^ w2s_vswz(li));
#pragma unroll
        for (int r = 0; r < NP; r++) v[r] = V[r * W::V_PART_U4];
    };

    f32x4 Y[2][2][MT];                               // [i][j]: output (2ty + i, 2tx + j) of the wtile

    // output transform of one tile step: Y[i][j][t] += At[i][p] * At[j][q] * m,  At = (1 1 1 0 ; 0 1 -1 -1), as PLAIN adds /
    // subtracts in the shadow of the next tile step's MFMAs (round 3, tools/ubench/mfma_f16_valu3.hip: a plain vector
    // instruction issues beside another wave's -- and, two per MFMA, its own wave's -- MFMAs for free; a packed fp32 one
    // never does, and the optimiser packs every vector addition it sees: hence the asm).  The `groups` calls after a tile
    // step's MFMAs each take every groups-th of the up to 16 additions.
    f32x4 mr[3];
    auto out_tf = [&](int p, int q, int t, const f32x4& m, int groups, int g) {
        if (W2S_DIAG & 1) {
            if (g == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(Y[0][0][t][0]) : "v"(m[0]));
            return;
        }
        int n = 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int ci = (i == 0) ? (p < 3 ? 1 : 0) : (p == 0 ? 0 : (p == 1 ? 1 : -1));
                const int cj = (j == 0) ? (q < 3 ? 1 : 0) : (q == 0 ? 0 : (q == 1 ? 1 : -1));
                if (ci * cj == 0) continue;
#pragma unroll
                for (int e = 0; e < 4; e++, n++) {
                    if (n % groups != g) continue;
                    if (ci * cj > 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(Y[i][j][t][e]) : "v"(m[e]));
                    else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(Y[i][j][t][e]) : "v"(m[e]));
                }
            }
    };

#pragma unroll 1
    for (int k = 0; k < n_convs; k++) {
    const float* bias = bias_all + k * 128;
    const int wexp = NP <= 2 ? wexp_all[k] : 0;
    // k > 0: every wave is through the previous epilogue (its stores are visible to the workgroup, the exchange area is dead)
    __syncthreads();
    tidv = tid;
    asm volatile("" : "+v"(tidv));
    W2S_STAMP(0);
    if (k > 0) irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)b_first * P * 128, 0, nb_boards * P * 512, 0x00020000);
    load_img(0);
    load_b(0, wf[0]);
    if (W2S_WAHEAD == 2) load_b(1, wf[1]);
    // V rows WT..ROWS-1 and the image's border cells are never written by the loop below: they stay zero (the V rows only
    // reach accumulator columns that are never stored); the epilogue's exchange area lies over both, hence once per convolution
    for (int i = tid; i < W::IMG_U4 + W::V_U4; i += 512) lds[i] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    if (NP <= 2 && tid < 16) {                       // the bounds the previous epilogue (or the prologue) collected
        amax_in[tid] = amax_out[tid];
        amax_out[tid] = 0u;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int t = 0; t < MT; t++) Y[i][j][t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    __syncthreads();                                 // the zero fills are complete
    store_img();
    W2S_STAMP(1);

#pragma unroll 1
    for (int chunk = 0; chunk < 4; chunk++) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            __syncthreads();                         // the image is staged; every wave has read the previous row of V
            if (p == 2 && chunk + 1 < 4 && !(W2S_DIAG & 8)) load_img(chunk + 1);      // in flight during this row
            transform(p);
            __syncthreads();                         // V complete; (p == 3) the image is dead
            W2S_STAMP(2 + 2 * (4 * chunk + p));
            if (p == 3 && chunk + 1 < 4 && !(W2S_DIAG & 8)) store_img();
            uint4 vf[2][NP];                          // (plain words: a bf16 vector across a branch is re-packed element-wise)
            v_read(0, 0, vf[0]);
            // the row's four stages unrolled (round 3): the output transform's factors are compile-time signs, so it is
            // packed adds / subtracts of the products that count only (63 per stage on average; with q a run-time value
            // it was 84 multiply-adds, a third of them by 0.0)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                {
                    const int pos = p * 4 + q;
                    const int qq = W2S_WAHEAD == 2 ? pos % 3 : (q & 1);
                    const int s = chunk * 16 + p * 4 + q;
                    if (W2S_WAHEAD == 2) {
                        if (pos + 2 < 16 && !(W2S_DIAG & 16)) load_b(s + 2, wf[(pos + 2) % 3]);
                    } else
                    if (!(W2S_DIAG & 16)) load_b(s + 1 < 64 ? s + 1 : 63, wf[(qq + 1) & 1]);   // unconditional: the wait counts stay exact
#pragma unroll
                    for (int t = 0; t < MT; t++) {
                        const int n = q * MT + t;    // tile step of the row; its fragments sit in vf[n & 1], its products go to mr[n % 3]
                        if (t + 1 < MT) v_read(q, t + 1, vf[(n + 1) & 1]);
                        else v_read(q < 3 ? q + 1 : 3, 0, vf[(n + 1) & 1]);          // (q == 3: a harmless re-read)
                        // the products of tile step n - 2 (stage (p, qp), tile tp) are added to Y behind this one's MFMAs: nothing
                        // tells the compiler that the asm reads an MFMA's result, so the distance (>= 3 MFMAs = 48 cycles) is
                        // what keeps the read behind the matrix pipe's write
                        const int qp = (n - 2) / MT, tp = (n - 2) % MT;
                        f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
                        if constexpr (NP == 1) {
                            mr[n % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[qq][0]),
                                                                               __builtin_bit_cast(f16x8, vf[n & 1][0]), acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 1, 0);
                        } else if constexpr (NP == 2) {
                            const f16x8 u0 = __builtin_bit_cast(f16x8, wf[qq][0]), u1 = __builtin_bit_cast(f16x8, wf[qq][1]);
                            const f16x8 v0 = __builtin_bit_cast(f16x8, vf[n & 1][0]), v1 = __builtin_bit_cast(f16x8, vf[n & 1][1]);
                            // the three products of weight up to 2^-11, smallest first
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(u1, v0, acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 3, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v1, acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 3, 1);
                            mr[n % 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(u0, v0, acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 3, 2);
                        } else {
                            const bf16x8 u0 = __builtin_bit_cast(bf16x8, wf[qq][0]), u1 = __builtin_bit_cast(bf16x8, wf[qq][1]),
                                         u2 = __builtin_bit_cast(bf16x8, wf[qq][NP - 1]);
                            const bf16x8 v[3] = { __builtin_bit_cast(bf16x8, vf[n & 1][0]), __builtin_bit_cast(bf16x8, vf[n & 1][1]),
                                                  __builtin_bit_cast(bf16x8, vf[n & 1][NP - 1]) };
                            // the six products of weight up to 2^-16, smallest first
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u2, v[0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u0, v[2], acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 3, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u1, v[1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u1, v[0], acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 3, 1);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u0, v[1], acc, 0, 0, 0);
                            mr[n % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u0, v[0], acc, 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (n > 1) out_tf(p, qp, tp, mr[(n + 1) % 3], 3, 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);             // keep the pipeline as written
                    }
                }
            }
            // the row's last two tile steps (numbers 4 * MT - 2 and 4 * MT - 1), the last one at a safe distance from its MFMAs
            out_tf(p, 3, MT - 2, mr[(4 * MT - 2) % 3], 1, 0);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
            out_tf(p, 3, MT - 1, mr[(4 * MT - 1) % 3], 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (W2S_WAHEAD == 2 && p == 3 && !(W2S_DIAG & 16)) {      // the chunk's two idle positions: the next chunk's first two stages
                load_b(chunk * 16 + 16 < 64 ? chunk * 16 + 16 : 63, wf[0]);
                load_b(chunk * 16 + 17 < 64 ? chunk * 16 + 17 : 63, wf[1]);
            }
            W2S_STAMP(3 + 2 * (4 * chunk + p));
        }
    }

    // epilogue.  The products are formed transposed (A operand = weights, B operand = V), so in the C/D layout of
    // a 16x16 tile (col = lane & 15, row = 4 * (lane >> 4) + reg) a lane holds FOUR CONSECUTIVE output channels
    // of ONE wtile -- but NEIGHBOURING LANES hold different wtiles, cells 512 bytes apart in memory: stored from there,
    // every 16-byte piece is a cache-line access of its own, and the texture path's one line per clock, not HBM, sets
    // the epilogue's length (12-13 us of a workgroup's 67 at 13x13).  Round 3: each tile's cells go through a
    // wave-private kilobyte of LDS and come back with four consecutive lanes on the 64 bytes a wave owns of one
    // cell (lane L: wtile 16 t + L / 4, channels 4 (L & 3) ..): residual loads and stores are then 64-byte runs.
    // Residual and output are buffer accesses relative to the workgroup's first board: 32-bit offsets, and a cell off
    // the board (or a tile / board that does not exist) gets an offset beyond the descriptor's range -- its load
    // returns zeros and its store is dropped, without a branch.
    auto epilogue = [&](auto has_resid, auto to_heads) {
    constexpr bool RESID = decltype(has_resid)::value, HEADS = decltype(to_heads)::value;
    const int lane_e = tidv & 63;
    const int li = lane_e & 15, lq = lane_e >> 4;            // (shadow the kernel's: see tidv)
    const int eq = lane_e & 3, ew = lane_e >> 2;             // the lane's channel quad and wtile (of 16) after the exchange
    const int col = wave * 16 + 4 * eq;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(&bias[col]);
    f32x4 hwv[3];
    float* hp = reinterpret_cast<float*>(lds) + (size_t)((wave * 4 + eq) * (W::NBW * P)) * 3;
    if constexpr (HEADS) {
#pragma unroll
        for (int c = 0; c < 3; c++) hwv[c] = *reinterpret_cast<const f32x4*>(&head_w[c * 128 + col]);
    }
    __syncthreads();                                         // every wave has read the last row of V: LDS is free
    W2S_STAMP(37);
    f32x4* xch = HEADS ? lds + W::HP_U4 + wave * W::XCH_HEADS_WAVE_U4 : lds + wave * W::XCH_WAVE_U4;
    // tile t's four cells from lane (li, lq) = (wtile, quad) into the exchange area (17: no bank conflicts on either side)
    auto xch_put = [&](int t) {
#pragma unroll
        for (int ij = 0; ij < 4; ij++) xch[((t & 1) * 4 + ij) * 72 + lq * 17 + li] = Y[ij >> 1][ij & 1][t];
        asm volatile("" ::: "memory");
    };
    const int nb_here = nboards - b_first < W::NBW ? nboards - b_first : W::NBW;
    const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RESID ? resid : bias) + (RESID && !(W2S_DIAG & 64) ? (size_t)b_first * P * 128 : 0), 0,
                                                         RESID ? nb_here * P * 512 : 0, 0x00020000);
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + ((W2S_DIAG & 64) ? 0 : (size_t)b_first * P * 128), 0, HEADS ? 0 : nb_here * P * 512, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    int cur_b = -1;                                          // NP <= 2: the board this lane's tiles have been on, and the
    float cur_m = 0.0f;                                      // largest value it stored for it
    auto flush_bound = [&]() {
        if (cur_b >= 0 && __float_as_uint(cur_m) > amax_out[cur_b]) atomicMax(&amax_out[cur_b], __float_as_uint(cur_m));
    };
    // offsets, board and residual of tile t in the lane's new place
    auto tile_at = [&](int t, unsigned* vo, int& brd, f32x4* rv) {
        const int w = t * 16 + ew;
        const int bl = w / W::T2, rem = w - bl * W::T2;
        const bool tile_ok = w < W::WT && bl < nb_here;
        brd = tile_ok ? bl : -1;
        const int ty = rem / W::T, tx = rem - ty * W::T;
        const unsigned tb = (unsigned)(((bl * P + 2 * ty * S + 2 * tx) * 128 + col) * 4);
#pragma unroll
        for (int ij = 0; ij < 4; ij++) {
            const bool ok = tile_ok && 2 * ty + (ij >> 1) < S && 2 * tx + (ij & 1) < S;
            vo[ij] = ok ? tb + (unsigned)(((ij >> 1) * S + (ij & 1)) * 512) : OOB;
            if (RESID && !(W2S_DIAG & 4)) rv[ij] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vo[ij], 0, 0));
            else rv[ij] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
    };
    unsigned vo[2][4];
    int brd[2];
    f32x4 rv[2][4];
    tile_at(0, vo[0], brd[0], rv[0]);
    if constexpr (!HEADS) xch_put(0);
#pragma unroll
    for (int t = 0; t < MT; t++) {
        if (t + 1 < MT) {                                    // the next tile's residual and its cells are on their way
            tile_at(t + 1, vo[(t + 1) & 1], brd[(t + 1) & 1], rv[(t + 1) & 1]);
            if constexpr (!HEADS) xch_put(t + 1);
        }
        const int tb = t & 1;
        // NP <= 2: the accumulators are scaled back by 2^-(input scale of the tile's board + weight scale)
        const int back = NP <= 2 ? -(f16_scale_exp(amax_in[brd[tb] < 0 ? 0 : brd[tb]], 4) + wexp) : 0;
        float mx = 0.0f;
#pragma unroll
        for (int ij = 0; ij < 4; ij++) {
            f32x4 v;                                                         // the cell as lane L sees it: wtile L / 4, quad L & 3
            if constexpr (HEADS) {
                xch[(ij & 1) * 72 + lq * 17 + li] = Y[ij >> 1][ij & 1][t];
                asm volatile("" ::: "memory");
                v = xch[(ij & 1) * 72 + eq * 17 + ew];
                asm volatile("" ::: "memory");
            } else
                v = xch[((t & 1) * 4 + ij) * 72 + eq * 17 + ew];
            if constexpr (NP <= 2) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_ldexpf(v[e], back);
            }
            v += bv;
            if (RESID) v += rv[tb][ij];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.0f);
            if constexpr (HEADS) {
                if (vo[tb][ij] != OOB) {
                    const int lcell = (int)(vo[tb][ij] >> 9);                            // (offset = (cell * 128 + col) * 4, col < 128)
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        hp[lcell * 3 + c] = fmaf(v[3], hwv[c][3], fmaf(v[2], hwv[c][2], fmaf(v[1], hwv[c][1], v[0] * hwv[c][0])));
                }
            } else if (!(W2S_DIAG & 32))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w2s_v4u, v), orsrc, vo[tb][ij], 0, 0);
            if constexpr (NP <= 2) {
                const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                mx = fmaxf(mx, vo[tb][ij] != OOB ? m4 : 0.0f);
            }
        }
        asm volatile("" ::: "memory");
        if constexpr (NP <= 2) {
            // the largest value stored for the tile's board; into LDS when the lane's tiles move on to the next board
            if (brd[tb] != cur_b) {
                flush_bound();
                cur_b = brd[tb];
                cur_m = 0.0f;
            }
            cur_m = fmaxf(cur_m, mx);
        }
    }
    if constexpr (NP <= 2) flush_bound();
    W2S_STAMP(38);
    if constexpr (HEADS) {
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
    };
    const bool heads_now = head_out != nullptr && k == n_convs - 1;
    if (heads_now) epilogue(std::true_type{}, std::true_type{});
    else if (resid) epilogue(std::true_type{}, std::false_type{});
    else epilogue(std::false_type{}, std::false_type{});
    if (NP <= 2 && amax_io && k == n_convs - 1 && !heads_now) {
        __syncthreads();
        if (tid < W::NBW && b_first + tid < nboards) amax_io[b_first + tid] = __uint_as_float(amax_out[tid]);
    }
    W2S_STAMP(34);
    // next convolution of the launch: swap the buffers, step the weights
    if (k & 1) { in = x; out = y; resid = resid0; } else { in = y; out = x; resid = x; }
    wt += 64 * 8 * 64 * NP;
    wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wt), 0, 64 * W_STAGE_BYTES, 0x00020000);
    }
}

template <int S, int NP>
static int launch_wino2s_size(float* x, float* y, const float* resid0, const void* wt_split, const float* bias, int nboards, int n_convs,
                              hipStream_t st, const int* wexp, float* amax_io, float amax_first, const float* head_w,
                              const float* head_b, float* head_out)
{
    const int grid = (nboards + Wino2S<S, NP>::NBW - 1) / Wino2S<S, NP>::NBW;
    k_conv3x3_wino2s<S, NP><<<dim3(grid), dim3(512), 0, st>>>(x, y, resid0, reinterpret_cast<const uint4*>(wt_split), bias, nboards, n_convs, wexp,
                                                              amax_io, amax_first, head_w, head_b, head_out);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

// n_convs = 1: y = relu(conv(x) + bias (+ resid0)); n_convs = 2 * blocks, resid0 = NULL: the whole residual trunk in one
// launch (weights / biases / scale exponents of consecutive convolutions contiguous), result in x.
// wt_split: [n_convs][64 stages][8 waves][parts][64 lanes][8 x 16 bit] (nn_host.hip).
// parts = 3: bf16; parts = 2 / 1: fp16, which also takes the weights' scale exponents wexp[n_convs] (device memory) and the
// boards' bounds: the first input's are amax_first > 0 for every board or amax_io[nboards]; amax_io (may be NULL when
// amax_first > 0) receives the last output's.
// head_out != NULL: it receives [board][3][P] = the heads' 1x1 convolutions (head_w, head_b) of the last output (see the
// kernel) instead of the output going to memory; the last convolution must be one with a residual.
int launch_conv3x3_wino2s(int S_rt, float* x, float* y, const float* resid0, const void* wt_split, const float* bias, int nboards,
                          int n_convs, hipStream_t st, int parts, const int* wexp, float* amax_io, float amax_first,
                          const float* head_w, const float* head_b, float* head_out)
{
    if (nboards < 1 || n_convs < 1 || (n_convs > 1 && ((n_convs & 1) || resid0))) {
        set_error("the F(2x2,3x3) trunk launch runs whole residual blocks (an even number of convolutions, no outer residual) or one convolution");
        return GOAI_ERR_ARG;
    }
    if (head_out && (!head_w || !head_b || (n_convs == 1 && !resid0))) {
        set_error("the heads' 1x1 convolutions are formed by a block's second convolution (with its residual)");
        return GOAI_ERR_ARG;
    }
    if (parts <= 2 && (!wexp || (!amax_io && !(amax_first > 0.0f)))) {
        set_error("the fp16 F(2x2,3x3) convolution needs the weights' scale exponents and the boards' bounds");
        return GOAI_ERR_ARG;
    }
    if (parts == 1) {
        GOAI_DISPATCH_SIZE(S_rt, return (launch_wino2s_size<S, 1>(x, y, resid0, wt_split, bias, nboards, n_convs, st, wexp, amax_io, amax_first, head_w, head_b, head_out)));
        return GOAI_OK;
    }
    if (parts == 2) {
        GOAI_DISPATCH_SIZE(S_rt, return (launch_wino2s_size<S, 2>(x, y, resid0, wt_split, bias, nboards, n_convs, st, wexp, amax_io, amax_first, head_w, head_b, head_out)));
        return GOAI_OK;
    }
    GOAI_DISPATCH_SIZE(S_rt, return (launch_wino2s_size<S, 3>(x, y, resid0, wt_split, bias, nboards, n_convs, st, nullptr, nullptr, 0.0f, head_w, head_b, head_out)));
    return GOAI_OK;
}

}  // namespace goai
