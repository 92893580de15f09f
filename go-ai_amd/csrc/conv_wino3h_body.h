// conv_wino3h_body.h -- the workgroup-level body of round 3's fp16 F(3x3,3x3) trunk kernel for 9x9 boards (see
// conv_wino3h_kernels.hip, whose kernel is a call of wino3h_trunk), as a header so that the search's persistent step kernel
// (mcts_kernels.hip) can run the same code between its own phases.  Translation units that include this need
// -mllvm -pragma-unroll-threshold=200000 (csrc/Makefile).
#pragma once
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

}  // namespace goai
