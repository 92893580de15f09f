// conv_bf16_kernels.hip -- the trunk's conv3x3 128 -> 128 (+ folded BatchNorm bias, optional
// residual, ReLU; /root/reference/go_ai/models/__init__.py:234-251) on the bf16 matrix cores
// with SPLIT operands (GOAI_DTYPE_BF16X3): every fp32 activation x and weight w is split into
// two bf16 numbers, x = xh + xl, w = wh + wl (xh = bf16(x), xl = bf16(x - xh); 16 significant
// bits together), and each product is taken as  xh*wh + xl*wh + xh*wl  with fp32 accumulation
// (v_mfma_f32_16x16x32_bf16).  The dropped term xl*wl and the split residues are ~2^-17
// relative, so the outputs agree with the exact-fp32 kernel (conv_kernels.hip) to ~1e-5 --
// inside BASELINE.json's fp32 tolerance (1e-3) -- at a third of the bf16 MFMA rate (5.3x the
// fp32 MFMA rate).  A single bf16 product per term (plain bf16) was measured at 3e-2 maximum
// error on the policy logits of the 3-block net and is therefore not offered.
//
// Same tensors as the fp32 kernel: activations stay fp32 [M = boards*P][128] in HBM; they are
// split while they are staged into LDS; weights are split once on the host.
//
// Workgroup = 4 waves as a 2 x 2 grid over a BM = 32*RTW rows x 128 columns tile: wave (wr, wc)
// owns RTW row tiles x 4 column tiles of 16x16 (16*RTW accumulator registers); an A fragment
// pair (two ds_read_b128) feeds 12 MFMAs and the 8 B fragments of a stage stay in registers
// across the row tiles.
// K = 9 taps x 128 channels is walked as 4 channel chunks x 9 taps; a stage = one tap of one
// 32-channel chunk = one K = 32 MFMA block per operand pair:
//   - the A image (tile rows + a halo of S+1 rows either side; a row = 32 hi | 32 lo bf16 =
//     128 B) is staged once per chunk and re-read per tap with a row shift; off-board taps read
//     a zero row;
//   - the B image (W_tap^T, [128 out][32 hi | 32 lo] = 16 KiB, stored in HBM in this order) is
//     double-buffered through registers.
// LDS rows are 128 B (eight 16-byte slots), stored in transposed blocks of 16 rows (swz16 below).
// A workgroup needs <= 79 KiB, so two share a CU (2 waves per SIMD).
#include <cstdlib>

#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// LDS images are stored "transposed" in blocks of 16 rows: the 16-byte slot s of rows 16b .. 16b+15
// is one contiguous 256-byte bank line, so the 16 lanes of a ds_read_b128 lane group -- which
// always hold 16 consecutive rows, whatever slots their k quarters select -- hit 16 different
// bank slots for every tap shift (an XOR swizzle of row-major rows conflicts for odd shifts).
// Index (in 16-byte units) of slot s of row r:
__device__ __forceinline__ int swz16(int row, int slot) { return (((row >> 4) << 3) + slot) * 16 + (row & 15); }

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const f32x2 x = { a, b };
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}

// x = hi + lo: hi = bf16(x) (RNE), lo = bf16(x - hi); x - hi is exact in fp32
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo)
{
    hi = pack_bf16(a, b);
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    lo = pack_bf16(a - ah, b - bh);
}

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, uint4& vh, uint4& vl)
{
    split2(x0[0], x0[1], vh.x, vl.x);
    split2(x0[2], x0[3], vh.y, vl.y);
    split2(x1[0], x1[1], vh.z, vl.z);
    split2(x1[2], x1[3], vh.w, vl.w);
}

#ifdef GOAI_CONV_STAMPS
__device__ unsigned long long g_bf_stamps[8 * 4096];   // diagnostic build only: start / loop start / loop end / end
extern "C" int goai_debug_bf_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bf_stamps), (size_t)n * sizeof(unsigned long long));
}
#define BF_STAMP(i) if (threadIdx.x == 0) g_bf_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define BF_STAMP(i)
#endif

template <int S, int RTW>
struct ConvBfCfg {
    static constexpr int BM = 32 * RTW;
    static constexpr int HALO = S + 1;
    static constexpr int ROWS_A = BM + 2 * HALO;               // data rows; row ROWS_A is the zero row
    static constexpr int A_U4 = ((ROWS_A + 1 + 15) / 16) * 16 * 8;   // whole blocks of 16 rows
    static constexpr int B_U4 = 128 * 8;
    static constexpr int NA = (((ROWS_A + 15) / 16) * 64 + 255) / 256;   // 8-channel pieces per thread per A chunk
};

template <int S, int RTW, bool RESID>
__global__ __launch_bounds__(256, 2) void k_conv3x3_bf16(const float* __restrict__ in, const uint4* __restrict__ wt,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ resid, float* __restrict__ out, int M)
{
    using G = Geom<S>;
    using C = ConvBfCfg<S, RTW>;
    __shared__ uint4 lds[C::A_U4 + 2 * C::B_U4];
    uint4* As = lds;
    uint4* Bs = lds + C::A_U4;

    BF_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * C::BM;
    const int li = lane & 15, lq = lane >> 4;

    if (tid < 8) As[swz16(C::ROWS_A, tid)] = uint4{ 0u, 0u, 0u, 0u };   // the zero row

    // bit `tap` of vmask[t]: the tap's source cell of row 16*(wr*RTW + t) + li is on the board
    int vmask[RTW];
#pragma unroll
    for (int t = 0; t < RTW; t++) {
        const int m = m0 + (wr * RTW + t) * 16 + li;
        const int p = m % G::P, r = p / S, c = p % S;
        int mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(r + dy) < (unsigned)S && (unsigned)(c + dx) < (unsigned)S) mask |= 1 << tap;
        }
        vmask[t] = (m < M) ? mask : 0;
    }
    int brow[4];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) brow[ct] = wc * 64 + ct * 16 + li;

    f32x4 acc[RTW][4];
#pragma unroll
    for (int t = 0; t < RTW; t++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[t][ct] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    // raw fp32 pieces of the next chunk's A image; they are split into bf16 halves only when they are
    // stored (store_a), so the wait for these loads comes after the stage's MFMAs, not before
    f32x4 araw0[C::NA], araw1[C::NA];

    auto load_a = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < C::NA; i++) {
            const int idx = tid + i * 256;
            const int row = (idx & 15) | ((idx >> 6) << 4), piece = (idx >> 4) & 3;
            const long grow = (long)m0 - C::HALO + row;
            f32x4 x0 = { 0.0f, 0.0f, 0.0f, 0.0f }, x1 = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (row < C::ROWS_A && grow >= 0 && grow < M) {
                const float* src = &in[grow * 128 + chunk * 32 + piece * 8];
                x0 = *reinterpret_cast<const f32x4*>(src);
                x1 = *reinterpret_cast<const f32x4*>(src + 4);
            }
            araw0[i] = x0;
            araw1[i] = x1;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < C::NA; i++) {
            const int idx = tid + i * 256;
            const int row = (idx & 15) | ((idx >> 6) << 4), piece = (idx >> 4) & 3;
            if (row < C::ROWS_A) {
                uint4 vh, vl;
                split8(araw0[i], araw1[i], vh, vl);
                As[swz16(row, piece)] = vh;
                As[swz16(row, 4 + piece)] = vl;
            }
        }
    };
    // (four named registers, not an array: hipcc "promotes" a uint4 breg[4] that lives across the
    // stage into LDS, which turns the prefetch into load -> wait -> LDS -> LDS)
    const int bsrc0 = ((((tid & 15) | ((tid >> 7) << 4)) << 3) + ((tid >> 4) & 7));     // idx = tid
    const int bdst0 = swz16((tid & 15) | ((tid >> 7) << 4), (tid >> 4) & 7);
    uint4 breg0, breg1, breg2, breg3;
    auto load_b = [&](int stage) {
        const uint4* src = wt + (size_t)stage * C::B_U4 + bsrc0;     // idx = tid + 256 i: rows advance by 32 i
        breg0 = src[0];
        breg1 = src[32 * 8];
        breg2 = src[64 * 8];
        breg3 = src[96 * 8];
    };
    auto store_b = [&](int buf) {
        uint4* dst = Bs + buf * C::B_U4 + bdst0;                     // 32 rows = 2 blocks of 16 rows = 256 units
        dst[0] = breg0;
        dst[256] = breg1;
        dst[512] = breg2;
        dst[768] = breg3;
    };
    // A fragments (hi, lo) of this wave's row tile t for the tap whose flat row offset is toff
    auto a_frag = [&](int tap, int toff, int t, bf16x8& hi, bf16x8& lo) {
        const int row = ((vmask[t] >> tap) & 1) ? ((wr * RTW + t) * 16 + li + C::HALO + toff) : C::ROWS_A;
        hi = __builtin_bit_cast(bf16x8, As[swz16(row, lq)]);
        lo = __builtin_bit_cast(bf16x8, As[swz16(row, 4 + lq)]);
    };
    auto tap_off = [](int tap) { return (tap / 3 - 1) * S + (tap - (tap / 3) * 3 - 1); };

    load_a(0);
    load_b(0);
    store_a();
    store_b(0);
    __syncthreads();

    // a stage is RTW steps (one per row tile), each = one A fragment pair x 12 MFMAs; the pair of
    // step i+1 is in flight while step i computes -- across stage boundaries too, as long as the
    // A image stays (same chunk)
    constexpr int NSTAGE = 4 * 9;
    BF_STAMP(1);
    const int prio_slot = (blockIdx.x / 256) & 1;
    bf16x8 ah[2], al[2];
    a_frag(0, tap_off(0), 0, ah[0], al[0]);
    for (int s = 0; s < NSTAGE; s++) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const int buf = s & 1;
#ifndef ABL_NOPRIO
        if ((s + prio_slot) & 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
        if (s + 1 < NSTAGE) load_b(s + 1);
        const bool next_chunk = (tap == 8) && (chunk + 1 < 4);
        if (next_chunk) load_a(chunk + 1);

        const int toff = tap_off(tap), toff_next = tap_off(tap + 1);
        const uint4* Bb = Bs + buf * C::B_U4;
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
            bh[ct] = __builtin_bit_cast(bf16x8, Bb[swz16(brow[ct], lq)]);
            bl[ct] = __builtin_bit_cast(bf16x8, Bb[swz16(brow[ct], 4 + lq)]);
        }
#pragma unroll
        for (int t = 0; t < RTW; t++) {
            const int cur = t & 1;                        // ring slot of this step
            if (t + 1 < RTW) a_frag(tap, toff, t + 1, ah[cur ^ 1], al[cur ^ 1]);
            else if (tap != 8) a_frag(tap + 1, toff_next, 0, ah[cur ^ 1], al[cur ^ 1]);
            const bf16x8 xh = ah[cur], xl = al[cur];
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
                acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh[ct], acc[t][ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
                acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh[ct], acc[t][ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
                acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl[ct], acc[t][ct], 0, 0, 0);
#ifndef ABL_NOSCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // keep the LDS stores of the next stage's weights BEHIND this stage's MFMAs: left alone, hipcc
        // hoists them to just after their global loads and exposes the L2 latency in every stage
        asm volatile("" ::: "memory");
        if (s + 1 < NSTAGE) store_b(buf ^ 1);      // buf^1 was last read in stage s-1
        if (next_chunk) {
            __syncthreads();                        // everyone is done with this chunk's A image
            store_a();
        }
        __syncthreads();
        // the pair prefetched by the last step sits in slot (RTW & 1); the next stage starts at slot 0
        if (tap != 8) {
            if (RTW & 1) { ah[0] = ah[1]; al[0] = al[1]; }
        } else if (s + 1 < NSTAGE) {
            a_frag(0, tap_off(0), 0, ah[0], al[0]);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    BF_STAMP(2);

    // epilogue.  C/D layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg.
    const bool full = m0 + C::BM <= M;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        const int col = wc * 64 + ct * 16 + li;
        const float bv = bias[col];
        if (full) {
            f32x4 rv[RTW];
            if (RESID) {
#pragma unroll
                for (int t = 0; t < RTW; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        rv[t][r] = resid[((long)m0 + (wr * RTW + t) * 16 + 4 * lq + r) * 128 + col];
            }
#pragma unroll
            for (int t = 0; t < RTW; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[t][ct][r] + bv;
                    if (RESID) v += rv[t][r];
                    out[((long)m0 + (wr * RTW + t) * 16 + 4 * lq + r) * 128 + col] = fmaxf(v, 0.0f);
                }
        } else {
#pragma unroll
            for (int t = 0; t < RTW; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const long row = (long)m0 + (wr * RTW + t) * 16 + 4 * lq + r;
                    if (row < M) {
                        float v = acc[t][ct][r] + bv;
                        if (RESID) v += resid[row * 128 + col];
                        out[row * 128 + col] = fmaxf(v, 0.0f);
                    }
                }
        }
    }
    BF_STAMP(3);
}

// Rows per workgroup = 32 * RTW.  Pick RTW in 3..6 minimising (rounds over the chip's workgroup
// slots, two per CU) x (RTW + 3): a workgroup's time is its rows plus a fixed part (weight
// streaming, prologue, epilogue) worth about three row tiles; ties go to the taller tile.
static int pick_rtw(int M)
{
    const long slots = 256 * 2;
    int best = 6;
    long best_cost = -1;
    for (int rtw = 6; rtw >= 3; rtw--) {
        const long wgs = (M + 32 * rtw - 1) / (32 * rtw);
        const long rounds = (wgs + slots - 1) / slots;
        const long rem = wgs - (rounds - 1) * slots;
        const long depth = (rounds - 1) * 2 + (rem + 255) / 256;
        const long cost = depth * (rtw + 3);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rtw; }
    }
    return best;
}

template <int S, int RTW>
static void launch_rtw(const float* in, const uint4* wt, const float* bias, const float* resid, float* out, int M,
                       hipStream_t st)
{
    const int grid = (M + 32 * RTW - 1) / (32 * RTW);
    if (resid) k_conv3x3_bf16<S, RTW, true><<<dim3(grid), dim3(256), 0, st>>>(in, wt, bias, resid, out, M);
    else k_conv3x3_bf16<S, RTW, false><<<dim3(grid), dim3(256), 0, st>>>(in, wt, bias, resid, out, M);
}

template <int S>
static void launch_bf_s(int rtw, const float* in, const uint4* wt, const float* bias, const float* resid, float* out,
                        int M, hipStream_t st)
{
    switch (rtw) {
        case 3: launch_rtw<S, 3>(in, wt, bias, resid, out, M, st); break;
        case 4: launch_rtw<S, 4>(in, wt, bias, resid, out, M, st); break;
        case 5: launch_rtw<S, 5>(in, wt, bias, resid, out, M, st); break;
        default: launch_rtw<S, 6>(in, wt, bias, resid, out, M, st); break;
    }
}

int launch_conv3x3_bf16(int S_rt, const float* in, const void* wt_bf16, const float* bias, const float* resid,
                        float* out, int M, hipStream_t st)
{
    static const int forced = [] { const char* e = getenv("GOAI_CONV_RTW"); return e ? atoi(e) : 0; }();   // tuning aid
    const int rtw = (forced >= 3 && forced <= 6) ? forced : pick_rtw(M);
    GOAI_DISPATCH_SIZE(S_rt, launch_bf_s<S>(rtw, in, (const uint4*)wt_bf16, bias, resid, out, M, st));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
