// conv_kernels.hip -- the dominant kernel: conv3x3 128 -> 128 (+ folded BatchNorm bias,
// optional residual, ReLU) of the trunk (/root/reference/go_ai/models/__init__.py:234-251)
// as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma
// chains, 64 FLOP/clk/SIMD).
//
//   out[M = boards*P][128] = relu( sum_{tap, cin} shift_tap(act)[M][cin] * W[tap][cin][cout] + b (+ resid) )
//
// Workgroup = 4 waves = BM = 16*RT flat rows x 128 output channels (tiles may span boards).
// Wave w owns all RT row tiles x columns [32w, 32w+32): RT x 2 accumulator tiles of 16x16.
// K = 9 taps x 128 input channels is walked as 4 channel chunks x 9 taps ("stages"):
//   - the A image (tile rows + a halo of S+1 rows either side, 32 channels) is staged in LDS
//     once per chunk and re-read per tap with a row shift; rows whose tap falls off the board
//     read a dedicated zero row, so the inner loop is branch-free;
//   - the B image (W_tap^T, [128 out][32 in], 16 KiB, stored in HBM already in this order) is
//     double-buffered through registers: the next stage's weights are in flight while this
//     stage's MFMAs run.
// LDS rows are 32 floats (eight 16-byte slots); slot s of row r lives at slot s ^ ((r >> 1) & 7):
// every ds_read_b128 lane group touches 16 rows that are distinct mod 16, hence 16 distinct
// bank slots, with no padding.  At RT <= 8 a workgroup needs <= 52 KiB and <= 168 VGPRs, so
// three workgroups (12 waves, 3 per SIMD) share a CU.
//
// RT is chosen per launch (5..8) so that the grid fills the chip's 256 x 3 workgroup slots
// with the fewest, fullest rounds: e.g. 1024 boards of 9x9 = 82,944 rows run as 741
// workgroups of 112 rows in ONE round instead of 648 workgroups of 128 rows that leave a
// third of the CUs one workgroup short.
#include <cstdlib>

#include "go_board.h"
#include "goai_internal.h"
#include "nn.h"

namespace goai {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;      // input channels per stage
constexpr int LD = 32;      // floats per LDS row

__device__ __forceinline__ int swz(int row, int slot) { return row * LD + ((slot ^ ((row >> 1) & 7)) << 2); }

#ifdef GOAI_CONV_STAMPS
__device__ unsigned long long g_conv_stamps[8 * 4096];   // diagnostic build only: start / loop end / end / xcc per workgroup
#endif

template <int S, int RT>
struct ConvCfg {
    static constexpr int BM = 16 * RT;
    static constexpr int HALO = S + 1;
    static constexpr int ROWS_A = BM + 2 * HALO;               // data rows; row ROWS_A is the zero row
    static constexpr int A_FLOATS = (ROWS_A + 1) * LD;
    static constexpr int B_FLOATS = 128 * LD;
    static constexpr int NA = (ROWS_A * 8 + 255) / 256;        // float4 per thread per A chunk
    static constexpr int LDS_BYTES = (A_FLOATS + 2 * B_FLOATS) * 4;
};

// SPLIT: blockIdx.y = one 32-channel chunk of the input; the workgroup walks that chunk's 9 stages only and
// writes its raw partial sums to out + blockIdx.y * M * 128 (k_conv3x3_sum4 adds the four up in a fixed order).  For
// batches whose row tiles alone cannot fill the chip (the training step at the reference's batch size).
template <int S, int RT, bool RESID, bool SPLIT = false>
__global__ __launch_bounds__(256, 3) void k_conv3x3(const float* __restrict__ in, const float* __restrict__ wt,
                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                    float* __restrict__ out_base, int M, float floor_v)
{
    using G = Geom<S>;
    float* __restrict__ out = SPLIT ? out_base + (size_t)blockIdx.y * M * 128 : out_base;
    const int s_begin = SPLIT ? blockIdx.y * 9 : 0, s_end = SPLIT ? s_begin + 9 : 4 * 9;
    using C = ConvCfg<S, RT>;
    __shared__ __attribute__((aligned(16))) float lds[C::A_FLOATS + 2 * C::B_FLOATS];
    float* As = lds;
    float* Bs = lds + C::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * C::BM;
#ifdef GOAI_CONV_STAMPS
    if (tid == 0) {
        g_conv_stamps[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        g_conv_stamps[blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memtime();
    }
#endif
    const int li = lane & 15, lq = lane >> 4;      // fragment row / column, k quarter

    if (tid < LD) As[C::ROWS_A * LD + tid] = 0.0f;  // the zero row

    // per-lane geometry of the RT row tiles (lane li supplies row 16*t + li of each)
    // bit `tap` of vmask[t]: the tap's source cell is on the board (and the row exists)
    int vmask[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int m = m0 + t * 16 + li;
        const int p = m % G::P, r = p / S, c = p % S;
        int mask = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(r + dy) < (unsigned)S && (unsigned)(c + dx) < (unsigned)S) mask |= 1 << tap;
        }
        vmask[t] = (m < M) ? mask : 0;
    }
    int brow[2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) brow[tn] = wave * 32 + tn * 16 + li;

    f32x4 acc[RT][2];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) acc[t][tn] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };

    f32x4 breg[4];
    f32x4 areg[C::NA];

    auto load_a = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < C::NA; i++) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, piece = idx & 7;
            const long grow = (long)m0 - C::HALO + row;
            f32x4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (row < C::ROWS_A && grow >= 0 && grow < M)
                v = *reinterpret_cast<const f32x4*>(&in[grow * 128 + chunk * CK + piece * 4]);
            areg[i] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < C::NA; i++) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, piece = idx & 7;
            if (row < C::ROWS_A) *reinterpret_cast<f32x4*>(&As[swz(row, piece)]) = areg[i];
        }
    };
    auto load_b = [&](int stage) {
        const float* src = wt + (size_t)stage * 128 * CK;
#pragma unroll
        for (int i = 0; i < 4; i++) breg[i] = *reinterpret_cast<const f32x4*>(&src[(tid + i * 256) * 4]);
    };
    auto store_b = [&](int buf) {
        float* dst = Bs + buf * C::B_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(&dst[swz(idx >> 3, idx & 7)]) = breg[i];
        }
    };

    // A fragment of row tile t, k block kk, for the tap whose flat row offset is toff
    auto a_frag = [&](int tap, int toff, int t, int kk) -> f32x4 {
#if defined(ABL_NO_AREAD)
        return f32x4{ (float)tap, (float)toff, (float)t, (float)kk };
#elif defined(ABL_FIXED_ADDR)
        return *reinterpret_cast<const f32x4*>(&As[(t * 16 + li) * LD + kk * 16 + lq * 4]);
#else
        const int row = ((vmask[t] >> tap) & 1) ? (t * 16 + li + C::HALO + toff) : C::ROWS_A;
        return *reinterpret_cast<const f32x4*>(&As[swz(row, kk * 4 + lq)]);
#endif
    };
    auto tap_off = [](int tap) { return (tap / 3 - 1) * S + (tap - (tap / 3) * 3 - 1); };

    load_a(s_begin / 9);
    load_b(s_begin);
    store_a();
    store_b(0);
    __syncthreads();

    // Software pipeline: a stage is 2*RT steps (k block kk = step / RT, row tile t = step % RT),
    // each = one A fragment x 8 MFMAs.  The A fragments of steps i+1 and i+2 are in flight while
    // step i computes -- across stage boundaries too, as long as the A image stays (same chunk).
    constexpr int NSTEP = 2 * RT;
    f32x4 a[3];
    a[0] = a_frag(0, tap_off(0), 0, 0);
    a[1] = a_frag(0, tap_off(0), 1 % RT, 1 / RT);
    // The SIMD arbiter serves the oldest wave first, so of the three workgroups sharing a CU
    // the first-dispatched one would finish ~30 % early and the last one would run its tail
    // alone on the SIMD, exposing every LDS latency.  Rotating the issue priority per stage
    // gives each resident workgroup the same share, so they finish together.
    const int prio_slot = blockIdx.x / 256;
    for (int s = s_begin; s < s_end; s++) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const int buf = (s - s_begin) & 1;
#ifndef ABL_NO_PRIO
        switch ((s + prio_slot) % 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(2); break;
        }
#endif
#ifndef ABL_NO_BLOAD
        if (s + 1 < s_end) load_b(s + 1);
#endif
        const bool next_chunk = (tap == 8) && (s + 1 < s_end);
        if (next_chunk) load_a(chunk + 1);

        const int toff = tap_off(tap), toff_next = tap_off(tap + 1);
        const float* Bb = Bs + buf * C::B_FLOATS;
        f32x4 b[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; kk++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
                b[kk][tn] = *reinterpret_cast<const f32x4*>(&Bb[swz(brow[tn], kk * 4 + lq)]);
#pragma unroll
        for (int step = 0; step < NSTEP; step++) {
            const int kk = step / RT, t = step % RT;
            const int pre = step + 2;                 // fragment to put in flight now
            if (pre < NSTEP) a[pre % 3] = a_frag(tap, toff, pre % RT, pre / RT);
            else if (tap != 8) a[pre % 3] = a_frag(tap + 1, toff_next, (pre - NSTEP) % RT, (pre - NSTEP) / RT);
            const f32x4 av = a[step % 3];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
                    acc[t][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[kk][tn][j], acc[t][tn], 0, 0, 0);
            // keep "read for step+2, then the 8 MFMAs of this step" in this order: left alone, hipcc
            // sinks every ds_read to just before its first use and exposes the LDS latency per step
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef ABL_NO_BLOAD
        if (s + 1 < s_end) store_b(buf ^ 1);       // buf^1 was last read in stage s-1
#endif
        if (next_chunk) {
            __syncthreads();                        // everyone is done with this chunk's A image
            store_a();
        }
#ifndef ABL_NO_BARRIER
        __syncthreads();
#endif
        // rotate the two prefetched fragments into slots 0 / 1 for the next stage
        if (tap != 8) {
            const f32x4 n0 = a[NSTEP % 3], n1 = a[(NSTEP + 1) % 3];
            a[0] = n0;
            a[1] = n1;
        } else if (s + 1 < s_end) {
            a[0] = a_frag(0, tap_off(0), 0, 0);
            a[1] = a_frag(0, tap_off(0), 1 % RT, 1 / RT);
        }
    }

#ifdef GOAI_CONV_STAMPS
    if (tid == 0) {
        g_conv_stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memrealtime();
        g_conv_stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime();
    }
#endif
    // epilogue.  C/D layout of a 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg.
    const bool full = m0 + C::BM <= M;
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int col = wave * 32 + tn * 16 + li;
        const float bv = SPLIT ? 0.0f : bias[col];
        if (full) {
            // batch all residual loads of this column strip, then all stores
            f32x4 rv[RT];
            if (RESID) {
#pragma unroll
                for (int t = 0; t < RT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) rv[t][r] = resid[((long)m0 + t * 16 + 4 * lq + r) * 128 + col];
            }
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[t][tn][r] + bv;
                    if (RESID) v += rv[t][r];
                    out[((long)m0 + t * 16 + 4 * lq + r) * 128 + col] = fmaxf(v, floor_v);
                }
        } else {
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const long row = (long)m0 + t * 16 + 4 * lq + r;
                    if (row < M) {
                        float v = acc[t][tn][r] + bv;
                        if (RESID) v += resid[row * 128 + col];
                        out[row * 128 + col] = fmaxf(v, floor_v);
                    }
                }
        }
    }
#ifdef GOAI_CONV_STAMPS
    if (tid == 0) {
        g_conv_stamps[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_conv_stamps[blockIdx.x * 8 + 3] = xcc;
    }
#endif
}

#ifdef GOAI_CONV_STAMPS
#endif
// Rows per workgroup = 16 * RT.  Pick RT in 5..8 minimising (rounds over the chip's workgroup
// slots) x RT, i.e. the time of the slowest CU; ties go to the taller tile (less halo).
static int pick_rt(int S, int M)
{
    const int cus = 256;
    int best_rt = 8;
    long best_cost = -1;
    for (int rt = 8; rt >= 5; rt--) {
        const long lds = ((long)(16 * rt + 2 * (S + 1) + 1) * LD + 2 * 128 * LD) * 4;
        long per_cu = (160 * 1024) / ((lds + 511) / 512 * 512);
        if (per_cu > 3) per_cu = 3;                       // 3 waves per SIMD by registers
        const long wgs = (M + 16 * rt - 1) / (16 * rt);
        const long rounds = (wgs + cus * per_cu - 1) / (cus * per_cu);
        // within the last round CUs hold ceil(remaining / cus) workgroups
        const long rem = wgs - (rounds - 1) * cus * per_cu;
        const long depth = (rounds - 1) * per_cu + (rem + cus - 1) / cus;
        const long cost = depth * rt;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rt = rt; }
    }
    return best_rt;
}

template <int S, int RT>
static void launch_rt(const float* in, const float* wt, const float* bias, const float* resid, float* out, int M,
                      float floor_v, hipStream_t st)
{
    const int grid = (M + 16 * RT - 1) / (16 * RT);
    if (resid) k_conv3x3<S, RT, true><<<dim3(grid), dim3(256), 0, st>>>(in, wt, bias, resid, out, M, floor_v);
    else k_conv3x3<S, RT, false><<<dim3(grid), dim3(256), 0, st>>>(in, wt, bias, resid, out, M, floor_v);
}

template <int S>
static void launch_s(int rt, const float* in, const float* wt, const float* bias, const float* resid, float* out, int M,
                     float floor_v, hipStream_t st)
{
    switch (rt) {
        case 5: launch_rt<S, 5>(in, wt, bias, resid, out, M, floor_v, st); break;
        case 6: launch_rt<S, 6>(in, wt, bias, resid, out, M, floor_v, st); break;
        case 7: launch_rt<S, 7>(in, wt, bias, resid, out, M, floor_v, st); break;
        default: launch_rt<S, 8>(in, wt, bias, resid, out, M, floor_v, st); break;
    }
}

// out[i] = p0[i] + p1[i] + p2[i] + p3[i] + bias[col] (+ resid[i]), four floats per thread
__global__ __launch_bounds__(256) void k_conv3x3_sum4(const float* __restrict__ partial, const float* __restrict__ bias,
                                                      const float* __restrict__ resid, float* __restrict__ out, long n4, long stride4)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(partial);
    f32x4 v = ((p[i] + p[i + stride4]) + p[i + 2 * stride4]) + p[i + 3 * stride4];
    v += *reinterpret_cast<const f32x4*>(&bias[(i & 31) * 4]);
    if (resid) v += reinterpret_cast<const f32x4*>(resid)[i];
    reinterpret_cast<f32x4*>(out)[i] = v;
}

#ifdef GOAI_CONV_STAMPS
extern "C" int goai_debug_conv_stamps(unsigned long long* host_out, int n)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif

static int launch_conv3x3_floor(int S_rt, const float* in, const float* wt, const float* bias, const float* resid, float* out,
                                int M, float floor_v, hipStream_t st)
{
    static const int forced = [] { const char* e = getenv("GOAI_CONV_RT"); return e ? atoi(e) : 0; }();   // tuning aid
    const int rt = (forced >= 5 && forced <= 8) ? forced : pick_rt(S_rt, M);
    GOAI_DISPATCH_SIZE(S_rt, launch_s<S>(rt, in, wt, bias, resid, out, M, floor_v, st));
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

int launch_conv3x3(int S_rt, const float* in, const float* wt, const float* bias, const float* resid, float* out,
                   int M, hipStream_t st)
{
    return launch_conv3x3_floor(S_rt, in, wt, bias, resid, out, M, 0.0f, st);
}

// The same implicit GEMM without the ReLU: out = conv(in) + bias (+ resid).  The training path uses it for the
// convolutions in front of a train-mode BatchNorm and, with transposed / flipped weights, for the data gradient.
int launch_conv3x3_linear(int S_rt, const float* in, const float* wt, const float* bias, const float* resid, float* out,
                          int M, hipStream_t st)
{
    return launch_conv3x3_floor(S_rt, in, wt, bias, resid, out, M, -INFINITY, st);
}

// The same convolution (no ReLU) for batches of few rows: every workgroup takes one 32-channel chunk of the input
// (4 x the workgroups, a quarter of the stages each) into `partial` [4][M][128]; the chunks are summed in a fixed order.
// Rounds differently from launch_conv3x3_linear (another summation order): a caller uses one or the other per shape.
int launch_conv3x3_linear_split(int S_rt, const float* in, const float* wt, const float* bias, const float* resid, float* out,
                                float* partial, int M, hipStream_t st)
{
    constexpr int RT = 5;
    const int grid = (M + 16 * RT - 1) / (16 * RT);
    GOAI_DISPATCH_SIZE(S_rt, (k_conv3x3<S, RT, false, true><<<dim3(grid, 4), dim3(256), 0, st>>>(in, wt, nullptr, nullptr, partial, M, -INFINITY)));
    const long n4 = (long)M * 32;
    k_conv3x3_sum4<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st>>>(partial, bias, resid, out, n4, n4);
    GOAI_HIP_CHECK(hipGetLastError());
    return GOAI_OK;
}

}  // namespace goai
