// nn.h -- device-side weight images of the policy/value CNN and its launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace goai {

struct HeadWeights {
    const float* conv_w;       // [3][128]: act ch0, act ch1, crit (BN folded); value net uses slot 2 only
    const float* conv_b;       // [3]
    const float* act_lin_wt;   // [2P][A]  (transposed Linear weight), AC only
    const float* act_lin_b;    // [A]
    const float* crit_w1t;     // [P][256]
    const float* crit_b1;      // [256]
    const float* crit_w2;      // [256]
    const float* crit_b2;      // [1]
};

struct Net {
    int kind = -1;             // 0 actor-critic, 1 value
    int layers = 0;            // residual blocks (reference: 3)
    int max_batch = 0;
    bool loaded = false;
    float shift = 0.0f;        // input shift: 0.5 for the AC net (ac_net.py:41), 0 for the value net
    float* w0 = nullptr;       // [54][128]   conv0, k = ci*9 + tap
    float* b0 = nullptr;       // [128]
    void* w0_split = nullptr;  // conv0 on the bf16 matrix cores: [3 bf16 parts of w][3 k-steps][128 out][4 taps x 8 ch] (null: VALU kernel)
    int dtype = 0;             // GOAI_DTYPE_F32 / GOAI_DTYPE_BF16X3: arithmetic of the 3x3 trunk convolutions
    float* wt = nullptr;       // f32:  [2*layers][4 chunks][9 taps][128 out][32 in]
    float* wt_wino = nullptr;  // f32 Winograd F(2x2,3x3): [2*layers][4 chunks][16 xi][128 out][32 in] = G g Gt (null: direct only)
    int wino_mode = 0;         // 0: direct kernel, 2: Winograd F(2x2,3x3), 3: Winograd F(3x3,3x3) (chosen in nn_init)
    bool wino3_paired = false; // F(3x3,3x3) with two chunks per output transform (conv_wino3p_kernels.hip): max_batch <= 4 * CUs
    void* wino3p_tab = nullptr;
    bool trunk_fused = false;  // F(3x3,3x3) and the split F(2x2,3x3): all 2 * layers convolutions of a forward in one launch
    float* wt_wino3 = nullptr; // F(3x3,3x3): [2*layers][4 chunks][25 xi][128 out][32 in]
    void* wino3_tab[2] = { nullptr, nullptr };   // input-transform address tables for 4 / 5 boards per workgroup
    int n_cu = 256;
    bool split6 = false;       // GOAI_DTYPE_BF16X6: F(3x3,3x3) with its products on the bf16 matrix cores, operands split in three
    void* wt_split6 = nullptr; // [2*layers][stages][8 waves][3 parts][64 lanes][8 bf16]: 100 stages for F(3x3,3x3) on 9x9
                               // (conv_wino3s_kernels.hip), 64 for F(2x2,3x3) on the other boards (conv_wino2s_kernels.hip)
    bool wino3h = false;       // fp16 parts on 9x9: conv_wino3h_kernels.hip (GOAI_WINO3S_OLD=1: round 2's conv_wino3s_kernels.hip)
    int parts = 3;             // with split6: 3 bf16 parts per operand, or 2 fp16 parts (GOAI_DTYPE_F16X3: same kernels, same image
                               // with 2 parts, operands scaled by powers of two into fp16's range)
    int* wexp = nullptr;       // parts == 2: [2*layers] scale exponents of the weight images (device; wexp_host: the same)
    int wexp_host[128] = {};
    float* amax = nullptr;     // parts == 2: [max_batch] upper bound of each board's activations between two launches
    float amax0 = 0.0f;        // parts == 2: upper bound of conv0's output, from its weights (the input planes are 0 / 1)
    void* wt_bf16 = nullptr;   // bf16x3: [2*layers][4 chunks][9 taps][128 out][32 hi | 32 lo]
    float* bt = nullptr;       // [2*layers][128]
    float* head_blob = nullptr;
    HeadWeights hw{};
    float* act[3] = { nullptr, nullptr, nullptr };   // activation buffers [max_batch*P][128]
    float* hbuf = nullptr;     // heads_in_trunk: [max_batch][3][P], the heads' 1x1 convolutions as left by the trunk launch
    bool heads_in_trunk = false;   // split F(3x3,3x3) trunk in one launch: its last epilogue forms the heads' 1x1 convolutions
};

// GOAI_DTYPE_F16X3 (conv_wino3s_kernels.hip): exponent of the power of two that scales a board's input activations, all
// in [0, amax] (amax = the fp32 with bits `amax_bits`), so that Winograd-transformed values, at most `growth` x amax
// (36 for F(3x3,3x3), 4 for F(2x2,3x3)), stay below fp16's largest number: amax < 2^(E+1) -> 36 * 2^(E+1) * 2^(9-E) = 36864,
// 4 * 2^(E+1) * 2^(12-E) = 32768.
__device__ __forceinline__ int f16_scale_exp(unsigned amax_bits, int growth)
{
    int e = (int)(amax_bits >> 23);                 // biased exponent: E = e - 127
    e = e < 16 ? 16 : (e > 250 ? 250 : e);          // the scale stays a normal number
    return (growth == 36 ? 136 : 139) - e;
}

int launch_conv0(int S, const uint8_t* states, const float* w0, const void* w0_split, const float* b0, float shift,
                 float* out, int nboards, hipStream_t st);
int launch_conv3x3(int S, const float* in, const float* wt, const float* bias, const float* resid, float* out, int M,
                   hipStream_t st);
int launch_conv3x3_linear(int S, const float* in, const float* wt, const float* bias, const float* resid, float* out, int M,
                          hipStream_t st);
int launch_conv3x3_linear_split(int S, const float* in, const float* wt, const float* bias, const float* resid, float* out,
                                float* partial, int M, hipStream_t st);
int launch_conv3x3_bf16(int S, const float* in, const void* wt_bf16, const float* bias, const float* resid, float* out,
                        int M, hipStream_t st);
int wino_boards_per_wg(int S);
int wino3_supported(int S);
int wino3_boards_per_wg(int S);
size_t wino3_build_table(int S, int nbw, uint16_t* out);
// n_convs = 1: y = relu(conv(x) + bias (+ resid0)); n_convs = 2 * blocks, resid0 = NULL: the whole
// residual trunk in one launch (weights / biases of consecutive layers contiguous), result in x
int launch_conv3x3_wino3(int S, float* x, float* y, const float* resid0, const float* wt_wino3, const float* bias,
                         const void* const* tabs, int nboards, int n_cu, int n_convs, hipStream_t st);
int wino3p_supported(int S, int max_batch, int n_cu);
size_t wino3p_build_table(int S, uint16_t* out);
int launch_conv3x3_wino3p(int S, float* x, float* y, const float* resid0, const float* wt_wino3, const float* bias,
                          const void* tab, int nboards, int n_cu, int n_convs, hipStream_t st);
size_t wino3s_build_table(int S, uint16_t* out);
int launch_conv3x3_wino3s(int S, float* x, float* y, const float* resid0, const void* wt_split, const float* bias,
                          const void* tab, int nboards, int n_convs, hipStream_t st, int parts = 3, const int* wexp = nullptr,
                          float* amax_io = nullptr, float amax_first = 0.0f, const float* head_w = nullptr,
                          const float* head_b = nullptr, float* head_out = nullptr);
// round 3's form of the fp16 F(3x3,3x3) trunk (conv_wino3h_kernels.hip): parts = 2 (f16x3) or 1 (f16)
int launch_conv3x3_wino3h(int S, float* x, float* y, const float* resid0, const void* wt_split, const float* bias, int nboards,
                          int n_convs, hipStream_t st, int parts, const int* wexp, float* amax_io, float amax_first,
                          const float* head_w = nullptr, const float* head_b = nullptr, float* head_out = nullptr);
// the split F(2x2,3x3) convolution of the other board sizes (conv_wino2s_kernels.hip); same launch contract as
// launch_conv3x3_wino3h: one convolution, or the whole residual trunk (n_convs = 2 * blocks) in one launch
int launch_conv3x3_wino2s(int S, float* x, float* y, const float* resid0, const void* wt_split, const float* bias, int nboards,
                          int n_convs, hipStream_t st, int parts, const int* wexp, float* amax_io, float amax_first,
                          const float* head_w = nullptr, const float* head_b = nullptr, float* head_out = nullptr);
int launch_conv3x3_wino(int S, const float* in, const float* wt_wino, const float* bias, const float* resid, float* out,
                        int nboards, hipStream_t st);
// act_is_h: `act` is not the trunk's output [board][P][128] but the heads' 1x1 convolutions of it, [board][3][P]
// (formed by the trunk launch itself: launch_conv3x3_wino3s with head_out)
int launch_heads(int S, const float* act, const HeadWeights& hw, int is_ac, float* pi, float* val, int nboards,
                 hipStream_t st, bool act_is_h = false);
int launch_numpy_shim(int S, const uint8_t* states, float* pi, float* val, int n, hipStream_t st);

}  // namespace goai
