// nn_host.hip -- host side of the CNN: buffer set-up, state_dict ingestion with BatchNorm
// folding and weight re-layout, and the per-batch launch sequence.
//
// Weight names are the reference's own state_dict keys:
//   main.0 / main.1                      conv0 + BN      (/root/reference/go_ai/models/__init__.py:22-26)
//   main.{3+i}.main.{0,1,3,4}            BasicBlock i    (/root/reference/go_ai/models/__init__.py:234-251)
//   act_head.{0,1,4}, crit_head.{0,1,4,6}  AC heads      (/root/reference/go_ai/models/ac_net.py:15-31)
//   convs.{0,1,4,6}                       value head     (/root/reference/go_ai/models/val_net.py:12-20)
// game_head.* and *.num_batches_tracked are ignored.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.h"
#include "goai_internal.h"

namespace goai {

static int dev_alloc(goai_ctx* c, void** p, size_t bytes)
{
    GOAI_HIP_CHECK(hipMalloc(p, bytes));
    c->allocs.push_back(*p);
    return GOAI_OK;
}

constexpr size_t WT_BF16_PER_CONV = (size_t)2 * 9 * 128 * 128;   // hi and lo halves
// split Winograd images, 16-bit elements per convolution: [stages][8 waves][parts][64 lanes][8]; 100 stages for
// F(3x3,3x3) (4 chunks x 5 x 5), 64 for F(2x2,3x3) (4 chunks x 4 x 4); parts = 3 (bf16x6) or 2 (f16x3)
static size_t wt_split_per_conv(const Net& n) { return (size_t)(n.wino_mode == 3 ? 100 : 64) * 8 * n.parts * 64 * 8; }

// float -> bfloat16 bits, round to nearest even (what v_cvt_pk_bf16_f32 does on the device)
static uint16_t bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

static int nn_init_impl(goai_ctx* c, int kind, int layers, int max_batch, int dtype);

// A failed init leaves the context as it was (its partial allocations are freed), so it can be retried.
int nn_init(goai_ctx* c, int kind, int layers, int max_batch, int dtype)
{
    GOAI_REQUIRE(kind == 0 || kind == 1, "net_kind must be 0 (actor-critic) or 1 (value)");
    GOAI_REQUIRE(layers >= 0 && layers <= 64, "n_blocks out of range");
    GOAI_REQUIRE(max_batch > 0, "max_batch must be positive");
    GOAI_REQUIRE(c->net.kind < 0, "goai_nn_init was already called on this context");
    const size_t mark = c->allocs.size(), host_mark = c->host_allocs.size();
    const int rc = nn_init_impl(c, kind, layers, max_batch, dtype);
    if (rc) {
        c->release_from(mark, host_mark);
        c->net = Net();
    }
    return rc;
}

static int nn_init_impl(goai_ctx* c, int kind, int layers, int max_batch, int dtype)
{
    Net& n = c->net;
    n.kind = kind;
    n.layers = layers;
    n.max_batch = max_batch;
    n.dtype = dtype;
    n.shift = kind == 0 ? 0.5f : 0.0f;
    int rc;
    if ((rc = dev_alloc(c, (void**)&n.w0, 54 * 128 * sizeof(float)))) return rc;
    if ((rc = dev_alloc(c, (void**)&n.b0, 128 * sizeof(float)))) return rc;
    {
        const char* c0 = getenv("GOAI_CONV0");            // "valu": the fp32 VALU kernel (A/B timing)
        if (!(c0 && std::string(c0) == "valu") && (rc = dev_alloc(c, &n.w0_split, 3 * 3 * 128 * 32 * sizeof(uint16_t))))
            return rc;
    }
    if (layers > 0) {
        // fp32 trunk: Winograd F(2x2,3x3) (conv_wino_kernels.hip) or the direct implicit GEMM
        // (conv_kernels.hip).  Winograd does 1.8-2x less matrix work from 9x9 up but runs as whole
        // workgroups of ~100 output tiles, one per CU, so it only pays when those fill the chip.
        // The choice is made HERE, once, from max_batch, not per launch: a position's
        // network output must not depend on how many other positions share its batch (the two
        // kernels round differently).  GOAI_CONV_ALGO=direct / wino forces one of them.
        const char* algo = getenv("GOAI_CONV_ALGO");
        n.wino_mode = 0;
        {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                prop.multiProcessorCount > 0)
                n.n_cu = prop.multiProcessorCount;
        }
        if (dtype == GOAI_DTYPE_BF16X6 || dtype == GOAI_DTYPE_F16X3 || dtype == GOAI_DTYPE_F16) {
            // F(3x3,3x3) where it tiles the board exactly (9x9), F(2x2,3x3) elsewhere
            n.wino_mode = wino3_supported(c->S) ? 3 : 2;
            // GOAI_DTYPE_F16 (one fp16 part, the 1e-2 tier): F(3x3,3x3)'s transforms (factors up to 4) amplify the operands'
            // 2^-11 rounding to 1.1e-2 .. 1.4e-2 on the 3-block nets, F(2x2,3x3)'s to 2e-3 .. 6e-3: the tier runs F(2x2,3x3) on
            // every board size; GOAI_F16_F3=1 keeps F(3x3,3x3) on 9x9 (faster, outside the tolerance: measurements only)
            if (dtype == GOAI_DTYPE_F16 && !getenv("GOAI_F16_F3")) n.wino_mode = 2;
            n.split6 = true;
            n.parts = dtype == GOAI_DTYPE_F16 ? 1 : dtype == GOAI_DTYPE_F16X3 ? 2 : 3;
            // fp16 parts on 9x9: round 3's kernel (GOAI_WINO3S_OLD=1 keeps round 2's for the two-part form: A / B timing)
            n.wino3h = n.wino_mode == 3 && (n.parts == 1 || (n.parts == 2 && !getenv("GOAI_WINO3S_OLD")));
        }
        if (dtype == GOAI_DTYPE_F32) {
            if (algo && std::string(algo) == "wino") n.wino_mode = 2;
            else if (algo && std::string(algo) == "wino3" && wino3_supported(c->S)) n.wino_mode = 3;
            else if (!(algo && std::string(algo) == "direct") && c->S >= 9) {
                // one workgroup per CU and round: Winograd when the last round is at least 3/4 full
                const int nbw = wino_boards_per_wg(c->S);
                const int grid = (max_batch + nbw - 1) / nbw, rounds = (grid + n.n_cu - 1) / n.n_cu;
                if (4 * grid >= 3 * rounds * n.n_cu) n.wino_mode = 2;
                // 9x9: F(3x3,3x3) tiles the board exactly (1.8x fewer products again); same one-workgroup-
                // per-CU shape, 4 boards each
                if (wino3_supported(c->S)) {
                    const int nb3 = wino3_boards_per_wg(c->S);
                    const int g3 = (max_batch + nb3 - 1) / nb3, r3 = (g3 + n.n_cu - 1) / n.n_cu;
                    if (4 * g3 >= 3 * r3 * n.n_cu) n.wino_mode = 3;
                }
            }
        }
        if (n.wino_mode == 2 && !n.split6 && (rc = dev_alloc(c, (void**)&n.wt_wino, (size_t)2 * layers * 16 * 128 * 128 * sizeof(float))))
            return rc;
        // a workgroup owns whole boards in the F(3x3,3x3) kernels and in the split F(2x2,3x3) kernel: the trunk is one launch
        n.trunk_fused = (n.wino_mode == 3 || (n.wino_mode == 2 && n.split6)) && !getenv("GOAI_NO_TRUNK_FUSION");
        {
            const char* e = getenv("GOAI_WINO3_PAIRED");
            n.wino3_paired = n.wino_mode == 3 && !n.split6 && wino3p_supported(c->S, max_batch, n.n_cu) && !(e && e[0] == '0');
        }
        if (n.wino3_paired || (n.split6 && n.wino_mode == 3)) {
            std::vector<uint16_t> tab(wino3p_build_table(c->S, nullptr));
            if (n.split6) wino3s_build_table(c->S, tab.data());
            else wino3p_build_table(c->S, tab.data());
            if ((rc = dev_alloc(c, &n.wino3p_tab, tab.size() * sizeof(uint16_t)))) return rc;
            GOAI_HIP_CHECK(hipMemcpy(n.wino3p_tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        if (n.split6) {
            if ((rc = dev_alloc(c, &n.wt_split6, (size_t)2 * layers * wt_split_per_conv(n) * sizeof(uint16_t)))) return rc;
            if (n.parts <= 2) {
                if ((rc = dev_alloc(c, (void**)&n.wexp, (size_t)2 * layers * sizeof(int)))) return rc;
                if ((rc = dev_alloc(c, (void**)&n.amax, (size_t)max_batch * sizeof(float)))) return rc;
            }
        } else if (n.wino_mode == 3) {
            if ((rc = dev_alloc(c, (void**)&n.wt_wino3, (size_t)2 * layers * 25 * 128 * 128 * sizeof(float)))) return rc;
            for (int i = 0; i < 2; i++) {
                std::vector<uint16_t> tab(wino3_build_table(c->S, 4 + i, nullptr));
                wino3_build_table(c->S, 4 + i, tab.data());
                if ((rc = dev_alloc(c, &n.wino3_tab[i], tab.size() * sizeof(uint16_t)))) return rc;
                GOAI_HIP_CHECK(hipMemcpy(n.wino3_tab[i], tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            }
        }
        if (n.split6) {
            // no direct-convolution image: this context only ever runs the split F(3x3,3x3) kernel
        } else if (dtype == GOAI_DTYPE_BF16X3) {
            if ((rc = dev_alloc(c, &n.wt_bf16, (size_t)2 * layers * WT_BF16_PER_CONV * sizeof(uint16_t)))) return rc;
        } else if ((rc = dev_alloc(c, (void**)&n.wt, (size_t)2 * layers * 9 * 128 * 128 * sizeof(float)))) return rc;
        if ((rc = dev_alloc(c, (void**)&n.bt, (size_t)2 * layers * 128 * sizeof(float)))) return rc;
    }
    const size_t P = c->P, A = c->A;
    const size_t head_floats = 3 * 128 + 4 + 2 * P * A + A + P * 256 + 256 + 256 + 4;
    if ((rc = dev_alloc(c, (void**)&n.head_blob, head_floats * sizeof(float)))) return rc;
    for (int i = 0; i < 3; i++)
        if ((rc = dev_alloc(c, (void**)&n.act[i], (size_t)max_batch * P * 128 * sizeof(float)))) return rc;
    // The split F(3x3,3x3) trunk in one launch ends with the heads' three 1x1 convolutions (conv_wino3s_kernels.hip): its
    // last convolution's 128 channels never go to memory.  GOAI_NO_HEAD_FUSION=1: the heads do them, as everywhere else.
    n.heads_in_trunk = layers > 0 && n.split6 && (n.wino_mode == 2 || (n.wino_mode == 3 && n.trunk_fused)) &&
                       !getenv("GOAI_NO_HEAD_FUSION");
    if (n.heads_in_trunk && (rc = dev_alloc(c, (void**)&n.hbuf, (size_t)max_batch * 3 * P * sizeof(float)))) return rc;
    return GOAI_OK;
}

namespace {
struct Tensor { const float* p; int64_t n; };
typedef std::map<std::string, Tensor> TensorMap;

bool get(const TensorMap& m, const std::string& k, int64_t numel, const float*& out)
{
    auto it = m.find(k);
    if (it == m.end()) { set_error("state_dict is missing " + k); return false; }
    if (it->second.n != numel) {
        set_error("state_dict tensor " + k + " has " + std::to_string(it->second.n) + " elements, expected " +
                  std::to_string(numel));
        return false;
    }
    out = it->second.p;
    return true;
}

// scale[c] = gamma / sqrt(var + eps), shift[c] = (conv_bias - mean) * scale + beta
bool bn_fold(const TensorMap& m, const std::string& conv, const std::string& bn, int C, std::vector<double>& scale,
             std::vector<double>& shift)
{
    const float *cb, *g, *b, *mu, *var;
    if (!get(m, conv + ".bias", C, cb) || !get(m, bn + ".weight", C, g) || !get(m, bn + ".bias", C, b) ||
        !get(m, bn + ".running_mean", C, mu) || !get(m, bn + ".running_var", C, var))
        return false;
    scale.resize(C);
    shift.resize(C);
    for (int c = 0; c < C; c++) {
        scale[c] = (double)g[c] / std::sqrt((double)var[c] + 1e-5);
        shift[c] = ((double)cb[c] - (double)mu[c]) * scale[c] + (double)b[c];
    }
    return true;
}
}  // namespace

int nn_load_weights(goai_ctx* c, const char* const* names, const float* const* tensors, const int64_t* numels, int n)
{
    Net& net = c->net;
    GOAI_REQUIRE(net.kind >= 0, "call goai_nn_init first");
    GOAI_REQUIRE(names && tensors && numels && n > 0, "NULL / empty weight table");
    TensorMap m;
    for (int i = 0; i < n; i++) {
        GOAI_REQUIRE(names[i] && tensors[i], "NULL entry in weight table");
        m[names[i]] = Tensor{ tensors[i], numels[i] };
    }
    const int P = c->P, A = c->A;
    std::vector<double> sc, sh;
    const float* w;

    // conv0 -> [k = ci*9 + tap][cout]
    std::vector<float> w0(54 * 128), b0(128);
    if (!get(m, "main.0.weight", 128 * 6 * 9, w) || !bn_fold(m, "main.0", "main.1", 128, sc, sh)) return GOAI_ERR_ARG;
    for (int co = 0; co < 128; co++) {
        for (int k = 0; k < 54; k++) w0[k * 128 + co] = (float)((double)w[co * 54 + k] * sc[co]);
        b0[co] = (float)sh[co];
    }
    GOAI_HIP_CHECK(hipMemcpy(net.w0, w0.data(), w0.size() * sizeof(float), hipMemcpyHostToDevice));
    GOAI_HIP_CHECK(hipMemcpy(net.b0, b0.data(), b0.size() * sizeof(float), hipMemcpyHostToDevice));
    {
        // no output of conv0 + ReLU exceeds this: its inputs are the six 0 / 1 planes (minus the AC net's shift)
        double bound = 0.0;
        for (int co = 0; co < 128; co++) {
            double hi = b0[co];
            for (int k = 0; k < 54; k++) {
                const double wk = w0[k * 128 + co];
                hi += std::fmax(wk * (1.0 - net.shift), -wk * net.shift);
            }
            bound = std::fmax(bound, hi);
        }
        net.amax0 = (float)(bound * 1.001) + 1e-30f;
    }
    if (net.w0_split) {
        // every folded fp32 weight as the exact sum of three bf16 numbers; k = tap * 8 + channel
        std::vector<uint16_t> ws((size_t)3 * 3 * 128 * 32, 0);
        auto bf_to_f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
        for (int co = 0; co < 128; co++)
            for (int ci = 0; ci < 6; ci++)
                for (int tap = 0; tap < 9; tap++) {
                    float r = w0[(ci * 9 + tap) * 128 + co];
                    for (int sp = 0; sp < 3; sp++) {
                        const uint16_t h = bf16_rne(r);
                        ws[(((size_t)sp * 3 + tap / 4) * 128 + co) * 32 + (tap % 4) * 8 + ci] = h;
                        r -= bf_to_f(h);
                    }
                    // r == 0 here unless a part fell into the bf16 denormal range (|w| < 2^-110)
                }
        GOAI_HIP_CHECK(hipMemcpy(net.w0_split, ws.data(), ws.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }

    // trunk convs -> [conv][chunk][tap][cout][32 cin]
    std::vector<float> wt((size_t)9 * 128 * 128), bt(128);
    std::vector<uint16_t> wb(WT_BF16_PER_CONV);
    for (int l = 0; l < 2 * net.layers; l++) {
        const std::string base = "main." + std::to_string(3 + l / 2) + ".main.";
        const std::string conv = base + (l % 2 ? "3" : "0"), bn = base + (l % 2 ? "4" : "1");
        if (!get(m, conv + ".weight", 128 * 128 * 9, w) || !bn_fold(m, conv, bn, 128, sc, sh)) return GOAI_ERR_ARG;
        for (int co = 0; co < 128; co++) bt[co] = (float)sh[co];
        GOAI_HIP_CHECK(hipMemcpy(net.bt + (size_t)l * 128, bt.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
        if (net.dtype == GOAI_DTYPE_BF16X3) {
            // [4 chunks][9 taps][128 out][32 hi | 32 lo]: BN scale folded in fp64, rounded to fp32 as in the
            // fp32 path, then split w = hi + lo with hi = bf16(w), lo = bf16(w - hi)
            for (int chunk = 0; chunk < 4; chunk++)
                for (int tap = 0; tap < 9; tap++)
                    for (int co = 0; co < 128; co++)
                        for (int k = 0; k < 32; k++) {
                            const int ci = chunk * 32 + k;
                            const float wf = (float)((double)w[((size_t)co * 128 + ci) * 9 + tap] * sc[co]);
                            const uint16_t hi = bf16_rne(wf);
                            uint32_t hu = (uint32_t)hi << 16;
                            float hf;
                            memcpy(&hf, &hu, 4);
                            const size_t base = (((size_t)chunk * 9 + tap) * 128 + co) * 64;
                            wb[base + k] = hi;
                            wb[base + 32 + k] = bf16_rne(wf - hf);
                        }
            GOAI_HIP_CHECK(hipMemcpy((uint16_t*)net.wt_bf16 + (size_t)l * WT_BF16_PER_CONV, wb.data(),
                                     wb.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            continue;
        }
        if (net.split6) {
            // U = G g Gt (4x4 for F(2x2,3x3), 5x5 for F(3x3,3x3)), BN scale folded in fp64 and rounded to fp32 as for the
            // fp32 kernels, then cut into parts; [stage = (chunk * T + p) * T + q][wave][part][lane (li, lq)][j]:
            // output channel 16 wave + li, input channel 32 chunk + 8 lq + j -- the lanes' MFMA A operands.
            //   parts = 3: bf16 -- top 16 bits, top 16 bits of the rest, the rest: exact;
            //   parts = 2: fp16 -- U scaled by 2^wexp so that the largest |U| of the convolution lies in [2^14, 2^15),
            //              h = fp16(U), l = fp16(U - h), round to nearest.
            static const double G2[4][3] = { { 1, 0, 0 }, { 0.5, 0.5, 0.5 }, { 0.5, -0.5, 0.5 }, { 0, 0, 1 } };
            static const double G3[5][3] = { { 0.5, 0, 0 }, { -0.5, -0.5, -0.5 }, { -1.0 / 6, 1.0 / 6, -1.0 / 6 },
                                             { 1.0 / 6, 1.0 / 3, 2.0 / 3 }, { 0, 0, 1 } };
            const int T = net.wino_mode == 3 ? 5 : 4, parts = net.parts;
            const size_t per_conv = wt_split_per_conv(net);
            std::vector<float> uf((size_t)T * T * 128 * 128);                 // [p][q][co][ci]
            float umax = 0.0f;
            for (int co = 0; co < 128; co++)
                for (int ci = 0; ci < 128; ci++) {
                    const float* g = &w[((size_t)co * 128 + ci) * 9];
                    for (int pp = 0; pp < T; pp++)
                        for (int qq = 0; qq < T; qq++) {
                            double u = 0.0;
                            for (int ky = 0; ky < 3; ky++)
                                for (int kx = 0; kx < 3; kx++)
                                    u += (T == 5 ? G3[pp][ky] * G3[qq][kx] : G2[pp][ky] * G2[qq][kx]) * (double)g[ky * 3 + kx];
                            const float r = (float)(u * sc[co]);
                            uf[(((size_t)pp * T + qq) * 128 + co) * 128 + ci] = r;
                            umax = std::fmax(umax, std::fabs(r));
                        }
                }
            int wexp = 0;
            if (parts <= 2) {
                GOAI_REQUIRE(std::isfinite(umax), "non-finite convolution weights");
                int ex = 0;
                if (umax > 0.0f) std::frexp(umax, &ex);                        // umax in [2^(ex-1), 2^ex)
                wexp = 15 - ex;
                wexp = wexp > 100 ? 100 : wexp;
                net.wexp_host[l] = wexp;
                GOAI_HIP_CHECK(hipMemcpy(net.wexp + l, &wexp, sizeof(int), hipMemcpyHostToDevice));
            }
            std::vector<uint16_t> ws(per_conv);
            for (int pp = 0; pp < T; pp++)
                for (int qq = 0; qq < T; qq++)
                    for (int co = 0; co < 128; co++)
                        for (int ci = 0; ci < 128; ci++) {
                            float r = uf[(((size_t)pp * T + qq) * 128 + co) * 128 + ci];
                            const int chunk = ci / 32, k = ci % 32;
                            const size_t stage = (size_t)(chunk * T + pp) * T + qq;
                            const size_t lane = (size_t)(k / 8) * 16 + co % 16;
                            uint16_t* dst = &ws[(((stage * 8 + co / 16) * parts) * 64 + lane) * 8 + k % 8];
                            if (parts <= 2) {
                                r = std::ldexp(r, wexp);
                                const _Float16 h = (_Float16)r, lo = (_Float16)(r - (float)h);
                                memcpy(dst, &h, 2);
                                if (parts == 2) memcpy(dst + 64 * 8, &lo, 2);
                            } else
                                for (int part = 0; part < 3; part++) {
                                    uint32_t bits;
                                    memcpy(&bits, &r, 4);
                                    bits &= 0xffff0000u;
                                    float hf;
                                    memcpy(&hf, &bits, 4);
                                    dst[(size_t)part * 64 * 8] = (uint16_t)(bits >> 16);
                                    r -= hf;                      // exact
                                }
                        }
            GOAI_HIP_CHECK(hipMemcpy((uint16_t*)net.wt_split6 + (size_t)l * per_conv, ws.data(), ws.size() * sizeof(uint16_t),
                                     hipMemcpyHostToDevice));
            continue;
        }
        if (net.wino_mode == 3) {
            // F(3x3,3x3): U = G g Gt (5x5) per (cout, cin) in fp64, BN scale folded
            static const double G3[5][3] = { { 0.5, 0, 0 }, { -0.5, -0.5, -0.5 }, { -1.0 / 6, 1.0 / 6, -1.0 / 6 },
                                             { 1.0 / 6, 1.0 / 3, 2.0 / 3 }, { 0, 0, 1 } };
            std::vector<float> wu((size_t)25 * 128 * 128);
            for (int co = 0; co < 128; co++)
                for (int ci = 0; ci < 128; ci++) {
                    const float* g = &w[((size_t)co * 128 + ci) * 9];
                    for (int pp = 0; pp < 5; pp++)
                        for (int qq = 0; qq < 5; qq++) {
                            double u = 0.0;
                            for (int ky = 0; ky < 3; ky++)
                                for (int kx = 0; kx < 3; kx++) u += G3[pp][ky] * G3[qq][kx] * (double)g[ky * 3 + kx];
                            // [stage][wave = co / 16][kk][lane = 16 lq + li][jj], k = 16 kk + 4 lq + jj: exactly the MFMA
                            // operand registers of the kernels' lanes, so a wave's weight load is 1 KB of whole lines
                            const int chunk = ci / 32, k = ci % 32;
                            const size_t lane = (size_t)((k % 16) / 4) * 16 + co % 16;
                            wu[((size_t)chunk * 25 + pp * 5 + qq) * 4096 + (size_t)(co / 16) * 512 + (k / 16) * 256 + lane * 4 + k % 4] =
                                (float)(u * sc[co]);
                        }
                }
            GOAI_HIP_CHECK(hipMemcpy(net.wt_wino3 + (size_t)l * 25 * 128 * 128, wu.data(), wu.size() * sizeof(float),
                                     hipMemcpyHostToDevice));
        }
        if (net.wino_mode == 2) {
            // U = G g Gt per (cout, cin), G = (1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1), in fp64, BN scale folded
            static const double Gm[4][3] = { { 1, 0, 0 }, { 0.5, 0.5, 0.5 }, { 0.5, -0.5, 0.5 }, { 0, 0, 1 } };
            std::vector<float> wu((size_t)16 * 128 * 128);
            for (int co = 0; co < 128; co++)
                for (int ci = 0; ci < 128; ci++) {
                    const float* g = &w[((size_t)co * 128 + ci) * 9];
                    for (int pp = 0; pp < 4; pp++)
                        for (int qq = 0; qq < 4; qq++) {
                            double u = 0.0;
                            for (int ky = 0; ky < 3; ky++)
                                for (int kx = 0; kx < 3; kx++) u += Gm[pp][ky] * Gm[qq][kx] * (double)g[ky * 3 + kx];
                            const int chunk = ci / 32, k = ci % 32;
                            // the lanes' own order, as for F(3x3,3x3) above
                            const size_t lane = (size_t)((k % 16) / 4) * 16 + co % 16;
                            wu[((size_t)chunk * 16 + pp * 4 + qq) * 4096 + (size_t)(co / 16) * 512 + (k / 16) * 256 + lane * 4 + k % 4] =
                                (float)(u * sc[co]);
                        }
                }
            GOAI_HIP_CHECK(hipMemcpy(net.wt_wino + (size_t)l * 16 * 128 * 128, wu.data(), wu.size() * sizeof(float),
                                     hipMemcpyHostToDevice));
        }
        for (int chunk = 0; chunk < 4; chunk++)
            for (int tap = 0; tap < 9; tap++)
                for (int co = 0; co < 128; co++)
                    for (int k = 0; k < 32; k++) {
                        int ci = chunk * 32 + k;
                        wt[(((size_t)chunk * 9 + tap) * 128 + co) * 32 + k] =
                            (float)((double)w[((size_t)co * 128 + ci) * 9 + tap] * sc[co]);
                    }
        GOAI_HIP_CHECK(hipMemcpy(net.wt + (size_t)l * 9 * 128 * 128, wt.data(), wt.size() * sizeof(float),
                                 hipMemcpyHostToDevice));
    }

    // heads
    const size_t o_conv_w = 0, o_conv_b = 3 * 128, o_alw = o_conv_b + 4, o_alb = o_alw + (size_t)2 * P * A,
                 o_w1 = o_alb + A, o_b1 = o_w1 + (size_t)P * 256, o_w2 = o_b1 + 256, o_b2 = o_w2 + 256;
    std::vector<float> blob(o_b2 + 4, 0.0f);
    const std::string crit = net.kind == 0 ? "crit_head" : "convs";
    if (net.kind == 0) {
        if (!get(m, "act_head.0.weight", 2 * 128, w) || !bn_fold(m, "act_head.0", "act_head.1", 2, sc, sh))
            return GOAI_ERR_ARG;
        for (int ch = 0; ch < 2; ch++) {
            for (int k = 0; k < 128; k++) blob[o_conv_w + ch * 128 + k] = (float)((double)w[ch * 128 + k] * sc[ch]);
            blob[o_conv_b + ch] = (float)sh[ch];
        }
        const float* lb;
        if (!get(m, "act_head.4.weight", (int64_t)A * 2 * P, w) || !get(m, "act_head.4.bias", A, lb)) return GOAI_ERR_ARG;
        for (int a = 0; a < A; a++) {
            for (int j = 0; j < 2 * P; j++) blob[o_alw + (size_t)j * A + a] = w[(size_t)a * 2 * P + j];
            blob[o_alb + a] = lb[a];
        }
    }
    if (!get(m, crit + ".0.weight", 128, w) || !bn_fold(m, crit + ".0", crit + ".1", 1, sc, sh)) return GOAI_ERR_ARG;
    for (int k = 0; k < 128; k++) blob[o_conv_w + 2 * 128 + k] = (float)((double)w[k] * sc[0]);
    blob[o_conv_b + 2] = (float)sh[0];
    const float *l1b, *l2w, *l2b;
    if (!get(m, crit + ".4.weight", (int64_t)256 * P, w) || !get(m, crit + ".4.bias", 256, l1b) ||
        !get(m, crit + ".6.weight", 256, l2w) || !get(m, crit + ".6.bias", 1, l2b))
        return GOAI_ERR_ARG;
    for (int o = 0; o < 256; o++) {
        for (int p = 0; p < P; p++) blob[o_w1 + (size_t)p * 256 + o] = w[(size_t)o * P + p];
        blob[o_b1 + o] = l1b[o];
        blob[o_w2 + o] = l2w[o];
    }
    blob[o_b2] = l2b[0];
    GOAI_HIP_CHECK(hipMemcpy(net.head_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* hb = net.head_blob;
    net.hw = HeadWeights{ hb + o_conv_w, hb + o_conv_b, hb + o_alw, hb + o_alb, hb + o_w1, hb + o_b1, hb + o_w2, hb + o_b2 };
    net.loaded = true;
    return GOAI_OK;
}

// heads: the trunk's last convolution of a Net::heads_in_trunk context (split F(2x2,3x3)): `out` = net.hbuf receives the
// heads' 1x1 convolutions of the output instead of the output
static int conv_any(goai_ctx* c, const float* in, int l, const float* b, const float* resid, float* out, int M, bool heads = false)
{
    const Net& net = c->net;
    if (net.dtype == GOAI_DTYPE_BF16X3)
        return launch_conv3x3_bf16(c->S, in, (const uint16_t*)net.wt_bf16 + (size_t)l * WT_BF16_PER_CONV, b, resid, out, M,
                                   c->stream);
    const bool wino = net.wino_mode == 2;
    // fp16 parts: the first convolution's input is bounded by amax0, every launch leaves its output's bounds in net.amax
    if (net.split6 && net.wino_mode == 2)
        return launch_conv3x3_wino2s(c->S, const_cast<float*>(in), out, resid,
                                     (const uint16_t*)net.wt_split6 + (size_t)l * wt_split_per_conv(net), b, M / c->P, 1, c->stream,
                                     net.parts, net.parts <= 2 ? net.wexp + l : nullptr, net.amax, l == 0 ? net.amax0 : 0.0f,
                                     heads ? net.hw.conv_w : nullptr, net.hw.conv_b, heads ? out : nullptr);
    if (net.split6 && net.wino3h)
        return launch_conv3x3_wino3h(c->S, const_cast<float*>(in), out, resid,
                                     (const uint16_t*)net.wt_split6 + (size_t)l * wt_split_per_conv(net), b, M / c->P, 1, c->stream,
                                     net.parts, net.wexp + l, net.amax, l == 0 ? net.amax0 : 0.0f);
    if (net.split6)
        return launch_conv3x3_wino3s(c->S, const_cast<float*>(in), out, resid,
                                     (const uint16_t*)net.wt_split6 + (size_t)l * wt_split_per_conv(net), b, net.wino3p_tab, M / c->P, 1,
                                     c->stream, net.parts, net.wexp + l, net.amax, l == 0 ? net.amax0 : 0.0f);
    if (net.wino3_paired)
        return launch_conv3x3_wino3p(c->S, const_cast<float*>(in), out, resid, net.wt_wino3 + (size_t)l * 25 * 128 * 128, b,
                                     net.wino3p_tab, M / c->P, net.n_cu, 1, c->stream);
    if (net.wino_mode == 3)
        return launch_conv3x3_wino3(c->S, const_cast<float*>(in), out, resid, net.wt_wino3 + (size_t)l * 25 * 128 * 128, b,
                                    net.wino3_tab, M / c->P, net.n_cu, 1, c->stream);
    if (wino)
        return launch_conv3x3_wino(c->S, in, net.wt_wino + (size_t)l * 16 * 128 * 128, b, resid, out, M / c->P, c->stream);
    return launch_conv3x3(c->S, in, net.wt + (size_t)l * 9 * 128 * 128, b, resid, out, M, c->stream);
}

static int timed_conv(goai_ctx* c, const float* in, int l, const float* b, const float* resid, float* out, int M, bool heads = false)
{
    if (!c->prof_enabled) return conv_any(c, in, l, b, resid, out, M, heads);
    hipEvent_t e0, e1;
    GOAI_HIP_CHECK(hipEventCreate(&e0));
    GOAI_HIP_CHECK(hipEventCreate(&e1));
    c->prof_events.push_back(e0);
    c->prof_events.push_back(e1);
    GOAI_HIP_CHECK(hipEventRecord(e0, c->stream));
    int rc = conv_any(c, in, l, b, resid, out, M, heads);
    GOAI_HIP_CHECK(hipEventRecord(e1, c->stream));
    return rc;
}

// Raw network outputs for n states (pi may be NULL; always NULL-able for the value net), through the three
// activation buffers `act` of `cap` boards each.  `timed`: bracket the trunk launches with HIP events
// when the context's profile is on (creates events, so not for the allocation-free entry point).
// The residual trunk over the nb boards whose first-layer activations are in act[0]; *out = the buffer (one of the
// three) that holds the result.
// *out_is_h: *out is not an activation buffer but net.hbuf, the heads' 1x1 convolutions (Net::heads_in_trunk)
static int nn_trunk_on(goai_ctx* c, float* const act[3], int nb, bool prof, float** out, bool* out_is_h)
{
    *out_is_h = false;
    Net& net = c->net;
    const int M = nb * c->P;
    int rc;
    float *x = act[0], *y = act[1], *z = act[2];
    if (net.trunk_fused && net.layers > 0) {
        // F(3x3,3x3): a workgroup owns whole boards, so all 2 * layers convolutions run in one launch
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof) {
            GOAI_HIP_CHECK(hipEventCreate(&e0));
            GOAI_HIP_CHECK(hipEventCreate(&e1));
            c->prof_events.push_back(e0);
            c->prof_events.push_back(e1);
            GOAI_HIP_CHECK(hipEventRecord(e0, c->stream));
        }
        rc = net.wino_mode == 2
                 ? launch_conv3x3_wino2s(c->S, x, y, nullptr, net.wt_split6, net.bt, nb, 2 * net.layers, c->stream, net.parts, net.wexp,
                                         nullptr, net.amax0, net.heads_in_trunk ? net.hw.conv_w : nullptr, net.hw.conv_b,
                                         net.heads_in_trunk ? net.hbuf : nullptr)
             : net.wino3h
                 ? launch_conv3x3_wino3h(c->S, x, y, nullptr, net.wt_split6, net.bt, nb, 2 * net.layers, c->stream, net.parts, net.wexp,
                                         nullptr, net.amax0, net.heads_in_trunk ? net.hw.conv_w : nullptr, net.hw.conv_b,
                                         net.heads_in_trunk ? net.hbuf : nullptr)
             : net.split6
                 ? launch_conv3x3_wino3s(c->S, x, y, nullptr, net.wt_split6, net.bt, net.wino3p_tab, nb, 2 * net.layers, c->stream,
                                         net.parts, net.wexp, nullptr, net.amax0, net.heads_in_trunk ? net.hw.conv_w : nullptr,
                                         net.hw.conv_b, net.heads_in_trunk ? net.hbuf : nullptr)
             : net.wino3_paired
                 ? launch_conv3x3_wino3p(c->S, x, y, nullptr, net.wt_wino3, net.bt, net.wino3p_tab, nb, net.n_cu,
                                         2 * net.layers, c->stream)
                 : launch_conv3x3_wino3(c->S, x, y, nullptr, net.wt_wino3, net.bt, net.wino3_tab, nb, net.n_cu,
                                        2 * net.layers, c->stream);
        if (e1) GOAI_HIP_CHECK(hipEventRecord(e1, c->stream));
        if (rc) return rc;
        if (net.heads_in_trunk) {
            *out = net.hbuf;
            *out_is_h = true;
            return GOAI_OK;
        }
    } else
    for (int l = 0; l < net.layers; l++) {
        const bool heads = net.heads_in_trunk && net.wino_mode == 2 && l == net.layers - 1;
        if (prof) {
            if ((rc = timed_conv(c, x, 2 * l, net.bt + (size_t)(2 * l) * 128, nullptr, y, M))) return rc;
            if ((rc = timed_conv(c, y, 2 * l + 1, net.bt + (size_t)(2 * l + 1) * 128, x, heads ? net.hbuf : z, M, heads))) return rc;
        } else {
            if ((rc = conv_any(c, x, 2 * l, net.bt + (size_t)(2 * l) * 128, nullptr, y, M))) return rc;
            if ((rc = conv_any(c, y, 2 * l + 1, net.bt + (size_t)(2 * l + 1) * 128, x, heads ? net.hbuf : z, M, heads))) return rc;
        }
        float* t = x; x = z; z = t;
    }
    *out = x;
    if (net.heads_in_trunk && net.wino_mode == 2 && net.layers > 0) {
        *out = net.hbuf;
        *out_is_h = true;
    }
    return GOAI_OK;
}

static int nn_forward_on(goai_ctx* c, float* const act[3], int cap, bool timed, const uint8_t* states, float* pi,
                         float* val, int n)
{
    Net& net = c->net;
    GOAI_REQUIRE(net.loaded, "weights not loaded (goai_load_weights)");
    const int P = c->P, A = c->A;
    const bool prof = timed && c->prof_enabled;
    for (int done = 0; done < n; done += cap) {
        const int nb = (n - done) < cap ? (n - done) : cap;
        const int M = nb * P;
        int rc;
        if ((rc = launch_conv0(c->S, states + (size_t)done * 6 * P, net.w0, net.w0_split, net.b0, net.shift, act[0], nb, c->stream)))
            return rc;
        float* x = nullptr;
        bool x_is_h = false;
        if ((rc = nn_trunk_on(c, act, nb, prof, &x, &x_is_h))) return rc;
        if ((rc = launch_heads(c->S, x, net.hw, net.kind == 0 && pi != nullptr, pi ? pi + (size_t)done * A : nullptr, val + done, nb,
                               c->stream, x_is_h)))
            return rc;
    }
    return GOAI_OK;
}

// The forward in three pieces on the context's own activation buffers, for the search's fused step (mcts_host.hip):
// n <= max_batch boards, first convolution into act[0], trunk (timed like nn_forward_raw's), heads.
int nn_piece_conv0(goai_ctx* c, const uint8_t* states, int n)
{
    Net& net = c->net;
    return launch_conv0(c->S, states, net.w0, net.w0_split, net.b0, net.shift, net.act[0], n, c->stream);
}
int nn_piece_trunk(goai_ctx* c, int n, float** out, bool* out_is_h)
{
    return nn_trunk_on(c, c->net.act, n, c->prof_enabled, out, out_is_h);
}
int nn_piece_heads(goai_ctx* c, const float* x, bool x_is_h, float* pi, float* val, int n)
{
    return launch_heads(c->S, x, c->net.hw, c->net.kind == 0 && pi != nullptr, pi, val, n, c->stream, x_is_h);
}

int nn_forward_raw(goai_ctx* c, const uint8_t* states, float* pi, float* val, int n)
{
    return nn_forward_on(c, c->net.act, c->net.max_batch, true, states, pi, val, n);
}

// ---- the allocation-free forward (goai_nn_forward_static) --------------------------------------
// The three activation buffers of a small batch live in a file-scope device array instead of the
// context's hipMalloc'ed ones: NN_STATIC_ROWS board cells x 128 channels x fp32 x 3 = 12 MiB, held by
// every GPU the library runs on until the process ends.  `scratch` is only a host-side handle here: its
// device address is looked up once per device (nn_static_setup, called by goai_ctx_create with that
// device current) and everything below uses that pointer.
static __device__ __align__(256) float scratch[3 * GOAI_NN_STATIC_MAX_CELLS * 128];
static std::mutex g_scratch_mutex;
static float* g_scratch_ptr[64];          // per device

int nn_static_setup(int device)
{
    GOAI_REQUIRE(device >= 0 && device < 64, "device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return GOAI_OK;
    float* p = nullptr;
    GOAI_HIP_CHECK(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    GOAI_REQUIRE(p != nullptr, "hipGetSymbolAddress returned NULL for the static activation buffers");
    g_scratch_ptr[device] = p;
    return GOAI_OK;
}

int nn_forward_static(goai_ctx* c, const uint8_t* states, float* pi, float* val, int n)
{
    GOAI_REQUIRE(c->device >= 0 && c->device < 64 && g_scratch_ptr[c->device] != nullptr,
                 "static activation buffers were not set up on this device");
    if ((size_t)n * c->P > (size_t)GOAI_NN_STATIC_MAX_CELLS) {
        set_error("goai_nn_forward_static: n * S * S = " + std::to_string((size_t)n * c->P) + " board cells exceed "
                  "GOAI_NN_STATIC_MAX_CELLS = " + std::to_string(GOAI_NN_STATIC_MAX_CELLS) + "; use goai_nn_forward");
        return GOAI_ERR_ARG;
    }
    float* base = g_scratch_ptr[c->device];
    float* const act[3] = { base, base + (size_t)GOAI_NN_STATIC_MAX_CELLS * 128, base + (size_t)2 * GOAI_NN_STATIC_MAX_CELLS * 128 };
    const int cap_static = GOAI_NN_STATIC_MAX_CELLS / c->P;
    const int cap = cap_static < c->net.max_batch ? cap_static : c->net.max_batch;
    return nn_forward_on(c, act, cap, false, states, pi, val, n);
}

}  // namespace goai
