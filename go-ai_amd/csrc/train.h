// train.h -- device state of the optimisation step (train_host.hip) and the launchers of train_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

namespace goai {

constexpr int TR_SPLIT_ROWS = 10240;   // below this many rows the 128 -> 128 convolutions run split over input-channel chunks
constexpr int TR_MAX_SLICES = 64;      // row slices of the deterministic reductions (partials, then an ordered sum)

// one named tensor of the reference's state_dict inside the flat buffers
struct TrainSlot {
    bool stat;       // BatchNorm running statistic (lives in `stats`) rather than a parameter (`params`)
    size_t off;      // in floats
    size_t numel;
};

struct Train {
    bool ready = false;
    int kind = -1, layers = 0, max_batch = 0;
    int64_t steps = 0;                  // Adam's t
    size_t n_params = 0, n_stats = 0;
    std::vector<std::string> order;     // parameter names in parameters() order, then the statistics
    std::map<std::string, TrainSlot> slots;
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *stats = nullptr;
    float *img_f = nullptr, *img_d = nullptr;   // [2*layers][4][9][128][32]: forward / data-gradient weight images
    float* zero128 = nullptr;
    std::vector<float*> z;              // [2*layers+1] pre-BatchNorm convolution outputs [M][128]
    std::vector<float*> x;              // [layers+1] block inputs / trunk output
    std::vector<float*> y;              // [layers] mid-block activations
    float* g[4] = { nullptr, nullptr, nullptr, nullptr };   // gradient work buffers [M][128]
    float* bn_saved = nullptr;          // [2*layers+2][2][128]: batch mean / invstd per BatchNorm (last: the head's)
    float *col_partial = nullptr, *bn_sums = nullptr, *wgrad_partial = nullptr, *conv_partial = nullptr;
    float *hz = nullptr, *dhz = nullptr, *f = nullptr, *df = nullptr;   // head: [M][2], [B][2P]
    float *h1 = nullptr, *dh1 = nullptr, *logits = nullptr, *dlogits = nullptr, *metrics = nullptr;
    // The step as a HIP graph (round 3; GOAI_NO_TRAIN_GRAPH=1 at goai_train_init: plain launches): a step is ~55 launches per
    // residual block and pass, all with arguments that only depend on the batch size, once the caller's arrays are copied
    // to these staging buffers and the step's scalars (lr, weight decay, Adam's bias corrections) are read from `scal`.
    bool graph_on = false;
    uint8_t *in_states = nullptr, *in_next = nullptr;   // [max_batch][6][P]
    float *in_wins = nullptr, *in_pi = nullptr;         // [max_batch], [max_batch][A]
    float* scal = nullptr;                              // device: lr, weight decay, 1 - 0.9^t, sqrt(1 - 0.999^t)
    float scal_host[4] = { 0.0f, 0.0f, 1.0f, 1.0f };
    std::map<int, hipGraphExec_t> graphs;               // by batch size
    hipStream_t gstream = nullptr;                      // the step's own stream when the context runs on the null stream (which cannot be captured)
};

int tr_conv0_fwd(int S, const uint8_t* states, const float* w, const float* b, float shift, float* z, int M, hipStream_t st);
int tr_conv0_bwd(int S, const uint8_t* states, const float* dz, float shift, float* dw, float* db, int M, hipStream_t st);
int tr_relayout(const float* w, float* img_f, float* img_d, hipStream_t st);
int tr_bn_stats(const float* z, float* partial, float* mean, float* invstd, float* run_mean, float* run_var, int M, hipStream_t st);
int tr_bn_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, const float* resid,
              float* a, int M, hipStream_t st);
int tr_bn_bwd(const float* d, const float* a, const float* z, const float* mean, const float* invstd, const float* gamma,
              float* partial, float* sums, float* dgamma, float* dbeta, float* dz, float* dres, int M, hipStream_t st);
int tr_colsum_add(const float* x, float* partial, float* dst, int M, hipStream_t st);
int tr_wgrad(int S, const float* a_in, const float* dz, float* partial, float* dw, int M, hipStream_t st);
int tr_head_fwd(const float* a, const float* w, const float* b, const float* gamma, const float* beta, float* hz, float* mean,
                float* invstd, float* run_mean, float* run_var, float* f, int M, int C, int P, hipStream_t st);
int tr_head_bwd(const float* df, const float* f, const float* hz, const float* mean, const float* invstd, const float* gamma,
                const float* a, const float* w, float* partial, float* dgamma, float* dbeta, float* dw, float* db, float* dhz,
                float* dA, int M, int C, int P, hipStream_t st);
int tr_linear_fwd(const float* f, const float* W, const float* bias, float* out, int B, int O, int J, int relu, hipStream_t st);
int tr_linear_bwd(const float* dout, const float* f, const float* W, const float* act_mask, float* dW, float* db, float* df, int B,
                  int O, int J, hipStream_t st);
int tr_critic_loss(const float* val, const float* wins, float* dval, float* metrics, int B, float weight, hipStream_t st);
int tr_actor_loss(const float* logits, const float* pi, float* dlogits, float* metrics, int B, int A, hipStream_t st);
int tr_fill(float* dst, long n, float v, hipStream_t st);
// scal (device): lr, weight decay, bc1, sqrt(bc2)
int tr_adam(float* p, const float* g, float* m, float* v, long n, const float* scal, hipStream_t st);

}  // namespace goai
