// train_host.hip -- host side of the optimisation step: flat parameter / gradient / Adam buffers addressed by the
// reference's state_dict keys, and the launch sequence of one train_step.
//
//   ActorCriticNet.train_step  /root/reference/go_ai/models/ac_net.py:67-86   critic pass on next_states (target -wins),
//                                                                             actor pass on states, loss 2 * cl + al
//   ValueNet.train_step        /root/reference/go_ai/models/val_net.py:29-43  critic pass only, loss cl
//   average_model              /root/reference/go_ai/models/__init__.py:279-282  (the caller all-reduces `params`)
//
// The two passes of an actor-critic step share the activation buffers: forward + backward of the critic pass, then
// forward + backward of the actor pass, gradients accumulating in `grads` (fp32 addition of the two contributions
// commutes, so the order autograd would have used does not matter).  BatchNorm running statistics move once per
// forward pass, critic pass first, as in the reference.
#include <cmath>
#include <algorithm>
#include <cstring>
#include <cstdlib>

#include "ctx.h"
#include "goai_internal.h"

namespace goai {

static int tr_alloc(goai_ctx* c, float** p, size_t floats)
{
    GOAI_HIP_CHECK(hipMalloc((void**)p, floats * sizeof(float)));
    c->allocs.push_back(*p);
    return GOAI_OK;
}

static void add_slot(Train& t, const std::string& name, bool stat, size_t numel)
{
    size_t& total = stat ? t.n_stats : t.n_params;
    t.slots[name] = TrainSlot{ stat, total, numel };
    t.order.push_back(name);
    total += numel;
}

static void add_conv_bn(Train& t, const std::string& conv, const std::string& bn, int cout, int cin_k)
{
    add_slot(t, conv + ".weight", false, (size_t)cout * cin_k);
    add_slot(t, conv + ".bias", false, cout);
    add_slot(t, bn + ".weight", false, cout);
    add_slot(t, bn + ".bias", false, cout);
    add_slot(t, bn + ".running_mean", true, cout);
    add_slot(t, bn + ".running_var", true, cout);
}

static std::string block_prefix(int i) { return "main." + std::to_string(3 + i) + ".main."; }

static int train_init_impl(goai_ctx* c, int kind, int layers, int max_batch);

// A failed init leaves the context as it was (its partial allocations are freed), so it can be retried.
int train_init(goai_ctx* c, int kind, int layers, int max_batch)
{
    GOAI_REQUIRE(kind == 0 || kind == 1, "net_kind must be 0 (actor-critic) or 1 (value)");
    GOAI_REQUIRE(layers >= 0 && layers <= 64, "n_blocks out of range");
    GOAI_REQUIRE(max_batch > 1, "max_batch must be at least 2 (BatchNorm needs a batch)");
    GOAI_REQUIRE((long)max_batch * c->P * 128 < (1L << 31), "max_batch * board cells too large");
    GOAI_REQUIRE(!c->train.ready && c->train.kind < 0, "goai_train_init was already called on this context");
    const size_t mark = c->allocs.size(), host_mark = c->host_allocs.size();
    const int rc = train_init_impl(c, kind, layers, max_batch);
    if (rc) {
        c->release_from(mark, host_mark);
        c->train = Train();
    }
    return rc;
}

static int train_init_impl(goai_ctx* c, int kind, int layers, int max_batch)
{
    Train& t = c->train;
    t.kind = kind;
    t.layers = layers;
    t.max_batch = max_batch;
    const size_t P = c->P, A = c->A;
    // parameters() order of the reference's modules (RLNet.main, then the heads in definition order)
    add_conv_bn(t, "main.0", "main.1", 128, 6 * 9);
    for (int i = 0; i < layers; i++) {
        add_conv_bn(t, block_prefix(i) + "0", block_prefix(i) + "1", 128, 128 * 9);
        add_conv_bn(t, block_prefix(i) + "3", block_prefix(i) + "4", 128, 128 * 9);
    }
    if (kind == 0) {
        add_conv_bn(t, "act_head.0", "act_head.1", 2, 128);
        add_slot(t, "act_head.4.weight", false, A * 2 * P);
        add_slot(t, "act_head.4.bias", false, A);
    }
    const std::string head = kind == 0 ? "crit_head" : "convs";
    add_conv_bn(t, head + ".0", head + ".1", 1, 128);
    add_slot(t, head + ".4.weight", false, 256 * P);
    add_slot(t, head + ".4.bias", false, 256);
    add_slot(t, head + ".6.weight", false, 256);
    add_slot(t, head + ".6.bias", false, 1);

    int rc;
    float** flat[] = { &t.params, &t.grads, &t.adam_m, &t.adam_v };
    for (float** p : flat) {
        if ((rc = tr_alloc(c, p, t.n_params))) return rc;
        GOAI_HIP_CHECK(hipMemsetAsync(*p, 0, t.n_params * sizeof(float), c->stream));
    }
    if ((rc = tr_alloc(c, &t.stats, t.n_stats))) return rc;
    GOAI_HIP_CHECK(hipMemsetAsync(t.stats, 0, t.n_stats * sizeof(float), c->stream));
    const size_t img = (size_t)9 * 128 * 128, MB = (size_t)max_batch * P * 128;
    if (layers > 0) {
        if ((rc = tr_alloc(c, &t.img_f, 2 * layers * img)) || (rc = tr_alloc(c, &t.img_d, 2 * layers * img))) return rc;
    }
    if ((rc = tr_alloc(c, &t.zero128, 128))) return rc;
    GOAI_HIP_CHECK(hipMemsetAsync(t.zero128, 0, 128 * sizeof(float), c->stream));
    t.z.assign(2 * layers + 1, nullptr);
    t.x.assign(layers + 1, nullptr);
    t.y.assign(layers, nullptr);
    for (auto* vec : { &t.z, &t.x, &t.y })
        for (float*& p : *vec)
            if ((rc = tr_alloc(c, &p, MB))) return rc;
    for (float*& p : t.g)
        if ((rc = tr_alloc(c, &p, MB))) return rc;
    if ((rc = tr_alloc(c, &t.bn_saved, (size_t)(2 * layers + 2) * 2 * 128))) return rc;
    if ((rc = tr_alloc(c, &t.col_partial, (size_t)TR_MAX_SLICES * 2 * 128))) return rc;
    if ((rc = tr_alloc(c, &t.bn_sums, 2 * 128))) return rc;
    if ((rc = tr_alloc(c, &t.wgrad_partial, (size_t)TR_MAX_SLICES * img))) return rc;
    if (layers > 0 && (rc = tr_alloc(c, &t.conv_partial, (size_t)4 * std::min<size_t>(max_batch * P, TR_SPLIT_ROWS) * 128))) return rc;
    const size_t B = max_batch;
    if ((rc = tr_alloc(c, &t.hz, B * P * 2)) || (rc = tr_alloc(c, &t.dhz, B * P * 2)) || (rc = tr_alloc(c, &t.f, B * P * 2)) ||
        (rc = tr_alloc(c, &t.df, B * P * 2)) || (rc = tr_alloc(c, &t.h1, B * 256)) || (rc = tr_alloc(c, &t.dh1, B * 256)) ||
        (rc = tr_alloc(c, &t.logits, B * A)) || (rc = tr_alloc(c, &t.dlogits, B * A)) || (rc = tr_alloc(c, &t.metrics, 4)))
        return rc;
    {
        float* stage = nullptr;                       // the caller's arrays, copied here in front of the step's graph
        const size_t st_floats = (B * 6 * P + 3) / 4;
        if ((rc = tr_alloc(c, &stage, st_floats))) return rc;
        t.in_states = reinterpret_cast<uint8_t*>(stage);
        if ((rc = tr_alloc(c, &stage, st_floats))) return rc;
        t.in_next = reinterpret_cast<uint8_t*>(stage);
        if ((rc = tr_alloc(c, &t.in_wins, B)) || (rc = tr_alloc(c, &t.in_pi, B * A)) || (rc = tr_alloc(c, &t.scal, 4))) return rc;
    }
    t.graph_on = !getenv("GOAI_NO_TRAIN_GRAPH");
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    t.steps = 0;
    t.ready = true;
    return GOAI_OK;
}

static int find_slot(const Train& t, const char* name, int64_t numel, const TrainSlot*& out)
{
    GOAI_REQUIRE(name != nullptr, "NULL tensor name");
    auto it = t.slots.find(name);
    if (it == t.slots.end()) { set_error(std::string("the training net has no tensor ") + name); return GOAI_ERR_ARG; }
    if ((int64_t)it->second.numel != numel) {
        set_error(std::string("tensor ") + name + " has " + std::to_string(numel) + " elements, expected " +
                  std::to_string(it->second.numel));
        return GOAI_ERR_ARG;
    }
    out = &it->second;
    return GOAI_OK;
}

static bool ignored_key(const std::string& k)
{
    return k.rfind("game_head.", 0) == 0 || (k.size() >= 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0);
}

// Parameters and BatchNorm statistics from host tensors by state_dict key (game_head.* and *.num_batches_tracked are
// skipped, as in goai_load_weights).  strict: every tensor of the net must be present.  Adam's state is left alone
// (train_reset_optimizer).
int train_load(goai_ctx* c, const char* const* names, const float* const* tensors, const int64_t* numels, int n, int strict)
{
    Train& t = c->train;
    GOAI_REQUIRE(t.ready, "call goai_train_init first");
    GOAI_REQUIRE(names && tensors && numels && n > 0, "NULL / empty weight table");
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    size_t seen = 0;
    for (int i = 0; i < n; i++) {
        GOAI_REQUIRE(names[i] && tensors[i], "NULL entry in weight table");
        if (ignored_key(names[i])) continue;
        const TrainSlot* s;
        const int rc = find_slot(t, names[i], numels[i], s);
        if (rc) return rc;
        GOAI_HIP_CHECK(hipMemcpy((s->stat ? t.stats : t.params) + s->off, tensors[i], s->numel * sizeof(float), hipMemcpyHostToDevice));
        seen++;
    }
    if (strict && seen != t.slots.size()) {
        set_error("state_dict covers " + std::to_string(seen) + " of the net's " + std::to_string(t.slots.size()) + " tensors");
        return GOAI_ERR_ARG;
    }
    return GOAI_OK;
}

// a fresh torch.optim.Adam: moments and step count to zero
int train_reset_optimizer(goai_ctx* c)
{
    Train& t = c->train;
    GOAI_REQUIRE(t.ready, "call goai_train_init first");
    GOAI_HIP_CHECK(hipMemsetAsync(t.adam_m, 0, t.n_params * sizeof(float), c->stream));
    GOAI_HIP_CHECK(hipMemsetAsync(t.adam_v, 0, t.n_params * sizeof(float), c->stream));
    GOAI_HIP_CHECK(hipMemsetAsync(t.grads, 0, t.n_params * sizeof(float), c->stream));
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    t.steps = 0;
    return GOAI_OK;
}

// what: 0 value (parameter or statistic), 1 gradient of the last step, 2 / 3 Adam's first / second moment
int train_export(goai_ctx* c, int what, const char* const* names, float* const* tensors, const int64_t* numels, int n)
{
    Train& t = c->train;
    GOAI_REQUIRE(t.ready, "call goai_train_init first");
    GOAI_REQUIRE(what >= 0 && what <= 3, "what must be 0 (values), 1 (gradients), 2 or 3 (Adam moments)");
    GOAI_REQUIRE(names && tensors && numels && n > 0, "NULL / empty tensor table");
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    const float* flat[] = { t.params, t.grads, t.adam_m, t.adam_v };
    for (int i = 0; i < n; i++) {
        GOAI_REQUIRE(tensors[i] != nullptr, "NULL entry in tensor table");
        const TrainSlot* s;
        const int rc = find_slot(t, names[i], numels[i], s);
        if (rc) return rc;
        GOAI_REQUIRE(!s->stat || what == 0, "BatchNorm statistics have no gradient / moments");
        GOAI_HIP_CHECK(hipMemcpy(tensors[i], (s->stat ? t.stats : flat[what]) + s->off, s->numel * sizeof(float), hipMemcpyDeviceToHost));
    }
    return GOAI_OK;
}

namespace {
struct Names {
    std::string conv, bn;
};
struct Pass {
    goai_ctx* c;
    Train& t;
    int S, P, A, n, M;
    hipStream_t st;
    float* par(const std::string& k) const { return t.params + t.slots.at(k).off; }
    float* grad(const std::string& k) const { return t.grads + t.slots.at(k).off; }
    float* stat(const std::string& k) const { return t.stats + t.slots.at(k).off; }
    float* mean(int bn) const { return t.bn_saved + (size_t)bn * 256; }
    float* invstd(int bn) const { return t.bn_saved + (size_t)bn * 256 + 128; }
};

Names trunk_names(int l)   // l = 0: conv0; l = 1 + 2 i / 2 + 2 i: first / second convolution of block i
{
    if (l == 0) return { "main.0", "main.1" };
    const int i = (l - 1) / 2, second = (l - 1) % 2;
    return { block_prefix(i) + (second ? "3" : "0"), block_prefix(i) + (second ? "4" : "1") };
}

#define TR_TRY(expr) do { const int _rc = (expr); if (_rc) return _rc; } while (0)

// out = conv3x3(in) + bias (+ resid), no ReLU.  Few rows: split over the four input-channel chunks (4 x the workgroups).
int conv_linear(const Pass& p, const float* in, const float* img, const float* bias, const float* resid, float* out)
{
    if (p.M < TR_SPLIT_ROWS) return launch_conv3x3_linear_split(p.S, in, img, bias, resid, out, p.t.conv_partial, p.M, p.st);
    return launch_conv3x3_linear(p.S, in, img, bias, resid, out, p.M, p.st);
}

// statistics of z[l], then a = relu(bn(z[l]) (+ resid))
int bn_layer_fwd(const Pass& p, int l, const float* resid, float* a)
{
    const Names nm = trunk_names(l);
    TR_TRY(tr_bn_stats(p.t.z[l], p.t.col_partial, p.mean(l), p.invstd(l), p.stat(nm.bn + ".running_mean"),
                       p.stat(nm.bn + ".running_var"), p.M, p.st));
    return tr_bn_fwd(p.t.z[l], p.mean(l), p.invstd(l), p.par(nm.bn + ".weight"), p.par(nm.bn + ".bias"), resid, a, p.M, p.st);
}

int trunk_forward(const Pass& p, const uint8_t* states)
{
    Train& t = p.t;
    TR_TRY(tr_conv0_fwd(p.S, states, p.par("main.0.weight"), p.par("main.0.bias"), t.kind == 0 ? 0.5f : 0.0f, t.z[0], p.M, p.st));
    TR_TRY(bn_layer_fwd(p, 0, nullptr, t.x[0]));
    const size_t img = (size_t)9 * 128 * 128;
    for (int i = 0; i < t.layers; i++) {
        const int l1 = 1 + 2 * i, l2 = 2 + 2 * i;
        TR_TRY(conv_linear(p, t.x[i], t.img_f + (size_t)(l1 - 1) * img, p.par(trunk_names(l1).conv + ".bias"), nullptr, t.z[l1]));
        TR_TRY(bn_layer_fwd(p, l1, nullptr, t.y[i]));
        TR_TRY(conv_linear(p, t.y[i], t.img_f + (size_t)(l2 - 1) * img, p.par(trunk_names(l2).conv + ".bias"), nullptr, t.z[l2]));
        TR_TRY(bn_layer_fwd(p, l2, t.x[i], t.x[i + 1]));
    }
    return GOAI_OK;
}

// d(loss) / d(trunk output) is in g[0] on entry
int bn_layer_bwd(const Pass& p, int l, const float* d, const float* a, float* dz, float* dres)
{
    const Names nm = trunk_names(l);
    return tr_bn_bwd(d, a, p.t.z[l], p.mean(l), p.invstd(l), p.par(nm.bn + ".weight"), p.t.col_partial, p.t.bn_sums,
                     p.grad(nm.bn + ".weight"), p.grad(nm.bn + ".bias"), dz, dres, p.M, p.st);
}

int trunk_backward(const Pass& p, const uint8_t* states)
{
    Train& t = p.t;
    const size_t img = (size_t)9 * 128 * 128;
    for (int i = t.layers - 1; i >= 0; i--) {
        const int l1 = 1 + 2 * i, l2 = 2 + 2 * i;
        const Names n1 = trunk_names(l1), n2 = trunk_names(l2);
        // x[i+1] = relu(bn(z[l2]) + x[i]): g[0] -> dz2 in g[1], the residual branch's share in g[2]
        TR_TRY(bn_layer_bwd(p, l2, t.g[0], t.x[i + 1], t.g[1], t.g[2]));
        TR_TRY(tr_wgrad(p.S, t.y[i], t.g[1], t.wgrad_partial, p.grad(n2.conv + ".weight"), p.M, p.st));
        TR_TRY(tr_colsum_add(t.g[1], t.col_partial, p.grad(n2.conv + ".bias"), p.M, p.st));
        TR_TRY(conv_linear(p, t.g[1], t.img_d + (size_t)(l2 - 1) * img, t.zero128, nullptr, t.g[3]));
        // y[i] = relu(bn(z[l1])): g[3] -> dz1 in g[1]
        TR_TRY(bn_layer_bwd(p, l1, t.g[3], t.y[i], t.g[1], nullptr));
        TR_TRY(tr_wgrad(p.S, t.x[i], t.g[1], t.wgrad_partial, p.grad(n1.conv + ".weight"), p.M, p.st));
        TR_TRY(tr_colsum_add(t.g[1], t.col_partial, p.grad(n1.conv + ".bias"), p.M, p.st));
        TR_TRY(conv_linear(p, t.g[1], t.img_d + (size_t)(l1 - 1) * img, t.zero128, t.g[2], t.g[0]));
    }
    TR_TRY(bn_layer_bwd(p, 0, t.g[0], t.x[0], t.g[1], nullptr));
    return tr_conv0_bwd(p.S, states, t.g[1], t.kind == 0 ? 0.5f : 0.0f, p.grad("main.0.weight"), p.grad("main.0.bias"), p.M, p.st);
}

// critic_step (models/__init__.py:115-134) through crit_head / convs, loss weight `weight` in the total
int critic_pass(const Pass& p, const uint8_t* states, const float* wins, float weight)
{
    Train& t = p.t;
    const std::string h = t.kind == 0 ? "crit_head" : "convs";
    const int hb = 2 * t.layers + 1;
    float* xl = t.x[t.layers];
    TR_TRY(trunk_forward(p, states));
    TR_TRY(tr_head_fwd(xl, p.par(h + ".0.weight"), p.par(h + ".0.bias"), p.par(h + ".1.weight"), p.par(h + ".1.bias"), t.hz,
                       p.mean(hb), p.invstd(hb), p.stat(h + ".1.running_mean"), p.stat(h + ".1.running_var"), t.f, p.M, 1, p.P, p.st));
    TR_TRY(tr_linear_fwd(t.f, p.par(h + ".4.weight"), p.par(h + ".4.bias"), t.h1, p.n, 256, p.P, 1, p.st));
    TR_TRY(tr_linear_fwd(t.h1, p.par(h + ".6.weight"), p.par(h + ".6.bias"), t.logits, p.n, 1, 256, 0, p.st));
    TR_TRY(tr_critic_loss(t.logits, wins, t.dlogits, t.metrics, p.n, weight, p.st));
    TR_TRY(tr_linear_bwd(t.dlogits, t.h1, p.par(h + ".6.weight"), t.h1, p.grad(h + ".6.weight"), p.grad(h + ".6.bias"), t.dh1, p.n, 1,
                         256, p.st));
    TR_TRY(tr_linear_bwd(t.dh1, t.f, p.par(h + ".4.weight"), nullptr, p.grad(h + ".4.weight"), p.grad(h + ".4.bias"), t.df, p.n, 256,
                         p.P, p.st));
    TR_TRY(tr_head_bwd(t.df, t.f, t.hz, p.mean(hb), p.invstd(hb), p.par(h + ".1.weight"), xl, p.par(h + ".0.weight"), t.col_partial,
                       p.grad(h + ".1.weight"), p.grad(h + ".1.bias"), p.grad(h + ".0.weight"), p.grad(h + ".0.bias"), t.dhz, t.g[0],
                       p.M, 1, p.P, p.st));
    return trunk_backward(p, states);
}

// actor_step (models/__init__.py:167-186) through act_head
int actor_pass(const Pass& p, const uint8_t* states, const float* pi)
{
    Train& t = p.t;
    const std::string h = "act_head";
    const int hb = 2 * t.layers + 1;
    float* xl = t.x[t.layers];
    TR_TRY(trunk_forward(p, states));
    TR_TRY(tr_head_fwd(xl, p.par(h + ".0.weight"), p.par(h + ".0.bias"), p.par(h + ".1.weight"), p.par(h + ".1.bias"), t.hz,
                       p.mean(hb), p.invstd(hb), p.stat(h + ".1.running_mean"), p.stat(h + ".1.running_var"), t.f, p.M, 2, p.P, p.st));
    TR_TRY(tr_linear_fwd(t.f, p.par(h + ".4.weight"), p.par(h + ".4.bias"), t.logits, p.n, p.A, 2 * p.P, 0, p.st));
    TR_TRY(tr_actor_loss(t.logits, pi, t.dlogits, t.metrics, p.n, p.A, p.st));
    TR_TRY(tr_linear_bwd(t.dlogits, t.f, p.par(h + ".4.weight"), nullptr, p.grad(h + ".4.weight"), p.grad(h + ".4.bias"), t.df, p.n,
                         p.A, 2 * p.P, p.st));
    TR_TRY(tr_head_bwd(t.df, t.f, t.hz, p.mean(hb), p.invstd(hb), p.par(h + ".1.weight"), xl, p.par(h + ".0.weight"), t.col_partial,
                       p.grad(h + ".1.weight"), p.grad(h + ".1.bias"), p.grad(h + ".0.weight"), p.grad(h + ".0.bias"), t.dhz, t.g[0],
                       p.M, 2, p.P, p.st));
    return trunk_backward(p, states);
}
}  // namespace

// One train_step.  states / next_states: device uint8 [n][6][S][S]; wins: device float [n]; pi: device float [n][A]
// (actor-critic only; states and pi are ignored for the value net).  metrics_host: critic loss, critic accuracy,
// actor loss, actor accuracy (NaN for the value net's actor entries).
int train_step(goai_ctx* c, const uint8_t* states, const uint8_t* next_states, const float* wins, const float* pi, int n, float lr,
               float weight_decay, float* metrics_host)
{
    Train& t = c->train;
    GOAI_REQUIRE(t.ready, "call goai_train_init first");
    GOAI_REQUIRE(n >= 2 && n <= t.max_batch, "batch size must be in 2 .. max_batch");
    GOAI_REQUIRE(next_states && wins && metrics_host, "NULL array argument");
    GOAI_REQUIRE(t.kind == 1 || (states && pi), "the actor-critic step needs states and pi");
    hipStream_t st = c->stream;
    if (t.graph_on && st == nullptr) {
        // the null stream cannot be captured: the step runs on a stream of its own, behind everything queued so far
        if (!t.gstream) GOAI_HIP_CHECK(hipStreamCreateWithFlags(&t.gstream, hipStreamNonBlocking));
        GOAI_HIP_CHECK(hipStreamSynchronize(nullptr));
        st = t.gstream;
    }
    const Pass p{ c, t, c->S, c->P, c->A, n, n * c->P, st };
    t.steps++;
    const double bc1 = 1.0 - std::pow(0.9, (double)t.steps), bc2 = 1.0 - std::pow(0.999, (double)t.steps);
    t.scal_host[0] = lr;
    t.scal_host[1] = weight_decay;
    t.scal_host[2] = (float)bc1;
    t.scal_host[3] = (float)std::sqrt(bc2);
    GOAI_HIP_CHECK(hipMemcpyAsync(t.scal, t.scal_host, sizeof(t.scal_host), hipMemcpyHostToDevice, p.st));
    // everything between the caller's arrays and the metrics: the same launches for every step of one batch size
    auto body = [&](const uint8_t* st_in, const uint8_t* nx_in, const float* wins_in, const float* pi_in) -> int {
        const size_t img = (size_t)9 * 128 * 128;
        TR_TRY(tr_fill(t.grads, (long)t.n_params, 0.0f, p.st));                                  // optimizer.zero_grad()
        TR_TRY(tr_fill(t.metrics, 4, NAN, p.st));
        for (int l = 1; l <= 2 * t.layers; l++)
            TR_TRY(tr_relayout(p.par(trunk_names(l).conv + ".weight"), t.img_f + (size_t)(l - 1) * img, t.img_d + (size_t)(l - 1) * img, p.st));
        TR_TRY(critic_pass(p, nx_in, wins_in, t.kind == 0 ? 2.0f : 1.0f));
        if (t.kind == 0) TR_TRY(actor_pass(p, st_in, pi_in));
        return tr_adam(t.params, t.grads, t.adam_m, t.adam_v, (long)t.n_params, t.scal, p.st);
    };
    int rc = GOAI_OK;
    if (t.graph_on) {
        const size_t sb = (size_t)n * 6 * c->P;
        if (t.kind == 0) {
            GOAI_HIP_CHECK(hipMemcpyAsync(t.in_states, states, sb, hipMemcpyDeviceToDevice, p.st));
            GOAI_HIP_CHECK(hipMemcpyAsync(t.in_pi, pi, (size_t)n * c->A * sizeof(float), hipMemcpyDeviceToDevice, p.st));
        }
        GOAI_HIP_CHECK(hipMemcpyAsync(t.in_next, next_states, sb, hipMemcpyDeviceToDevice, p.st));
        GOAI_HIP_CHECK(hipMemcpyAsync(t.in_wins, wins, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, p.st));
        auto it = t.graphs.find(n);
        if (it == t.graphs.end()) {
            // first step of this batch size: record the launches (nothing runs while the stream is capturing)
            hipGraph_t graph = nullptr;
            GOAI_HIP_CHECK(hipStreamBeginCapture(p.st, hipStreamCaptureModeThreadLocal));
            rc = body(t.in_states, t.in_next, t.in_wins, t.in_pi);
            const hipError_t ce = hipStreamEndCapture(p.st, &graph);
            if (rc == GOAI_OK && ce != hipSuccess) {
                set_error(std::string("hipStreamEndCapture (training step): ") + hipGetErrorString(ce));
                rc = GOAI_ERR_HIP;
            }
            hipGraphExec_t exec = nullptr;
            if (rc == GOAI_OK) {
                const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                if (ie != hipSuccess) {
                    set_error(std::string("hipGraphInstantiate (training step): ") + hipGetErrorString(ie));
                    rc = GOAI_ERR_HIP;
                }
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (rc == GOAI_OK) it = t.graphs.emplace(n, exec).first;
        }
        if (rc == GOAI_OK) GOAI_HIP_CHECK(hipGraphLaunch(it->second, p.st));
    } else
        rc = body(states, next_states, wins, pi);
    if (rc != GOAI_OK) {
        t.steps--;
        (void)hipStreamSynchronize(p.st);
        return rc;
    }
    GOAI_HIP_CHECK(hipMemcpyAsync(metrics_host, t.metrics, 4 * sizeof(float), hipMemcpyDeviceToHost, p.st));
    GOAI_HIP_CHECK(hipStreamSynchronize(p.st));
    return GOAI_OK;
}

// The trained weights into this context's inference images (goai_nn_forward, the search): the same BatchNorm folding
// and re-layout as goai_load_weights, from the training buffers.
int train_sync_inference(goai_ctx* c)
{
    Train& t = c->train;
    GOAI_REQUIRE(t.ready, "call goai_train_init first");
    GOAI_REQUIRE(c->net.kind == t.kind && c->net.layers == t.layers, "goai_nn_init of the same net kind / n_blocks is needed first");
    GOAI_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<float> hp(t.n_params), hs(t.n_stats);
    GOAI_HIP_CHECK(hipMemcpy(hp.data(), t.params, hp.size() * sizeof(float), hipMemcpyDeviceToHost));
    GOAI_HIP_CHECK(hipMemcpy(hs.data(), t.stats, hs.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<const char*> names;
    std::vector<const float*> ptrs;
    std::vector<int64_t> numels;
    for (const std::string& k : t.order) {
        const TrainSlot& s = t.slots.at(k);
        names.push_back(k.c_str());
        ptrs.push_back((s.stat ? hs.data() : hp.data()) + s.off);
        numels.push_back((int64_t)s.numel);
    }
    return nn_load_weights(c, names.data(), ptrs.data(), numels.data(), (int)names.size());
}

}  // namespace goai
