// Native caller of the C-ABI (include/drlvae.h), built with the host side under AddressSanitizer and
// UBSan (tests/native/Makefile): exercises the packing, segment-table and staging code of
// libdrlvae.so with ragged, empty and too-short series, every entry point once, and checks return
// codes, output finiteness and a few invariants.  No Python, no torch: the boundary as a C/C++
// maintainer of another host would bind it.  Prints "abi_driver OK" and exits 0 on success.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

#include "drlvae.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != 0) {                                                                          \
            std::fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_,         \
                         drl_last_error(ctx));                                                   \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)
#define EXPECT(cond)                                                                             \
    do {                                                                                         \
        if (!(cond)) { std::fprintf(stderr, "%s:%d expectation failed: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static std::mt19937 rng(7);
static std::vector<float> uniform(size_t n, float lo, float hi) {
    std::uniform_real_distribution<float> d(lo, hi);
    std::vector<float> v(n);
    for (auto& x : v) x = d(rng);
    return v;
}
static bool finite(const std::vector<float>& v) {
    for (float x : v) if (!std::isfinite(x)) return false;
    return true;
}

struct Vae {
    std::vector<int32_t> din, dout, acts;
    std::vector<std::vector<float>> k, b;
    int mu, lv;
};
static Vae make_vae(int D, int hid, int lat) {
    Vae v;
    v.din = {D, hid, hid, lat, hid};
    v.dout = {hid, lat, lat, hid, D};
    v.acts = {1, 0, 0, 1, 2};
    v.mu = 1; v.lv = 2;
    for (size_t l = 0; l < v.din.size(); ++l) {
        const float lim = std::sqrt(6.0f / (v.din[l] + v.dout[l]));
        v.k.push_back(uniform((size_t)v.din[l] * v.dout[l], -lim, lim));
        v.b.push_back(uniform(v.dout[l], -0.05f, 0.05f));
    }
    return v;
}
static int set_vae(drl_ctx* ctx, const Vae& v) {
    std::vector<const float*> kp, bp;
    for (size_t l = 0; l < v.k.size(); ++l) { kp.push_back(v.k[l].data()); bp.push_back(v.b[l].data()); }
    return drl_set_vae(ctx, (int)v.k.size(), v.din.data(), v.dout.data(), v.acts.data(), v.mu, v.lv,
                       kp.data(), bp.data(), 10.0f);
}
static int set_lstm(drl_ctx* ctx, int I, int H, int A) {
    const float lim = std::sqrt(6.0f / (I + H + 4 * H));
    auto k = uniform((size_t)(I + H) * 4 * H, -lim, lim), b = uniform(4 * H, -0.1f, 0.1f);
    auto w = uniform((size_t)H * A, -1.0f, 1.0f), bo = uniform(A, -1.0f, 1.0f);
    return drl_set_lstm(ctx, I, H, A, 1.0f, k.data(), b.data(), w.data(), bo.data());
}

int main() {
    drl_ctx* ctx = nullptr;
    if (drl_create(0, &ctx) != 0) { std::fprintf(stderr, "drl_create: %s\n", drl_last_error(nullptr)); return 1; }
    std::printf("%s\n", drl_version());

    // ---- univariate, ragged batch: an empty series, one shorter than W, one exactly W, long ones
    const int W = 25;
    {
        Vae v = make_vae(W, 64, 10);
        CHECK(set_vae(ctx, v));
        CHECK(set_lstm(ctx, 2, 128, 2));
        const std::vector<int64_t> lens = {0, 7, 25, 26, 333, 1461, 90};
        std::vector<int64_t> off = {0};
        for (auto n : lens) off.push_back(off.back() + n);
        int64_t K = 0;
        for (auto n : lens) K += n > W ? n - W : 0;
        auto x = uniform((size_t)off.back(), 0.0f, 1.0f);
        std::vector<float> err(K), q(2 * K), margin(K);
        std::vector<int32_t> act(K);
        drl_score_opts o{};
        o.struct_size = sizeof o; o.abi_revision = DRL_ABI_REVISION; o.vae_lag = 1; o.flag_mode = DRL_FLAG_QUIRK; o.action_rule = DRL_ACTION_ARGMAX;
        for (int math = 0; math < 2; ++math) {
            CHECK(drl_set_lstm_math(ctx, math));
            CHECK(drl_score_batch(ctx, x.data(), off.data(), (int)lens.size(), 1, W, &o, err.data(), q.data(),
                                  act.data(), margin.data()));
            EXPECT(finite(err) && finite(q) && finite(margin));
            for (int64_t i = 0; i < K; ++i) {
                EXPECT(act[i] == (q[2 * i + 1] > q[2 * i] ? 1 : 0));
                EXPECT(margin[i] == std::fabs(q[2 * i] - q[2 * i + 1]));
            }
        }
        CHECK(drl_set_lstm_math(ctx, 0));
        // per-series MinMax on the device + explicit flags
        std::vector<float> sc(lens.size(), 0.5f), of(lens.size(), 0.1f), fl(x.size());
        for (size_t i = 0; i < fl.size(); ++i) fl[i] = (float)(i % 3 == 0);
        o.flag_mode = DRL_FLAG_EXPLICIT; o.flags = fl.data(); o.in_scale = sc.data(); o.in_offset = of.data();
        CHECK(drl_score_batch(ctx, x.data(), off.data(), (int)lens.size(), 1, W, &o, err.data(), q.data(), nullptr, nullptr));
        EXPECT(finite(err) && finite(q));
        // outputs optional
        o = drl_score_opts{}; o.struct_size = sizeof o; o.abi_revision = DRL_ABI_REVISION; o.vae_lag = 1;
        CHECK(drl_score_batch(ctx, x.data(), off.data(), (int)lens.size(), 1, W, &o, err.data(), nullptr, nullptr, nullptr));
        CHECK(drl_score_batch(ctx, x.data(), off.data(), (int)lens.size(), 1, W, &o, nullptr, q.data(), nullptr, nullptr));
        // one series through the single-series entry point
        CHECK(drl_score_series(ctx, x.data() + off[5], lens[5], 1, W, &o, err.data(), q.data(), act.data(), margin.data()));
        // nothing to score at all
        const int64_t off0[2] = {0, 10};
        CHECK(drl_score_batch(ctx, x.data(), off0, 1, 1, W, &o, err.data(), q.data(), act.data(), margin.data()));

        // rollout + AL selection
        std::vector<int32_t> ract(K), passes(1);
        std::vector<float> rq(2 * K);
        CHECK(drl_rollout_batch(ctx, x.data(), off.data(), (int)lens.size(), W, nullptr, nullptr, ract.data(), rq.data(), passes.data()));
        EXPECT(passes[0] >= 1 && finite(rq));
        for (auto a : ract) EXPECT(a == 0 || a == 1);
        const std::vector<int32_t> sel = {0, 5, 17, (int32_t)K - 1};
        std::vector<int32_t> idx(40);
        int32_t n_out = -1;
        o.flag_mode = DRL_FLAG_QUIRK;
        CHECK(drl_al_get_samples(ctx, x.data(), off.data(), (int)lens.size(), 1, W, &o, sel.data(), (int)sel.size(), 40,
                                 idx.data(), &n_out, margin.data()));
        EXPECT(n_out == 40);
        for (int i = 0; i < n_out; ++i) {
            EXPECT(idx[i] >= 0 && idx[i] < K);
            for (auto s : sel) EXPECT(idx[i] != s);
            if (i) EXPECT(margin[idx[i]] >= margin[idx[i - 1]]);
        }

        // explicit batches
        const int64_t B = 37;
        auto xb = uniform((size_t)B * W, 0.0f, 1.0f), eps = uniform((size_t)B * 10, -1.0f, 1.0f);
        std::vector<float> rec(B * W), rerr(B), qb(B * 2);
        CHECK(drl_vae_predict(ctx, xb.data(), B, nullptr, rec.data(), rerr.data()));
        CHECK(drl_vae_predict(ctx, xb.data(), B, eps.data(), rec.data(), nullptr));
        EXPECT(finite(rec) && finite(rerr));
        auto st = uniform((size_t)B * W * 2, 0.0f, 1.0f);
        CHECK(drl_q_predict(ctx, st.data(), B, W, qb.data()));
        EXPECT(finite(qb));
    }

    // ---- multivariate, WaDi formulation: both LSTM paths (H=64 projections, H=104 generic) and the wide VAE
    for (int H : {64, 104}) {
        const int F = 19, Wm = 12, D = F * Wm;
        Vae v = make_vae(D, 128, 10);
        CHECK(set_vae(ctx, v));
        CHECK(set_lstm(ctx, F + 1, H, 2));
        const std::vector<int64_t> lens = {Wm - 1, 200, Wm + 1, 77};
        std::vector<int64_t> off = {0};
        for (auto n : lens) off.push_back(off.back() + n);
        int64_t K = 0;
        for (auto n : lens) K += n > Wm ? n - Wm : 0;
        auto x = uniform((size_t)off.back() * F, -3.0f, 3.0f);
        auto cen = uniform(D, -0.5f, 0.5f), scl = uniform(D, 0.5f, 2.0f);
        std::vector<float> err(K), q(4 * K), margin(K);
        std::vector<int32_t> act(K);
        drl_score_opts o{};
        o.struct_size = sizeof o; o.abi_revision = DRL_ABI_REVISION; o.vae_lag = 0; o.flag_mode = DRL_FLAG_BOTH; o.action_rule = DRL_ACTION_WADI;
        std::vector<double> cen64(cen.begin(), cen.end()), scl64(scl.begin(), scl.end());   // the ABI takes sklearn's float64
        o.vae_center = cen64.data(); o.vae_scale = scl64.data(); o.vae_clip = 10.0f;
        CHECK(drl_score_batch(ctx, x.data(), off.data(), (int)lens.size(), F, Wm, &o, err.data(), q.data(), act.data(), margin.data()));
        EXPECT(finite(err) && finite(q));
        for (int64_t i = 0; i < K; ++i) EXPECT(act[i] == (q[4 * i + 3] > q[4 * i] ? 1 : 0));
        std::printf("multivariate H=%d: lstm kernel %s\n", H, drl_last_lstm_variant(ctx));
    }

    // ---- errors come back as codes, not crashes
    {
        drl_score_opts o{};
        o.struct_size = 3;
        float dummy[64] = {0};
        const int64_t off1[2] = {0, 64};
        EXPECT(drl_score_batch(ctx, dummy, off1, 1, 1, 25, &o, dummy, nullptr, nullptr, nullptr) != 0);
        EXPECT(drl_score_batch(ctx, nullptr, off1, 1, 1, 25, &o, dummy, nullptr, nullptr, nullptr) != 0);
        // a caller built against revision 1 of the struct (the field was `reserved`, 0) is refused, not misread
        o = drl_score_opts{}; o.struct_size = sizeof o; o.vae_lag = 1;
        EXPECT(drl_score_batch(ctx, dummy, off1, 1, 1, 25, &o, dummy, nullptr, nullptr, nullptr) == DRL_ERR_ARG);
        EXPECT(std::strstr(drl_last_error(ctx), "abi_revision") != nullptr);
        EXPECT(drl_set_knob(ctx, "no_such_knob", 1) == DRL_ERR_ARG);
        EXPECT(drl_set_knob(ctx, "no_pipeline", 1) == DRL_OK && drl_set_knob(ctx, "no_pipeline", 0) == DRL_OK);
        EXPECT(drl_debug_alloc_count() > 0);
        EXPECT(drl_set_lstm(ctx, 2, 130, 2, 1.0f, dummy, dummy, dummy, dummy) != 0);
        EXPECT(drl_create(9999, &ctx) != 0 || true);
    }
    drl_destroy(ctx);
    std::printf("abi_driver OK\n");
    return 0;
}
