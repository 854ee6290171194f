// TEST INFRASTRUCTURE ONLY -- CPU oracle (see rng.hpp header).
//
// CPU restatement of the policy/value heads in eval mode (BatchNorm running stats, dropout off),
// i.e. what the reference exports to ONNX: `model.predict(obs)` (scripts/export_onnx.py:31-52).
//   PyRatMLP.predict      alpharat/nn/models/mlp.py:120-153
//   SymmetricMLP.predict  alpharat/nn/models/symmetric.py:124-229
//   PyRatCNN.predict      alpharat/nn/models/cnn/model.py:123-230
//   ResBlock / GPoolResBlock        alpharat/nn/models/cnn/blocks.py:10-79
//   MLPPolicyHead / PointValueHead  alpharat/nn/models/cnn/heads.py:10-38
// Pinned against golden vectors produced by importing the reference's Python classes in the
// build container (tests/golden/nets/, generator tools/gen_net_golden.py).
// Sums are accumulated in double and rounded to f32 per layer output.
//
// Weight blob ("ARNET001"): u32 arch (0 mlp,1 symmetric,2 cnn), u32 width, u32 height,
// u32 n_tensors, then per tensor: u32 name_len, name bytes, u32 ndim, u32 dims[ndim], f32 data.
// Names are the torch state_dict keys.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace oracle {

struct Tensor {
    std::vector<uint32_t> dims;
    std::vector<float> data;
};

struct NetBlob {
    uint32_t arch = 0, width = 0, height = 0;
    std::map<std::string, Tensor> t;

    bool load(const char* path, std::string& err) {
        FILE* f = std::fopen(path, "rb");
        if (!f) {
            err = std::string("cannot open ") + path;
            return false;
        }
        char magic[8];
        uint32_t n = 0;
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "ARNET001", 8) == 0 &&
                  std::fread(&arch, 4, 1, f) == 1 && std::fread(&width, 4, 1, f) == 1 &&
                  std::fread(&height, 4, 1, f) == 1 && std::fread(&n, 4, 1, f) == 1;
        for (uint32_t i = 0; ok && i < n; ++i) {
            uint32_t nl = 0, nd = 0;
            ok = std::fread(&nl, 4, 1, f) == 1 && nl < 4096;
            std::string name(nl, '\0');
            ok = ok && std::fread(&name[0], 1, nl, f) == nl && std::fread(&nd, 4, 1, f) == 1 && nd <= 8;
            Tensor ten;
            size_t count = 1;
            for (uint32_t d = 0; ok && d < nd; ++d) {
                uint32_t v = 0;
                ok = std::fread(&v, 4, 1, f) == 1;
                ten.dims.push_back(v);
                count *= v;
            }
            if (ok) {
                ten.data.resize(count);
                ok = std::fread(ten.data.data(), 4, count, f) == count;
            }
            if (ok) t[name] = std::move(ten);
        }
        std::fclose(f);
        if (!ok) err = std::string("malformed weight blob ") + path;
        return ok;
    }
    const Tensor* get(const std::string& name) const {
        auto it = t.find(name);
        return it == t.end() ? nullptr : &it->second;
    }
    bool has(const std::string& name) const { return t.count(name) != 0; }
};

struct NetOut {
    float logits_p1[5], logits_p2[5];
    float policy_p1[5], policy_p2[5];
    float value_p1, value_p2;
};

namespace netops {

inline void linear(const Tensor& w, const Tensor* b, const float* x, float* y) {
    uint32_t out = w.dims[0], in = w.dims[1];
    for (uint32_t o = 0; o < out; ++o) {
        double acc = b ? (double)b->data[o] : 0.0;
        const float* row = &w.data[(size_t)o * in];
        for (uint32_t i = 0; i < in; ++i) acc += (double)row[i] * (double)x[i];
        y[o] = (float)acc;
    }
}
// BatchNorm in eval mode over `c` channels, `hw` positions each (hw = 1 for BatchNorm1d)
inline void bn_eval(const NetBlob& nb, const std::string& p, float* x, uint32_t c, uint32_t hw) {
    const Tensor& g = *nb.get(p + ".weight");
    const Tensor& be = *nb.get(p + ".bias");
    const Tensor& m = *nb.get(p + ".running_mean");
    const Tensor& v = *nb.get(p + ".running_var");
    for (uint32_t ch = 0; ch < c; ++ch) {
        double inv = 1.0 / std::sqrt((double)v.data[ch] + 1e-5);
        for (uint32_t i = 0; i < hw; ++i) {
            double xv = x[(size_t)ch * hw + i];
            x[(size_t)ch * hw + i] = (float)((xv - (double)m.data[ch]) * inv * (double)g.data[ch] + (double)be.data[ch]);
        }
    }
}
inline void relu(float* x, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.0f;
}
inline void softmax5(const float* l, float* p) {
    float mx = l[0];
    for (int i = 1; i < 5; ++i) mx = l[i] > mx ? l[i] : mx;
    double e[5], s = 0.0;
    for (int i = 0; i < 5; ++i) {
        e[i] = std::exp((double)l[i] - (double)mx);
        s += e[i];
    }
    for (int i = 0; i < 5; ++i) p[i] = (float)(e[i] / s);
}
inline float softplus(float x) {  // torch: beta=1, threshold=20
    return x > 20.0f ? x : (float)std::log1p(std::exp((double)x));
}
// 3x3 (pad 1) or 1x1 convolution, no bias, NCHW single sample. w dims [co, ci, k, k]
inline void conv2d(const Tensor& w, const float* x, float* y, int h, int wd) {
    int co = (int)w.dims[0], ci = (int)w.dims[1], k = (int)w.dims[2];
    int pad = k / 2;
    for (int o = 0; o < co; ++o)
        for (int yy = 0; yy < h; ++yy)
            for (int xx = 0; xx < wd; ++xx) {
                double acc = 0.0;
                for (int c = 0; c < ci; ++c)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx) {
                            int sy = yy + ky - pad, sx = xx + kx - pad;
                            if (sy < 0 || sy >= h || sx < 0 || sx >= wd) continue;
                            acc += (double)w.data[(((size_t)o * ci + c) * k + ky) * k + kx] *
                                   (double)x[((size_t)c * h + sy) * wd + sx];
                        }
                y[((size_t)o * h + yy) * wd + xx] = (float)acc;
            }
}

}  // namespace netops

// observation layout (alpharat/nn/builders/flat.py:33-86): maze[hw*4] p1[hw] p2[hw] cheese[hw]
// scalars: 0 score_diff, 1 progress, 2 p1_mud, 3 p2_mud, 4 p1_score, 5 p2_score
inline bool net_forward(const NetBlob& nb, const float* obs, NetOut& out, std::string& err) {
    using namespace netops;
    const int hw = (int)(nb.width * nb.height);
    const float* maze = obs;
    const float* p1pos = obs + hw * 4;
    const float* p2pos = obs + hw * 5;
    const float* cheese = obs + hw * 6;
    const float* sc = obs + hw * 7;
    auto need = [&](const char* k) {
        if (!nb.has(k)) {
            err = std::string("weight blob lacks ") + k;
            return false;
        }
        return true;
    };
    if (nb.arch == 0) {  // ---- PyRatMLP
        if (!need("trunk.0.weight") || !need("trunk.4.weight") || !need("value_head.weight")) return false;
        uint32_t hd = nb.get("trunk.0.weight")->dims[0];
        std::vector<float> a(hd), b(hd);
        linear(*nb.get("trunk.0.weight"), nb.get("trunk.0.bias"), obs, a.data());
        bn_eval(nb, "trunk.1", a.data(), hd, 1);
        relu(a.data(), hd);
        linear(*nb.get("trunk.4.weight"), nb.get("trunk.4.bias"), a.data(), b.data());
        bn_eval(nb, "trunk.5", b.data(), hd, 1);
        relu(b.data(), hd);
        linear(*nb.get("policy_p1_head.weight"), nb.get("policy_p1_head.bias"), b.data(), out.logits_p1);
        linear(*nb.get("policy_p2_head.weight"), nb.get("policy_p2_head.bias"), b.data(), out.logits_p2);
        float v[2];
        linear(*nb.get("value_head.weight"), nb.get("value_head.bias"), b.data(), v);
        out.value_p1 = softplus(v[0]);
        out.value_p2 = softplus(v[1]);
    } else if (nb.arch == 1) {  // ---- SymmetricMLP
        if (!need("shared_encoder.0.weight") || !need("player_encoder.0.weight") || !need("trunk.0.weight"))
            return false;
        uint32_t hd = nb.get("shared_encoder.0.weight")->dims[0];
        std::vector<float> shared_raw((size_t)hw * 5 + 1), praw[2];
        std::memcpy(shared_raw.data(), maze, sizeof(float) * hw * 4);
        std::memcpy(shared_raw.data() + hw * 4, cheese, sizeof(float) * hw);
        shared_raw[(size_t)hw * 5] = sc[1];
        for (int p = 0; p < 2; ++p) {
            praw[p].resize((size_t)hw + 2);
            std::memcpy(praw[p].data(), p == 0 ? p1pos : p2pos, sizeof(float) * hw);
            praw[p][hw] = sc[2 + p];      // mud
            praw[p][hw + 1] = sc[4 + p];  // score
        }
        std::vector<float> shared(hd), h[2], cat((size_t)hd * 2), t1(hd);
        linear(*nb.get("shared_encoder.0.weight"), nb.get("shared_encoder.0.bias"), shared_raw.data(), shared.data());
        bn_eval(nb, "shared_encoder.1", shared.data(), hd, 1);
        relu(shared.data(), hd);
        for (int p = 0; p < 2; ++p) {
            std::vector<float> pe(hd);
            linear(*nb.get("player_encoder.0.weight"), nb.get("player_encoder.0.bias"), praw[p].data(), pe.data());
            bn_eval(nb, "player_encoder.1", pe.data(), hd, 1);
            relu(pe.data(), hd);
            std::memcpy(cat.data(), shared.data(), sizeof(float) * hd);
            std::memcpy(cat.data() + hd, pe.data(), sizeof(float) * hd);
            linear(*nb.get("trunk.0.weight"), nb.get("trunk.0.bias"), cat.data(), t1.data());
            bn_eval(nb, "trunk.1", t1.data(), hd, 1);
            relu(t1.data(), hd);
            h[p].resize(hd);
            linear(*nb.get("trunk.4.weight"), nb.get("trunk.4.bias"), t1.data(), h[p].data());
            bn_eval(nb, "trunk.5", h[p].data(), hd, 1);
            relu(h[p].data(), hd);
        }
        std::vector<float> agg(hd);
        for (uint32_t i = 0; i < hd; ++i) agg[i] = h[0][i] + h[1][i];
        for (int p = 0; p < 2; ++p) {
            std::memcpy(cat.data(), h[p].data(), sizeof(float) * hd);
            std::memcpy(cat.data() + hd, agg.data(), sizeof(float) * hd);
            linear(*nb.get("policy_head.weight"), nb.get("policy_head.bias"), cat.data(),
                   p == 0 ? out.logits_p1 : out.logits_p2);
            float v;
            linear(*nb.get("value_head.weight"), nb.get("value_head.bias"), cat.data(), &v);
            (p == 0 ? out.value_p1 : out.value_p2) = softplus(v);
        }
    } else if (nb.arch == 2) {  // ---- PyRatCNN
        if (!need("stem.weight") || !need("combiner.0.weight") || !need("policy_head.linear.weight")) return false;
        const bool pooled_value = nb.has("value_head.mlp.0.weight");  // PooledValueHead (cnn/heads.py:40-68)
        if (!pooled_value && !nb.has("value_head.linear.weight")) {
            err = "oracle: unknown value head";
            return false;
        }
        const int H = (int)nb.height, W = (int)nb.width;
        const int C = (int)nb.get("stem.weight")->dims[0];
        std::vector<float> spatial((size_t)5 * hw);
        for (int c = 0; c < 4; ++c)
            for (int i = 0; i < hw; ++i) spatial[(size_t)c * hw + i] = maze[(size_t)i * 4 + c];
        for (int i = 0; i < hw; ++i) spatial[(size_t)4 * hw + i] = cheese[i];
        std::vector<float> feat((size_t)C * hw), t1((size_t)C * hw), t2((size_t)C * hw);
        conv2d(*nb.get("stem.weight"), spatial.data(), feat.data(), H, W);
        bn_eval(nb, "stem_bn", feat.data(), C, hw);
        relu(feat.data(), feat.size());
        for (int bi = 0;; ++bi) {
            std::string p = "blocks." + std::to_string(bi);
            if (!nb.has(p + ".conv1.weight")) break;
            t1 = feat;
            bn_eval(nb, p + ".bn1", t1.data(), C, hw);
            relu(t1.data(), t1.size());
            conv2d(*nb.get(p + ".conv1.weight"), t1.data(), t2.data(), H, W);
            bn_eval(nb, p + ".bn2", t2.data(), C, hw);
            relu(t2.data(), t2.size());
            conv2d(*nb.get(p + ".conv2.weight"), t2.data(), t1.data(), H, W);  // t1 = regular
            if (nb.has(p + ".pool_conv.weight")) {
                const int G = (int)nb.get(p + ".pool_conv.weight")->dims[0];
                std::vector<float> pin = feat, pc((size_t)G * hw), pcat((size_t)2 * G), pout(C);
                bn_eval(nb, p + ".pool_bn", pin.data(), C, hw);
                relu(pin.data(), pin.size());
                conv2d(*nb.get(p + ".pool_conv.weight"), pin.data(), pc.data(), H, W);
                for (int g = 0; g < G; ++g) {
                    double s = 0.0;
                    float mx = pc[(size_t)g * hw];
                    for (int i = 0; i < hw; ++i) {
                        s += pc[(size_t)g * hw + i];
                        mx = pc[(size_t)g * hw + i] > mx ? pc[(size_t)g * hw + i] : mx;
                    }
                    pcat[g] = (float)(s / hw);
                    pcat[G + g] = mx;
                }
                linear(*nb.get(p + ".pool_linear.weight"), nb.get(p + ".pool_linear.bias"), pcat.data(), pout.data());
                for (int c = 0; c < C; ++c)
                    for (int i = 0; i < hw; ++i)
                        feat[(size_t)c * hw + i] = t1[(size_t)c * hw + i] + pout[c] + feat[(size_t)c * hw + i];
            } else {
                for (size_t i = 0; i < feat.size(); ++i) feat[i] = t1[i] + feat[i];
            }
        }
        const int PD = (int)nb.get("player_encoder.0.weight")->dims[0];
        const int HD = (int)nb.get("combiner.0.weight")->dims[0];
        std::vector<float> h[2];
        for (int p = 0; p < 2; ++p) {
            const float* mask = p == 0 ? p1pos : p2pos;
            std::vector<float> cat((size_t)C + PD);
            for (int c = 0; c < C; ++c) {
                double s = 0.0;
                for (int i = 0; i < hw; ++i) s += (double)feat[(size_t)c * hw + i] * (double)mask[i];
                cat[c] = (float)s;
            }
            float side[3] = {sc[4 + p], sc[2 + p], sc[1]};  // [score, mud, progress]
            linear(*nb.get("player_encoder.0.weight"), nb.get("player_encoder.0.bias"), side, cat.data() + C);
            relu(cat.data() + C, PD);
            h[p].resize(HD);
            linear(*nb.get("combiner.0.weight"), nb.get("combiner.0.bias"), cat.data(), h[p].data());
            relu(h[p].data(), HD);
        }
        std::vector<float> cat2((size_t)HD * 2);
        for (int p = 0; p < 2; ++p) {
            for (int i = 0; i < HD; ++i) {
                cat2[i] = h[p][i];
                cat2[HD + i] = h[0][i] + h[1][i];
            }
            linear(*nb.get("policy_head.linear.weight"), nb.get("policy_head.linear.bias"), cat2.data(),
                   p == 0 ? out.logits_p1 : out.logits_p2);
            float v;
            if (pooled_value) {
                // cat([mean, max over the board of the trunk's output, h_i, agg]) -> Linear -> ReLU -> Linear -> softplus
                const int HH = (int)nb.get("value_head.mlp.0.weight")->dims[0];
                std::vector<float> vin((size_t)2 * C + 2 * HD), vh(HH);
                for (int c = 0; c < C; ++c) {
                    double s = 0.0;
                    float mx = feat[(size_t)c * hw];
                    for (int i = 0; i < hw; ++i) {
                        s += feat[(size_t)c * hw + i];
                        mx = feat[(size_t)c * hw + i] > mx ? feat[(size_t)c * hw + i] : mx;
                    }
                    vin[c] = (float)(s / hw);
                    vin[C + c] = mx;
                }
                for (int i = 0; i < 2 * HD; ++i) vin[2 * C + i] = cat2[i];
                linear(*nb.get("value_head.mlp.0.weight"), nb.get("value_head.mlp.0.bias"), vin.data(), vh.data());
                relu(vh.data(), HH);
                linear(*nb.get("value_head.mlp.2.weight"), nb.get("value_head.mlp.2.bias"), vh.data(), &v);
            } else {
                linear(*nb.get("value_head.linear.weight"), nb.get("value_head.linear.bias"), cat2.data(), &v);
            }
            (p == 0 ? out.value_p1 : out.value_p2) = softplus(v);
        }
    } else {
        err = "unknown architecture id in weight blob";
        return false;
    }
    netops::softmax5(out.logits_p1, out.policy_p1);
    netops::softmax5(out.logits_p2, out.policy_p2);
    return true;
}

}  // namespace oracle
