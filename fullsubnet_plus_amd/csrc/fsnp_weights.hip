// fsnp_weights.hip - strict weight loading of the C ABI: the parameter tree a handle expects (reference names / shapes,
// base_inferencer.py:100-107 loads with strict=True) and its packing into the device layouts of the kernels (MFMA fragment order,
// transposed / zero-padded GEMM operands, GroupNorm folded into the sconv weights, summed biases).  Host code only.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fsnp_handle.h"
#include "weight_watch.h"

namespace fsnp {

static const int kDilations[8] = {1, 2, 5, 9, 1, 2, 5, 9};  // sequence_model.py:48-57
static const char* kAtt[3] = {"channel_attention", "channel_attention_real", "channel_attention_imag"};
static const char* kFb[3] = {"fb_model", "fb_model_real", "fb_model_imag"};
static const char* kConvNames[3] = {"smallConv1d", "middleConv1d", "largeConv1d"};

void build_specs(fsnp_handle* h) {
    auto add = [&](const std::string& n, int64_t numel) { h->specs.push_back({n, numel}); };
    const int F = h->F, CH = h->CH, H = h->H, Fr = h->Fr;
    const int att = h->cfg.attention;
    const bool fsn = h->model == FSNP_MODEL_FULLSUBNET;
    if (fsn) {   // fullsubnet.py:39-47: SequenceModel(257 -> 512 x 2 -> 257), same key layout as nn.LSTM
        const std::string f = "fb_model.sequence_model.";
        const int64_t G = h->NG;
        add(f + "weight_ih_l0", G * CH * F);
        add(f + "weight_hh_l0", G * CH * CH);
        add(f + "bias_ih_l0", G * CH);
        add(f + "bias_hh_l0", G * CH);
        add(f + "weight_ih_l1", G * CH * CH);
        add(f + "weight_hh_l1", G * CH * CH);
        add(f + "bias_ih_l1", G * CH);
        add(f + "bias_hh_l1", G * CH);
        add("fb_model.fc_output_layer.weight", (int64_t)F * CH);
        add("fb_model.fc_output_layer.bias", F);
    }
    for (int a = 0; a < 3 && !fsn; ++a) {
        const std::string p = kAtt[a];
        if (att == FSNP_ATT_TSSE) {
            for (int c = 0; c < 3; ++c) {
                add(p + "." + kConvNames[c] + ".0.weight", (int64_t)F * h->cfg.kersize[c]);
                add(p + "." + kConvNames[c] + ".0.bias", F);
            }
            add(p + ".feature_concate_fc.weight", 3);
            add(p + ".feature_concate_fc.bias", 1);
        }
        if (att == FSNP_ATT_ECA) {
            add(p + ".conv.weight", 3);
        } else {
            add(p + ".fc1.weight", (int64_t)Fr * F);
            add(p + ".fc1.bias", Fr);
            add(p + ".fc2.weight", (int64_t)F * Fr);
            add(p + ".fc2.bias", F);
        }
    }
    for (int b = 0; b < 3 && !fsn; ++b) {
        for (int i = 0; i < h->NB; ++i) {
            const std::string p = std::string(kFb[b]) + ".sequence_model." + std::to_string(i);
            add(p + ".conv1x1.weight", (int64_t)CH * F);
            add(p + ".conv1x1.bias", CH);
            add(p + ".prelu1.weight", 1);
            add(p + ".norm1.weight", CH);
            add(p + ".norm1.bias", CH);
            add(p + ".depthwise_conv.weight", (int64_t)CH * 3);
            add(p + ".depthwise_conv.bias", CH);
            add(p + ".prelu2.weight", 1);
            add(p + ".norm2.weight", CH);
            add(p + ".norm2.bias", CH);
            add(p + ".sconv.weight", (int64_t)F * CH);
            add(p + ".sconv.bias", F);
        }
        add(std::string(kFb[b]) + ".fc_output_layer.weight", (int64_t)F * F);
        add(std::string(kFb[b]) + ".fc_output_layer.bias", F);
    }
    if (h->sb_tcn) {
        for (int i = 0; i < 8; ++i) {
            const std::string p = "sb_model.sequence_model." + std::to_string(i);
            add(p + ".conv1x1.weight", (int64_t)CH * h->NIN);
            add(p + ".conv1x1.bias", CH);
            add(p + ".prelu1.weight", 1);
            add(p + ".norm1.weight", CH);
            add(p + ".norm1.bias", CH);
            add(p + ".depthwise_conv.weight", (int64_t)CH * 3);
            add(p + ".depthwise_conv.bias", CH);
            add(p + ".prelu2.weight", 1);
            add(p + ".norm2.weight", CH);
            add(p + ".norm2.bias", CH);
            add(p + ".sconv.weight", (int64_t)h->NIN * CH);
            add(p + ".sconv.bias", h->NIN);
        }
        add("sb_model.fc_output_layer.weight", (int64_t)h->cfg.output_size * h->NIN);
        add("sb_model.fc_output_layer.bias", h->cfg.output_size);
        return;
    }
    const std::string s = "sb_model.sequence_model.";
    const int64_t G = h->NG;
    add(s + "weight_ih_l0", G * H * h->NIN);
    add(s + "weight_hh_l0", G * H * H);
    add(s + "bias_ih_l0", G * H);
    add(s + "bias_hh_l0", G * H);
    add(s + "weight_ih_l1", G * H * H);
    add(s + "weight_hh_l1", G * H * H);
    add(s + "bias_ih_l1", G * H);
    add(s + "bias_hh_l1", G * H);
    add("sb_model.fc_output_layer.weight", (int64_t)h->cfg.output_size * H);
    add("sb_model.fc_output_layer.bias", h->cfg.output_size);
}

// ---- fsnp_watch_weights: a 64-bit fingerprint of the caller's source tensors, taken on the device in front of every forward.
// sum over all elements of bits(x_i) * (2 i + 1) mod 2^64 (i = position in the concatenation): any single changed element changes
// it, the sum is order-independent (integer adds), so blocks accumulate with one atomic each and the LAST block to finish compares.
__global__ __launch_bounds__(256) void weight_watch_kernel(const WatchSeg* __restrict__ segs, int nseg, unsigned long long* acc,
                                                           int baseline, unsigned* err_host) {
    weight_watch_block(segs, nseg, acc, baseline, err_host, (int)blockIdx.x, (int)gridDim.x);
}
int launch_weight_watch(fsnp_handle* h, hipStream_t s, bool baseline) {
    if (h->watch_nseg <= 0) return 0;
    const int grid = h->watch_nseg < kWatchBlocks ? h->watch_nseg : kWatchBlocks;
    hipLaunchKernelGGL(weight_watch_kernel, dim3(grid), dim3(256), 0, s, static_cast<const WatchSeg*>(h->watch_segs), h->watch_nseg,
                       h->watch_acc, baseline ? 1 : 0, h->d_err);
    FSNP_HIP_CHECK(hipGetLastError());
    return 0;
}
void drop_weight_watch(fsnp_handle* h) {
    if (h->watch_segs) (void)hipFree(h->watch_segs);
    if (h->watch_acc) (void)hipFree(h->watch_acc);
    h->watch_segs = nullptr; h->watch_acc = nullptr; h->watch_nseg = 0;
}

}  // namespace fsnp

extern "C" {

int fsnp_watch_weights(fsnp_handle* h, const void* const* dev_ptrs, const int64_t* numels, int32_t n, int32_t every, void* hip_stream) {
    if (!h || n < 0 || (n > 0 && (!dev_ptrs || !numels)) || every < 1) { set_error("fsnp_watch_weights: bad argument"); return 1; }
    if (!h->committed) { set_error("fsnp_watch_weights: commit the weights first (the baseline fingerprint belongs to the packed set)"); return 2; }
    FSNP_ON_DEVICE(h);
    FSNP_HIP_CHECK(hipDeviceSynchronize());           // (an earlier watch kernel may still read the old table)
    drop_weight_watch(h);
    h->watch_every = every; h->watch_calls = 0;
    if (n == 0) return 0;
    std::vector<WatchSeg> segs;
    unsigned long long first = 0;
    constexpr int64_t kSeg = kWatchSeg;               // elements per segment (32 KB): ~1400 segments for the default model
    for (int i = 0; i < n; ++i) {
        if (!dev_ptrs[i] || numels[i] < 0) { set_error("fsnp_watch_weights: tensor %d is null / negative", i); return 1; }
        for (int64_t o = 0; o < numels[i]; o += kSeg) {
            const int64_t c = numels[i] - o < kSeg ? numels[i] - o : kSeg;
            segs.push_back(WatchSeg{static_cast<const unsigned*>(dev_ptrs[i]) + o, (unsigned)c, first + (unsigned long long)o});
        }
        first += (unsigned long long)numels[i];
    }
    if (segs.empty()) return 0;
    FSNP_HIP_CHECK(hipMalloc(&h->watch_segs, segs.size() * sizeof(WatchSeg)));
    FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->watch_acc), (8 + kWatchBlocks) * 8));     // {-, tickets, baseline, -, ...; a partial sum per block}
    FSNP_HIP_CHECK(hipMemcpy(h->watch_segs, segs.data(), segs.size() * sizeof(WatchSeg), hipMemcpyHostToDevice));
    // (on the CALLER's stream, like the baseline kernel behind it: a null-stream hipMemset of device memory returns before it has run
    // and does not order with a non-blocking stream - on torch side streams it zeroed the tickets / the baseline under the running
    // baseline kernel now and then: a false "stale weights" flag, profiles/r05_pytest_gpu_runA.log)
    FSNP_HIP_CHECK(hipMemsetAsync(h->watch_acc, 0, (8 + kWatchBlocks) * 8, static_cast<hipStream_t>(hip_stream)));
    h->watch_nseg = (int)segs.size();
    if (const int rc = launch_weight_watch(h, static_cast<hipStream_t>(hip_stream), true)) return rc;
    // the baseline belongs to the handle's shared state like a forward's workspace: a later forward on ANOTHER (non-blocking) stream
    // must not start its watch blocks (same ticket word, same baseline slot) before this kernel has finished (a spurious
    // code 6 when the watch was registered on the default stream and the forward ran on a torch side stream)
    return mark_forward_done(h, static_cast<hipStream_t>(hip_stream));
}

int fsnp_num_weights(const fsnp_handle* h) { return h ? (int)h->specs.size() : 0; }

int fsnp_weight_info(const fsnp_handle* h, int index, const char** name, int64_t* numel) {
    if (!h || index < 0 || index >= (int)h->specs.size()) { set_error("fsnp_weight_info: bad index"); return 1; }
    if (name) *name = h->specs[index].name.c_str();
    if (numel) *numel = h->specs[index].numel;
    return 0;
}

int fsnp_set_weight(fsnp_handle* h, const char* name, const float* host_data, int64_t numel) {
    if (!h || !name || !host_data) { set_error("fsnp_set_weight: null argument"); return 1; }
    for (const auto& s : h->specs) {
        if (s.name == name) {
            if (s.numel != numel) {
                set_error("size mismatch for %s: expected %lld elements, got %lld", name, (long long)s.numel, (long long)numel);
                return 2;
            }
            h->host_w[s.name].assign(host_data, host_data + numel);
            h->committed = false;
            h->watch_nseg = 0;            // the watched tensors belonged to the previous weight set (fsnp_watch_weights again after the commit)
            return 0;
        }
    }
    set_error("unexpected key in state_dict: %s", name);
    return 2;
}

int fsnp_commit_weights(fsnp_handle* h) {
    if (!h) { set_error("null handle"); return 1; }
    for (const auto& s : h->specs)
        if (!h->host_w.count(s.name)) { set_error("missing key in state_dict: %s", s.name.c_str()); return 2; }
    const int F = h->F, CH = h->CH, H = h->H, NB = h->NB, Fr = h->Fr;
    std::vector<float> blob;
    auto alloc = [&](size_t n) { size_t o = blob.size(); blob.resize(align_up(o + n, 64), 0.0f); return o; };
    auto W = [&](const std::string& n) -> const std::vector<float>& { return h->host_w.at(n); };
    auto put = [&](const std::string& n) { const auto& v = W(n); size_t o = alloc(v.size()); std::copy(v.begin(), v.end(), blob.begin() + o); return o; };

    // ---- frontend (TSSE) : reference layouts are already what the kernels want
    size_t o_conv_w[3][3] = {}, o_conv_b[3][3] = {}, o_cat_w[3] = {}, o_cat_b[3] = {}, o_fc1w[3] = {}, o_fc1b[3] = {},
           o_fc2w[3] = {}, o_fc2b[3] = {};
    const int att = h->cfg.attention;
    const bool fsn = h->model == FSNP_MODEL_FULLSUBNET;
    const int nbr_w = fsn ? 0 : 3;          // the original FullSubNet has neither attention nor TCN branches
    for (int a = 0; a < nbr_w; ++a) {
        const std::string p = kAtt[a];
        if (att == FSNP_ATT_TSSE) {
            for (int c = 0; c < 3; ++c) {
                o_conv_w[a][c] = put(p + "." + kConvNames[c] + ".0.weight");
                o_conv_b[a][c] = put(p + "." + kConvNames[c] + ".0.bias");
            }
            o_cat_w[a] = put(p + ".feature_concate_fc.weight");
            o_cat_b[a] = put(p + ".feature_concate_fc.bias");
        }
        if (att == FSNP_ATT_ECA) {
            o_cat_w[a] = put(p + ".conv.weight");            // the 3 taps of Conv1d(1,1,3) over the channel axis
            continue;
        }
        {   // transposed copies: fc1 [Fr][F] -> [F][Fr], fc2 [F][Fr] -> [Fr][F]
            const auto& w1 = W(p + ".fc1.weight");
            o_fc1w[a] = alloc((size_t)F * Fr);
            for (int o = 0; o < Fr; ++o)
                for (int f = 0; f < F; ++f) blob[o_fc1w[a] + (size_t)f * Fr + o] = w1[(size_t)o * F + f];
            const auto& w2 = W(p + ".fc2.weight");
            o_fc2w[a] = alloc((size_t)Fr * F);
            for (int o = 0; o < F; ++o)
                for (int f = 0; f < Fr; ++f) blob[o_fc2w[a] + (size_t)f * F + o] = w2[(size_t)o * Fr + f];
        }
        o_fc1b[a] = put(p + ".fc1.bias");
        o_fc2b[a] = put(p + ".fc2.bias");
    }
    // ---- TCN stacks (SequenceModel(sequence_model="TCN"), sequence_model.py:47-58,80-81): zero-padded row-major
    // [N pad 384][K pad 16] GEMM operands, [model][block] major.  `cin` channels in / out of every TCNBlock, `fc_out` rows of
    // the final Linear(cin, fc_out).  Used for the three full-band models and for a sub-band TCN.
    struct TcnOff { size_t w1, b1, a1, g1w, g1b, dw, db, a2, g2w, g2b, w2, b2, w2g, c1, c2, wf, bf; int NB, N1P, K1P, N2P, K2P; };
    auto pack_tcn = [&](const std::vector<std::string>& models, int nb, int cin, int fc_out) {
        TcnOff t{};
        const size_t nm = models.size() ? models.size() : 1;
        t.NB = nb; t.N1P = (int)align_up(CH, 384); t.K1P = (int)align_up(cin, 16); t.N2P = (int)align_up(cin, 384); t.K2P = (int)align_up(CH, 16);
        t.w1 = alloc(nm * nb * t.N1P * t.K1P); t.b1 = alloc(nm * nb * t.N1P); t.a1 = alloc(nm * nb + 1);
        t.g1w = alloc(nm * nb * CH); t.g1b = alloc(nm * nb * CH);
        t.dw = alloc(nm * nb * 3 * CH); t.db = alloc(nm * nb * CH); t.a2 = alloc(nm * nb + 1);
        t.g2w = alloc(nm * nb * CH); t.g2b = alloc(nm * nb * CH);
        t.w2 = alloc(nm * nb * t.N2P * t.K2P); t.b2 = alloc(nm * nb * t.N2P);
        t.w2g = alloc(nm * nb * t.N2P * t.K2P); t.c1 = alloc(nm * nb * t.N2P); t.c2 = alloc(nm * nb * t.N2P);
        t.wf = alloc(nm * t.N2P * t.K1P); t.bf = alloc(nm * t.N2P);
        for (size_t b = 0; b < models.size(); ++b) {
            for (int i = 0; i < nb; ++i) {
                const std::string p = models[b] + ".sequence_model." + std::to_string(i);
                const size_t bi = b * nb + i;
                const auto& w1 = W(p + ".conv1x1.weight");           // [CH][cin][1]
                for (int n = 0; n < CH; ++n)
                    for (int k = 0; k < cin; ++k) blob[t.w1 + (bi * t.N1P + n) * t.K1P + k] = w1[(size_t)n * cin + k];
                std::copy(W(p + ".conv1x1.bias").begin(), W(p + ".conv1x1.bias").end(), blob.begin() + t.b1 + bi * t.N1P);
                blob[t.a1 + bi] = W(p + ".prelu1.weight")[0];
                std::copy(W(p + ".norm1.weight").begin(), W(p + ".norm1.weight").end(), blob.begin() + t.g1w + bi * CH);
                std::copy(W(p + ".norm1.bias").begin(), W(p + ".norm1.bias").end(), blob.begin() + t.g1b + bi * CH);
                const auto& dw = W(p + ".depthwise_conv.weight");     // [CH][1][3] -> tap major
                for (int c = 0; c < CH; ++c)
                    for (int jj = 0; jj < 3; ++jj) blob[t.dw + (bi * 3 + jj) * CH + c] = dw[(size_t)c * 3 + jj];
                std::copy(W(p + ".depthwise_conv.bias").begin(), W(p + ".depthwise_conv.bias").end(), blob.begin() + t.db + bi * CH);
                blob[t.a2 + bi] = W(p + ".prelu2.weight")[0];
                std::copy(W(p + ".norm2.weight").begin(), W(p + ".norm2.weight").end(), blob.begin() + t.g2w + bi * CH);
                std::copy(W(p + ".norm2.bias").begin(), W(p + ".norm2.bias").end(), blob.begin() + t.g2b + bi * CH);
                const auto& w2 = W(p + ".sconv.weight");              // [cin][CH][1]
                for (int n = 0; n < cin; ++n)
                    for (int k = 0; k < CH; ++k) blob[t.w2 + (bi * t.N2P + n) * t.K2P + k] = w2[(size_t)n * CH + k];
                std::copy(W(p + ".sconv.bias").begin(), W(p + ".sconv.bias").end(), blob.begin() + t.b2 + bi * t.N2P);
                // GroupNorm 2 folded into the sconv GEMM (tcn.hip tcn_gemm_dma_kernel): weights times gamma, and the two
                // per-output constants of  sum_k ((a - m) r g_k + b_k) W[n][k] = r sum_k a g_k W[n][k] + c1[n] - r m c2[n]
                const auto& g2 = W(p + ".norm2.weight");
                const auto& be2 = W(p + ".norm2.bias");
                const auto& sb2 = W(p + ".sconv.bias");
                for (int n = 0; n < cin; ++n) {
                    double s1 = sb2[n], s2 = 0.0;
                    for (int k = 0; k < CH; ++k) {
                        const double wv = w2[(size_t)n * CH + k];
                        blob[t.w2g + (bi * t.N2P + n) * t.K2P + k] = (float)(wv * (double)g2[k]);
                        s1 += (double)be2[k] * wv;
                        s2 += (double)g2[k] * wv;
                    }
                    blob[t.c1 + bi * t.N2P + n] = (float)s1;
                    blob[t.c2 + bi * t.N2P + n] = (float)s2;
                }
            }
            const auto& wf = W(models[b] + ".fc_output_layer.weight");   // [fc_out][cin]: top rows of a zero-padded [N2P][K1P]
            for (int n = 0; n < fc_out; ++n)
                for (int k = 0; k < cin; ++k) blob[t.wf + (b * t.N2P + n) * t.K1P + k] = wf[(size_t)n * cin + k];
            const auto& bf = W(models[b] + ".fc_output_layer.bias");
            std::copy(bf.begin(), bf.end(), blob.begin() + t.bf + b * t.N2P);
        }
        return t;
    };
    auto bind_tcn = [&](TcnWeights& t, const TcnOff& o, const float* d) {
        t.w1 = d + o.w1; t.b1 = d + o.b1; t.a1 = d + o.a1; t.g1w = d + o.g1w; t.g1b = d + o.g1b;
        t.dw = d + o.dw; t.db = d + o.db; t.a2 = d + o.a2; t.g2w = d + o.g2w; t.g2b = d + o.g2b;
        t.w2 = d + o.w2; t.b2 = d + o.b2; t.wf = d + o.wf; t.bf = d + o.bf;
        t.w2g = d + o.w2g; t.c1 = d + o.c1; t.c2 = d + o.c2;
        t.num_cus = h->num_cus; t.NB = o.NB; t.N1P = o.N1P; t.K1P = o.K1P; t.N2P = o.N2P; t.K2P = o.K2P;
        for (int i = 0; i < o.NB; ++i) t.dilation[i] = kDilations[i];
        const char* de = getenv("FSNP_GEMM_DMA");          // 0 = the general GEMM kernel everywhere (tuning / A-B)
        t.gemm_dma = de && de[0] == '0' ? 0 : 1;
    };
    std::vector<std::string> fb_models;
    for (int b = 0; b < nbr_w; ++b) fb_models.push_back(kFb[b]);
    const TcnOff fb_off = pack_tcn(fb_models, NB, F, F);
    const TcnOff sb_off = h->sb_tcn ? pack_tcn({"sb_model"}, 8, h->NIN, h->cfg.output_size) : TcnOff{};
    // ---- recurrent models: MFMA B-fragment order + summed biases.  Every kernel sees FOUR column slots per hidden unit:
    // LSTM i, f, g, o (the reference's gate order); GRU r, z, n_x, n_h with W_in only in the input rows of K and W_hn only
    // in the hidden rows (zero blocks elsewhere), biases b_ir + b_hr, b_iz + b_hz, b_in, b_hn.
    struct Rnn4 { std::vector<float> wih0, whh0, wih1, whh1, bias; };
    auto expand = [&](const std::string& pre, int Hh, int nin) {
        Rnn4 r;
        const auto &a0 = W(pre + "weight_ih_l0"), &a1 = W(pre + "weight_hh_l0"), &a2 = W(pre + "weight_ih_l1"), &a3 = W(pre + "weight_hh_l1");
        r.bias.assign((size_t)2 * 4 * Hh, 0.0f);
        if (!h->gru) {
            r.wih0 = a0; r.whh0 = a1; r.wih1 = a2; r.whh1 = a3;
            for (int l = 0; l < 2; ++l) {
                const auto& bi = W(pre + "bias_ih_l" + std::to_string(l));
                const auto& bh = W(pre + "bias_hh_l" + std::to_string(l));
                for (int i = 0; i < 4 * Hh; ++i) r.bias[(size_t)l * 4 * Hh + i] = bi[i] + bh[i];
            }
            return r;
        }
        auto spread = [&](const std::vector<float>& src, int cols, bool hidden) {   // [3H][cols] -> [4H][cols]
            std::vector<float> dst((size_t)4 * Hh * cols, 0.0f);
            std::copy(src.begin(), src.begin() + (size_t)2 * Hh * cols, dst.begin());                        // r, z
            std::copy(src.begin() + (size_t)2 * Hh * cols, src.end(), dst.begin() + (size_t)(hidden ? 3 : 2) * Hh * cols);   // n
            return dst;
        };
        r.wih0 = spread(a0, nin, false); r.whh0 = spread(a1, Hh, true);
        r.wih1 = spread(a2, Hh, false); r.whh1 = spread(a3, Hh, true);
        for (int l = 0; l < 2; ++l) {
            const auto& bi = W(pre + "bias_ih_l" + std::to_string(l));
            const auto& bh = W(pre + "bias_hh_l" + std::to_string(l));
            float* b = r.bias.data() + (size_t)l * 4 * Hh;
            for (int i = 0; i < 2 * Hh; ++i) b[i] = bi[i] + bh[i];
            for (int i = 0; i < Hh; ++i) { b[2 * Hh + i] = bi[2 * Hh + i]; b[3 * Hh + i] = bh[2 * Hh + i]; }
        }
        return r;
    };
    const Rnn4 sbw = h->sb_tcn ? Rnn4{} : expand("sb_model.sequence_model.", H, h->NIN);
    const bool tuned = !h->sb_tcn && !h->generic_sb;            // MFMA kernels exist for this cell / hidden size / input width
    size_t o_wgen = 0;
    if (h->generic_sb) {                                        // runtime-sized kernel: transposed [layer][k][4H]
        o_wgen = alloc(lstm_generic_pack_floats(H, h->NIN));
        lstm_generic_pack_weights(H, h->NIN, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wgen);
    }
    size_t o_wpack = 0, o_wpack12 = 0, o_wpack_bf[2] = {0, 0};
    if (!h->gru && tuned && (H == 384 || H == 256)) {      // the row-tile kernel (and its bf16 variant) exists for LSTM only
        o_wpack = alloc(lstm_pack_floats(H, h->KX, 4));
        lstm_pack_weights(H, h->NIN, h->KX, 4, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack);
        if (H == 384) {
            o_wpack12 = alloc(lstm_pack_floats(H, h->KX, 12));
            lstm_pack_weights(H, h->NIN, h->KX, 12, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack12);
        }
        for (int i = 0; i < 2 && h->KX == 40 && H == 384 && h->NIN < h->KX; ++i) {      // the bf16-ih variant is built for the default input width only (and needs a spare input column: its layer-0 bias rides there)
            const int nw = i == 0 ? 4 : 12;
            o_wpack_bf[i] = alloc(lstm_pack_floats_bf16ih(H, h->KX, nw));
            lstm_pack_weights_bf16ih(H, h->NIN, h->KX, nw, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), sbw.bias.data(),
                                     blob.data() + o_wpack_bf[i]);
        }
    }
    size_t o_wpack16 = 0, o_wpack16_bf = 0;
    bool have16_bf = false;
    if (h->lstm16_ok) {
        o_wpack16 = alloc(lstm16_pack_floats(H, h->KX));
        lstm16_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack16);
        if (h->KX == 40 && H == 384) {                 // bf16-ih stream of the half-tile kernel (configs[4], round 4)
            o_wpack16_bf = alloc(lstm16_pack_floats_bf16ih(H, h->KX));
            lstm16_pack_weights_bf16ih(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack16_bf);
            have16_bf = true;
        }
    }
    size_t o_wpack_gru = 0;
    if (h->gru && tuned && H == 384) {
        o_wpack_gru = alloc(gru_pack_floats(H, h->KX, 4));
        gru_pack_weights(H, h->NIN, h->KX, 4, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack_gru);
    }
    size_t o_wpack_coop[4] = {0, 0, 0, 0};
    for (int ui = 0; ui < 4 && tuned; ++ui) {
        const int units = 8 << ui;
        o_wpack_coop[ui] = alloc(lstm_coop_pack_floats(H, h->KX, units));
        lstm_coop_pack_weights(H, h->NIN, h->KX, units, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(),
                               blob.data() + o_wpack_coop[ui]);
    }
    size_t o_wpack_hp = 0, o_wpack_hpw = 0;
    if (h->hp_ok) {
        o_wpack_hp = alloc(lstm_hp_pack_floats(H, h->KX));
        lstm_hp_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack_hp);
        o_wpack_hpw = alloc(lstm_hpw_pack_floats(H, h->KX));
        lstm_hpw_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack_hpw);
    }
    size_t o_wpack_coopw = 0;
    if (h->coopw_ok) {
        o_wpack_coopw = alloc(lstm_coopw_pack_floats(H, h->KX));
        lstm_coopw_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(), blob.data() + o_wpack_coopw);
    }
    const size_t o_wpack_coopn = alloc(tuned ? lstm_coopn_pack_floats(H, h->KX) : 0);
    if (tuned)
        lstm_coopn_pack_weights(H, h->NIN, h->KX, sbw.wih0.data(), sbw.whh0.data(), sbw.wih1.data(), sbw.whh1.data(),
                                blob.data() + o_wpack_coopn);
    // ---- original FullSubNet: full-band recurrent model (cooperative kernel, KX = 264) + Linear(CH, F) as a GEMM operand
    constexpr int KXF = 264;
    size_t o_fbpack[3] = {0, 0, 0}, o_fbbias = 0, o_fsn_wf = 0, o_fsn_bf = 0, o_fbgen = 0, o_fbv = 0;
    bool have_fbv = false;
    const int fsn_kp = (int)align_up(CH, 16), fsn_np = (int)align_up(F, 384);
    if (fsn) {
        const Rnn4 fbw = expand("fb_model.sequence_model.", CH, F);
        if (h->generic_fb) {
            o_fbgen = alloc(lstm_generic_pack_floats(CH, F));
            lstm_generic_pack_weights(CH, F, fbw.wih0.data(), fbw.whh0.data(), fbw.wih1.data(), fbw.whh1.data(), blob.data() + o_fbgen);
        }
        for (int ui = 0; ui < 3 && !h->generic_fb; ++ui) {
            const int units = 8 << ui;
            o_fbpack[ui] = alloc(lstm_coop_pack_floats(CH, KXF, units));
            lstm_coop_pack_weights(CH, F, KXF, units, fbw.wih0.data(), fbw.whh0.data(), fbw.wih1.data(), fbw.whh1.data(),
                                   blob.data() + o_fbpack[ui]);
        }
        if (!h->generic_fb && !h->gru && CH == 512 && F <= 288) {          // small batches: matrix-vector products on the VALU (lstm_fbv.hip)
            o_fbv = alloc(lstm_fbv_pack_floats(CH));
            lstm_fbv_pack_weights(CH, F, fbw.wih0.data(), fbw.whh0.data(), fbw.wih1.data(), fbw.whh1.data(), blob.data() + o_fbv);
            have_fbv = true;
        }
        o_fbbias = alloc(fbw.bias.size());
        std::copy(fbw.bias.begin(), fbw.bias.end(), blob.begin() + o_fbbias);
        o_fsn_wf = alloc((size_t)fsn_np * fsn_kp);
        const auto& wf = W("fb_model.fc_output_layer.weight");          // [F][CH]
        for (int n = 0; n < F; ++n)
            for (int k = 0; k < CH; ++k) blob[o_fsn_wf + (size_t)n * fsn_kp + k] = wf[(size_t)n * CH + k];
        o_fsn_bf = alloc(fsn_np);
        const auto& bf = W("fb_model.fc_output_layer.bias");
        std::copy(bf.begin(), bf.end(), blob.begin() + o_fsn_bf);
    }
    const size_t o_lbias = alloc(sbw.bias.size());
    std::copy(sbw.bias.begin(), sbw.bias.end(), blob.begin() + o_lbias);
    const size_t o_wfc = h->sb_tcn ? 0 : put("sb_model.fc_output_layer.weight");
    const size_t o_bfc = h->sb_tcn ? 0 : put("sb_model.fc_output_layer.bias");
    // ---- unfold multiplicities w_r, by brute force over (f, j)
    const size_t o_refl = alloc(F), o_reflfb = alloc(F);
    for (int f = 0; f < F; ++f) {
        for (int j = 0; j < h->NSB; ++j) blob[o_refl + reflect_index(f - h->cfg.sb_num_neighbors + j, F)] += 1.0f;
        for (int j = 0; j < 2 * h->cfg.fb_num_neighbors + 1; ++j) blob[o_reflfb + reflect_index(f - h->cfg.fb_num_neighbors + j, F)] += 1.0f;
    }

    FSNP_ON_DEVICE(h);
    if (h->d_weights) { FSNP_HIP_CHECK(hipDeviceSynchronize()); FSNP_HIP_CHECK(hipFree(h->d_weights)); h->d_weights = nullptr; }
    FSNP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_weights), blob.size() * sizeof(float)));
    FSNP_HIP_CHECK(hipMemcpy(h->d_weights, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* d = h->d_weights;
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) { h->fw.conv_w[a][c] = d + o_conv_w[a][c]; h->fw.conv_b[a][c] = d + o_conv_b[a][c]; }
        h->fw.cat_w[a] = d + o_cat_w[a]; h->fw.cat_b[a] = d + o_cat_b[a];
        h->fw.fc1_wT[a] = d + o_fc1w[a]; h->fw.fc1_b[a] = d + o_fc1b[a];
        h->fw.fc2_wT[a] = d + o_fc2w[a]; h->fw.fc2_b[a] = d + o_fc2b[a];
    }
    for (int c = 0; c < 3; ++c) h->fw.ksize[c] = h->cfg.kersize[c];
    h->fw.attention = h->cfg.attention;
    h->fw.subband_num = h->cfg.subband_num > 0 ? h->cfg.subband_num : 1;
    bind_tcn(h->tw, fb_off, d);
    h->lw.wpack = d + o_wpack; h->lw.wpack12 = d + o_wpack12; for (int ui = 0; ui < 4; ++ui) h->lw.wpack_coop[ui] = d + o_wpack_coop[ui];
    h->lw.wpack_coopn = d + o_wpack_coopn;
    h->lw.wpack_hp = d + o_wpack_hp;
    h->lw.wpack_hpw = h->hp_ok ? d + o_wpack_hpw : nullptr;
    h->lw.hp_wave = h->hp_wave;
    h->lw.wpack_coopw = h->coopw_ok ? d + o_wpack_coopw : nullptr;
    h->lw.wpack_gru = d + o_wpack_gru;
    h->lw.wpack16 = d + o_wpack16;
    h->lw.wpack16_bf = have16_bf ? d + o_wpack16_bf : nullptr;
    h->lw.wpack_bf[0] = d + o_wpack_bf[0]; h->lw.wpack_bf[1] = d + o_wpack_bf[1]; h->lw.ih_bf16 = h->ih_bf16; h->lw.waves = h->lstm_waves; h->lw.bias = d + o_lbias; h->lw.wfc = d + o_wfc; h->lw.bfc = d + o_bfc;
    h->lw.H = H; h->lw.NIN = h->NIN; h->lw.KX = h->KX; h->lw.OUT = h->cfg.output_size; h->lw.gru = h->gru;
    h->lw.wgen = d + o_wgen;
    if (h->sb_tcn) bind_tcn(h->sbt, sb_off, d);
    if (fsn) {
        h->fbw = LstmWeights{};
        for (int ui = 0; ui < 3; ++ui) h->fbw.wpack_coop[ui] = d + o_fbpack[ui];
        h->fbw.bias = d + o_fbbias;
        h->fbw.H = CH; h->fbw.NIN = F; h->fbw.KX = KXF; h->fbw.OUT = 0; h->fbw.gru = h->gru;
        h->fbw.wgen = d + o_fbgen;
        h->fbw.wpack_fbv = have_fbv ? d + o_fbv : nullptr;
        h->fsn_wf = d + o_fsn_wf; h->fsn_bf = d + o_fsn_bf; h->fsn_kp = fsn_kp;
    }
    h->d_refl_w = d + o_refl;
    h->d_refl_wfb = d + o_reflfb;
    if (tuned) {                                // which column-split instantiations fit twice on a CU (registers, LDS)
        for (int ui = 0; ui < 4; ++ui) h->occ_ksplit[ui] = std::max(1, lstm_coop_occupancy(h->lw, 8 << ui));
        for (int rpg = 1; rpg <= 2; ++rpg) h->occ_coopn[rpg - 1] = std::max(1, lstm_coopn_occupancy(h->lw, rpg));
    }
    if (h->generic_sb && lstm_generic_check(h->H, h->NIN, false)) return 2;
    if (h->generic_fb && lstm_generic_check(h->CH, h->F, true)) return 2;
    h->committed = true;
    (void)Fr;
    return 0;
}

}  // extern "C"
