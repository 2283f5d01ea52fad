// Weight intake: checkpoint tensors (named as the reference loader reads them,
// WeightLoading.swift:92-127, 151-176, 243-347) -> kernel-friendly device layouts.
//   * q/k/v projections are concatenated row-wise into one GEMM operand;
//   * gate/up rows are interleaved in blocks of 8 so the SwiGLU product is formed in the epilogue;
//   * conv_out columns are permuted from the reference's channel-major flatten (c*F + f,
//     AudioEncoder.swift:395-396) to (f*C + c), the order conv3's epilogue writes;
//   * norm weights and biases are widened to fp32.
// Missing or mis-shaped tensors raise an error naming the key.
#include "model.h"
#include <cmath>
#include <cstring>

namespace {

const HostTensor& need(q3asr_model* m, const std::string& name, int dtype, std::vector<int64_t> shape) {
    auto it = m->host.find(name);
    if (it == m->host.end()) throw Q3Error("q3asr: missing checkpoint tensor '" + name + "'");
    const HostTensor& t = it->second;
    if (t.dtype != dtype) throw Q3Error("q3asr: tensor '" + name + "' has the wrong dtype");
    if (t.shape != shape) {
        std::string got, want;
        for (auto s : t.shape) got += std::to_string(s) + " ";
        for (auto s : shape) want += std::to_string(s) + " ";
        throw Q3Error("q3asr: tensor '" + name + "' has shape [ " + got + "], expected [ " + want + "]");
    }
    return t;
}

inline float bf16_host(uint16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

float* upload_f32_from_bf16(q3asr_model* m, const HostTensor& t) {
    std::vector<float> v((size_t)t.numel());
    const uint16_t* s = (const uint16_t*)t.data.data();
    for (size_t i = 0; i < v.size(); ++i) v[i] = bf16_host(s[i]);
    float* d = m->dalloc<float>(v.size());
    HIP_CHECK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return d;
}

float* vec_f32(q3asr_model* m, const std::string& name, int64_t n) { return upload_f32_from_bf16(m, need(m, name, Q3ASR_BF16, {n})); }

template <class T>
T* upload_vec(q3asr_model* m, const std::vector<T>& v) {
    T* d = m->dalloc<T>(v.size());
    HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

struct RowRef {
    std::string name;  // linear prefix
    int64_t row;
};

// Builds one device GEMM operand from rows of (possibly several) checkpoint linears.
LinearW build_linear(q3asr_model* m, const std::vector<RowRef>& rows, int64_t K, bool q4, const std::vector<int>* colperm = nullptr) {
    LinearW L;
    L.N = (int)rows.size();
    L.K = (int)K;
    const int g = m->cfg.group_size;
    std::map<std::string, int64_t> nrows;
    for (auto& r : rows) nrows[r.name] = std::max(nrows[r.name], r.row + 1);
    if (!q4) {
        std::vector<uint16_t> buf((size_t)L.N * K);
        std::map<std::string, const uint16_t*> src;
        for (auto& kv : nrows) {
            auto it = m->host.find(kv.first + ".weight");
            if (it == m->host.end()) throw Q3Error("q3asr: missing checkpoint tensor '" + kv.first + ".weight'");
            const HostTensor& t = it->second;
            if (t.dtype != Q3ASR_BF16 || t.numel() % K != 0 || t.numel() / K < kv.second)
                throw Q3Error("q3asr: tensor '" + kv.first + ".weight' has the wrong dtype or shape");
            src[kv.first] = (const uint16_t*)t.data.data();
        }
        for (size_t i = 0; i < rows.size(); ++i) {
            const uint16_t* s = src[rows[i].name] + rows[i].row * K;
            uint16_t* d = buf.data() + i * K;
            if (colperm) for (int64_t c = 0; c < K; ++c) d[c] = s[(*colperm)[c]];
            else memcpy(d, s, K * 2);
        }
        L.w = upload_vec(m, buf);
        L.fmt = W_BF16;
    } else {
        if (K % g != 0 || g != 64 || m->cfg.bits != 4) throw Q3Error("q3asr: q4 weights need group_size 64, bits 4");
        const int64_t kw = K / 8, kg = K / g;
        std::vector<uint32_t> wbuf((size_t)L.N * kw);
        std::vector<uint16_t> sbuf((size_t)L.N * kg), bbuf((size_t)L.N * kg);
        for (auto& kv : nrows) {
            auto it = m->host.find(kv.first + ".weight");
            if (it == m->host.end()) throw Q3Error("q3asr: missing checkpoint tensor '" + kv.first + ".weight'");
            if (it->second.dtype != Q3ASR_U32 || it->second.shape.size() != 2 || it->second.shape[1] != kw ||
                it->second.shape[0] < kv.second)
                throw Q3Error("q3asr: tensor '" + kv.first + ".weight' is not a [rows, K/8] u32 q4 weight of the expected size");
            need(m, kv.first + ".scales", Q3ASR_BF16, {it->second.shape[0], kg});
            need(m, kv.first + ".biases", Q3ASR_BF16, {it->second.shape[0], kg});
        }
        for (size_t i = 0; i < rows.size(); ++i) {
            const auto& w = m->host.at(rows[i].name + ".weight");
            const auto& s = m->host.at(rows[i].name + ".scales");
            const auto& b = m->host.at(rows[i].name + ".biases");
            memcpy(wbuf.data() + i * kw, (const uint32_t*)w.data.data() + rows[i].row * kw, kw * 4);
            memcpy(sbuf.data() + i * kg, (const uint16_t*)s.data.data() + rows[i].row * kg, kg * 2);
            memcpy(bbuf.data() + i * kg, (const uint16_t*)b.data.data() + rows[i].row * kg, kg * 2);
        }
        L.w = upload_vec(m, wbuf);
        L.scales = upload_vec(m, sbuf);
        L.biases = upload_vec(m, bbuf);
        L.fmt = W_Q4;
    }
    // every checkpoint tensor feeds exactly one operand: its host copy is released as soon as that operand is on the
    // device, so a model being loaded never holds much more than one layer twice (a finalize that fails half-way
    // leaves a model that cannot be finalized again; it could not have been used anyway)
    for (auto& kv : nrows)
        for (const char* part : {".weight", ".scales", ".biases"}) m->host.erase(kv.first + part);
    return L;
}

std::vector<RowRef> all_rows(const std::string& name, int64_t n) {
    std::vector<RowRef> r;
    for (int64_t i = 0; i < n; ++i) r.push_back({name, i});
    return r;
}

void check_dense(q3asr_model* m, const std::string& name, int64_t n, int64_t k) { need(m, name + ".weight", Q3ASR_BF16, {n, k}); }

}  // namespace

void model_finalize(q3asr_model* m) {
    if (m->finalized) throw Q3Error("q3asr: model already finalized");
    const q3asr_config& c = m->cfg;
    HIP_CHECK(hipSetDevice(m->device));
    const std::string a = "audio_tower.";
    const int d = c.enc_d_model, C = c.conv_channels;
    int f3 = c.n_mels;
    for (int i = 0; i < 3; ++i) f3 = (f3 - 1) / 2 + 1;
    m->conv_f3 = f3;

    // ---- encoder ----
    {
        const HostTensor& w1 = need(m, a + "conv2d1.weight", Q3ASR_BF16, {C, 3, 3, 1});
        m->conv1_w = upload_f32_from_bf16(m, w1);
        m->conv1_b = vec_f32(m, a + "conv2d1.bias", C);
        need(m, a + "conv2d2.weight", Q3ASR_BF16, {C, 3, 3, C});
        need(m, a + "conv2d3.weight", Q3ASR_BF16, {C, 3, 3, C});
        m->conv2 = build_linear(m, all_rows(a + "conv2d2", C), 9 * C, false);
        m->conv3 = build_linear(m, all_rows(a + "conv2d3", C), 9 * C, false);
        m->conv2_b = vec_f32(m, a + "conv2d2.bias", C);
        m->conv3_b = vec_f32(m, a + "conv2d3.bias", C);
        check_dense(m, a + "conv_out", d, (int64_t)C * f3);
        std::vector<int> perm((size_t)C * f3);
        for (int f = 0; f < f3; ++f)
            for (int ch = 0; ch < C; ++ch) perm[(size_t)f * C + ch] = ch * f3 + f;
        m->conv_out = build_linear(m, all_rows(a + "conv_out", d), (int64_t)C * f3, false, &perm);
        // sinusoidal positions 0..12, restarted per chunk: AudioEncoder.swift:141-169, 407-408
        const int half = d / 2;
        std::vector<float> pe((size_t)13 * d);
        const float inc = logf(10000.0f) / (float)(half - 1);
        for (int t = 0; t < 13; ++t)
            for (int i = 0; i < half; ++i) {
                const float st = (float)t * expf(-inc * (float)i);
                pe[(size_t)t * d + i] = sinf(st);
                pe[(size_t)t * d + half + i] = cosf(st);
            }
        m->pe_tab = upload_vec(m, pe);
        m->enc.resize(c.enc_layers);
        for (int i = 0; i < c.enc_layers; ++i) {
            const std::string p = a + "layers." + std::to_string(i) + ".";
            EncLayerW& L = m->enc[i];
            L.ln1_w = vec_f32(m, p + "self_attn_layer_norm.weight", d);
            L.ln1_b = vec_f32(m, p + "self_attn_layer_norm.bias", d);
            L.ln2_w = vec_f32(m, p + "final_layer_norm.weight", d);
            L.ln2_b = vec_f32(m, p + "final_layer_norm.bias", d);
            std::vector<RowRef> rows;
            std::vector<float> qb;
            for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
                check_dense(m, p + "self_attn." + n, d, d);
                auto r = all_rows(p + "self_attn." + n, d);
                rows.insert(rows.end(), r.begin(), r.end());
                const HostTensor& b = need(m, p + "self_attn." + n + ".bias", Q3ASR_BF16, {d});
                for (int j = 0; j < d; ++j) qb.push_back(bf16_host(((const uint16_t*)b.data.data())[j]));
            }
            L.qkv = build_linear(m, rows, d, false);
            L.qkv_b = upload_vec(m, qb);
            check_dense(m, p + "self_attn.out_proj", d, d);
            L.out = build_linear(m, all_rows(p + "self_attn.out_proj", d), d, false);
            L.out_b = vec_f32(m, p + "self_attn.out_proj.bias", d);
            check_dense(m, p + "fc1", c.enc_ffn, d);
            L.fc1 = build_linear(m, all_rows(p + "fc1", c.enc_ffn), d, false);
            L.fc1_b = vec_f32(m, p + "fc1.bias", c.enc_ffn);
            check_dense(m, p + "fc2", d, c.enc_ffn);
            L.fc2 = build_linear(m, all_rows(p + "fc2", d), c.enc_ffn, false);
            L.fc2_b = vec_f32(m, p + "fc2.bias", d);
        }
        m->ln_post_w = vec_f32(m, a + "ln_post.weight", d);
        m->ln_post_b = vec_f32(m, a + "ln_post.bias", d);
        check_dense(m, a + "proj1", d, d);
        m->proj1 = build_linear(m, all_rows(a + "proj1", d), d, false);
        m->proj1_b = vec_f32(m, a + "proj1.bias", d);
        check_dense(m, a + "proj2", c.enc_out_dim, d);
        m->proj2 = build_linear(m, all_rows(a + "proj2", c.enc_out_dim), d, false);
        m->proj2_b = vec_f32(m, a + "proj2.bias", c.enc_out_dim);
    }

    // ---- decoder ----
    {
        const bool q4 = c.weight_format == 1;
        const int H = c.hidden, hd = c.head_dim;
        if (!q4) check_dense(m, "model.embed_tokens", c.vocab, H);
        m->embed = build_linear(m, all_rows("model.embed_tokens", c.vocab), H, q4);
        m->final_norm = vec_f32(m, "model.norm.weight", H);
        m->dec.resize(c.layers);
        for (int i = 0; i < c.layers; ++i) {
            const std::string p = "model.layers." + std::to_string(i) + ".";
            DecLayerW& L = m->dec[i];
            L.in_norm = vec_f32(m, p + "input_layernorm.weight", H);
            L.post_norm = vec_f32(m, p + "post_attention_layernorm.weight", H);
            L.q_norm = vec_f32(m, p + "self_attn.q_norm.weight", hd);
            L.k_norm = vec_f32(m, p + "self_attn.k_norm.weight", hd);
            std::vector<RowRef> rows = all_rows(p + "self_attn.q_proj", (int64_t)c.heads * hd);
            auto rk = all_rows(p + "self_attn.k_proj", (int64_t)c.kv_heads * hd);
            auto rv = all_rows(p + "self_attn.v_proj", (int64_t)c.kv_heads * hd);
            rows.insert(rows.end(), rk.begin(), rk.end());
            rows.insert(rows.end(), rv.begin(), rv.end());
            if (!q4) {
                check_dense(m, p + "self_attn.q_proj", (int64_t)c.heads * hd, H);
                check_dense(m, p + "self_attn.k_proj", (int64_t)c.kv_heads * hd, H);
                check_dense(m, p + "self_attn.v_proj", (int64_t)c.kv_heads * hd, H);
                check_dense(m, p + "self_attn.o_proj", H, (int64_t)c.heads * hd);
                check_dense(m, p + "mlp.gate_proj", c.inter, H);
                check_dense(m, p + "mlp.up_proj", c.inter, H);
                check_dense(m, p + "mlp.down_proj", H, c.inter);
            }
            L.qkv = build_linear(m, rows, H, q4);
            L.o = build_linear(m, all_rows(p + "self_attn.o_proj", H), (int64_t)c.heads * hd, q4);
            std::vector<RowRef> gu;
            for (int j = 0; j < c.inter; j += 8) {
                for (int k = 0; k < 8; ++k) gu.push_back({p + "mlp.gate_proj", j + k});
                for (int k = 0; k < 8; ++k) gu.push_back({p + "mlp.up_proj", j + k});
            }
            L.gate_up = build_linear(m, gu, H, q4);
            L.down = build_linear(m, all_rows(p + "mlp.down_proj", H), c.inter, q4);
        }
        // RoPE tables: inv_freq and angles in fp32 as QuantizedTextDecoder.swift:219-230 evaluates them
        const int half = std::min(c.rotary_dim, hd) / 2;
        m->rope_max_pos = 4096;
        std::vector<float> ct((size_t)m->rope_max_pos * half), st(ct.size());
        const float lt = logf(c.rope_theta) / (float)half;
        for (int pos = 0; pos < m->rope_max_pos; ++pos)
            for (int i = 0; i < half; ++i) {
                const float ang = (float)pos * expf(-(float)i * lt);
                ct[(size_t)pos * half + i] = cosf(ang);
                st[(size_t)pos * half + i] = sinf(ang);
            }
        m->rope_cos = upload_vec(m, ct);
        m->rope_sin = upload_vec(m, st);
    }
    m->host.clear();
    m->finalized = true;
}
