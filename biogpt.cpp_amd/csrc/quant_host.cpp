// See quant_host.h.  One encoder template covers the four nibble formats; the knobs per format are
// (bits, asymmetric?) -- symmetric formats scale by the signed extreme value divided by
// -2^(bits-1), asymmetric ones by (max-min)/(2^bits-1) with the minimum stored.
#include "quant_host.h"

#include <algorithm>
#include <cmath>
#include <thread>

#include "model_file.h"

namespace bg {

namespace {

template <int BITS, bool ASYM>
void encode_nibble_block(const float *x, uint8_t *out) {
    constexpr int LEVELS = 1 << BITS;
    float scale, base = 0.0f;
    if (ASYM) {
        float lo = x[0], hi = x[0];
        for (int j = 1; j < QK; j++) { lo = std::min(lo, x[j]); hi = std::max(hi, x[j]); }
        scale = (hi - lo) / (float)(LEVELS - 1);
        base = lo;
    } else {
        float extreme = 0.0f, mag = 0.0f;
        for (int j = 0; j < QK; j++)
            if (std::fabs(x[j]) > mag) { mag = std::fabs(x[j]); extreme = x[j]; }
        scale = extreme / (float)(-(LEVELS / 2));
    }
    const float inv = scale != 0.0f ? 1.0f / scale : 0.0f;

    uint8_t *p = out;
    const uint16_t hs = f32_to_f16(scale);
    std::memcpy(p, &hs, 2); p += 2;
    if (ASYM) { const uint16_t hb = f32_to_f16(base); std::memcpy(p, &hb, 2); p += 2; }
    uint8_t *qh = nullptr;
    if (BITS == 5) { qh = p; p += 4; }
    uint8_t *qs = p;

    uint32_t high_bits = 0;
    for (int j = 0; j < QK / 2; j++) {
        int q[2];
        for (int half = 0; half < 2; half++) {
            const float v = x[j + half * (QK / 2)];
            int code;
            if (ASYM) {
                const float t = (v - base) * inv + 0.5f;
                code = (BITS == 4) ? std::min(LEVELS - 1, (int)(int8_t)t) : (int)(uint8_t)t;
            } else {
                const float t = v * inv + ((float)(LEVELS / 2) + 0.5f);  // one rounding: x*id + 8.5f / 16.5f
                code = std::min(LEVELS - 1, (int)(int8_t)t);
            }
            q[half] = code;
        }
        qs[j] = (uint8_t)((q[0] & 0x0F) | ((q[1] & 0x0F) << 4));
        if (BITS == 5) {
            high_bits |= (uint32_t)((q[0] >> 4) & 1) << j;
            high_bits |= (uint32_t)((q[1] >> 4) & 1) << (j + QK / 2);
        }
    }
    if (BITS == 5) std::memcpy(qh, &high_bits, 4);
}

void encode_q8_0_block(const float *x, uint8_t *out) {
    float mag = 0.0f;
    for (int j = 0; j < QK; j++) mag = std::max(mag, std::fabs(x[j]));
    const float scale = mag / 127.0f;
    const float inv = scale != 0.0f ? 1.0f / scale : 0.0f;
    const uint16_t hs = f32_to_f16(scale);
    std::memcpy(out, &hs, 2);
    int8_t *q = reinterpret_cast<int8_t *>(out + 2);
    for (int j = 0; j < QK; j++) q[j] = (int8_t)std::round(x[j] * inv);  // half away from zero
}

void encode_row(int32_t type, const float *x, int64_t k, uint8_t *dst) {
    const size_t bb = file_block_bytes(type);
    for (int64_t b = 0; b < k / QK; b++) {
        const float *xb = x + b * QK;
        uint8_t *ob = dst + (size_t)b * bb;
        switch (type) {
            case T_Q4_0: encode_nibble_block<4, false>(xb, ob); break;
            case T_Q4_1: encode_nibble_block<4, true>(xb, ob); break;
            case T_Q5_0: encode_nibble_block<5, false>(xb, ob); break;
            case T_Q5_1: encode_nibble_block<5, true>(xb, ob); break;
            case T_Q8_0: encode_q8_0_block(xb, ob); break;
            default: break;
        }
    }
}

unsigned worker_count() {
    unsigned n = std::thread::hardware_concurrency();
    return n == 0 ? 1 : std::min(n, 32u);
}

}  // namespace

size_t quantize_rows(int32_t type, const float *src, int64_t nrows, int64_t k, uint8_t *dst) {
    const size_t rb = file_row_bytes(type, k);
    if (type == T_F32) {
        std::memcpy(dst, src, rb * (size_t)nrows);
    } else if (type == T_F16) {
        uint16_t *h = reinterpret_cast<uint16_t *>(dst);
        for (int64_t i = 0; i < nrows * k; i++) h[i] = f32_to_f16(src[i]);
    } else {
        const unsigned nw = (nrows * k > (1 << 18)) ? worker_count() : 1;
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nw; w++) {
            const int64_t r0 = nrows * w / nw, r1 = nrows * (w + 1) / nw;
            auto job = [=] { for (int64_t r = r0; r < r1; r++) encode_row(type, src + r * k, k, dst + (size_t)r * rb); };
            if (nw == 1) job(); else pool.emplace_back(job);
        }
        for (auto &t : pool) t.join();
    }
    return rb * (size_t)nrows;
}

void dequantize_row(int32_t type, const uint8_t *src, int64_t k, float *dst) {
    if (type == T_F32) { std::memcpy(dst, src, (size_t)k * 4); return; }
    if (type == T_F16) {
        const uint16_t *h = reinterpret_cast<const uint16_t *>(src);
        for (int64_t i = 0; i < k; i++) dst[i] = f16_to_f32(h[i]);
        return;
    }
    const size_t bb = file_block_bytes(type);
    for (int64_t b = 0; b < k / QK; b++) {
        const uint8_t *p = src + (size_t)b * bb;
        float *o = dst + b * QK;
        uint16_t hd; std::memcpy(&hd, p, 2); p += 2;
        const float d = f16_to_f32(hd);
        float m = 0.0f;
        if (type == T_Q4_1 || type == T_Q5_1) { uint16_t hm; std::memcpy(&hm, p, 2); p += 2; m = f16_to_f32(hm); }
        if (type == T_Q8_0) {
            const int8_t *q = reinterpret_cast<const int8_t *>(p);
            for (int j = 0; j < QK; j++) o[j] = (float)q[j] * d;
            continue;
        }
        uint32_t qh = 0;
        const bool five = (type == T_Q5_0 || type == T_Q5_1);
        if (five) { std::memcpy(&qh, p, 4); p += 4; }
        const int off = (type == T_Q4_0) ? 8 : (type == T_Q5_0 ? 16 : 0);
        for (int j = 0; j < QK / 2; j++) {
            int lo = p[j] & 0x0F, hi = p[j] >> 4;
            if (five) { lo |= (int)((qh >> j) & 1u) << 4; hi |= (int)((qh >> (j + 16)) & 1u) << 4; }
            if (off) { o[j] = (float)(lo - off) * d; o[j + 16] = (float)(hi - off) * d; }
            else     { o[j] = (float)lo * d + m;     o[j + 16] = (float)hi * d + m; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// file -> file quantizer
// ------------------------------------------------------------------------------------------------
namespace {
struct OutFile {
    FILE *f = nullptr;
    ~OutFile() { if (f) fclose(f); }
    bool put(const void *p, size_t n) { return fwrite(p, 1, n, f) == n; }
    bool put_i32(int32_t v) { return put(&v, 4); }
    bool put_str(const std::string &s) { return put_i32((int32_t)s.size()) && put(s.data(), s.size()); }
};

bool write_preamble(OutFile &out, const biogpt_hip_hparams &hp, int32_t ftype,
                    const std::vector<std::string> &vocab, const std::vector<std::string> &merges) {
    bool ok = out.put_i32((int32_t)FILE_MAGIC) && out.put_i32(hp.n_vocab) && out.put_i32(hp.n_layer) &&
              out.put_i32(hp.n_head) && out.put_i32(hp.n_positions) && out.put_i32(hp.d_ff) &&
              out.put_i32(hp.d_model) && out.put_i32(ftype);
    ok = ok && out.put_i32((int32_t)vocab.size());
    for (const auto &s : vocab) ok = ok && out.put_str(s);
    ok = ok && out.put_i32((int32_t)merges.size());
    for (const auto &s : merges) ok = ok && out.put_str(s);
    return ok;
}

bool write_tensor_header(OutFile &out, const std::string &name, int32_t type, int n_dims, int64_t ne0, int64_t ne1) {
    bool ok = out.put_i32(n_dims) && out.put_i32((int32_t)name.size()) && out.put_i32(type) && out.put_i32((int32_t)ne0);
    if (n_dims == 2) ok = ok && out.put_i32((int32_t)ne1);
    return ok && out.put(name.data(), name.size());
}
}  // namespace

bool quantize_file(const std::string &in_path, const std::string &out_path, int32_t ftype) {
    const TensorType qt = ftype_to_type(ftype);
    if (qt == T_INVALID || !is_quantized(qt)) BG_FAIL(false, "invalid model type %d", ftype);  // biogpt.cpp:468-484
    ModelFile mf;
    if (!mf.open(in_path)) return false;
    OutFile out;
    out.f = fopen(out_path.c_str(), "wb");
    if (!out.f) BG_FAIL(false, "failed to open '%s' for writing", out_path.c_str());
    if (!write_preamble(out, mf.hp, ftype, mf.vocab, mf.merges))  // quantize.cpp:43-122
        BG_FAIL(false, "write failed on '%s'", out_path.c_str());

    std::vector<uint8_t> raw, packed;
    std::vector<float> f32;
    for (const TensorEntry &t : mf.tensors) {
        raw.resize(t.nbytes);
        if (!mf.read_payload(t, raw.data())) return false;
        // selection rule biogpt.cpp:523: name contains "weight" and the tensor is 2-D
        const bool quantize = t.name.find("weight") != std::string::npos && t.ne1 != 1;
        if (!quantize) {
            if (!write_tensor_header(out, t.name, t.type, t.n_dims, t.ne0, t.ne1) || !out.put(raw.data(), raw.size()))
                BG_FAIL(false, "write failed on '%s'", out_path.c_str());
            continue;
        }
        if (t.type != T_F32 && t.type != T_F16)
            BG_FAIL(false, "unsupported ttype %d for integer quantization of '%s'", t.type, t.name.c_str());
        if (t.ne0 % QK) BG_FAIL(false, "tensor '%s': row length %lld not a multiple of %d", t.name.c_str(), (long long)t.ne0, QK);
        const int64_t nel = t.ne0 * t.ne1;
        f32.resize((size_t)nel);
        if (t.type == T_F16) {
            const uint16_t *h = reinterpret_cast<const uint16_t *>(raw.data());
            for (int64_t i = 0; i < nel; i++) f32[(size_t)i] = f16_to_f32(h[i]);
        } else {
            std::memcpy(f32.data(), raw.data(), (size_t)nel * 4);
        }
        packed.resize(file_row_bytes(qt, t.ne0) * (size_t)t.ne1);
        quantize_rows(qt, f32.data(), t.ne1, t.ne0, packed.data());
        if (!write_tensor_header(out, t.name, qt, t.n_dims, t.ne0, t.ne1) || !out.put(packed.data(), packed.size()))
            BG_FAIL(false, "write failed on '%s'", out_path.c_str());
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// synthetic model writer: counter-based generator so any element can be regenerated
// independently: u64 stream = splitmix64(seed, tensor index, element pair index) -> Box-Muller.
// ------------------------------------------------------------------------------------------------
namespace {
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// two independent N(0,1) values for pair index p of stream `key`
inline void normal_pair(uint64_t key, uint64_t p, float &a, float &b) {
    const uint64_t r1 = splitmix64(key ^ (2 * p + 1) * 0xD6E8FEB86659FD93ull);
    const uint64_t r2 = splitmix64(r1 ^ key);
    const double u1 = ((double)(r1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0,1]
    const double u2 = (double)(r2 >> 11) * (1.0 / 9007199254740992.0);          // [0,1)
    const double rad = std::sqrt(-2.0 * std::log(u1));
    a = (float)(rad * std::cos(6.283185307179586 * u2));
    b = (float)(rad * std::sin(6.283185307179586 * u2));
}

void fill_normal(float *dst, int64_t n, uint64_t key, float mean, float std_) {
    const unsigned nw = n > (1 << 18) ? worker_count() : 1;
    const int64_t pairs = (n + 1) / 2;
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < nw; w++) {
        const int64_t p0 = pairs * w / nw, p1 = pairs * (w + 1) / nw;
        auto job = [=] {
            for (int64_t p = p0; p < p1; p++) {
                float a, b;
                normal_pair(key, (uint64_t)p, a, b);
                dst[2 * p] = mean + std_ * a;
                if (2 * p + 1 < n) dst[2 * p + 1] = mean + std_ * b;
            }
        };
        if (nw == 1) job(); else pool.emplace_back(job);
    }
    for (auto &t : pool) t.join();
}
}  // namespace

bool write_synthetic(const std::string &path, const biogpt_hip_hparams &hp, uint64_t seed) {
    if (hp.ftype != 0 && hp.ftype != 1) BG_FAIL(false, "synthetic models are written as f32 (0) or f16 (1), got ftype %d", hp.ftype);
    if (hp.d_model % hp.n_head) BG_FAIL(false, "d_model %% n_head != 0");
    OutFile out;
    out.f = fopen(path.c_str(), "wb");
    if (!out.f) BG_FAIL(false, "failed to open '%s' for writing", path.c_str());
    std::vector<std::string> vocab((size_t)hp.n_vocab), merges((size_t)std::max(0, hp.n_merges));
    for (int i = 0; i < hp.n_vocab; i++) vocab[(size_t)i] = (i == 2 ? std::string("</s>") : "t" + std::to_string(i) + "</w>");
    for (int i = 0; i < hp.n_merges; i++) merges[(size_t)i] = "a" + std::to_string(i) + " b" + std::to_string(i);
    if (!write_preamble(out, hp, hp.ftype, vocab, merges)) BG_FAIL(false, "write failed on '%s'", path.c_str());

    // file order = HF state-dict order (convert.py:55): embeddings, layers, final LN, lm head
    std::vector<ExpectedTensor> order;
    {
        auto all = expected_tensors(hp);
        auto pick = [&](const std::string &n) { for (auto &e : all) if (e.name == n) { order.push_back(e); return; } };
        pick("biogpt.embed_tokens.weight");
        pick("biogpt.embed_positions.weight");
        for (auto &e : all) if (e.name.rfind("biogpt.layers.", 0) == 0) order.push_back(e);
        pick("biogpt.layer_norm.weight");
        pick("biogpt.layer_norm.bias");
        pick("output_projection.weight");
    }
    std::vector<float> buf;
    std::vector<uint16_t> half;
    uint64_t tensor_idx = 0;
    for (auto &e : order) {
        const int64_t ne0 = e.ne0;
        const int64_t ne1 = e.ne1 < 0 ? (int64_t)hp.n_positions + 2 : e.ne1;  // embed_positions: P+2 rows (F5)
        const int64_t n = ne0 * ne1;
        buf.resize((size_t)n);
        const uint64_t key = splitmix64(seed ^ (0x5851F42D4C957F2Dull * (++tensor_idx)));
        const bool ln_gain = !e.matrix && e.name.find("layer_norm.weight") != std::string::npos;
        fill_normal(buf.data(), n, key, ln_gain ? 1.0f : 0.0f, 0.02f);
        if (e.name == "biogpt.embed_tokens.weight" && hp.n_vocab > 1)
            std::fill(buf.begin() + ne0, buf.begin() + 2 * ne0, 0.0f);  // pad row (id 1) is zero in HF
        const int n_dims = e.matrix ? 2 : 1;
        const int32_t type = (e.matrix && hp.ftype == 1) ? T_F16 : T_F32;
        if (!write_tensor_header(out, e.name, type, n_dims, ne0, ne1)) BG_FAIL(false, "write failed on '%s'", path.c_str());
        bool ok;
        if (type == T_F16) {
            half.resize((size_t)n);
            for (int64_t i = 0; i < n; i++) half[(size_t)i] = f32_to_f16(buf[(size_t)i]);
            ok = out.put(half.data(), (size_t)n * 2);
        } else {
            ok = out.put(buf.data(), (size_t)n * 4);
        }
        if (!ok) BG_FAIL(false, "write failed on '%s'", path.c_str());
    }
    return true;
}

}  // namespace bg
