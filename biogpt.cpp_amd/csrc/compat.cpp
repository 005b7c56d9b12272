// C++ wrappers with the reference's signatures (include/biogpt_compat.h) over the C-ABI, plus the
// shims for the 11 ggml symbols examples/main/main.cpp calls directly.
#include "../../include/biogpt_compat.h"
#include "host_common.h"
#include "quant_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace {
// distinct non-null addresses for the opaque handles callers only pass around / free
char g_backend_tag, g_buffer_tag, g_allocr_tag, g_graph_tag;
std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
}  // namespace

extern "C" {
void ggml_time_init(void) { g_t0 = std::chrono::steady_clock::now(); }
int64_t ggml_time_us(void) { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - g_t0).count(); }
size_t ggml_backend_get_alignment(ggml_backend_t) { return 256; }
struct ggml_allocr *ggml_allocr_new_measure(size_t) { return reinterpret_cast<ggml_allocr *>(&g_allocr_tag); }
size_t ggml_allocr_alloc_graph(struct ggml_allocr *, struct ggml_cgraph *) { return 0; }  // scratch lives in the engine
void ggml_allocr_free(struct ggml_allocr *) {}
ggml_backend_buffer_t ggml_backend_alloc_buffer(ggml_backend_t, size_t) { return reinterpret_cast<ggml_backend_buffer_t>(&g_buffer_tag); }
struct ggml_allocr *ggml_allocr_new_from_buffer(ggml_backend_buffer_t) { return reinterpret_cast<ggml_allocr *>(&g_allocr_tag); }
void ggml_free(struct ggml_context *ctx) { biogpt_hip_free(reinterpret_cast<biogpt_hip_ctx *>(ctx)); }
void ggml_backend_buffer_free(ggml_backend_buffer_t) {}
void ggml_backend_free(ggml_backend_t) {}
}

bool biogpt_model_load(const std::string &fname, biogpt_model &model, biogpt_vocab &vocab, const uint8_t verbosity) {
    const char *dev = std::getenv("BIOGPT_HIP_DEVICE");
    biogpt_hip_ctx *ctx = biogpt_hip_load(fname.c_str(), dev ? std::atoi(dev) : 0, verbosity);
    if (!ctx) return false;  // message already printed, reference-style
    biogpt_hip_hparams hp;
    biogpt_hip_get_hparams(ctx, &hp);
    model.hparams.n_vocab = hp.n_vocab; model.hparams.n_layer = hp.n_layer; model.hparams.n_head = hp.n_head;
    model.hparams.n_positions = hp.n_positions; model.hparams.d_ff = hp.d_ff; model.hparams.d_model = hp.d_model;
    model.hparams.ftype = hp.ftype; model.hparams.n_merges = hp.n_merges;
    model.layers_decoder.assign((size_t)hp.n_layer, biogpt_layer_decoder());
    model.n_loaded = biogpt_hip_n_tensors(ctx);
    model.ctx = reinterpret_cast<ggml_context *>(ctx);
    model.backend = reinterpret_cast<ggml_backend_t>(&g_backend_tag);
    model.buffer_w = model.buffer_kv = reinterpret_cast<ggml_backend_buffer_t>(&g_buffer_tag);

    vocab.n_vocab = hp.n_vocab;
    vocab.n_merges = hp.n_merges;
    for (int32_t i = 0; i < hp.n_vocab; i++) {  // biogpt.cpp:86-100
        const char *p = nullptr; int32_t n = 0;
        if (biogpt_hip_vocab_token(ctx, i, &p, &n) != 0) break;
        std::string w(p, (size_t)n);
        vocab.token_to_id[w] = i;
        vocab.id_to_token[i] = w;
    }
    word_pair last_pair;
    for (int32_t r = 0; r < hp.n_merges; r++) {  // biogpt.cpp:131-155: "left right" -> rank
        const char *p = nullptr; int32_t n = 0;
        if (biogpt_hip_merge(ctx, r, &p, &n) != 0) break;
        if (n > 0) {   // first two whitespace-separated words; an empty record re-ranks the previous pair (biogpt.cpp:138-152)
            std::stringstream ss(std::string(p, (size_t)n));
            std::string left, right;
            ss >> left >> right;
            last_pair = word_pair(left, right);
        }
        vocab.bpe_ranks[last_pair] = r;
    }
    return true;
}

// ---- tensor-section quantizer on open streams (biogpt.cpp:459-621) -----------------------------------
// examples/quantize/quantize.cpp copies the header, vocab and merges itself and hands the two streams over
// for the tensors.  Same selection rule (name contains "weight" and ne[1] != 1), same records, same
// progress lines; payload bytes come from the block encoders of quant_host.cpp (byte-identical to the oracle).
void biogpt_model_quantize_internal(std::ifstream &fin, std::ofstream &fout, const ggml_ftype ftype) {
    const bg::TensorType target = bg::ftype_to_type((int32_t)ftype);
    if (target == bg::T_INVALID || target == bg::T_F32 || target == bg::T_F16) {
        fprintf(stderr, "%s: invalid model type %d\n", __func__, (int)ftype);
        throw std::runtime_error("invalid model type");
    }
    auto get32 = [&fin]() { int32_t v = 0; fin.read(reinterpret_cast<char *>(&v), 4); return v; };
    auto put32 = [&fout](int32_t v) { fout.write(reinterpret_cast<const char *>(&v), 4); };
    const double MB = 1024.0 * 1024.0;
    size_t bytes_as_f32 = 0, bytes_out = 0;
    std::vector<char> raw;
    std::vector<float> values;
    std::vector<uint8_t> blocks;
    for (;;) {
        const int32_t n_dims = get32(), name_len = get32();
        int32_t ttype = get32();
        if (fin.eof()) break;
        int32_t ne[2] = {1, 1}, count = 1;
        for (int d = 0; d < n_dims; d++) { ne[d] = get32(); count *= ne[d]; }
        std::string name((size_t)name_len, '\0');
        fin.read(&name[0], name_len);
        printf("%64s - [%5d, %5d], type = %6s ", name.c_str(), ne[0], ne[1], bg::type_name(ttype));

        const bool pick = name.find("weight") != std::string::npos && ne[1] != 1;
        if (pick) {
            if (ttype != bg::T_F32 && ttype != bg::T_F16) throw std::runtime_error("unsupported ttype for integer quantization");
            values.resize((size_t)count);
            if (ttype == bg::T_F16) {
                raw.resize((size_t)count * 2);
                fin.read(raw.data(), (std::streamsize)raw.size());
                const uint16_t *h = reinterpret_cast<const uint16_t *>(raw.data());
                for (int32_t i = 0; i < count; i++) values[(size_t)i] = bg::f16_to_f32(h[i]);
            } else {
                fin.read(reinterpret_cast<char *>(values.data()), (std::streamsize)count * 4);
            }
            ttype = target;
        } else {
            raw.resize((size_t)count * (ttype == bg::T_F32 ? 4 : 2));
            fin.read(raw.data(), (std::streamsize)raw.size());
        }
        put32(n_dims); put32(name_len); put32(ttype);
        for (int d = 0; d < n_dims; d++) put32(ne[d]);
        fout.write(name.data(), name_len);
        if (pick) {
            blocks.resize(bg::file_row_bytes(target, ne[0]) * (size_t)ne[1]);
            const size_t n = bg::quantize_rows(target, values.data(), ne[1], ne[0], blocks.data());
            fout.write(reinterpret_cast<const char *>(blocks.data()), (std::streamsize)n);
            bytes_out += n;
            printf("size = %8.2f MB -> %8.2f MB\n", count * 4.0 / MB, n / MB);
        } else {
            fout.write(raw.data(), (std::streamsize)raw.size());
            bytes_out += raw.size();
            printf("size = %8.3f MB\n", raw.size() / MB);
        }
        bytes_as_f32 += (size_t)count * 4;
    }
    printf("%s: model size  = %8.2f MB\n", __func__, bytes_as_f32 / MB);
    printf("%s: quant size  = %8.2f MB | ftype = %d (%s)\n", __func__, bytes_out / MB, (int)ftype, bg::type_name(target));
}

// ---- tokenizer (biogpt.cpp:850-906) over the C-ABI ---------------------------------------------------
// The maps in `vocab` are the caller's to edit, so the handle is rebuilt from them on every call (a prompt is
// tokenized once per run): id_to_token in id order, bpe_ranks in rank order.
namespace {
struct VocabHandle {
    biogpt_hip_vocab *h = nullptr;
    explicit VocabHandle(biogpt_vocab &vocab) {
        int32_t n_tok = 0;
        for (const auto &kv : vocab.id_to_token) n_tok = std::max(n_tok, kv.first + 1);
        std::vector<std::string> tokens((size_t)n_tok), merges;
        for (const auto &kv : vocab.id_to_token) if (kv.first >= 0) tokens[(size_t)kv.first] = kv.second;
        for (const auto &kv : vocab.bpe_ranks) {
            if (kv.second < 0) continue;
            if ((size_t)kv.second >= merges.size()) merges.resize((size_t)kv.second + 1);
            merges[(size_t)kv.second] = kv.first.first + " " + kv.first.second;
        }
        std::vector<const char *> tp, mp;
        std::vector<int32_t> tl, ml;
        for (const std::string &t : tokens) { tp.push_back(t.data()); tl.push_back((int32_t)t.size()); }
        for (const std::string &m : merges) { mp.push_back(m.data()); ml.push_back((int32_t)m.size()); }
        h = biogpt_hip_vocab_create(tp.data(), tl.data(), n_tok, mp.data(), ml.data(), (int32_t)merges.size());
    }
    ~VocabHandle() { biogpt_hip_vocab_free(h); }
};
}  // namespace

token_sequence gpt_tokenize(biogpt_vocab &vocab, const std::string &text, const std::string &lang) {
    VocabHandle v(vocab);
    if (!v.h) throw std::runtime_error(biogpt_hip_last_error());
    token_sequence ids(64);
    int n = biogpt_hip_tokenize(v.h, text.c_str(), lang.c_str(), ids.data(), (int32_t)ids.size());
    if (n > (int)ids.size()) {
        ids.resize((size_t)n);
        n = biogpt_hip_tokenize(v.h, text.c_str(), lang.c_str(), ids.data(), (int32_t)ids.size());
    }
    if (n == BIOGPT_HIP_E_LENGTH) throw std::length_error("basic_string::_M_create");   // what the reference throws (mosestokenizer.cpp:264)
    if (n < 0) throw std::runtime_error(biogpt_hip_last_error());
    ids.resize((size_t)n);
    return ids;
}

std::string gpt_decode(std::vector<std::string> &tokens, const std::string &lang) {
    std::string joined;
    for (size_t i = 0; i < tokens.size(); i++) { if (i) joined += '\n'; joined += tokens[i]; }
    std::string out(joined.size() + 64, '\0');
    int n = biogpt_hip_decode_strings(joined.c_str(), lang.c_str(), &out[0], (int32_t)out.size());
    if (n + 1 > (int)out.size()) {
        out.assign((size_t)n + 1, '\0');
        n = biogpt_hip_decode_strings(joined.c_str(), lang.c_str(), &out[0], (int32_t)out.size());
    }
    if (n < 0) throw std::runtime_error(biogpt_hip_last_error());
    out.resize((size_t)n);
    // side effect of the reference: the caller's vector is rewritten in place (std::transform onto itself, biogpt.cpp:879-884)
    for (std::string &t : tokens) {
        t.erase(std::remove(t.begin(), t.end(), ' '), t.end());
        for (const char *tag : {"</w>", "</s>"})
            for (size_t at; (at = t.find(tag)) != std::string::npos;) t.replace(at, 4, " ");
    }
    return out;
}

struct ggml_cgraph *biogpt_graph(const biogpt_model &, struct ggml_allocr *, const token_sequence &, const int) {
    return reinterpret_cast<ggml_cgraph *>(&g_graph_tag);  // main.cpp:59 only feeds it to ggml_allocr_alloc_graph
}

bool biogpt_eval(const biogpt_model &model, const token_sequence &embed_inp, std::vector<float> &logits,
                 struct ggml_allocr *, const int n_past, const int /*n_threads: CPU-backend knob*/) {
    biogpt_hip_ctx *ctx = biogpt_model_hip(model);
    logits.resize((size_t)model.hparams.n_vocab);  // biogpt.cpp:843
    return biogpt_hip_eval(ctx, embed_inp.data(), (int32_t)embed_inp.size(), n_past, logits.data()) == 0;
}

// temperature -> top-k (partial sort) -> softmax in double -> top-p cut + renormalise -> one draw
// (biogpt.cpp:908-980).  With top_k == 1 this is arg-max, the draw is still consumed.
namespace {
typedef std::pair<double, biogpt_vocab::id> scored;
// everything after the partial sort: cand = the top_k (scaled logit, id) pairs, best first (biogpt.cpp:938-980)
biogpt_vocab::id sample_from_top(std::vector<scored> &cand, double top_p, std::mt19937 &rng) {
    int top_k = (int)cand.size();
    double peak = -INFINITY;
    for (const scored &c : cand) peak = std::max(peak, c.first);
    std::vector<double> prob(cand.size());
    double total = 0.0;
    for (size_t i = 0; i < cand.size(); i++) { prob[i] = std::exp(cand[i].first - peak); total += prob[i]; }
    for (double &p : prob) p /= total;
    if (top_p < 1.0f) {
        double run = 0.0;
        for (int i = 0; i < top_k; i++) {
            run += prob[(size_t)i];
            if (run >= top_p) { prob.resize((size_t)i + 1); cand.resize((size_t)i + 1); break; }
        }
        const double renorm = 1.0 / run;
        for (double &p : prob) p *= renorm;
    }
    std::discrete_distribution<> pick(prob.begin(), prob.end());
    return cand[(size_t)pick(rng)].second;
}
}  // namespace

// biogpt.cpp:908-980.  The reference copies all n logits into (score, id) pairs and std::partial_sort's them (~0.3 ms for 42 k entries: more than a decode
// step takes on the MI355X).  Same selection in ONE pass over the floats: the k best so far are kept in descending order (equal values: lower id first, the
// rule of oracle/sampler.py and of the device-side top-k; the reference's partial_sort leaves ties unspecified: with EQUAL logits at the k-th place the
// sampled id can differ from what a particular build of the reference draws -- a deviation inside what the reference itself leaves open), a block of 16 logits is only looked at when
// its maximum beats the current k-th value.  Scaling by 1 / temp is monotone, so it is applied to the k survivors only.
biogpt_vocab::id biogpt_sample_top_k_top_p(const biogpt_vocab &vocab, const float *logits, int top_k, double top_p,
                                           double temp, std::mt19937 &rng) {
    const int n = (int)vocab.id_to_token.size();
    top_k = std::max(1, std::min(top_k, n));
    const double inv_t = 1.0 / temp;
    if (!(inv_t > 0.0) || top_k > 256) {      // temp <= 0 / NaN reverses or destroys the order, large k: the reference's own way
        std::vector<scored> cand((size_t)n);
        for (int i = 0; i < n; i++) cand[(size_t)i] = scored(logits[i] * inv_t, i);
        std::partial_sort(cand.begin(), cand.begin() + top_k, cand.end(), [](const scored &a, const scored &b) { return a.first > b.first; });
        cand.resize((size_t)top_k);
        return sample_from_top(cand, top_p, rng);
    }
    float bv[256];
    int bi[256];
    int have = 0;
    float thr = -INFINITY;      // value of the k-th best once k are held; a later (higher-id) logit must be strictly greater to enter
    auto offer = [&](float v, int i) {
        if (have == top_k && !(v > thr)) return;
        if (v != v) return;      // NaN never compares greater (as in the reference's comparator)
        int pos = have < top_k ? have : top_k - 1;
        while (pos > 0 && bv[pos - 1] < v) { bv[pos] = bv[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
        bv[pos] = v; bi[pos] = i;
        if (have < top_k) have++;
        if (have == top_k) thr = bv[top_k - 1];
    };
    int i = 0;
    for (; i < n && have < top_k; i++) offer(logits[i], i);
    for (; i + 16 <= n; i += 16) {
        float m = -INFINITY;      // (not logits[i]: a NaN there would hide the whole block -- every comparison with it is false)
        for (int j = 0; j < 16; j++) m = logits[i + j] > m ? logits[i + j] : m;
        if (m > thr)
            for (int j = 0; j < 16; j++) offer(logits[i + j], i + j);
    }
    for (; i < n; i++) offer(logits[i], i);
    std::vector<scored> cand((size_t)have);
    for (int k = 0; k < have; k++) cand[(size_t)k] = scored(bv[k] * inv_t, bi[k]);
    return sample_from_top(cand, top_p, rng);
}

// Extension (no reference counterpart): biogpt_eval + biogpt_sample_top_k_top_p in one call, the top-k selection done on
// the device (biogpt_hip_eval_topk) so that 512 bytes instead of the 170 KB logits row cross PCIe per token.  Same ids
// as the two reference calls whenever the k-th and (k+1)-th logits differ (ties are "lower id first" here and
// unspecified in the reference's std::partial_sort).  top_k > 64 falls back to the two-call form.
biogpt_vocab::id biogpt_eval_sample_top_k_top_p(const biogpt_model &model, const biogpt_vocab &vocab, const token_sequence &embed_inp,
                                                const int n_past, int top_k, double top_p, double temp, std::mt19937 &rng) {
    biogpt_hip_ctx *ctx = biogpt_model_hip(model);
    const int n = (int)vocab.id_to_token.size();
    top_k = std::max(1, std::min(top_k, n));
    if (top_k > 64) {
        std::vector<float> logits;
        if (!biogpt_eval(model, embed_inp, logits, nullptr, n_past, 1)) throw std::runtime_error(biogpt_hip_last_error());
        return biogpt_sample_top_k_top_p(vocab, logits.data(), top_k, top_p, temp, rng);
    }
    float vals[64];
    int32_t ids[64];
    const int got = biogpt_hip_eval_topk(ctx, embed_inp.data(), (int32_t)embed_inp.size(), n_past, top_k, vals, ids);
    if (got < 0) throw std::runtime_error(biogpt_hip_last_error());
    std::vector<scored> cand((size_t)got);
    const double inv_t = 1.0 / temp;
    for (int i = 0; i < got; i++) cand[(size_t)i] = scored(vals[i] * inv_t, ids[i]);
    return sample_from_top(cand, top_p, rng);
}

// ---- CLI flags of the reference's `biogpt` tool (biogpt.cpp:982-1040), table-driven -------------------
namespace {
enum FlagKind { F_I32, F_F32, F_U8, F_STR };
struct Flag {
    const char *short_name, *long_name, *meta, *help;
    FlagKind kind;
    size_t offset;
};
#define FLAG(sn, ln, meta, help, kind, member) {sn, ln, meta, help, kind, offsetof(biogpt_params, member)}
const Flag kFlags[] = {
    FLAG("-s", "--seed", "SEED", "RNG seed", F_I32, seed),
    FLAG("-t", "--threads", "N", "number of threads (ignored by the MI355X engine)", F_I32, n_threads),
    FLAG("-p", "--prompt", "PROMPT", "prompt to start generation with", F_STR, prompt),
    FLAG("-l", "--lang", "LANG", "language of the prompt; like the reference this overwrites the PROMPT (F9)", F_STR, prompt),
    FLAG("-n", "--n_predict", "N", "number of tokens to predict", F_I32, n_predict),
    FLAG("-v", "--verbosity", "V", "verbosity level", F_U8, verbosity),
    FLAG(nullptr, "--top_k", "N", "top-k sampling", F_I32, top_k),
    FLAG(nullptr, "--top_p", "N", "top-p sampling", F_F32, top_p),
    FLAG(nullptr, "--temp", "N", "temperature", F_F32, temp),
    FLAG("-b", "--batch_size", "N", "batch size for prompt processing", F_I32, n_batch),
    FLAG("-m", "--model", "FNAME", "model path", F_STR, model),
};
#undef FLAG
}  // namespace

bool biogpt_params_parse(int argc, char **argv, biogpt_params &params) {
    for (int i = 1; i < argc; i++) {
        const std::string arg = argv[i];
        if (arg == "-h" || arg == "--help") { biogpt_print_usage(argv, params); exit(0); }
        const Flag *hit = nullptr;
        for (const Flag &f : kFlags)
            if ((f.short_name && arg == f.short_name) || arg == f.long_name) { hit = &f; break; }
        if (!hit) {  // biogpt.cpp:1011-1015
            fprintf(stderr, "error: unknown argument: %s\n", arg.c_str());
            biogpt_print_usage(argv, params);
            exit(0);
        }
        if (i + 1 >= argc) {  // the reference reads argv[++i] unchecked; fail cleanly instead
            fprintf(stderr, "error: missing value for %s\n", arg.c_str());
            return false;
        }
        char *base = reinterpret_cast<char *>(&params) + hit->offset;
        const char *val = argv[++i];
        switch (hit->kind) {
            case F_I32: *reinterpret_cast<int32_t *>(base) = std::stoi(val); break;
            case F_F32: *reinterpret_cast<float *>(base) = std::stof(val); break;
            case F_U8: *reinterpret_cast<uint8_t *>(base) = (uint8_t)std::stoi(val); break;
            case F_STR: *reinterpret_cast<std::string *>(base) = val; break;
        }
    }
    return true;
}

void biogpt_print_usage(char **argv, const biogpt_params &params) {
    fprintf(stderr, "usage: %s [options]\n\noptions:\n  -h, --help            show this help message and exit\n", argv[0]);
    for (const Flag &f : kFlags) {
        std::string names = f.short_name ? std::string(f.short_name) + " " + f.meta + ", " : std::string();
        names += std::string(f.long_name) + " " + f.meta;
        const char *base = reinterpret_cast<const char *>(&params) + f.offset;
        std::string dflt;
        switch (f.kind) {
            case F_I32: dflt = std::to_string(*reinterpret_cast<const int32_t *>(base)); break;
            case F_F32: { char b[32]; snprintf(b, sizeof b, "%.1f", *reinterpret_cast<const float *>(base)); dflt = b; } break;
            case F_U8: dflt = std::to_string((int)*reinterpret_cast<const uint8_t *>(base)); break;
            case F_STR: dflt = *reinterpret_cast<const std::string *>(base); break;
        }
        fprintf(stderr, "  %-28s %s (default: %s)\n", names.c_str(), f.help, dflt.c_str());
    }
    fprintf(stderr, "\n");
}
