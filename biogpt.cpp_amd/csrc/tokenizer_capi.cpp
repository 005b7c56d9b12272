// extern "C" entry points of the tokenizer (include/biogpt_hip.h, "text <-> ids").  Host-only.
#include <algorithm>
#include <cstring>
#include <mutex>

#include "../../include/biogpt_hip.h"
#include "host_common.h"
#include "model_file.h"
#include "tokenizer.h"

struct biogpt_hip_vocab {
    bgtok::Vocab v;
    biogpt_hip_vocab(const std::vector<std::string> &t, const std::vector<std::string> &m) : v(t, m) {}
};

namespace {
std::mutex g_tok_mutex;   // the shared MosesTokenizer caches prefix lists

// strings: returns the byte length (without NUL); the bytes + NUL are written only if they fit
int hand_out(const std::string &s, char *out, int32_t cap) {
    if (out && cap > 0 && (size_t)cap >= s.size() + 1) {
        std::memcpy(out, s.data(), s.size());
        out[s.size()] = 0;
    }
    return (int)s.size();
}
std::vector<std::string> lines_of(const char *nl_joined) {
    std::vector<std::string> out;
    if (!nl_joined || !*nl_joined) return out;
    const char *p = nl_joined;
    for (;;) {
        const char *e = std::strchr(p, '\n');
        if (!e) { out.push_back(p); break; }
        out.push_back(std::string(p, (size_t)(e - p)));
        p = e + 1;
    }
    return out;
}
}  // namespace

namespace bg {
biogpt_hip_vocab *make_vocab(const std::vector<std::string> &tokens, const std::vector<std::string> &merges) {
    return new biogpt_hip_vocab(tokens, merges);
}
void drop_vocab(biogpt_hip_vocab *v) { delete v; }
}  // namespace bg

extern "C" {

biogpt_hip_vocab *biogpt_hip_vocab_load(const char *fname) {
    bg::clear_error();
    if (!fname) BG_FAIL(nullptr, "null file name");
    bg::ModelFile mf;
    if (!mf.open(fname)) return nullptr;
    return new biogpt_hip_vocab(mf.vocab, mf.merges);
}

biogpt_hip_vocab *biogpt_hip_vocab_create(const char *const *tokens, const int32_t *token_lens, int32_t n_tokens,
                                          const char *const *merges, const int32_t *merge_lens, int32_t n_merges) {
    bg::clear_error();
    if (n_tokens < 0 || n_merges < 0 || (n_tokens && (!tokens || !token_lens)) || (n_merges && (!merges || !merge_lens)))
        BG_FAIL(nullptr, "bad argument");
    std::vector<std::string> t((size_t)n_tokens), m((size_t)n_merges);
    for (int32_t i = 0; i < n_tokens; i++) t[(size_t)i].assign(tokens[i] ? tokens[i] : "", (size_t)std::max(0, token_lens[i]));
    for (int32_t i = 0; i < n_merges; i++) m[(size_t)i].assign(merges[i] ? merges[i] : "", (size_t)std::max(0, merge_lens[i]));
    return new biogpt_hip_vocab(t, m);
}

void biogpt_hip_vocab_free(biogpt_hip_vocab *v) { delete v; }

int biogpt_hip_tokenizer_set_data_dir(const char *dir) {
    if (!dir) return -1;
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    bgtok::default_moses().set_data_dir(dir);
    return 0;
}

int biogpt_hip_moses_tokenize(const char *text, const char *lang, char *out, int32_t cap) {
    bg::clear_error();
    if (!text || !lang) BG_FAIL(-1, "null argument");
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    try {
        const std::vector<std::string> t = bgtok::default_moses().tokenize(text, lang);
        std::string j;
        for (size_t i = 0; i < t.size(); i++) { if (i) j += '\n'; j += t[i]; }
        return hand_out(j, out, cap);
    } catch (const std::length_error &) {
        BG_FAIL(BIOGPT_HIP_E_LENGTH, "a sentence-final period is followed by a non-ASCII word (the reference throws std::length_error here)");
    }
}

int biogpt_hip_moses_detokenize(const char *tokens_nl, const char *lang, char *out, int32_t cap) {
    bg::clear_error();
    if (!tokens_nl || !lang) BG_FAIL(-1, "null argument");
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    return hand_out(bgtok::default_moses().detokenize(lines_of(tokens_nl), lang), out, cap);
}

int biogpt_hip_bpe(const biogpt_hip_vocab *v, const char *word, char *out, int32_t cap) {
    bg::clear_error();
    if (!v || !word) BG_FAIL(-1, "null argument");
    if (!*word) BG_FAIL(-1, "empty word");
    return hand_out(v->v.bpe(word), out, cap);
}

int biogpt_hip_tokenize(const biogpt_hip_vocab *v, const char *text, const char *lang, int32_t *out_ids, int32_t cap) {
    bg::clear_error();
    if (!v || !text || !lang) BG_FAIL(-1, "null argument");
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    try {
        const std::vector<int32_t> ids = v->v.encode(bgtok::default_moses(), text, lang);
        for (size_t i = 0; i < ids.size() && (int32_t)i < cap && out_ids; i++) out_ids[i] = ids[i];
        return (int)ids.size();
    } catch (const std::length_error &) {
        BG_FAIL(BIOGPT_HIP_E_LENGTH, "a sentence-final period is followed by a non-ASCII word (the reference throws std::length_error here)");
    }
}

int biogpt_hip_decode(const biogpt_hip_vocab *v, const int32_t *ids, int32_t n, const char *lang, char *out, int32_t cap) {
    bg::clear_error();
    if (!v || (!ids && n > 0) || n < 0 || !lang) BG_FAIL(-1, "bad argument");
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    return hand_out(v->v.decode(bgtok::default_moses(), ids, n, lang), out, cap);
}

int biogpt_hip_decode_strings(const char *tokens_nl, const char *lang, char *out, int32_t cap) {
    bg::clear_error();
    if (!tokens_nl || !lang) BG_FAIL(-1, "null argument");
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    return hand_out(bgtok::decode_token_strings(bgtok::default_moses(), lines_of(tokens_nl), lang), out, cap);
}

int biogpt_hip_tokenizer_byte_class(int which, uint8_t *out256) {
    if (!out256 || which < 0 || which > 4) return -1;
    std::lock_guard<std::mutex> lock(g_tok_mutex);
    const bgtok::CharClasses &c = bgtok::default_moses().classes();
    const bgtok::ByteSet *sets[5] = {&c.alnum, &c.alpha, &c.lower, &c.num, &c.sc};
    for (int b = 0; b < 256; b++) out256[b] = sets[which]->has((unsigned char)b) ? 1 : 0;
    return 0;
}
}
