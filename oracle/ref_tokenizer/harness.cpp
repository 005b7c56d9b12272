// oracle/ref_tokenizer/harness.cpp -- C entry points around the REFERENCE's own tokenizer sources.
//
// TEST INFRASTRUCTURE ONLY.  The reference's tokenizer stack (mosestokenizer.cpp, bpe.cpp) does not depend
// on ggml, so -- unlike the forward pass -- it compiles from its own two source files.  oracle/Makefile
// compiles them from where they lie under /root/reference (never copied into this repository) together
// with this harness into oracle/_ref/libref_tokenizer.so.  The harness only declares what the reference
// headers declare (mosestokenizer.h:15-17, bpe.h:8-10) and restates the two 25-line glue functions that
// live in biogpt.cpp (which needs ggml and cannot be compiled): gpt_tokenize biogpt.cpp:850-875 and
// gpt_decode biogpt.cpp:877-906 are mirrored below against the SAME moses_tokenize / bpe / moses_detokenize.
//
// The reference reads ../data/perluniprops/*.txt at static-initialisation time and
// ../data/nonbreaking_prefixes/* on every call, both relative to the current directory: the python side
// (oracle/ref_tokenizer.py) changes into <data root>/run before dlopen and around every call.
#include <cstring>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "bpe.h"
#include "mosestokenizer.h"

namespace {
int put(const std::string &s, char *out, int cap) {
    if ((int)s.size() + 1 > cap) return -(int)s.size() - 100;  // buffer too small
    std::memcpy(out, s.data(), s.size());
    out[s.size()] = 0;
    return (int)s.size();
}
struct RefVocab {
    std::map<word_pair, int> ranks;
    std::map<std::string, int> token_to_id;
};
}  // namespace

extern "C" {

// tokens joined by '\n'.  Returns bytes written; -1 std::length_error (the reference's
// std::string(char, 1) quirk, mosestokenizer.cpp:264), -2 any other exception.
int ref_moses_tokenize(const char *text, const char *lang, char *out, int cap) {
    try {
        std::vector<std::string> t = moses_tokenize(text, lang);
        std::string j;
        for (size_t i = 0; i < t.size(); i++) { if (i) j += '\n'; j += t[i]; }
        return put(j, out, cap);
    } catch (const std::length_error &) { return -1; } catch (...) { return -2; }
}

int ref_moses_detokenize(const char *tokens_nl, const char *lang, char *out, int cap) {
    try {
        std::vector<std::string> t;
        std::stringstream ss(tokens_nl);
        std::string line;
        while (std::getline(ss, line, '\n')) t.push_back(line);
        return put(moses_detokenize(t, lang), out, cap);
    } catch (...) { return -2; }
}

void *ref_vocab_new(void) { return new RefVocab(); }
void ref_vocab_free(void *v) { delete static_cast<RefVocab *>(v); }
void ref_vocab_add_merge(void *v, const char *a, const char *b, int rank) { static_cast<RefVocab *>(v)->ranks[word_pair(a, b)] = rank; }
void ref_vocab_add_token(void *v, const char *tok, int id) { static_cast<RefVocab *>(v)->token_to_id[tok] = id; }

int ref_bpe(void *v, const char *word, char *out, int cap) {
    try { return put(bpe(word, static_cast<RefVocab *>(v)->ranks), out, cap); } catch (...) { return -2; }
}

// gpt_tokenize (biogpt.cpp:850-875): ids written to out_ids (cap entries), returns the count, negative on exception
int ref_gpt_tokenize(void *v, const char *text, const char *lang, int *out_ids, int cap) {
    RefVocab *rv = static_cast<RefVocab *>(v);
    try {
        std::vector<std::string> words = moses_tokenize(text, lang);
        std::vector<int> ids(1, 2);
        for (const std::string &w : words) {
            std::stringstream ss(bpe(w, rv->ranks));
            std::string piece;
            while (ss >> piece) {
                std::map<std::string, int>::const_iterator it = rv->token_to_id.find(piece);
                if (it != rv->token_to_id.end()) ids.push_back(it->second);
            }
        }
        for (int i = 0; i < (int)ids.size() && i < cap; i++) out_ids[i] = ids[(size_t)i];
        return (int)ids.size();
    } catch (const std::length_error &) { return -1; } catch (...) { return -2; }
}

// gpt_decode (biogpt.cpp:877-906) on token strings joined by '\n'
int ref_gpt_decode(const char *tokens_nl, const char *lang, char *out, int cap) {
    try {
        std::string joined;
        std::stringstream in(tokens_nl);
        std::string t;
        while (std::getline(in, t, '\n')) {
            std::string u;
            for (char c : t) if (c != ' ') u += c;
            for (const char *tag : {"</w>", "</s>"}) {
                size_t p;
                while ((p = u.find(tag)) != std::string::npos) u.replace(p, 4, " ");
            }
            joined += u;
        }
        std::vector<std::string> clean;
        std::stringstream ss(joined);
        while (ss >> t) clean.push_back(t);
        return put(moses_detokenize(clean, lang), out, cap);
    } catch (...) { return -2; }
}
}
