// Text <-> token ids for BioGPT: Moses-style word splitting, byte-level BPE, vocabulary
// lookup and the inverse.  Host-only C++ (no device work: a prompt is tokenized once per request).
//
// Behavioural contract = the reference's tokenizer stack, byte for byte:
//   moses_tokenize    mosestokenizer.cpp:290-358      bpe            bpe.cpp:20-91
//   moses_detokenize  mosestokenizer.cpp:360-466      gpt_tokenize   biogpt.cpp:850-875
//   merges -> ranks   biogpt.cpp:131-155              gpt_decode     biogpt.cpp:877-906
// including its quirks (byte-wise "Unicode" classes, '#NUMERIC_ONLY#' never seen, XML un-escaping that
// does nothing, the std::length_error it throws in front of a non-ASCII word).  The reference drives all
// of it through std::regex objects rebuilt on every call over 30 KB bracket expressions; here every step
// is a single pass over the bytes with 256-bit membership tables, and the prefix lists are read once.
// tests/test_tokenizer.py checks it against the reference's own sources compiled in the build container
// (oracle/_ref) and against golden vectors generated from them.
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace bgtok {

// membership table over byte values
struct ByteSet {
    uint64_t w[4] = {0, 0, 0, 0};
    bool has(unsigned char c) const { return (w[c >> 6] >> (c & 63)) & 1u; }
    void add(unsigned char c) { w[c >> 6] |= (uint64_t)1 << (c & 63); }
    void add_range(int lo, int hi) { for (int c = lo; c <= hi; c++) add((unsigned char)c); }
    void add_bytes(const std::string &s) { for (unsigned char c : s) add(c); }
};

// The five perluniprops classes the reference splices into bracket expressions (mosestokenizer.cpp:99-104).
// std::regex over char matches BYTES, so each class is exactly the set of byte values occurring in its file.
struct CharClasses {
    ByteSet alnum, alpha, lower, num, sc;
    static CharClasses builtin();                       // the sets of the data files shipped with the reference
    bool load_dir(const std::string &perluniprops_dir); // recompute from <dir>/Is*.txt; false if any is missing
};

class MosesTokenizer {
public:
    MosesTokenizer();
    // Directory holding nonbreaking_prefixes/ (and optionally perluniprops/): the reference's data/.
    // Default: $BIOGPT_DATA_DIR, else "../data" relative to the working directory (mosestokenizer.cpp:11-12).
    void set_data_dir(const std::string &dir);
    const std::string &data_dir() const { return data_dir_; }

    // Throws std::length_error exactly where the reference does (a word ending in '.' that is neither an
    // abbreviation nor a listed prefix, followed by a word starting with a byte >= 0x80; mosestokenizer.cpp:264).
    std::vector<std::string> tokenize(const std::string &text, const std::string &lang);
    std::string detokenize(const std::vector<std::string> &tokens, const std::string &lang) const;

    const CharClasses &classes() const { return cls_; }

private:
    const std::vector<std::string> &prefixes(const std::string &lang);
    std::string split_sentence_final_periods(const std::string &text, const std::string &lang);

    CharClasses cls_;
    std::string data_dir_;
    std::map<std::string, std::vector<std::string>> prefix_cache_;  // by file name
};

struct PairHash {
    size_t operator()(const std::pair<std::string, std::string> &p) const {
        return std::hash<std::string>()(p.first) * 1000003u ^ std::hash<std::string>()(p.second);
    }
};

// vocabulary + merge ranks of one model file
class Vocab {
public:
    // tokens: id -> bytes; merges: rank -> raw merge string as stored in the file ("left right")
    Vocab(const std::vector<std::string> &tokens, const std::vector<std::string> &merges);

    std::string bpe(const std::string &word) const;   // pieces joined by ' ', last piece carries "</w>"
    // ids of `text`: always starts with 2 ("</s>"); unknown pieces are dropped with a warning on stderr
    std::vector<int32_t> encode(MosesTokenizer &moses, const std::string &text, const std::string &lang) const;
    // inverse used by the CLI: ids -> printable text (ids outside the table decode as empty, like the reference's map lookup)
    std::string decode(const MosesTokenizer &moses, const int32_t *ids, int32_t n, const std::string &lang) const;

    int32_t n_tokens() const { return (int32_t)id_to_token_.size(); }
    int32_t n_ranks() const { return (int32_t)ranks_.size(); }
    int rank_of(const std::string &a, const std::string &b) const;   // -1 if unranked

private:
    std::vector<std::string> id_to_token_;
    std::unordered_map<std::string, int32_t> token_to_id_;
    std::unordered_map<std::pair<std::string, std::string>, int, PairHash> ranks_;
};

// gpt_decode on token strings (biogpt.cpp:877-906)
std::string decode_token_strings(const MosesTokenizer &moses, const std::vector<std::string> &tokens, const std::string &lang);

MosesTokenizer &default_moses();   // process-wide instance used by the C-ABI (guarded by its own mutex there)

}  // namespace bgtok
