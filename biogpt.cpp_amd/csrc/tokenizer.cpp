// See tokenizer.h.  Every function names the reference lines whose observable behaviour it reproduces; the
// implementation shares nothing with them (no std::regex: each rewrite rule is one left-to-right pass that
// emulates "leftmost match, then continue after it", which is what regex_replace does for these patterns).
#include "tokenizer.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace bgtok {
namespace {

// \s of std::regex<char> / operator>> in the classic locale
inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

std::string squeeze_spaces(const std::string &s) {   // every whitespace run becomes one ' '
    std::string out;
    out.reserve(s.size());
    for (size_t i = 0; i < s.size();) {
        if (is_space((unsigned char)s[i])) {
            while (i < s.size() && is_space((unsigned char)s[i])) i++;
            out += ' ';
        } else {
            out += s[i++];
        }
    }
    return out;
}

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && is_space((unsigned char)s[a])) a++;
    while (b > a && is_space((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

void substitute(std::string &s, const std::string &from, const std::string &to) {   // all non-overlapping, left to right
    std::string out;
    size_t pos = 0;
    for (;;) {
        const size_t hit = s.find(from, pos);
        if (hit == std::string::npos) break;
        out.append(s, pos, hit - pos);
        out += to;
        pos = hit + from.size();
    }
    if (pos == 0) return;
    out.append(s, pos, std::string::npos);
    s.swap(out);
}

std::vector<std::string> words_of(const std::string &s) {   // operator>> splitting
    std::vector<std::string> out;
    size_t i = 0;
    const size_t n = s.size();
    while (i < n) {
        while (i < n && is_space((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < n && !is_space((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

// "..."-runs are hidden behind marker words while commas/apostrophes/periods are processed
// (mosestokenizer.cpp:183-199; only the first dot after the marker is ever folded into it, the rest stay)
void hide_dot_runs(std::string &s) {
    std::string a;
    for (size_t i = 0; i < s.size();) {
        if (s[i] == '.' && i + 1 < s.size() && s[i + 1] == '.') {
            size_t j = i;
            while (j < s.size() && s[j] == '.') j++;
            a += "DOTMULTI";
            a.append(j - i - 1, '.');
            i = j;
        } else {
            a += s[i++];
        }
    }
    static const std::string one = "DOTMULTI.";
    std::string b;
    for (size_t i = 0; i < a.size();) {
        const size_t nx = i + one.size();
        if (a.compare(i, one.size(), one) == 0 && nx < a.size() && a[nx] != '.') {
            b += "DOTDOTMULTI ";
            b += a[nx];
            i = nx + 1;
        } else {
            b += a[i++];
        }
    }
    substitute(b, one, "DOTDOTMULTI");
    s.swap(b);
}

void show_dot_runs(std::string &s) {   // mosestokenizer.cpp:200-206
    while (s.find("DOTDOTMULTI") != std::string::npos) substitute(s, "DOTDOTMULTI", "DOTMULTI.");
    substitute(s, "DOTMULTI", ".");
}

// one rewrite of  L ' R  triples; `mid` is what replaces the apostrophe between the two kept bytes
template <class L, class R>
std::string rewrite_apostrophes(const std::string &s, L left, R right, const char *mid) {
    std::string out;
    const size_t n = s.size();
    for (size_t i = 0; i < n;) {
        if (i + 2 < n && s[i + 1] == '\'' && left((unsigned char)s[i]) && right((unsigned char)s[i + 2])) {
            out += s[i];
            out += mid;
            out += s[i + 2];
            i += 3;
        } else {
            out += s[i++];
        }
    }
    return out;
}

std::string read_file(const std::string &path, bool *ok) {
    std::ifstream f(path, std::ios::binary);
    *ok = (bool)f;
    std::stringstream ss;
    if (f) ss << f.rdbuf();
    return ss.str();
}

}  // namespace

// ---- character classes -------------------------------------------------------------------------------
CharClasses CharClasses::builtin() {
    // Byte values occurring in data/perluniprops/Is{Alnum,Alpha,Lower,N,Sc}.txt as shipped with the reference
    // (tests/test_tokenizer.py re-derives them from the files when the reference tree is present).  The files
    // end in '\n', which is why 0x0A counts as a digit and as a currency sign.
    CharClasses c;
    auto high = [](ByteSet &s, std::initializer_list<std::pair<int, int>> r) { for (auto &p : r) s.add_range(p.first, p.second); };
    high(c.alpha, {{'A', 'Z'}, {'a', 'z'}, {0x80, 0xBF}, {0xC2, 0xCB}, {0xCD, 0xE3}, {0xEA, 0xEA}, {0xED, 0xED}, {0xEF, 0xEF}});
    c.alnum = c.alpha;
    c.alnum.add_range('0', '9');
    high(c.lower, {{'a', 'z'}, {0x80, 0xBF}, {0xC2, 0xCB}, {0xCD, 0xD6}, {0xE1, 0xE2}, {0xEA, 0xEA}, {0xEF, 0xF0}});
    high(c.num, {{0x0A, 0x0A}, {'0', '9'}, {0x80, 0xBF}, {0xC2, 0xC2}, {0xD9, 0xD9}, {0xDB, 0xDB}, {0xDF, 0xE3}, {0xEA, 0xEA}, {0xEF, 0xEF}});
    high(c.sc, {{0x0A, 0x0A}, {'$', '$'}, {0x82, 0x82}, {0x84, 0x84}, {0x8B, 0x8B}, {0x8F, 0x8F}, {0x9B, 0x9B}, {0x9F, 0xBD},
                {0xBF, 0xBF}, {0xC2, 0xC2}, {0xD6, 0xD6}, {0xD8, 0xD8}, {0xE0, 0xE2}, {0xEA, 0xEA}, {0xEF, 0xEF}});
    return c;
}

bool CharClasses::load_dir(const std::string &dir) {
    CharClasses c;
    struct { const char *name; ByteSet *dst; } files[] = {
        {"IsAlnum", &c.alnum}, {"IsAlpha", &c.alpha}, {"IsLower", &c.lower}, {"IsN", &c.num}, {"IsSc", &c.sc}};
    for (auto &f : files) {
        bool ok = false;
        const std::string bytes = read_file(dir + "/" + f.name + ".txt", &ok);
        if (!ok) return false;
        f.dst->add_bytes(bytes);
    }
    *this = c;
    return true;
}

// ---- Moses tokenizer ---------------------------------------------------------------------------------
MosesTokenizer::MosesTokenizer() : cls_(CharClasses::builtin()) {
    const char *env = std::getenv("BIOGPT_DATA_DIR");
    set_data_dir(env && *env ? env : "../data");
}

void MosesTokenizer::set_data_dir(const std::string &dir) {
    data_dir_ = dir;
    prefix_cache_.clear();
    cls_ = CharClasses::builtin();
    cls_.load_dir(dir + "/perluniprops");   // optional: a user's own class files win over the built-in sets
}

// mosestokenizer.cpp:14-61.  '#' starts a comment, which also swallows the "#NUMERIC_ONLY#" tag: the
// reference therefore never sees a numeric-only prefix and treats "No." like "Mr.".  With lang "" it would
// add every language named in data/nonbreaking_prefixes/AVAILABLE_LANGUAGES, a file its data/ does not have.
const std::vector<std::string> &MosesTokenizer::prefixes(const std::string &lang) {
    const std::string file = "nonbreaking_prefix." + (lang.empty() ? std::string("en") : lang);
    auto it = prefix_cache_.find(file);
    if (it != prefix_cache_.end()) return it->second;
    std::vector<std::string> &list = prefix_cache_[file];
    std::ifstream f(data_dir_ + "/nonbreaking_prefixes/" + file);
    std::string line;
    while (std::getline(f, line)) {
        line = line.substr(0, line.find('#'));
        if (!line.empty()) list.push_back(trim(line));
    }
    return list;
}

// mosestokenizer.cpp:237-287: "word." keeps its period when the word looks like an abbreviation ("e.g") or is a
// listed prefix; otherwise the period is split off.  The third escape of the reference -- next word starts
// in lower case -- builds std::string(first_char, 1), i.e. `first_char` copies of '\x01': never lower case
// for ASCII, and a length_error for a byte >= 0x80 (negative char -> huge count).
std::string MosesTokenizer::split_sentence_final_periods(const std::string &text, const std::string &lang) {
    std::vector<std::string> tok;   // pieces between whitespace runs: a leading empty piece is kept, a trailing one is not
    {
        size_t start = 0, i = 0;
        bool cut = false;
        while (i < text.size()) {
            if (is_space((unsigned char)text[i])) {
                tok.push_back(text.substr(start, i - start));
                while (i < text.size() && is_space((unsigned char)text[i])) i++;
                start = i;
                cut = true;
            } else {
                i++;
            }
        }
        if (start < text.size() || !cut) tok.push_back(text.substr(start));
    }
    const std::vector<std::string> &listed = prefixes(lang);
    for (size_t i = 0; i < tok.size(); i++) {
        const std::string &t = tok[i];
        if (t.size() < 2 || t.back() != '.') continue;
        const std::string stem = t.substr(0, t.size() - 1);
        bool abbreviation = false;
        if (stem.find('.') != std::string::npos)
            for (unsigned char c : stem) if (cls_.alpha.has(c)) { abbreviation = true; break; }
        if (abbreviation || std::find(listed.begin(), listed.end(), stem) != listed.end()) continue;
        if (i + 1 < tok.size() && !tok[i + 1].empty() && ((unsigned char)tok[i + 1][0] & 0x80))
            throw std::length_error("basic_string::_M_create");
        tok[i] = stem + " .";
    }
    std::string out;
    for (size_t i = 0; i < tok.size(); i++) { if (i) out += ' '; out += tok[i]; }
    return out;
}

std::vector<std::string> MosesTokenizer::tokenize(const std::string &text, const std::string &lang) {
    const CharClasses &k = cls_;
    std::string s = squeeze_spaces(text);                                   // :294
    s.erase(std::remove_if(s.begin(), s.end(), [](char c) { return (unsigned char)c < 0x20; }), s.end());  // :295
    s = trim(s);                                                            // :298
    {                                                                       // :301 pad what is not a word byte
        std::string o;
        for (unsigned char c : s) {
            if (k.alnum.has(c) || is_space(c) || c == '.' || c == '\'' || c == '`' || c == ',' || c == '-') o += (char)c;
            else { o += ' '; o += (char)c; o += ' '; }
        }
        s.swap(o);
    }
    {                                                                       // :304 a-b -> a @-@ b
        std::string o;
        for (size_t i = 0; i < s.size(); i++) {
            if (s[i] == '-' && i > 0 && i + 1 < s.size() && k.alnum.has((unsigned char)s[i - 1]) && k.alnum.has((unsigned char)s[i + 1])) o += " @-@ ";
            else o += s[i];
        }
        s.swap(o);
    }
    hide_dot_runs(s);                                                       // :307
    {                                                                       // :310-314 commas, except inside numbers
        std::string o;
        for (size_t i = 0; i < s.size();) {
            if (i + 1 < s.size() && s[i + 1] == ',' && !k.num.has((unsigned char)s[i])) { o += s[i]; o += " , "; i += 2; }
            else o += s[i++];
        }
        s.swap(o);
        o.clear();
        for (size_t i = 0; i < s.size();) {
            if (s[i] == ',' && i + 1 < s.size() && !k.num.has((unsigned char)s[i + 1])) { o += " , "; o += s[i + 1]; i += 2; }
            else o += s[i++];
        }
        s.swap(o);
        if (s.size() >= 2 && s.back() == ',' && k.num.has((unsigned char)s[s.size() - 2])) { s.pop_back(); s += " , "; }
    }
    auto alpha = [&k](unsigned char c) { return k.alpha.has(c); };
    auto not_alpha = [&k](unsigned char c) { return !k.alpha.has(c); };
    if (lang == "en") {                                                     // :317-322
        s = rewrite_apostrophes(s, not_alpha, not_alpha, " ' ");
        s = rewrite_apostrophes(s, [&k](unsigned char c) { return !k.alpha.has(c) && !k.num.has(c); }, alpha, " ' ");
        s = rewrite_apostrophes(s, alpha, not_alpha, " ' ");
        s = rewrite_apostrophes(s, alpha, alpha, " '");
        s = rewrite_apostrophes(s, [&k](unsigned char c) { return k.num.has(c); }, [](unsigned char c) { return c == 's'; }, " '");
    } else if (lang == "fr") {                                              // :323-328
        s = rewrite_apostrophes(s, not_alpha, not_alpha, " ' ");
        s = rewrite_apostrophes(s, not_alpha, alpha, " ' ");
        s = rewrite_apostrophes(s, alpha, not_alpha, " ' ");
        s = rewrite_apostrophes(s, alpha, alpha, "' ");
    } else {                                                                // :329-331
        substitute(s, "'", " ' ");
    }
    s = split_sentence_final_periods(s, lang);                              // :334
    s = trim(squeeze_spaces(s));                                            // :337-338
    {                                                                       // :341 a closing  .'  at the very end
        const size_t n = s.size();
        if (n >= 2 && s[n - 2] == '.' && s[n - 1] == '\'') s.replace(n - 2, 2, " . ' ");
        else if (n >= 3 && s[n - 3] == '.' && s[n - 2] == '\'' && s[n - 1] == ' ') s.replace(n - 3, 3, " . ' ");
    }
    show_dot_runs(s);                                                       // :344
    substitute(s, "&", "&amp;");                                            // :347, table :137-146 (order matters)
    substitute(s, "|", "&#124;");
    substitute(s, "<", "&lt;");
    substitute(s, ">", "&gt;");
    substitute(s, "'", "&apos;");
    substitute(s, "\"", "&quot;");
    substitute(s, "[", "&#91;");
    substitute(s, "]", "&#93;");
    return words_of(s);                                                     // :349-355
}

// mosestokenizer.cpp:360-466.  What the reference's patterns actually accept (several are not the bracket
// expressions they were meant to be) is what is implemented:
//   * " @-@" (leading space only) -> "-", so "a @-@ b" comes back as "a- b";
//   * XML escapes are NOT undone (the reference discards the result of its replacement loop, :376-380);
//   * the "closing punctuation" rule only fires on the literal word "[,.?!:;\%}]" followed by ')' characters;
//   * for fr/it/ga a word that is not an elision is dropped (its branch has no else).
std::string MosesTokenizer::detokenize(const std::vector<std::string> &in, const std::string &lang) const {
    std::string text = " ";
    for (const std::string &t : in) { text += t; text += ' '; }
    substitute(text, " @-@", "-");
    const std::vector<std::string> tok = words_of(text);

    ByteSet opener = cls_.sc;                       // currency signs and opening brackets attach to what follows
    opener.add_bytes("([{\xC2\xBF\xC2\xA1");
    ByteSet quote;                                  // ' " ` and the bytes of the two low/high double quotes
    quote.add_bytes("'\"`\xE2\x80\x9E\xE2\x80\x9C");
    ByteSet curly;
    curly.add_bytes("\xE2\x80\x9E\xE2\x80\x9C\xE2\x80\x9D");
    auto only = [](const std::string &t, const ByteSet &set) {
        for (unsigned char c : t) if (!set.has(c)) return false;
        return !t.empty();
    };
    static const std::string closer_stem = "[,.?!:;\\%}]";

    std::map<std::string, int> seen;                // quote word -> times seen (open/close alternation)
    std::string out, gap = " ";
    for (size_t i = 0; i < tok.size(); i++) {
        const std::string &t = tok[i];
        if (only(t, opener)) {
            out += gap + t;
            gap = "";
        } else if (t.size() > closer_stem.size() && t.compare(0, closer_stem.size(), closer_stem) == 0 &&
                   t.find_first_not_of(')', closer_stem.size()) == std::string::npos) {
            out += t;
            gap = " ";
        } else if (lang == "en" && i > 0 && t.size() >= 2 && t[0] == '\'' && cls_.alpha.has((unsigned char)t[1])) {
            out += t;                               // 's 're 't ... glue to the previous word
            gap = " ";
        } else if (lang == "fr" || lang == "it" || lang == "ga") {
            if (i + 2 <= tok.size() - 0 && i + 1 < tok.size() && t.size() >= 2 && t.back() == '\'' &&
                cls_.alpha.has((unsigned char)t[t.size() - 2]) && cls_.alpha.has((unsigned char)tok[i + 1][0])) {
                out += gap + t;                     // l' + word
                gap = "";
            }
        } else if (only(t, quote)) {
            const std::string key = only(t, curly) ? std::string("\"") : t;
            int &count = seen[key];
            if (count % 2 == 0) {
                if (lang == "en" && t == "'" && i > 0 && tok[i - 1].back() == 's') {
                    out += t;                       // plural possessive: s'
                    gap = " ";
                } else {
                    out += gap + t;                 // opening quote
                    gap = "";
                    count++;
                }
            } else {
                out += t;                           // closing quote
                gap = " ";
                count++;
            }
        } else {
            out += gap + t;
            gap = " ";
        }
    }
    {   // runs of two or more spaces -> one; then trim
        std::string o;
        for (size_t i = 0; i < out.size();) {
            if (out[i] == ' ') { while (i < out.size() && out[i] == ' ') i++; o += ' '; }
            else o += out[i++];
        }
        out.swap(o);
    }
    return trim(out);
}

// ---- vocabulary / BPE --------------------------------------------------------------------------------
Vocab::Vocab(const std::vector<std::string> &tokens, const std::vector<std::string> &merges) : id_to_token_(tokens) {
    for (size_t i = 0; i < tokens.size(); i++) token_to_id_[tokens[i]] = (int32_t)i;   // a repeated string keeps its last id (biogpt.cpp:98)
    // biogpt.cpp:131-155: the first two whitespace-separated words of each record; an empty record re-ranks
    // the previous pair (the reference leaves its pair variable untouched and stores the new rank under it)
    std::pair<std::string, std::string> pair;
    for (size_t r = 0; r < merges.size(); r++) {
        if (!merges[r].empty()) {
            const std::vector<std::string> w = words_of(merges[r]);
            pair.first = w.size() > 0 ? w[0] : std::string();
            pair.second = w.size() > 1 ? w[1] : std::string();
        }
        ranks_[pair] = (int)r;
    }
}

int Vocab::rank_of(const std::string &a, const std::string &b) const {
    auto it = ranks_.find(std::make_pair(a, b));
    return it == ranks_.end() ? -1 : it->second;
}

// bpe.cpp:20-91: bytes as initial symbols, "</w>" glued to the last; repeatedly merge the adjacent pair with
// the lowest rank (all its occurrences, left to right) until no adjacent pair is ranked.
std::string Vocab::bpe(const std::string &word) const {
    if (word.empty()) throw std::out_of_range("bpe: empty word");   // the reference's substr(size()-1) throws here
    std::vector<std::string> sym;
    for (size_t i = 0; i + 1 < word.size(); i++) sym.push_back(std::string(1, word[i]));
    sym.push_back(word.substr(word.size() - 1) + "</w>");
    if (sym.size() == 1) return word + "</w>";
    for (;;) {
        int best = -1;
        size_t at = 0;
        for (size_t i = 0; i + 1 < sym.size(); i++) {
            const int r = rank_of(sym[i], sym[i + 1]);
            if (r >= 0 && (best < 0 || r < best)) { best = r; at = i; }
        }
        if (best < 0) break;
        const std::string left = sym[at], right = sym[at + 1];
        std::vector<std::string> next;
        for (size_t i = 0; i < sym.size();) {
            if (i + 1 < sym.size() && sym[i] == left && sym[i + 1] == right) { next.push_back(left + right); i += 2; }
            else next.push_back(sym[i++]);
        }
        sym.swap(next);
        if (sym.size() == 1) break;
    }
    std::string out;
    for (size_t i = 0; i < sym.size(); i++) { if (i) out += ' '; out += sym[i]; }
    if (out == "\n  </w>") out = "\n</w>";   // bpe.cpp:86-88
    return out;
}

std::vector<int32_t> Vocab::encode(MosesTokenizer &moses, const std::string &text, const std::string &lang) const {
    std::vector<int32_t> ids(1, 2);   // "</s>" opens every prompt (biogpt.cpp:859)
    for (const std::string &w : moses.tokenize(text, lang)) {
        for (const std::string &piece : words_of(bpe(w))) {
            auto it = token_to_id_.find(piece);
            if (it != token_to_id_.end()) ids.push_back(it->second);
            else std::fprintf(stderr, "gpt_tokenize: unknown token '%s'\n", piece.c_str());   // dropped (biogpt.cpp:866-870)
        }
    }
    return ids;
}

std::string decode_token_strings(const MosesTokenizer &moses, const std::vector<std::string> &tokens, const std::string &lang) {
    std::string joined;
    for (std::string t : tokens) {
        t.erase(std::remove(t.begin(), t.end(), ' '), t.end());
        substitute(t, "</w>", " ");
        substitute(t, "</s>", " ");
        joined += t;
    }
    return moses.detokenize(words_of(joined), lang);
}

std::string Vocab::decode(const MosesTokenizer &moses, const int32_t *ids, int32_t n, const std::string &lang) const {
    std::vector<std::string> t;
    for (int32_t i = 0; i < n; i++)
        t.push_back(ids[i] >= 0 && ids[i] < n_tokens() ? id_to_token_[(size_t)ids[i]] : std::string());
    return decode_token_strings(moses, t, lang);
}

MosesTokenizer &default_moses() {
    static MosesTokenizer m;
    return m;
}

}  // namespace bgtok
