"""Generates tests/golden/tokenizer_vocab.bin and tests/golden/tokenizer_golden.json.

Runs ONLY in the build container: the expected outputs come from the REFERENCE's own tokenizer sources,
compiled where they lie by `make -C oracle ref_tokenizer` (oracle/_ref, never committed) and driven through
oracle/ref_tokenizer.py.  The committed files are data: a small BPE vocabulary trained below on sentences
written for this repository, input strings, and the reference's outputs for them.

    python tests/golden/make_tokenizer_golden.py
"""
import collections
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import modelfile_py  # noqa: E402
import ref_tokenizer  # noqa: E402

CORPUS = """
The patient was treated with 5,300 mg of metformin twice daily. Dr. Hale reported no adverse events.
COVID-19 is caused by SARS-CoV-2, a single-stranded RNA virus. Fig. 2 shows the spike protein.
Gene expression of TP53 was up-regulated in tumour tissue (p < 0.05), i.e. about two-fold.
Beta-blockers reduce heart rate; however, they don't suit every patient. It's a trade-off.
Insulin resistance precedes type 2 diabetes by 10-15 years... or longer.
The enzyme's active site binds ATP and Mg2+ ions. What is the binding affinity?
In vitro assays vs. in vivo models: results differ, e.g. for IL-6 and TNF-alpha signalling.
Approx. 3.5 million adults in the U.S. have hepatitis C. No. 7 on the list is aspirin.
Mrs. Green's biopsy showed "atypical cells" [grade II] & mild inflammation.
Patients' outcomes improved <30 days> after surgery | see Table 1.
the mitochondria is the powerhouse of the cell. protein kinase inhibitors block phosphorylation.
antibiotic resistance in bacteria spreads through plasmids, transposons and integrons.
"""

SPECIALS = [b"<s>", b"<pad>", b"</s>", b"<unk>"]     # BioGPT ids 0..3
N_MERGES = 420

CASES = [
    # the three known answers the reference carries (mosestokenizer.cpp:489-497; run there with lang "en")
    "Hello World!",
    "This ain't funny. It's actually hillarious, yet double Ls. | [] < > [ ] & You're gonna shake it off? Don't?",
    "this is a webpage https://stackoverflow.com/questions/6181381/how-to-print-variables-in-perl that kicks ass",
    # README.md:29 prompt
    "COVID-19 is",
    "The patient was treated with 5,300 mg of metformin. Dr. Hale reported no adverse events.",
    "Insulin resistance precedes diabetes by 10-15 years... or longer..",
    "e.g. IL-6 vs. TNF-alpha, i.e. cytokines. No. 7 is aspirin. pp. 12",
    "Mrs. Green's biopsy showed \"atypical cells\" [grade II] & mild inflammation.",
    "Patients' outcomes improved <30 days> after surgery | see Table 1.",
    "  leading and   trailing\twhitespace\n and a tab  ",
    "",
    " ",
    ".",
    "..",
    "a..b ...c.... d.",
    "1,2 ,3, 4,a b,5 6,",
    "x,",
    "7,",
    "it's 'quoted' ''double'' rock'n'roll 1990's",
    "l'homme qu'il d'une M. Dupont env. 3",
    "the end.'",
    "the end.' ",
    "well-known state-of-the-art -dash- a--b -",
    "price: $5.00 (approx.) {braces} 100% a/b \\ back",
    "DOTMULTI and DOTDOTMULTI. literal DOTMULTI.x",
    "café naïve µg/mL 5–6 “quoted” €10",
    "ends with accent café.",
    "period then accent. été",          # the reference throws std::length_error here
    "U.S. été",                          # abbreviation rule applies first: no throw
    "Dr. été",                           # listed prefix: no throw
    "control\x01chars\x02 here\x1f.",
    "&amp; already &lt;escaped&gt; 'x' \"y\"",
    "zzzqqq xqzj",                                  # pieces missing from the small vocabulary are dropped
]
LANGS = ["", "en", "fr", "de"]
DETOK_EXTRA = [
    ["Hello", "World", "!"],
    ["a", "@-@", "b"],
    ["$", "5", "(", "x", ")", "[", "y", "]"],
    ["he", "said", "\"", "yes", "\"", "and", "'", "no", "'"],
    ["the", "boys", "'", "toys", "it", "'s", "I", "'m"],
    ["l'", "homme", "qu'", "il", "dit"],
    ["[,.?!:;\\%}])", "[,.?!:;\\%}]))", "x"],
    ["„", "low", "“", "”", "``", "x", "''"],
    ["&amp;", "&lt;", "&apos;", "s"],
    ["¿", "qué", "?", "¡", "hola", "!"],
    [],
]


def train_bpe(text, n_merges):
    words = collections.Counter(text.split())
    seqs = {w: [bytes([b]) for b in w.encode("utf-8")[:-1]] + [w.encode("utf-8")[-1:] + b"</w>"] for w in words}
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, s in seqs.items():
            for a, b in zip(s, s[1:]):
                pairs[(a, b)] += words[w]
        if not pairs:
            break
        best = min(pairs, key=lambda p: (-pairs[p], p))
        if pairs[best] < 2:
            break
        merges.append(best)
        for w, s in seqs.items():
            out, i = [], 0
            while i < len(s):
                if i + 1 < len(s) and (s[i], s[i + 1]) == best:
                    out.append(s[i] + s[i + 1])
                    i += 2
                else:
                    out.append(s[i])
                    i += 1
            seqs[w] = out
    return merges


def lat(b):
    """bytes -> JSON-safe str (latin-1 keeps every byte value)"""
    return b.decode("latin-1")


def main():
    moses_ref = ref_tokenizer.RefTokenizer(prefix_dir=os.path.join(HERE, "tokenizer_data", "nonbreaking_prefixes"))
    # train on the reference's own word splitting so that the vocabulary covers "&apos;", "@-@", ...
    words = []
    for line in CORPUS.strip().split("\n"):
        for lang in ("", "en"):
            words += [w.decode("utf-8") for w in moses_ref.moses_tokenize(line, lang)]
    merges = train_bpe(" ".join(words), N_MERGES)
    vocab = list(SPECIALS)
    seen = set(vocab)
    for w in words:
        bs = w.encode("utf-8")
        for sym in [bytes([b]) for b in bs[:-1]] + [bs[-1:] + b"</w>"] + [bytes([b]) + b"</w>" for b in bs] + [bytes([b]) for b in bs]:
            if sym not in seen:
                seen.add(sym)
                vocab.append(sym)
    for a, b in merges:
        if a + b not in seen:
            seen.add(a + b)
            vocab.append(a + b)
    merge_records = [a + b" " + b for a, b in merges]
    merge_records.insert(5, b"")                 # an empty record: re-ranks the previous pair (biogpt.cpp:138-152)
    merge_records.append(b"x  y   z")            # only the first two words count
    hp = dict(n_vocab=len(vocab), n_layer=1, n_head=1, n_positions=32, d_ff=32, d_model=32, ftype=0)
    modelfile_py.write_model(os.path.join(HERE, "tokenizer_vocab.bin"), hp, vocab, merge_records, [])

    # the reference's loader semantics for the merge table (biogpt.cpp:131-155)
    pair, pairs = (b"", b""), {}
    for r, rec in enumerate(merge_records):
        if rec:
            w = rec.split()
            pair = (w[0] if w else b"", w[1] if len(w) > 1 else b"")
        pairs[pair] = r
    by_rank = sorted(pairs.items(), key=lambda kv: kv[1])
    moses_ref.L.ref_vocab_new.restype = __import__("ctypes").c_void_p
    moses_ref.vocab = moses_ref.L.ref_vocab_new()
    for i, t in enumerate(vocab):
        moses_ref.L.ref_vocab_add_token(moses_ref.vocab, t, i)
    for (a, b), r in by_rank:
        moses_ref.L.ref_vocab_add_merge(moses_ref.vocab, a, b, r)

    out = {"about": "expected outputs of the reference's tokenizer sources (oracle/_ref) with tests/golden/tokenizer_data "
                    "as its data directory; strings are latin-1 views of the raw bytes",
           "n_vocab": len(vocab), "n_merge_records": len(merge_records), "cases": [], "detok": []}
    for text in CASES:
        tb = text.encode("utf-8")
        for lang in LANGS:
            words_ = moses_ref.moses_tokenize(tb, lang)
            case = {"text": lat(tb), "lang": lang}
            if not isinstance(words_, list):
                case["throws"] = "length_error"
            else:
                ids = moses_ref.gpt_tokenize(tb, lang)
                case["moses"] = [lat(w) for w in words_]
                case["bpe"] = [lat(moses_ref.bpe(w)) for w in words_]
                case["ids"] = ids
                toks = [vocab[i] for i in ids]
                case["decode"] = {dl: lat(moses_ref.gpt_decode(toks, dl)) for dl in ("", "en", "fr")}
                case["detok"] = {dl: lat(moses_ref.moses_detokenize(words_, dl)) for dl in ("", "en", "fr", "it")}
            out["cases"].append(case)
    for toks in DETOK_EXTRA:
        tb = [t.encode("utf-8") for t in toks]
        out["detok"].append({"tokens": [lat(t) for t in tb],
                             "out": {dl: lat(moses_ref.moses_detokenize(tb, dl)) for dl in ("", "en", "fr", "it", "ga")}})
    with open(os.path.join(HERE, "tokenizer_golden.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=True)
    print("vocab %d tokens, %d merge records, %d cases" % (len(vocab), len(merge_records), len(out["cases"])))


if __name__ == "__main__":
    main()
