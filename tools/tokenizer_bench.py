"""Host-side throughput of text -> ids: the product tokenizer vs the reference's own sources (oracle/_ref, build
container only).  Same vocabulary (tests/golden/tokenizer_vocab.bin), same prefix fixture, same sentences.
    python tools/tokenizer_bench.py > profiles/tokenizer_r1.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkg  # noqa: E402
import modelfile_py  # noqa: E402
from oracle import ref_tokenizer  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_tokenizer_golden import CORPUS  # noqa: E402

pkg = _pkg.load()
DATA = os.path.join(ROOT, "tests", "golden", "tokenizer_data")
pkg.set_tokenizer_data_dir(DATA)
VOC = os.path.join(ROOT, "tests", "golden", "tokenizer_vocab.bin")
v = pkg.Vocab.load(VOC)
lines = [l.encode() for l in CORPUS.strip().split("\n")]
n_ids = sum(len(v.tokenize(l)) for l in lines)


def timed(fn, min_s=1.0):
    reps, t0 = 0, time.perf_counter()
    while True:
        for l in lines:
            fn(l)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= min_s:
            return dt / reps


t_mine = timed(lambda l: v.tokenize(l, ""))
print("sentences %d, ids per pass %d (vocabulary of %d tokens)" % (len(lines), n_ids, 554))
print("product  (csrc/tokenizer.cpp via ctypes): %8.3f ms per pass  %10.0f ids/s" % (t_mine * 1e3, n_ids / t_mine))
if ref_tokenizer.available():
    ref = ref_tokenizer.RefTokenizer(prefix_dir=os.path.join(DATA, "nonbreaking_prefixes"))
    _, toks, merges, _ = modelfile_py.read_model(VOC)
    pair, ranked = (b"", b""), {}
    for r, rec in enumerate(merges):
        if rec:
            w = rec.split()
            pair = (w[0] if w else b"", w[1] if len(w) > 1 else b"")
        ranked[pair] = r
    ref.vocab = ref.L.ref_vocab_new()
    for i, t in enumerate(toks):
        ref.L.ref_vocab_add_token(ref.vocab, t, i)
    for (a, c), r in ranked.items():
        ref.L.ref_vocab_add_merge(ref.vocab, a, c, r)
    assert all(ref.gpt_tokenize(l, "") == v.tokenize(l, "") for l in lines)
    t_ref = timed(lambda l: ref.gpt_tokenize(l, ""), min_s=3.0)
    print("reference (mosestokenizer.cpp + bpe.cpp, g++ -O2): %8.3f ms per pass  %10.0f ids/s" % (t_ref * 1e3, n_ids / t_ref))
    print("identical ids on every sentence; speed-up %.0fx" % (t_ref / t_mine))
