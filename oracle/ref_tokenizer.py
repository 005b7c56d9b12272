"""ctypes binding of oracle/_ref/libref_tokenizer.so = the REFERENCE's own tokenizer sources compiled here.

TEST INFRASTRUCTURE ONLY (tests/ and tests/golden/make_tokenizer_golden.py).  Exists only in the build
container (it needs /root/reference for the sources and for data/perluniprops); on a machine without the
reference `available()` is False and the tests fall back to the committed golden vectors.

The reference resolves its data files relative to the current directory ("../data/...",
mosestokenizer.cpp:11-12), at static-initialisation time and again on every call; `RefTokenizer` builds a
scratch tree <root>/data/{perluniprops -> reference, nonbreaking_prefixes/...} + <root>/run and changes
into <root>/run around dlopen and each call.
"""
import contextlib
import ctypes as C
import os
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("BIOGPT_REFERENCE_DIR", "/root/reference")
_LIB_PATH = os.path.join(_HERE, "_ref", "libref_tokenizer.so")
LENGTH_ERROR = -1      # the reference threw std::length_error (mosestokenizer.cpp:264)


def available():
    return os.path.exists(os.path.join(REFERENCE, "mosestokenizer.cpp")) and \
        os.path.isdir(os.path.join(REFERENCE, "data", "perluniprops"))


def build():
    if not available():
        return None
    subprocess.check_call(["make", "-C", _HERE, "ref_tokenizer", "REFERENCE=" + REFERENCE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


class RefTokenizer:
    """prefix_dir: directory holding nonbreaking_prefix.<lang> files (default: the reference's own)."""

    _lib = None

    def __init__(self, prefix_dir=None):
        if not available():
            raise RuntimeError("reference tokenizer sources/data not present")
        self.root = tempfile.mkdtemp(prefix="reftok_")
        os.makedirs(os.path.join(self.root, "data"))
        os.makedirs(os.path.join(self.root, "run"))
        os.symlink(os.path.join(REFERENCE, "data", "perluniprops"), os.path.join(self.root, "data", "perluniprops"))
        os.symlink(os.path.abspath(prefix_dir) if prefix_dir else os.path.join(REFERENCE, "data", "nonbreaking_prefixes"),
                   os.path.join(self.root, "data", "nonbreaking_prefixes"))
        self.run = os.path.join(self.root, "run")
        if RefTokenizer._lib is None:
            build()
            with _cwd(self.run):      # static initialisers read ../data/perluniprops/*.txt
                L = C.CDLL(_LIB_PATH)
            L.ref_moses_tokenize.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
            L.ref_moses_detokenize.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
            L.ref_vocab_new.restype = C.c_void_p
            L.ref_vocab_free.argtypes = [C.c_void_p]
            L.ref_vocab_add_merge.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
            L.ref_vocab_add_token.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
            L.ref_bpe.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
            L.ref_gpt_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int]
            L.ref_gpt_decode.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
            RefTokenizer._lib = L
        self.L = RefTokenizer._lib
        self.vocab = None

    @staticmethod
    def _b(s):
        return s if isinstance(s, bytes) else s.encode("utf-8")

    def _str_call(self, fn, *args):
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        with _cwd(self.run):
            n = fn(*args, buf, cap)
        return n, buf.raw[:max(n, 0)]

    def moses_tokenize(self, text, lang=""):
        """list of byte strings, or LENGTH_ERROR"""
        n, raw = self._str_call(self.L.ref_moses_tokenize, self._b(text), self._b(lang))
        if n < 0:
            return n
        return raw.split(b"\n") if n else []

    def moses_detokenize(self, tokens, lang=""):
        n, raw = self._str_call(self.L.ref_moses_detokenize, b"\n".join(self._b(t) for t in tokens), self._b(lang))
        assert n >= 0
        return raw

    def set_vocab(self, tokens, merges):
        """tokens: id -> bytes; merges: rank -> (left bytes, right bytes)"""
        self.vocab = self.L.ref_vocab_new()
        for i, t in enumerate(tokens):
            self.L.ref_vocab_add_token(self.vocab, self._b(t), i)
        for r, (a, b) in enumerate(merges):
            self.L.ref_vocab_add_merge(self.vocab, self._b(a), self._b(b), r)

    def bpe(self, word):
        n, raw = self._str_call(self.L.ref_bpe, self.vocab, self._b(word))
        assert n >= 0
        return raw

    def gpt_tokenize(self, text, lang=""):
        out = (C.c_int * 65536)()
        with _cwd(self.run):
            n = self.L.ref_gpt_tokenize(self.vocab, self._b(text), self._b(lang), out, 65536)
        return n if n < 0 else list(out[:n])

    def gpt_decode(self, tokens, lang=""):
        n, raw = self._str_call(self.L.ref_gpt_decode, b"\n".join(self._b(t) for t in tokens), self._b(lang))
        assert n >= 0
        return raw
