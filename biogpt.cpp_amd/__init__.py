"""Host-side Python mirror of the reference's model-library interface, bound over the C-ABI.

The reference is compiled C++ (biogpt.h:128-151); its drop-in C++ wrappers live in
include/biogpt_compat.h.  This module is the thin ctypes stub used by tests/, bench.py and
__graft_entry__.py -- it only marshals pointers and sizes into libbiogpt_hip.so
(include/biogpt_hip.h).  No compute happens here and there is NO fallback: if the HIP library
is missing or no GPU is present the calls raise.

Names mirror the reference: biogpt_model_load() -> BiogptModel, biogpt_eval() -> .eval(),
biogpt_model_quantize_internal()/quantize CLI -> quantize_file().
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BIOGPT_HIP_LIB") or os.path.join(_HERE, "libbiogpt_hip.so")   # override: profiling builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

FTYPES = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}
FTYPE_NAMES = {v: k for k, v in FTYPES.items()}
# bytes per 32-element block in the FILE layout
FILE_BLOCK_BYTES = {0: 128, 1: 64, 2: 18, 3: 20, 7: 34, 8: 22, 9: 24}


class HParams(C.Structure):
    """biogpt_hparams (biogpt.h:25-35) in file order + n_merges as found in the file."""
    _fields_ = [("n_vocab", C.c_int32), ("n_layer", C.c_int32), ("n_head", C.c_int32),
                ("n_positions", C.c_int32), ("d_ff", C.c_int32), ("d_model", C.c_int32),
                ("ftype", C.c_int32), ("n_merges", C.c_int32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


BIOGPT_BASE = dict(n_vocab=42384, n_layer=24, n_head=16, n_positions=1024, d_ff=4096, d_model=1024,
                   ftype=0, n_merges=40000)


class BiogptError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile libbiogpt_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))] + [os.path.join(_HERE, "..", "include", "biogpt_hip.h")]
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))] + (["-B"] if force else []) + ["all"]   # objects under csrc/obj/ (git-ignored)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL, stderr=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None

# every symbol include/biogpt_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("biogpt_hip_last_error", C.c_char_p, []),
    ("biogpt_hip_version", C.c_char_p, []),
    ("biogpt_hip_load", _P, [C.c_char_p, C.c_int, C.c_int]),
    ("biogpt_hip_load_into", _P, [C.c_char_p, C.c_int, C.c_int, _P, C.c_size_t]),
    ("biogpt_hip_attach", _P, [C.POINTER(HParams), C.c_int, _P, C.c_size_t]),
    ("biogpt_hip_arena_bytes_for", C.c_size_t, [C.POINTER(HParams)]),
    ("biogpt_hip_arena_ptr", _P, [_P]),
    ("biogpt_hip_arena_bytes", C.c_size_t, [_P]),
    ("biogpt_hip_free", None, [_P]),
    ("biogpt_hip_refresh_options", C.c_int, [_P]),
    ("biogpt_hip_xpipe_state", C.c_int, [_P]),
    ("biogpt_hip_share_vocab", C.c_int, [_P, _P]),
    ("biogpt_hip_replicas_load", _P, [C.c_char_p, _P, C.c_int, C.c_int]),
    ("biogpt_hip_replicas_count", C.c_int, [_P]),
    ("biogpt_hip_replicas_ctx", _P, [_P, C.c_int]),
    ("biogpt_hip_replicas_broadcast_seconds", C.c_double, [_P]),
    ("biogpt_hip_replicas_generate_greedy", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_double)]),
    ("biogpt_hip_replicas_free", None, [_P]),
    ("biogpt_hip_get_hparams", C.c_int, [_P, C.POINTER(HParams)]),
    ("biogpt_hip_n_tensors", C.c_int, [_P]),
    ("biogpt_hip_vocab_token", C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]),
    ("biogpt_hip_merge", C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]),
    ("biogpt_hip_eval", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    ("biogpt_hip_eval_inplace", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.POINTER(C.POINTER(C.c_float))]),
    ("biogpt_hip_resident_stats", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("biogpt_hip_chunk_launches", C.c_int64, [_P]),
    ("biogpt_hip_fpipe_launches", C.c_int64, [_P]),
    ("biogpt_hip_fpipe_stamps", C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    ("biogpt_hip_generate_launches", C.c_int, [_P, C.POINTER(C.c_int32), C.c_int]),
    ("biogpt_hip_lineage_stats", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("biogpt_hip_bench_sweep", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("biogpt_hip_bench_sweep_ex", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_size_t,
                                            C.POINTER(C.c_int8), C.POINTER(C.c_float)]),
    ("biogpt_hip_eval_device", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("biogpt_hip_eval_topk", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    ("biogpt_hip_logits_device", _P, [_P]),
    ("biogpt_hip_read_logits", C.c_int, [_P, _P]),
    ("biogpt_hip_synchronize", C.c_int, [_P]),
    ("biogpt_hip_eval_all", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    ("biogpt_hip_eval_prompt", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    ("biogpt_hip_generate_greedy", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    ("biogpt_hip_generate_greedy_batch", C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    ("biogpt_hip_read_kv", C.c_int, [_P, C.c_int, C.c_size_t, C.c_size_t, _P]),
    ("biogpt_hip_debug_stamps", C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(C.c_ulonglong)]),
    ("biogpt_hip_bench_matvec", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("biogpt_hip_bench_decode", C.c_int, [_P, C.c_int32, C.c_int, C.POINTER(C.c_double)]),
    ("biogpt_hip_bench_api_loop", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_double)]),
    ("biogpt_hip_bench_stream", C.c_int, [_P, C.c_int32, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("biogpt_hip_quantize_file", C.c_int, [C.c_char_p, C.c_char_p, C.c_int32]),
    ("biogpt_hip_quantize_rows_device", C.c_int, [C.c_int, C.c_int32, _P, C.c_int64, C.c_int64, _P]),
    ("biogpt_hip_write_synthetic", C.c_int, [C.c_char_p, C.POINTER(HParams), C.c_uint64]),
    # text <-> ids (host-only)
    ("biogpt_hip_vocab_load", _P, [C.c_char_p]),
    ("biogpt_hip_vocab_create", _P, [_P, _P, C.c_int32, _P, _P, C.c_int32]),
    ("biogpt_hip_vocab_free", None, [_P]),
    ("biogpt_hip_ctx_vocab", _P, [_P]),
    ("biogpt_hip_tokenizer_set_data_dir", C.c_int, [C.c_char_p]),
    ("biogpt_hip_tokenize", C.c_int, [_P, C.c_char_p, C.c_char_p, _P, C.c_int32]),
    ("biogpt_hip_decode", C.c_int, [_P, _P, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32]),
    ("biogpt_hip_decode_strings", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]),
    ("biogpt_hip_moses_tokenize", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]),
    ("biogpt_hip_moses_detokenize", C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]),
    ("biogpt_hip_bpe", C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int32]),
    ("biogpt_hip_tokenizer_byte_class", C.c_int, [C.c_int, _P]),
]
E_LENGTH = -7   # BIOGPT_HIP_E_LENGTH


def lib():
    """Load libbiogpt_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BiogptError("%s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _err():
    return lib().biogpt_hip_last_error().decode("utf-8", "replace")


def quantize_file(src, dst, ftype):
    """examples/quantize equivalent (quantize.cpp:8-135): ftype by name or ggml_ftype id."""
    ft = FTYPES[ftype] if isinstance(ftype, str) else int(ftype)
    if lib().biogpt_hip_quantize_file(os.fsencode(src), os.fsencode(dst), ft) != 0:
        raise BiogptError(_err())


def quantize_rows_device(rows, type_id, device=0):
    """ggml_quantize_* on the device: a [nrows, k] float32 array -> the file's block bytes (uint8 array)."""
    a = np.ascontiguousarray(rows, dtype=np.float32)
    nrows, k = a.shape
    bb = {2: 18, 3: 20, 6: 22, 7: 24, 8: 34}[int(type_id)]
    out = np.empty(nrows * (k // 32) * bb, dtype=np.uint8)
    if lib().biogpt_hip_quantize_rows_device(int(device), int(type_id), a.ctypes.data, nrows, k, out.ctypes.data) != 0:
        raise BiogptError(_err())
    return out


def write_synthetic(path, seed=0x42494F47, **hparams):
    """Write a seeded synthetic model file. hparams default to BioGPT-base."""
    hp = HParams(**{**BIOGPT_BASE, **hparams})
    if lib().biogpt_hip_write_synthetic(os.fsencode(path), C.byref(hp), C.c_uint64(seed)) != 0:
        raise BiogptError(_err())
    return hp


class TokenizerLengthError(BiogptError):
    """The reference's moses_tokenize throws std::length_error for this text (mosestokenizer.cpp:264)."""


def _bytes(s):
    return s if isinstance(s, bytes) else s.encode("utf-8")


def _string_result(fn, *args):
    """C-ABI string protocol: the call returns the needed length; retry once with a buffer that fits."""
    cap = 4096
    for _ in range(2):
        buf = C.create_string_buffer(cap)
        n = fn(*args, buf, cap)
        if n == E_LENGTH:
            raise TokenizerLengthError(_err())
        if n < 0:
            raise BiogptError(_err())
        if n + 1 <= cap:
            return buf.raw[:n]
        cap = n + 1
    raise BiogptError("string result did not fit")


def set_tokenizer_data_dir(path):
    """Directory holding nonbreaking_prefixes/ (the reference's data/); default $BIOGPT_DATA_DIR or ../data."""
    lib().biogpt_hip_tokenizer_set_data_dir(os.fsencode(path))


def moses_tokenize(text, lang=""):
    """moses_tokenize (mosestokenizer.cpp:290-358): list of byte strings."""
    raw = _string_result(lib().biogpt_hip_moses_tokenize, _bytes(text), _bytes(lang))
    return raw.split(b"\n") if raw else []


def moses_detokenize(tokens, lang=""):
    """moses_detokenize (mosestokenizer.cpp:360-466)."""
    return _string_result(lib().biogpt_hip_moses_detokenize, b"\n".join(_bytes(t) for t in tokens), _bytes(lang))


def decode_strings(tokens, lang=""):
    """gpt_decode (biogpt.cpp:877-906) on vocabulary strings."""
    return _string_result(lib().biogpt_hip_decode_strings, b"\n".join(_bytes(t) for t in tokens), _bytes(lang))


def byte_class(which):
    """256 flags of a perluniprops byte class: 0 IsAlnum, 1 IsAlpha, 2 IsLower, 3 IsN, 4 IsSc."""
    out = (C.c_uint8 * 256)()
    if lib().biogpt_hip_tokenizer_byte_class(which, out) != 0:
        raise BiogptError("bad class index")
    return bytes(out)


class Vocab:
    """biogpt_vocab (biogpt.h:37-48) for the tokenizer: gpt_tokenize / gpt_decode / bpe.  Host-only."""

    def __init__(self, handle, owned):
        self._h, self._owned = handle, owned

    @classmethod
    def load(cls, path):
        h = lib().biogpt_hip_vocab_load(os.fsencode(path))
        if not h:
            raise BiogptError(_err())
        return cls(h, True)

    @classmethod
    def create(cls, tokens, merges):
        """tokens: id -> bytes; merges: rank -> b"left right" records (what the model file stores)."""
        tokens = [_bytes(t) for t in tokens]
        merges = [_bytes(m) for m in merges]
        tp = (C.c_char_p * max(1, len(tokens)))(*tokens)
        tl = (C.c_int32 * max(1, len(tokens)))(*[len(t) for t in tokens])
        mp = (C.c_char_p * max(1, len(merges)))(*merges)
        ml = (C.c_int32 * max(1, len(merges)))(*[len(m) for m in merges])
        h = lib().biogpt_hip_vocab_create(tp, tl, len(tokens), mp, ml, len(merges))
        if not h:
            raise BiogptError(_err())
        return cls(h, True)

    def close(self):
        if self._h and self._owned:
            lib().biogpt_hip_vocab_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bpe(self, word):
        return _string_result(lib().biogpt_hip_bpe, self._h, _bytes(word))

    def tokenize(self, text, lang=""):
        """gpt_tokenize (biogpt.cpp:850-875): ids, starting with 2."""
        cap = 256
        for _ in range(2):
            out = (C.c_int32 * cap)()
            n = lib().biogpt_hip_tokenize(self._h, _bytes(text), _bytes(lang), out, cap)
            if n == E_LENGTH:
                raise TokenizerLengthError(_err())
            if n < 0:
                raise BiogptError(_err())
            if n <= cap:
                return list(out[:n])
            cap = n
        raise BiogptError("id result did not fit")

    def decode(self, ids, lang=""):
        arr = (C.c_int32 * max(1, len(ids)))(*ids)
        return _string_result(lib().biogpt_hip_decode, self._h, arr, len(ids), _bytes(lang))


def arena_bytes_for(hp):
    return int(lib().biogpt_hip_arena_bytes_for(C.byref(hp)))


class BiogptModel:
    """biogpt_model + biogpt_vocab handle (biogpt.h:78-107, :37-48) living on one HIP device."""

    def __init__(self, handle):
        if not handle:
            raise BiogptError(_err())
        self._h = handle
        self.hparams = HParams()
        lib().biogpt_hip_get_hparams(self._h, C.byref(self.hparams))
        self.n_vocab = self.hparams.n_vocab
        self.n_tensors = lib().biogpt_hip_n_tensors(self._h)

    # -- biogpt_model_load (biogpt.h:128-132) --
    @classmethod
    def load(cls, fname, device=0, verbosity=0, arena=None, arena_bytes=0):
        if arena is None:
            return cls(lib().biogpt_hip_load(os.fsencode(fname), device, verbosity))
        return cls(lib().biogpt_hip_load_into(os.fsencode(fname), device, verbosity, arena, arena_bytes))

    @classmethod
    def attach(cls, hparams, device, arena, arena_bytes):
        return cls(lib().biogpt_hip_attach(C.byref(hparams), device, arena, arena_bytes))

    # -- biogpt_eval (biogpt.h:145-151): logits of the last token --
    def eval(self, tokens, n_past):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.n_vocab, dtype=np.float32)
        if lib().biogpt_hip_eval(self._h, toks.ctypes.data, toks.size, int(n_past), out.ctypes.data) != 0:
            raise BiogptError(_err())
        return out

    def eval_topk(self, tokens, n_past, k):
        """biogpt_eval + device-side top-k: (values descending, ids) of the k <= 64 largest logits of the last token."""
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        vals = np.zeros(64, dtype=np.float32)
        ids = np.zeros(64, dtype=np.int32)
        got = lib().biogpt_hip_eval_topk(self._h, toks.ctypes.data, toks.size, int(n_past), int(k), vals.ctypes.data, ids.ctypes.data)
        if got < 0:
            raise BiogptError(_err())
        return vals[:got].copy(), ids[:got].copy()

    def eval_all(self, tokens, n_past):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty((toks.size, self.n_vocab), dtype=np.float32)
        if lib().biogpt_hip_eval_all(self._h, toks.ctypes.data, toks.size, int(n_past), out.ctypes.data) != 0:
            raise BiogptError(_err())
        return out

    def eval_prompt(self, tokens, n_past=0, n_batch=8, want_logits=True):
        """== eval() on consecutive chunks of n_batch tokens (main.cpp prompt loop), several chunks per pass.
        Returns the last token's logits, or None (asynchronous) with want_logits=False."""
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.n_vocab, dtype=np.float32) if want_logits else None
        if lib().biogpt_hip_eval_prompt(self._h, toks.ctypes.data, toks.size, int(n_past), int(n_batch),
                                        out.ctypes.data if want_logits else None) != 0:
            raise BiogptError(_err())
        return out

    def eval_device(self, tokens, n_past):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        if lib().biogpt_hip_eval_device(self._h, toks.ctypes.data, toks.size, int(n_past)) != 0:
            raise BiogptError(_err())

    def read_logits(self):
        """The device-side logits row of the last evaluated token (biogpt_hip_logits_device), copied to host memory."""
        out = np.empty(self.n_vocab, dtype=np.float32)
        if lib().biogpt_hip_read_logits(self._h, out.ctypes.data) != 0:
            raise BiogptError(_err())
        return out

    def resident_stats(self):
        """{hits, misses, streak, need} of the resident launch's speculative continuation (biogpt_hip_resident_stats)."""
        out = (C.c_int64 * 4)()
        if lib().biogpt_hip_resident_stats(self._h, out) != 0:
            raise BiogptError(_err())
        return dict(hits=int(out[0]), misses=int(out[1]), streak=int(out[2]), need=int(out[3]))

    def lineage_stats(self):
        """{graph_evals, stale_rows}: single-token evals replayed as captured five-launch steps / rows found to be another call's and repeated (biogpt_hip_lineage_stats)."""
        out = (C.c_int64 * 2)()
        if lib().biogpt_hip_lineage_stats(self._h, out) != 0:
            raise BiogptError(_err())
        return dict(graph_evals=int(out[0]), stale_rows=int(out[1]))

    def synchronize(self):
        if lib().biogpt_hip_synchronize(self._h) != 0:
            raise BiogptError(_err())

    # -- main.cpp:91-151 with --top_k 1 --
    def generate_greedy(self, prompt, n_predict, n_batch=8):
        pr = np.ascontiguousarray(prompt, dtype=np.int32)
        out = np.zeros(max(int(n_predict), 1), dtype=np.int32)
        secs = C.c_double(0.0)
        n = lib().biogpt_hip_generate_greedy(self._h, pr.ctypes.data, pr.size, int(n_batch), int(n_predict),
                                             out.ctypes.data, C.byref(secs))
        if n < 0:
            raise BiogptError(_err())
        return out[:n].copy(), secs.value

    def generate_greedy_batch(self, prompts, n_predict, n_batch=8):
        """Batched decode of several independent prompts (list of id lists) on this device."""
        lens = np.asarray([len(p) for p in prompts], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int32) for p in prompts]))
        out = np.zeros((len(prompts), max(int(n_predict), 1)), dtype=np.int32)
        secs = C.c_double(0.0)
        n = lib().biogpt_hip_generate_greedy_batch(self._h, flat.ctypes.data, lens.ctypes.data, len(prompts), int(n_batch),
                                                   int(n_predict), out.ctypes.data, C.byref(secs))
        if n < 0:
            raise BiogptError(_err())
        return out.reshape(-1)[:len(prompts) * n].reshape(len(prompts), n).copy(), secs.value

    def read_kv(self, which, offset, count):
        out = np.empty(int(count), dtype=np.float32)
        if lib().biogpt_hip_read_kv(self._h, int(which), int(offset), int(count), out.ctypes.data) != 0:
            raise BiogptError(_err())
        return out

    def debug_stamps(self, offset, count):
        """Profiling builds: `count` raw stage stamps (100 MHz ticks) from word `offset` of the context's stamp buffer."""
        out = (C.c_ulonglong * count)()
        if lib().biogpt_hip_debug_stamps(self._h, offset, count, out) != 0:
            raise BiogptError(_err())
        return np.frombuffer(out, dtype=np.uint64).copy()

    def bench_matvec(self, which, layer=0, reps=200):
        secs, nbytes = C.c_double(0.0), C.c_double(0.0)
        if lib().biogpt_hip_bench_matvec(self._h, int(which), int(layer), int(reps), C.byref(secs), C.byref(nbytes)) != 0:
            raise BiogptError(_err())
        return secs.value, nbytes.value

    def bench_stream(self, rows=1 << 19, reps=20, steps=8):
        secs, nbytes = C.c_double(0.0), C.c_double(0.0)
        if lib().biogpt_hip_bench_stream(self._h, int(rows), int(reps), int(steps), C.byref(secs), C.byref(nbytes)) != 0:
            raise BiogptError(_err())
        return secs.value, nbytes.value

    def bench_api_loop(self, prompt, n_predict, mode):
        """The reference's greedy host loop in C++ on this library (mode 0: full logits row per token, 1: device top-40)."""
        pr = np.ascontiguousarray(prompt, dtype=np.int32)
        out = np.zeros(int(n_predict), dtype=np.int32)
        secs = C.c_double(0.0)
        got = lib().biogpt_hip_bench_api_loop(self._h, pr.ctypes.data, pr.size, int(n_predict), int(mode), out.ctypes.data, C.byref(secs))
        if got < 0:
            raise BiogptError(_err())
        return out[:got], secs.value

    def generate_launches(self):
        """Tokens of each multi-token pipelined launch of the last generate_greedy call (biogpt_hip_generate_launches)."""
        buf = (C.c_int32 * 16)()
        n = int(lib().biogpt_hip_generate_launches(self._h, buf, 16))
        return [int(buf[i]) for i in range(max(0, min(n, 16)))]

    def fpipe_launches(self):
        """Single-token steps through the float-weight persistent launch (biogpt_hip_fpipe_launches); -1: not available to this context."""
        return int(lib().biogpt_hip_fpipe_launches(self._h))

    def fpipe_stamps(self):
        """Stage-border times of the last float-weight persistent launch (BIOGPT_HIP_FPIPE_STAMPS=1): numpy uint64 [3 workgroups][32 layers][32], 10 ns units; None when inactive."""
        import numpy as np
        buf = (C.c_uint64 * 3072)()
        n = int(lib().biogpt_hip_fpipe_stamps(self._h, buf, 3072))
        return None if n != 3072 else np.frombuffer(buf, dtype=np.uint64).reshape(3, 32, 32).copy()

    def chunk_launches(self):
        """Evals of 2 .. 8 tokens that went through the column-per-XCD launch (biogpt_hip_chunk_launches)."""
        return int(lib().biogpt_hip_chunk_launches(self._h))

    def xpipe_state(self):
        """1: single-token decode steps of this context run as the XCD-pipelined persistent launch; 0: off / path held by
        another context of the device; -1: unavailable or abandoned after a disturbed launch."""
        return int(lib().biogpt_hip_xpipe_state(self._h))

    def refresh_options(self):
        """Re-read the BIOGPT_HIP_* switches (they are cached at load time) and drop the captured graphs."""
        if lib().biogpt_hip_refresh_options(self._h) != 0:
            raise BiogptError(_err())

    def bench_sweep(self, reps=20, which=0):
        """(seconds per launch, algorithmic bytes per launch, max |device - host| over sampled rows) of the all-matrices mat-vec launch (biogpt_hip_bench_sweep)."""
        secs, nbytes, chk = C.c_double(0.0), C.c_double(0.0), C.c_double(-1.0)
        if lib().biogpt_hip_bench_sweep(self._h, int(which), int(reps), C.byref(secs), C.byref(nbytes), C.byref(chk)) != 0:
            raise BiogptError(_err())
        return secs.value, nbytes.value, chk.value

    def bench_sweep_rows(self, which=0, reps=4):
        """The sweep launch's output rows and the two Q8 activation vectors they were computed with (biogpt_hip_bench_sweep_ex): (rows float32 [sum of the selected
        matrices' rows], xq int8 [1024 + 4096], xd float32 [32 + 128])."""
        hp = self.hparams
        per_layer = {0: 3 * hp.d_model + hp.d_model + hp.d_ff + hp.d_model, 2: 4 * hp.d_model, 3: hp.d_model, 4: hp.d_ff, 1: 0}[int(which)]
        n = per_layer * hp.n_layer + (hp.n_vocab if which in (0, 1) else 0)
        rows = np.zeros(n, np.float32); xq = np.zeros(1024 + 4096, np.int8); xd = np.zeros(32 + 128, np.float32)
        secs, nbytes, chk = C.c_double(0.0), C.c_double(0.0), C.c_double(-1.0)
        if lib().biogpt_hip_bench_sweep_ex(self._h, int(which), int(reps), C.byref(secs), C.byref(nbytes), C.byref(chk), rows.ctypes.data_as(C.POINTER(C.c_float)), n,
                                           xq.ctypes.data_as(C.POINTER(C.c_int8)), xd.ctypes.data_as(C.POINTER(C.c_float))) != 0:
            raise BiogptError(_err())
        return rows, xq, xd

    def bench_decode(self, n_past, reps=50):
        secs = C.c_double(0.0)
        if lib().biogpt_hip_bench_decode(self._h, int(n_past), int(reps), C.byref(secs)) != 0:
            raise BiogptError(_err())
        return secs.value

    def vocab_token(self, i):
        p, n = C.c_char_p(), C.c_int32()
        if lib().biogpt_hip_vocab_token(self._h, int(i), C.byref(p), C.byref(n)) != 0:
            raise IndexError(i)
        return C.string_at(p, n.value)

    @property
    def vocab(self):
        """The context's vocabulary handle (borrowed: valid until close()); None for an attached context."""
        h = lib().biogpt_hip_ctx_vocab(self._h)
        return Vocab(h, False) if h else None

    @property
    def arena(self):
        return lib().biogpt_hip_arena_ptr(self._h), int(lib().biogpt_hip_arena_bytes(self._h))

    def close(self):
        if self._h:
            lib().biogpt_hip_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Replicas:
    """biogpt_hip_replicas_*: one process, one context per device, weights loaded once and RCCL-broadcast."""

    def __init__(self, fname, devices, verbosity=0):
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self._h = lib().biogpt_hip_replicas_load(_bytes(fname), devs.ctypes.data, int(devs.size), int(verbosity))
        if not self._h:
            raise BiogptError(_err())

    @property
    def count(self):
        return lib().biogpt_hip_replicas_count(self._h)

    @property
    def broadcast_seconds(self):
        return lib().biogpt_hip_replicas_broadcast_seconds(self._h)

    def vocab_of(self, i):
        """The tokenizer handle of replica i (attached replicas share the root's tables)."""
        ctx = lib().biogpt_hip_replicas_ctx(self._h, int(i))
        h = lib().biogpt_hip_ctx_vocab(ctx)
        return Vocab(h, False) if h else None

    def generate_greedy(self, prompts, n_predict, n_batch=8):
        flat = np.ascontiguousarray([t for p in prompts for t in p], dtype=np.int32)
        lens = np.ascontiguousarray([len(p) for p in prompts], dtype=np.int32)
        out = np.zeros((len(prompts), int(n_predict)), dtype=np.int32)
        counts = np.zeros(len(prompts), dtype=np.int32)
        secs = C.c_double(0.0)
        rc = lib().biogpt_hip_replicas_generate_greedy(self._h, flat.ctypes.data, lens.ctypes.data, len(prompts), int(n_batch), int(n_predict),
                                                       out.ctypes.data, counts.ctypes.data, C.byref(secs))
        if rc < 0:
            raise BiogptError(_err())
        return [out[g, :counts[g]].copy() for g in range(len(prompts))], secs.value

    def close(self):
        if self._h:
            lib().biogpt_hip_replicas_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_bytes_per_token(hp, T):
    """Algorithmic HBM bytes of one decoded token at context T:
    W(type) + 2*L*T*D*4 (KV read) + 2*L*D*4 (KV write) + V*4 (logits)."""
    D, F, V, L = hp.d_model, hp.d_ff, hp.n_vocab, hp.n_layer
    bb = FILE_BLOCK_BYTES[hp.ftype] / 32.0
    mats = L * (4 * D * D + 2 * D * F) * bb + V * D * bb
    vecs = (L * (4 * D + 4 * D + F + D) + 2 * D) * 4
    rows = 2 * D * bb
    return mats + vecs + rows + 2 * L * T * D * 4 + 2 * L * D * 4 + V * 4
