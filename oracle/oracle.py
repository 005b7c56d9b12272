"""ctypes binding of the CPU oracle (oracle/biogpt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See biogpt_oracle.h for the parity status
("parity unpinned" for the ggml arithmetic; format + model semantics pinned against
convert.py / HuggingFace BioGPT through tests/golden/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbiogpt_oracle.so")

TYPE_F32, TYPE_F16, TYPE_Q4_0, TYPE_Q4_1, TYPE_Q5_0, TYPE_Q5_1, TYPE_Q8_0, TYPE_Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
FTYPE_TO_TYPE = {0: TYPE_F32, 1: TYPE_F16, 2: TYPE_Q4_0, 3: TYPE_Q4_1, 7: TYPE_Q8_0, 8: TYPE_Q5_0, 9: TYPE_Q5_1}
FTYPE_NAMES = {0: "f32", 1: "f16", 2: "q4_0", 3: "q4_1", 7: "q8_0", 8: "q5_0", 9: "q5_1"}


def build(force=False):
    """Compile the C restatement (gcc, a few seconds)."""
    src = os.path.join(_HERE, "biogpt_oracle.c")
    hdr = os.path.join(_HERE, "biogpt_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libbiogpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Opts(C.Structure):
    _fields_ = [("gelu_erf", C.c_int), ("exp_f32", C.c_int), ("causal", C.c_int),
                ("ln_eps", C.c_float), ("n_threads", C.c_int), ("assoc", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.bo_fp32_to_fp16.restype = C.c_uint16
    L.bo_fp32_to_fp16.argtypes = [C.c_float]
    L.bo_fp16_to_fp32.restype = C.c_float
    L.bo_fp16_to_fp32.argtypes = [C.c_uint16]
    L.bo_gelu_table.restype = C.c_float
    L.bo_gelu_table.argtypes = [C.c_float]
    L.bo_exp_table.restype = C.c_float
    L.bo_exp_table.argtypes = [C.c_float]
    L.bo_type_block_bytes.restype = C.c_size_t
    L.bo_type_block_bytes.argtypes = [C.c_int]
    L.bo_row_bytes.restype = C.c_size_t
    L.bo_row_bytes.argtypes = [C.c_int, C.c_int64]
    L.bo_quantize.restype = C.c_size_t
    L.bo_quantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    L.bo_dequantize_row.restype = None
    L.bo_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    L.bo_vec_dot.restype = C.c_float
    L.bo_vec_dot_q.restype = C.c_float
    L.bo_vec_dot_q.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    L.bo_have_avx2.restype = C.c_int
    L.bo_have_avx2.argtypes = []
    L.bo_vec_dot.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    L.bo_load.restype = C.c_void_p
    L.bo_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.bo_free.restype = None
    L.bo_free.argtypes = [C.c_void_p]
    L.bo_hparams.restype = None
    L.bo_hparams.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.bo_set_opts.restype = None
    L.bo_set_opts.argtypes = [C.c_void_p, C.POINTER(_Opts)]
    L.bo_n_tensors.restype = C.c_int
    L.bo_n_tensors.argtypes = [C.c_void_p]
    L.bo_eval.restype = C.c_int
    L.bo_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.bo_tap.restype = C.c_int
    L.bo_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.bo_kv.restype = C.POINTER(C.c_float)
    L.bo_kv.argtypes = [C.c_void_p, C.c_int]
    L.bo_generate_greedy.restype = C.c_double
    L.bo_generate_greedy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.bo_quantize_file.restype = C.c_int
    L.bo_quantize_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def fp32_to_fp16(x):
    return int(lib().bo_fp32_to_fp16(float(x)))


def fp16_to_fp32(h):
    return float(lib().bo_fp16_to_fp32(int(h)))


def quantize(type_id, src, k):
    """Quantize a float32 array (rows of k) -> bytes, ggml block layout."""
    src = np.ascontiguousarray(src, dtype=np.float32).ravel()
    nbytes = lib().bo_row_bytes(type_id, k) * (src.size // k)
    dst = np.zeros(nbytes, dtype=np.uint8)
    n = lib().bo_quantize(type_id, src.ctypes.data, dst.ctypes.data, src.size, k)
    assert n == nbytes
    return dst


def dequantize_row(type_id, raw, k):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    out = np.zeros(k, dtype=np.float32)
    lib().bo_dequantize_row(type_id, raw.ctypes.data, out.ctypes.data, k)
    return out


def vec_dot(wtype, wrow_bytes, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(wrow_bytes, dtype=np.uint8)
    return float(lib().bo_vec_dot(wtype, x.size, w.ctypes.data, x.ctypes.data))


def quantize_file(src, dst, ftype):
    err = C.create_string_buffer(256)
    rc = lib().bo_quantize_file(src.encode(), dst.encode(), int(ftype), err, 256)
    if rc != 0:
        raise RuntimeError("oracle quantize_file: " + err.value.decode())


class OracleModel:
    """CPU restatement of biogpt_model_load + biogpt_eval (biogpt.cpp:27-453, :624-847)."""

    def __init__(self, path, n_threads=1, mode="ggml", assoc=0):
        self._assoc = int(assoc)
        err = C.create_string_buffer(256)
        self._h = lib().bo_load(path.encode(), err, 256)
        if not self._h:
            raise RuntimeError("oracle load failed: " + err.value.decode())
        hp = (C.c_int32 * 8)()
        lib().bo_hparams(self._h, hp)
        (self.n_vocab, self.n_layer, self.n_head, self.n_positions,
         self.d_ff, self.d_model, self.ftype, self.n_merges) = list(hp)
        self.n_tensors = lib().bo_n_tensors(self._h)
        self.set_mode(mode, n_threads)

    def set_mode(self, mode="ggml", n_threads=1, causal=None):
        o = _Opts()
        if mode == "hf":      # HuggingFace numerics: erf GELU, eps 1e-12, f32 exp, causal mask
            o.gelu_erf, o.exp_f32, o.causal, o.ln_eps = 1, 1, 1, 1e-12
        elif mode == "ggml":  # the reference's CPU path
            o.gelu_erf, o.exp_f32, o.causal, o.ln_eps = 0, 0, 0, 0.0
        else:
            raise ValueError(mode)
        if causal is not None:
            o.causal = int(causal)
        o.n_threads = int(n_threads)
        o.assoc = self._assoc     # 0: ggml's scalar fallbacks (the parity mode); 1: the shape of ggml's AVX2 kernels
        lib().bo_set_opts(self._h, C.byref(o))

    def eval(self, tokens, n_past, all_rows=False):
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        n = toks.size
        if all_rows:
            out = np.zeros((n, self.n_vocab), dtype=np.float32)
            rc = lib().bo_eval(self._h, toks.ctypes.data, n, int(n_past), None, out.ctypes.data)
        else:
            out = np.zeros(self.n_vocab, dtype=np.float32)
            rc = lib().bo_eval(self._h, toks.ctypes.data, n, int(n_past), out.ctypes.data, None)
        if rc != 0:
            raise RuntimeError("oracle eval failed rc=%d" % rc)
        return out

    def tap(self, layer):
        """Hidden state after `layer` (-1 = embeddings, n_layer = after final LayerNorm)."""
        buf = np.zeros((self.n_positions, self.d_model), dtype=np.float32)
        n = lib().bo_tap(self._h, int(layer), buf.ctypes.data)
        if n < 0:
            raise RuntimeError("no tap")
        return buf[:n].copy()

    def kv(self, which):
        p = lib().bo_kv(self._h, int(which))
        return np.ctypeslib.as_array(p, shape=(self.n_layer, self.n_positions, self.d_model))

    def generate_greedy(self, prompt, n_predict, n_batch=8):
        pr = np.ascontiguousarray(prompt, dtype=np.int32)
        n_predict = min(int(n_predict), self.n_positions - pr.size)
        out = np.zeros(n_predict, dtype=np.int32)
        t = lib().bo_generate_greedy(self._h, pr.ctypes.data, pr.size, int(n_batch), n_predict, out.ctypes.data)
        if t < 0:
            raise RuntimeError("oracle generate failed")
        return out, t

    def close(self):
        if self._h:
            lib().bo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def have_avx2():
    """True when the oracle's AVX2 + FMA forms of the SIMD-shaped dots run on this CPU (assoc bit 2 is then honoured)."""
    return bool(lib().bo_have_avx2())


def vec_dot_q(wtype, k, wrow_bytes, yblocks_bytes):
    """The reference's scalar vec_dot of one weight row (file-format bytes) against an activation row already quantized to Q8_0 / Q8_1 blocks."""
    return float(lib().bo_vec_dot_q(int(wtype), int(k), C.c_char_p(bytes(wrow_bytes)), C.c_char_p(bytes(yblocks_bytes))))
