"""Pure-Python reader/writer of ggml-model.bin for tests."""
import struct

import numpy as np

MAGIC = 0x67676D6C
BLOCK_BYTES = {2: 18, 3: 20, 6: 22, 7: 24, 8: 34}


def read_model(path):
    with open(path, "rb") as f:
        data = f.read()
    off = 0

    def i32():
        nonlocal off
        (v,) = struct.unpack_from("<i", data, off)
        off += 4
        return v

    assert i32() == MAGIC
    hp = dict(zip(["n_vocab", "n_layer", "n_head", "n_positions", "d_ff", "d_model", "ftype"], [i32() for _ in range(7)]))
    vocab, merges = [], []
    for lst in (vocab, merges):
        n = i32()
        for _ in range(n):
            ln = i32()
            lst.append(data[off:off + ln])
            off += ln
    tensors = []
    while off < len(data):
        n_dims, ln, ttype = i32(), i32(), i32()
        ne = [i32() for _ in range(n_dims)]
        name = data[off:off + ln].decode()
        off += ln
        nel = int(np.prod(ne))
        if ttype == 0:
            nb = nel * 4
        elif ttype == 1:
            nb = nel * 2
        else:
            nb = nel // 32 * BLOCK_BYTES[ttype]
        tensors.append(dict(name=name, type=ttype, ne=ne, raw=data[off:off + nb]))
        off += nb
    return hp, vocab, merges, tensors


def write_model(path, hp, vocab, merges, tensors):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", MAGIC))
        for k in ["n_vocab", "n_layer", "n_head", "n_positions", "d_ff", "d_model", "ftype"]:
            f.write(struct.pack("<i", hp[k]))
        for lst in (vocab, merges):
            f.write(struct.pack("<i", len(lst)))
            for s in lst:
                f.write(struct.pack("<i", len(s)))
                f.write(s)
        for t in tensors:
            nm = t["name"].encode()
            f.write(struct.pack("<iii", len(t["ne"]), len(nm), t["type"]))
            for e in t["ne"]:
                f.write(struct.pack("<i", e))
            f.write(nm)
            f.write(t["raw"])
