#!/usr/bin/env python3
"""HuggingFace BioGPT checkpoint -> ggml-model.bin (counterpart of the reference's
convert.py:28-119, same byte stream -- Appendix B of SURVEY.md):

    magic, 7 x int32 header (n_vocab, n_layer, n_head, n_positions, d_ff, d_model, ftype)
    vocab   : count, then (len, utf-8 bytes) sorted by id
    merges  : count, then (len, "left right") in file order (first two fields of each line)
    tensors : (n_dims, name_len, ttype, reversed dims, name, raw data) in state-dict order;
              --f16 stores 2-D "*.weight" tensors as float16, everything else float32

Reads `pytorch_model.bin` or `model.safetensors` + `config.json` + `vocab.json` + `merges.txt` from
--dir-model.  Pure host tool (numpy/torch on CPU), no GPU.

    python biogpt.cpp_amd/convert_hf.py --dir-model DIR --out FILE [--f16]
"""
import argparse
import json
import os
import struct

import numpy as np

MAGIC = 0x67676D6C


def load_state_dict(dir_model):
    st = os.path.join(dir_model, "model.safetensors")
    if os.path.exists(st):
        from safetensors.numpy import load_file
        sd = load_file(st)
        # safetensors drops tied duplicates: the reference file stores both
        if "output_projection.weight" not in sd and "biogpt.embed_tokens.weight" in sd:
            sd["output_projection.weight"] = sd["biogpt.embed_tokens.weight"]
        return sd
    import torch
    sd = torch.load(os.path.join(dir_model, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    return {k: v.float().numpy() if v.dtype not in (torch.float32, torch.float16) else v.numpy() for k, v in sd.items()}


def convert(dir_model, out_path, use_f16=False):
    cfg = json.load(open(os.path.join(dir_model, "config.json"), encoding="utf-8"))
    vocab = json.load(open(os.path.join(dir_model, "vocab.json"), encoding="utf-8"))
    with open(os.path.join(dir_model, "merges.txt"), encoding="utf-8") as f:
        merge_lines = f.read().split("\n")[:-1]
    sd = load_state_dict(dir_model)
    with open(out_path, "wb") as out:
        out.write(struct.pack("<8i", MAGIC, cfg["vocab_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"],
                              cfg["max_position_embeddings"], cfg["intermediate_size"], cfg["hidden_size"], int(use_f16)))
        tokens = sorted(vocab.items(), key=lambda kv: kv[1])
        out.write(struct.pack("<i", len(tokens)))
        for tok, _ in tokens:
            b = tok.encode("utf-8")
            out.write(struct.pack("<i", len(b)) + b)
        out.write(struct.pack("<i", len(merge_lines)))
        for line in merge_lines:
            b = " ".join(line.split()[:2]).encode("utf-8")
            out.write(struct.pack("<i", len(b)) + b)
        for name, arr in sd.items():
            a = np.squeeze(np.asarray(arr))
            is_matrix = a.ndim == 2 and name.endswith(".weight")
            a = a.astype(np.float16 if (use_f16 and is_matrix) else np.float32)
            nm = name.encode("utf-8")
            out.write(struct.pack("<3i", a.ndim, len(nm), 1 if a.dtype == np.float16 else 0))
            for d in reversed(a.shape):
                out.write(struct.pack("<i", d))
            out.write(nm)
            out.write(np.ascontiguousarray(a).tobytes())
    return out_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir-model", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--f16", action="store_true")
    args = ap.parse_args()
    print("wrote", convert(args.dir_model, args.out, args.f16))
