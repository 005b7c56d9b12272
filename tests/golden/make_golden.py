#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

Runs ONLY in the build container (needs /root/reference and HuggingFace transformers); the
fixtures it writes are data (inputs + expected outputs) and are what travels to the GPU box.

G1  tiny_f32.bin / tiny_f16.bin : a tiny seeded BioGPT written by the reference's OWN writer
    functions (convert.py:28-97 parse_hparams / parse_vocab / parse_bpe_merges / parse_model),
    i.e. the format oracle.  Dims are multiples of 32 so every quant type applies.
G1' tiny_hf_logits.npz          : HuggingFace BioGptForCausalLM fp32 logits for 3 prompts, every
    position (causal mask == incremental decode), the semantic cross-check for the restatement
    in "hf" mode.
G2  tiny_state.npz              : the seeded state dict itself (float32), so tests can rebuild
    files through the build's own writer and compare bytes with convert.py's output (G4).

Usage: python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")

CFG = dict(vocab_size=320, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
           intermediate_size=256, max_position_embeddings=64)
N_MERGES = 7
PROMPTS = [
    [2, 17, 45, 300, 9, 128, 64, 255, 31, 7, 199, 3],
    [2, 5, 5, 5, 77, 310, 12, 90],
    [2, 211, 19, 19, 250, 4, 101, 33, 280, 150, 61, 8, 2, 17, 45, 222],
]


def main():
    sys.path.insert(0, str(REF))
    import convert as refconv  # the reference's converter (module-level code only builds an argparser)
    from transformers import BioGptConfig, BioGptForCausalLM

    torch.manual_seed(1234)
    cfg = BioGptConfig(**CFG)
    model = BioGptForCausalLM(cfg).eval()

    # re-draw every parameter so that attention / FFN / biases all matter numerically
    g = torch.Generator().manual_seed(20260928)
    sd = model.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if name == "output_projection.weight":
                continue  # tied to embed_tokens below
            if t.ndim == 2:
                std = 0.35 if "embed" in name else 0.18
                t.copy_(torch.randn(t.shape, generator=g) * std)
            elif name.endswith("layer_norm.weight"):
                t.copy_(1.0 + 0.2 * torch.randn(t.shape, generator=g))
            else:
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
        sd["biogpt.embed_tokens.weight"][1].zero_()  # pad row, as HF keeps it
        sd["output_projection.weight"].copy_(sd["biogpt.embed_tokens.weight"])
    model.load_state_dict(sd)
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}

    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        hf_cfg = dict(vocab_size=CFG["vocab_size"], num_hidden_layers=CFG["num_hidden_layers"],
                      num_attention_heads=CFG["num_attention_heads"],
                      max_position_embeddings=CFG["max_position_embeddings"],
                      intermediate_size=CFG["intermediate_size"], hidden_size=CFG["hidden_size"])
        (td / "config.json").write_text(json.dumps(hf_cfg))
        vocab = {("tok%d</w>" % i if i % 3 else "t%d" % i): i for i in range(CFG["vocab_size"])}
        (td / "vocab.json").write_text(json.dumps(vocab))
        merges = ["#version: 0.2"] + ["a%d b%d 1" % (i, i) for i in range(N_MERGES - 1)]
        (td / "merges.txt").write_text("\n".join(merges) + "\n")
        for use_f16, fname in ((False, "tiny_f32.bin"), (True, "tiny_f16.bin")):
            with open(HERE / fname, "wb") as out:
                refconv.parse_hparams(td, out, use_f16)
                refconv.parse_vocab(td, out)
                refconv.parse_bpe_merges(td, out)
                refconv.parse_model(sd, out, use_f16)

    logits = {}
    with torch.no_grad():
        for i, p in enumerate(PROMPTS):
            out = model(input_ids=torch.tensor([p])).logits[0].float().numpy()
            logits["prompt%d" % i] = np.asarray(p, dtype=np.int32)
            logits["logits%d" % i] = out.astype(np.float32)
    np.savez_compressed(HERE / "tiny_hf_logits.npz", **logits)
    np.savez_compressed(HERE / "tiny_state.npz", **{k: v.numpy() for k, v in sd.items()})
    meta = dict(CFG, n_merges=N_MERGES, hidden_act=cfg.hidden_act, layer_norm_eps=cfg.layer_norm_eps,
                scale_embedding=cfg.scale_embedding, transformers=__import__("transformers").__version__,
                torch=torch.__version__)
    (HERE / "tiny_meta.json").write_text(json.dumps(meta, indent=1))
    for f in sorted(HERE.iterdir()):
        print("%-24s %8d bytes" % (f.name, f.stat().st_size))


if __name__ == "__main__":
    main()
