"""Converter counterpart: HF checkpoint directory -> ggml-model.bin."""
import json
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"


@pytest.fixture(scope="module")
def hf_dir(tmp_path_factory):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    d = tmp_path_factory.mktemp("hf")
    cfg = tr.BioGptConfig(vocab_size=96, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, max_position_embeddings=32)
    torch.manual_seed(3)
    model = tr.BioGptForCausalLM(cfg).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(0.05 * torch.randn_like(p))
    torch.save(model.state_dict(), d / "pytorch_model.bin")
    json.dump(dict(vocab_size=96, num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=32,
                   intermediate_size=128, hidden_size=64), open(d / "config.json", "w"))
    json.dump({("w%d</w>" % i): i for i in range(96)}, open(d / "vocab.json", "w"))
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join("x%d y%d 7" % (i, i) for i in range(9)) + "\n")
    return str(d), model


def test_converted_file_matches_hf_logits(pkg, oracle, hf_dir, tmp_path):
    import torch
    from biogpt_cpp_amd import convert_hf
    d, model = hf_dir
    out = convert_hf.convert(d, str(tmp_path / "m.bin"))
    o = oracle.OracleModel(out, mode="hf")
    assert (o.n_vocab, o.n_layer, o.n_head, o.n_positions, o.d_ff, o.d_model, o.ftype, o.n_merges) == (96, 2, 4, 32, 128, 64, 0, 10)
    toks = [2, 5, 17, 90, 33, 8]
    with torch.no_grad():
        ref = model(input_ids=torch.tensor([toks])).logits[0].numpy()
    got = o.eval(toks, 0, all_rows=True)
    # default HF init (std 0.02) makes the pre-LayerNorm variance tiny, which amplifies f32-vs-double round-off
    assert np.abs(got - ref).max() < 5e-4 and (got.argmax(1) == ref.argmax(1)).all()
    f16 = convert_hf.convert(d, str(tmp_path / "h.bin"), use_f16=True)
    assert oracle.OracleModel(f16).ftype == 1


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "convert.py")), reason="reference only exists in the build container")
@pytest.mark.parametrize("f16", [False, True])
def test_bytes_identical_to_reference_converter(pkg, hf_dir, tmp_path, f16):
    import torch
    from pathlib import Path
    from biogpt_cpp_amd import convert_hf
    sys.path.insert(0, REF)
    import convert as refconv
    d, _ = hf_dir
    mine = convert_hf.convert(d, str(tmp_path / "mine.bin"), use_f16=f16)
    ref = str(tmp_path / "ref.bin")
    sd = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu")
    with open(ref, "wb") as out:
        refconv.parse_hparams(Path(d), out, f16)
        refconv.parse_vocab(Path(d), out)
        refconv.parse_bpe_merges(Path(d), out)
        refconv.parse_model(sd, out, f16)
    assert open(mine, "rb").read() == open(ref, "rb").read()
