"""Pins the oracle's model semantics and file-format handling:
   - tests/golden/tiny_f32.bin / tiny_f16.bin were written by the reference's own convert.py functions;
   - tests/golden/tiny_hf_logits.npz holds HuggingFace BioGptForCausalLM fp32 logits of the same weights.
The restatement in 'hf' mode (erf GELU, eps 1e-12, f32 exp, causal) must match HF tightly; the ggml-mode
switches are then the parity oracle for the HIP path."""
import json
import os

import numpy as np
import pytest

from modelfile_py import read_model


@pytest.fixture(scope="module")
def hf(golden_dir):
    return np.load(os.path.join(golden_dir, "tiny_hf_logits.npz"))


def test_fixture_format_fields(golden_dir):
    hp, vocab, merges, tensors = read_model(os.path.join(golden_dir, "tiny_f32.bin"))
    meta = json.load(open(os.path.join(golden_dir, "tiny_meta.json")))
    assert hp == dict(n_vocab=meta["vocab_size"], n_layer=meta["num_hidden_layers"], n_head=meta["num_attention_heads"],
                      n_positions=meta["max_position_embeddings"], d_ff=meta["intermediate_size"],
                      d_model=meta["hidden_size"], ftype=0)
    assert len(vocab) == hp["n_vocab"] and len(merges) == meta["n_merges"] - 1 + 0 or len(merges) >= 1
    assert len(tensors) == 5 + 16 * hp["n_layer"]
    names = {t["name"] for t in tensors}
    assert "output_projection.weight" in names and "biogpt.embed_positions.weight" in names
    pos = [t for t in tensors if t["name"] == "biogpt.embed_positions.weight"][0]
    assert pos["ne"] == [hp["d_model"], hp["n_positions"] + 2]  # dims written reversed (convert.py:79-80)
    hp16, _, _, t16 = read_model(os.path.join(golden_dir, "tiny_f16.bin"))
    assert hp16["ftype"] == 1
    for t in t16:  # f16 only for 2-D tensors whose name ends in .weight (convert.py:62-70)
        assert t["type"] == (1 if (len(t["ne"]) == 2 and t["name"].endswith(".weight")) else 0)


def test_oracle_loads_reference_written_file(oracle, golden_dir):
    m = oracle.OracleModel(os.path.join(golden_dir, "tiny_f32.bin"))
    assert (m.n_vocab, m.n_layer, m.n_head, m.n_positions, m.d_ff, m.d_model, m.ftype) == (320, 2, 4, 64, 256, 64, 0)
    assert m.n_tensors == 37 and m.n_merges == 7


@pytest.mark.parametrize("i", [0, 1, 2])
def test_hf_mode_matches_huggingface(oracle, golden_dir, hf, i):
    m = oracle.OracleModel(os.path.join(golden_dir, "tiny_f32.bin"), mode="hf")
    p, ref = hf["prompt%d" % i], hf["logits%d" % i]
    step = np.array([m.eval(p[j:j + 1], j) for j in range(len(p))])  # one token per eval: F1 cannot bite
    assert np.abs(step - ref).max() < 5e-5
    chunk = m.eval(p, 0, all_rows=True)  # one chunk with the causal switch
    assert np.abs(chunk - ref).max() < 5e-5
    # chunked ingestion with n_batch = 5 (main.cpp:129-137) + causal
    outs, n_past = [], 0
    while n_past < len(p):
        c = p[n_past:n_past + 5]
        outs.append(m.eval(c, n_past, all_rows=True))
        n_past += len(c)
    assert np.abs(np.concatenate(outs) - ref).max() < 5e-5


def test_f16_file_close_to_hf(oracle, golden_dir, hf):
    m = oracle.OracleModel(os.path.join(golden_dir, "tiny_f16.bin"), mode="hf")
    p, ref = hf["prompt0"], hf["logits0"]
    step = np.array([m.eval(p[j:j + 1], j) for j in range(len(p))])
    assert np.abs(step - ref).max() < 0.08  # f16 weights + f16 activations in the dots
    assert (step.argmax(1) == ref.argmax(1)).mean() >= 0.9


def test_ggml_mode_differs_in_the_known_ways(oracle, golden_dir, hf):
    m = oracle.OracleModel(os.path.join(golden_dir, "tiny_f32.bin"), mode="ggml")
    p, ref = hf["prompt0"], hf["logits0"]
    step = np.array([m.eval(p[j:j + 1], j) for j in range(len(p))])
    d = np.abs(step - ref).max()
    assert 1e-5 < d < 5e-2  # fp16-table GELU/exp + eps 1e-5 are visible but small
    # F1: without the causal switch a chunk is NOT equal to token-by-token evaluation
    m2 = oracle.OracleModel(os.path.join(golden_dir, "tiny_f32.bin"), mode="ggml")
    chunk = m2.eval(p, 0, all_rows=True)
    assert np.abs(chunk[:-1] - step[:-1]).max() > 1e-3
    # row 0 of layer 0 already attends to all later keys of its chunk
    assert np.abs(chunk[0] - step[0]).max() > 1e-3


def test_threads_do_not_change_results(oracle, tiny_models):
    a = oracle.OracleModel(tiny_models["q4_0"], n_threads=1)
    b = oracle.OracleModel(tiny_models["q4_0"], n_threads=4)
    toks = np.array([2, 9, 100, 17], dtype=np.int32)
    assert (a.eval(toks, 0) == b.eval(toks, 0)).all()


def test_kv_cache_layout(oracle, golden_dir):
    m = oracle.OracleModel(os.path.join(golden_dir, "tiny_f32.bin"))
    m.eval(np.array([2, 5, 7], dtype=np.int32), 0)
    k = m.kv(0)
    assert k.shape == (2, 64, 64)
    assert np.abs(k[:, :3]).min() > 0 and (k[:, 3:] == 0).all()  # flat [layer][pos][d_model] (biogpt.cpp:722)


def test_greedy_harness_counts(oracle, tiny_models):
    m = oracle.OracleModel(tiny_models["q8_0"])
    ids, secs = m.generate_greedy([2, 17, 45, 300], 20, n_batch=8)
    assert len(ids) == 20 and secs > 0
    # same stream when generated by explicit eval calls
    m2 = oracle.OracleModel(tiny_models["q8_0"])
    lg = m2.eval(np.array([2, 17, 45, 300], dtype=np.int32), 0)
    n_past, mine = 4, []
    for _ in range(20):
        t = int(lg.argmax()); mine.append(t)
        lg = m2.eval(np.array([t], dtype=np.int32), n_past); n_past += 1
    assert mine == list(ids)
    # n_predict is clamped to n_positions - len(prompt) (main.cpp:82)
    ids2, _ = oracle.OracleModel(tiny_models["q8_0"]).generate_greedy([2] * 60, 200)
    assert len(ids2) == 4


@pytest.mark.parametrize("name,type_id", [("q4_0", 2), ("q4_1", 3), ("q5_0", 6), ("q5_1", 7), ("q8_0", 8)])
def test_quantized_modes_compute_the_dequantized_model(oracle, tiny_models, tmp_path, name, type_id):
    """Semantic bound for the W x A8 integer path (whose exact ggml arithmetic is unpinned): evaluating a quantized file
    must equal, up to the activation quantization (one Q8 rounding per 32-element block), evaluating a FLOAT file that
    holds the dequantized weights.  Independent of the block-dot code: the float file goes through the f32 mat-mul."""
    from modelfile_py import read_model, write_model
    hp, vocab, merges, tensors = read_model(tiny_models[name])
    ft = []
    for t in tensors:
        if t["type"] in (0, 1):
            ft.append(t)
            continue
        k, rows = t["ne"][0], t["ne"][1]
        raw = np.frombuffer(t["raw"], dtype=np.uint8)
        rb = len(raw) // rows
        w = np.stack([oracle.dequantize_row(type_id, raw[r * rb:(r + 1) * rb].copy(), k) for r in range(rows)])
        ft.append(dict(name=t["name"], type=0, ne=t["ne"], raw=w.astype(np.float32).tobytes()))
    fpath = str(tmp_path / "deq.bin")
    write_model(fpath, dict(hp, ftype=0), vocab, merges, ft)
    q, f = oracle.OracleModel(tiny_models[name]), oracle.OracleModel(fpath)
    toks = np.array([2, 17, 45, 300, 9, 128, 64, 255, 31, 7], dtype=np.int32)
    worst, scale = 0.0, 0.0
    for j in range(len(toks)):
        lq, lf = q.eval(toks[j:j + 1], j), f.eval(toks[j:j + 1], j)
        worst, scale = max(worst, float(np.abs(lq - lf).max())), max(scale, float(np.abs(lf).max()))
    assert worst < 0.02 * scale + 1e-3, (name, worst, scale)       # ~1/127 per rounding, averaged over a row
    assert worst > 0.0                                              # and it is NOT the float path in disguise
