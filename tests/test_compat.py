"""The C++ source-level boundary (include/biogpt_compat.h): reference-shaped programs compile against it
and run on the engine."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu

REF_MAIN = "/root/reference/examples/main/main.cpp"
INC = os.path.join(ROOT, "include", "compat")


def _build_driver(pkg, tmp_path):
    exe = str(tmp_path / "compat_driver")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", INC, os.path.join(ROOT, "tests", "compat_driver.cpp"),
                           "-L", libdir, "-lbiogpt_hip", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources only exist in the build container")
def test_reference_main_compiles_unmodified_against_compat_headers():
    # syntax + semantic check of the reference's own CLI source against OUR headers (no reference header is used)
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I", INC, REF_MAIN])


def test_driver_links_and_fails_loudly_without_model(pkg, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    r = subprocess.run([exe, str(tmp_path / "missing.bin"), "4", "1", "2", "5"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to open" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["q4_0", "q5_1"])
def test_driver_greedy_ids_match_oracle(pkg, oracle, tiny_models, tmp_path, name):
    exe = _build_driver(pkg, tmp_path)
    prompt = [2, 17, 45, 300, 9, 128, 64, 255, 31, 7, 199]        # 11 ids -> chunks of 8 + 3 (n_batch = 8)
    r = subprocess.run([exe, tiny_models[name], "24", "1"] + [str(t) for t in prompt], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ids = [int(t) for t in r.stdout.split()]
    ref, _ = oracle.OracleModel(tiny_models[name], n_threads=2).generate_greedy(prompt, 24, n_batch=8)
    assert ids == list(ref)
    assert "vocab 320 tokens, 7 merges, n_loaded 37" in r.stderr


def test_params_parse_flags(pkg, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    r = subprocess.run([exe, "--flags", "--bogus", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "unknown argument: --bogus" in r.stderr and "--top_k N" in r.stderr   # biogpt.cpp:1011-1015
    r = subprocess.run([exe, "--flags", "-m"], capture_output=True, text=True)
    assert r.returncode == 2 and "missing value" in r.stderr
    r = subprocess.run([exe, "--flags", "-m", str(tmp_path / "nope.bin"), "-n", "3", "--top_k", "1", "-p", "2 5"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to open" in r.stderr


@pytest.mark.gpu
def test_flags_drive_generation(pkg, oracle, tiny_models, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    r = subprocess.run([exe, "--flags", "-m", tiny_models["q4_1"], "-n", "12", "--top_k", "1", "-b", "4", "-p", "2 17 45 300 9 128"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref, _ = oracle.OracleModel(tiny_models["q4_1"]).generate_greedy([2, 17, 45, 300, 9, 128], 12, n_batch=4)
    assert [int(t) for t in r.stdout.split()] == list(ref)


@pytest.mark.gpu
def test_sampler_top_k_top_p_is_seeded_and_bounded(pkg, tiny_models, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    a = subprocess.run([exe, tiny_models["q8_0"], "16", "40", "2", "17"], capture_output=True, text=True)
    b = subprocess.run([exe, tiny_models["q8_0"], "16", "40", "2", "17"], capture_output=True, text=True)
    assert a.returncode == 0 and a.stdout == b.stdout          # mt19937(7): reproducible
    ids = [int(t) for t in a.stdout.split()]
    assert len(ids) == 16 and all(0 <= t < 320 for t in ids)


# ---- text in, text out: tokenizer + engine + sampler through the reference-shaped API -------------------------
GOLD = os.path.join(ROOT, "tests", "golden")
TOK_DATA = os.path.join(GOLD, "tokenizer_data")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "biogpt_ref_cli")
PROMPT = "The patient was treated with metformin. Dr. Hale reported no adverse events"


def _text_model(pkg, tmp_path, ftype="q8_0"):
    """A seeded tiny model carrying the tokenizer fixture's vocabulary and merge table."""
    import modelfile_py
    _, vocab, merges, _ = modelfile_py.read_model(os.path.join(GOLD, "tokenizer_vocab.bin"))
    raw = str(tmp_path / "raw.bin")
    pkg.write_synthetic(raw, n_vocab=len(vocab), n_layer=2, n_head=4, n_positions=96, d_ff=256, d_model=64, n_merges=3)
    hp, _, _, tensors = modelfile_py.read_model(raw)
    f32 = str(tmp_path / "text_f32.bin")
    modelfile_py.write_model(f32, hp, vocab, merges, tensors)
    out = str(tmp_path / ("text_%s.bin" % ftype))
    pkg.quantize_file(f32, out, ftype)
    return out, vocab


def _expected_text_run(pkg, oracle, model, vocab_strings, n_predict, n_batch):
    pkg.set_tokenizer_data_dir(TOK_DATA)
    v = pkg.Vocab.load(model)
    ids = v.tokenize(PROMPT, "")
    gen, _ = oracle.OracleModel(model, n_threads=2).generate_greedy(ids, n_predict, n_batch=n_batch)
    return v, ids, [int(t) for t in gen]


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources only exist in the build container")
def test_reference_cli_links_against_the_engine(pkg):
    """examples/main/main.cpp, unmodified, + include/compat + libbiogpt_hip.so = a complete program (every symbol it
    needs, tokenizer included, is exported); without a model it fails the way the reference does."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_cli"], stdout=subprocess.DEVNULL)
    r = subprocess.run([REF_CLI, "-m", "/nonexistent/model.bin", "-p", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "failed to load model from '/nonexistent/model.bin'" in r.stderr


@pytest.mark.gpu
def test_driver_text_in_text_out(pkg, oracle, tmp_path):
    exe = _build_driver(pkg, tmp_path)
    model, vocab = _text_model(pkg, tmp_path)
    v, ids, gen = _expected_text_run(pkg, oracle, model, vocab, 20, 8)
    assert len(ids) > 12
    r = subprocess.run([exe, "--text", "-m", model, "-n", "20", "--top_k", "1", "-b", "8", "-p", PROMPT], capture_output=True,
                       env=dict(os.environ, BIOGPT_DATA_DIR=TOK_DATA))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split(b"\n")
    assert lines[0] == b"prompt ids: " + " ".join(str(i) for i in ids).encode()
    assert [int(t) for t in lines[1].split()] == gen
    assert lines[2] == b"text: " + v.decode(ids + gen, "")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="oracle/_ref/biogpt_ref_cli is built from the reference's main.cpp in the build container")
def test_reference_cli_end_to_end_on_the_engine(pkg, oracle, tmp_path):
    """The reference's own CLI program (its main.cpp compiled against include/compat, oracle/Makefile ref_cli)
    running on the MI355X engine: prompt text in, generated text out, equal to tokenizer + oracle + decode."""
    model, vocab = _text_model(pkg, tmp_path, "q4_0")
    v, ids, gen = _expected_text_run(pkg, oracle, model, vocab, 16, 8)
    r = subprocess.run([REF_CLI, "-m", model, "-n", "16", "--top_k", "1", "-b", "8", "-s", "3", "-p", PROMPT], capture_output=True,
                       env=dict(os.environ, BIOGPT_DATA_DIR=TOK_DATA))
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert ("number of tokens in prompt = %d, first 8 tokens: %s \n" % (len(ids), " ".join(str(i) for i in ids[:8]))).encode() in out
    # main.cpp:139-145 prints one decoded chunk + ' ' per loop iteration: prompt chunks of n_batch ids, then one id at a time
    chunks = [ids[i:i + 8] for i in range(0, len(ids), 8)] + [[g] for g in gen]
    body = b"".join(v.decode(c, "") + b" " for c in chunks)
    assert body in out, (body, out)


REF_QUANTIZE = os.path.join(ROOT, "oracle", "_ref", "quantize_ref_cli")


@pytest.mark.skipif(not (os.path.exists("/root/reference/examples/quantize/quantize.cpp") or os.path.exists(REF_QUANTIZE)),
                    reason="needs the reference's quantize.cpp (build container) or the binary built from it")
@pytest.mark.parametrize("name,ftype", [("q4_0", 2), ("q4_1", 3), ("q8_0", 7), ("q5_0", 8), ("q5_1", 9)])
def test_reference_quantize_cli_on_our_library(pkg, oracle, tmp_path, name, ftype):
    """examples/quantize/quantize.cpp, unmodified, against include/compat + our biogpt_model_quantize_internal:
    output file byte-identical to the oracle's quantizer; F16 input too; progress lines as the reference prints."""
    if os.path.exists("/root/reference/examples/quantize/quantize.cpp"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_quantize"], stdout=subprocess.DEVNULL)
    for src_type in (0, 1):
        src = str(tmp_path / ("src%d.bin" % src_type))
        pkg.write_synthetic(src, ftype=src_type, n_vocab=96, n_layer=2, n_head=2, n_positions=32, d_ff=128, d_model=64)  # 40000 merges: the CLI insists (F6)
        out, want = str(tmp_path / "cli.bin"), str(tmp_path / "oracle.bin")
        r = subprocess.run([REF_QUANTIZE, "-f", src, "-o", out, "-t", str(ftype)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.rstrip().endswith("Done."), r.stderr
        oracle.quantize_file(src, want, ftype)
        with open(out, "rb") as a, open(want, "rb") as b:
            assert a.read() == b.read()
        assert "biogpt.embed_tokens.weight - [   64,    96], type =    %s size =     0.02 MB ->" % ("f32" if src_type == 0 else "f16") in r.stdout
        assert "ftype = %d (%s)" % (ftype, name) in r.stdout
    r = subprocess.run([REF_QUANTIZE, "-f", src, "-o", out, "-t", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "invalid model type 1" in r.stderr        # uncaught std::runtime_error, like the reference
