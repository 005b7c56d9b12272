"""Parity at BioGPT-base matrix shapes (d_model 1024, d_ff 4096, n_vocab 42384, n_positions 1024) on a
synthetic seeded model with a reduced layer count so that the CPU oracle finishes in seconds; the
full 24-layer configuration is exercised by bench.py and by smoke()."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-3
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("full")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, **KW)
    out = {"f32": f32}
    for name in ("q4_0", "q5_1", "q8_0", "q4_1", "q5_0"):
        out[name] = str(d / (name + ".bin"))
        pkg.quantize_file(f32, out[name], name)
    return out


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0", "q4_1", "q5_0", "f32"])
def test_fullshape_decode_and_prefill(pkg, oracle, files, name):
    g = pkg.BiogptModel.load(files[name])
    o = oracle.OracleModel(files[name], n_threads=8)
    rng = np.random.default_rng(11)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 11)]
    worst = 0.0
    lg, lo = g.eval(prompt[:8], 0), o.eval(prompt[:8], 0)           # N=8 chunk (n_batch default)
    worst = max(worst, float(np.abs(lg - lo).max()))
    lg, lo = g.eval(prompt[8:], 8), o.eval(prompt[8:], 8)           # ragged tail chunk N=4
    worst = max(worst, float(np.abs(lg - lo).max()))
    n_past = len(prompt)
    exact = 0
    for _ in range(6):                                              # teacher-forced single-token decode
        t = int(lo.argmax())
        assert int(lg.argmax()) == t
        lg, lo = g.eval([t], n_past), o.eval([t], n_past)
        worst = max(worst, float(np.abs(lg - lo).max()))
        exact += int((lg == lo).all())
        n_past += 1
    big = [int(v) for v in rng.integers(4, KW["n_vocab"], 19)]       # one eval of 19 tokens: three column groups of the 8-wide kernels
    lg, lo = g.eval(big, n_past), o.eval(big, n_past)
    worst = max(worst, float(np.abs(lg - lo).max()))
    print("%s: worst |diff| %.2e, %d/6 decode steps bit-identical" % (name, worst, exact))
    assert worst <= ATOL
    g.close()


def test_fullshape_long_context_roundtrip(pkg, oracle, files):
    """Size-independent property at full context: evaluating the same token at n_past = 1023 after the
    cache was filled by chunked prefill equals the oracle (which walks the same 1024 positions)."""
    g = pkg.BiogptModel.load(files["q4_0"])
    o = oracle.OracleModel(files["q4_0"], n_threads=8)
    rng = np.random.default_rng(3)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 1023)]
    n_past = 0
    while n_past < 1016:
        g.eval_device(toks[n_past:n_past + 8], n_past)
        o.eval(toks[n_past:n_past + 8], n_past)
        n_past += 8
    lg, lo = g.eval(toks[1016:1024], 1016), o.eval(toks[1016:1024], 1016)   # T = 1024 keys
    assert np.abs(lg - lo).max() <= ATOL and int(lg.argmax()) == int(lo.argmax())
    with pytest.raises(pkg.BiogptError):
        g.eval([2], 1024)
    g.close()


def test_full_context_generation_all_graph_buckets(pkg, oracle, files):
    """Greedy generation up to the last position (main.cpp:82 clamp): walks every captured decode graph
    (context buckets 64 ... n_positions) and the long-context attention path; ids must equal the oracle's."""
    g = pkg.BiogptModel.load(files["q4_0"])
    o = oracle.OracleModel(files["q4_0"], n_threads=8)
    prompt = [2, 901, 17000, 33, 4100]
    ids, secs = g.generate_greedy(prompt, 5000, n_batch=8)           # clamped to 1024 - 5 = 1019
    assert len(ids) == KW["n_positions"] - len(prompt)
    ref, _ = o.generate_greedy(prompt, 5000, n_batch=8)
    same = int((np.asarray(ids) == np.asarray(ref)).sum())
    print("full-context generation: %d/%d ids identical, %.0f tok/s" % (same, len(ids), len(ids) / secs))
    assert same == len(ids)
    g.close()


@pytest.mark.parametrize("name,n_seqs", [("q4_0", 8), ("q5_1", 3), ("q8_0", 11)])
def test_batched_decode_equals_single_stream(pkg, oracle, files, name, n_seqs):
    """biogpt_hip_generate_greedy_batch: independent sequences decoded together (one activation column per
    sequence, own KV cache and position) must give exactly the ids each sequence gets on its own."""
    g = pkg.BiogptModel.load(files[name])
    rng = np.random.default_rng(21)
    prompts = [[2] + [int(v) for v in rng.integers(4, KW["n_vocab"], int(rng.integers(1, 14)))] for _ in range(n_seqs)]
    n_predict = 70                                               # crosses the 64-key context bucket
    ids, secs = g.generate_greedy_batch(prompts, n_predict, n_batch=8)
    assert ids.shape == (n_seqs, n_predict) and secs > 0
    for s in (0, n_seqs - 1):                                     # against the oracle
        ref, _ = oracle.OracleModel(files[name], n_threads=8).generate_greedy(prompts[s], n_predict, n_batch=8)
        assert list(ids[s]) == list(ref)
    for s in range(n_seqs):                                       # against the single-stream device loop
        single, _ = g.generate_greedy(prompts[s], n_predict, n_batch=8)
        assert list(ids[s]) == list(single), "sequence %d" % s
    again, _ = g.generate_greedy_batch(prompts[:2], 5)            # smaller batch on the same context: graphs re-captured
    assert list(again[1]) == list(ids[1][:5])
    g.close()


@pytest.mark.parametrize("name", ["f32", "f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_biogpt_large_shapes_generic_path(pkg, oracle, tmp_path_factory, name):
    """BioGPT-large widths (d_model 1600 = 50 blocks, d_ff 6400 = 200 blocks, 25 heads of 64): not powers of two,
    so the shape-specialised kernels do not apply and the generic kernels run with masked lanes, 4 units per lane
    and the non-power-of-two LayerNorm division.  Same bit-parity bar."""
    d = tmp_path_factory.mktemp("large")
    kw = dict(n_vocab=2048, n_layer=2, n_head=25, n_positions=96, d_ff=6400, d_model=1600, n_merges=3)
    f32 = str(d / "f32.bin")
    if name in ("f32", "f16"):
        path = str(d / (name + ".bin"))
        pkg.write_synthetic(path, ftype=int(name == "f16"), **kw)
    else:
        pkg.write_synthetic(f32, **kw)
        path = str(d / (name + ".bin"))
        pkg.quantize_file(f32, path, name)
    g = pkg.BiogptModel.load(path)
    o = oracle.OracleModel(path, n_threads=8)
    rng = np.random.default_rng(9)
    toks = [2] + [int(v) for v in rng.integers(4, kw["n_vocab"], 20)]
    worst = 0.0
    lg, lo = g.eval(toks[:8], 0), o.eval(toks[:8], 0)
    worst = max(worst, float(np.abs(lg - lo).max()))
    lg, lo = g.eval(toks[8:13], 8), o.eval(toks[8:13], 8)
    worst = max(worst, float(np.abs(lg - lo).max()))
    for j in range(13, 21):
        lg, lo = g.eval([toks[j]], j), o.eval([toks[j]], j)
        worst = max(worst, float(np.abs(lg - lo).max()))
        assert int(lg.argmax()) == int(lo.argmax())
    ids, _ = g.generate_greedy(toks[:5], 12)
    ref, _ = oracle.OracleModel(path, n_threads=8).generate_greedy(toks[:5], 12)
    print("%s large-shape: worst |diff| %.2e" % (name, worst))
    assert worst <= ATOL and list(ids) == list(ref)
    g.close()


SHAPES = [
    # (n_vocab, n_layer, n_head, n_positions, d_ff, d_model)
    (77, 1, 3, 40, 160, 96),        # 3 blocks per row, odd vocab, dk = 32
    (130, 2, 2, 48, 1024, 256),     # dk = 128
    (64, 1, 1, 34, 32, 32),         # one block per row, one head
    (300, 1, 16, 36, 8192, 2048),   # dk = 128, fc2 rows of 256 blocks
    (1000, 1, 5, 33, 96, 320),      # d_ff < d_model, dk = 64, position table shorter than one 64-key tile
    (50, 1, 2, 200, 64, 128),       # dk = 64 with tables that are not multiples of 64 keys: every decode
    (50, 1, 2, 300, 64, 128),       #   attention variant (<=256, <=512, <=1024 keys) meets a ragged last tile
    (50, 1, 2, 90, 64, 128),        # dk = 64, a table that is not a multiple of 4 rows and a pass of >= 80 columns: the register-tiled pass attention
    (50, 1, 2, 201, 64, 128),       #   (one clamped base for 4 key rows) must not be taken; keys 88, 89 / 200 are scored against their own rows
    (50, 1, 2, 600, 64, 128),
    (50, 1, 2, 1000, 64, 128),
    (50, 1, 2, 1500, 64, 128),      # beyond 1024 keys: generic attention
    (500, 1, 16, 70, 4096, 1024),   # BioGPT-base widths (specialised chain) on a ragged 70-position table
    (500, 1, 8, 70, 2048, 1024),    # K = 1024 specialised mat-vecs mixed with generic fc2 and dk = 128 attention
    (500, 1, 32, 70, 4096, 1024),   # base widths but dk = 32: specialised mat-vecs without the Q8 hand-off chain
    (90, 1, 4, 70, 1024, 256),      # fc2 with K = 1024 writing a 256-wide residual
    (90, 1, 4, 70, 4096, 256),      # fc2 with K = 4096 writing a 256-wide residual
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "V%d_L%d_H%d_P%d_F%d_D%d" % s)
@pytest.mark.parametrize("name", ["f32", "f16", "q4_0", "q5_0", "q8_0"])
def test_odd_shapes(pkg, oracle, tmp_path_factory, shape, name):
    """Shapes the loader accepts but BioGPT-base never exercises (odd block counts, dk of 32/128, one head,
    d_ff < d_model): generic kernels, bit parity with the oracle for a prompt chunk, a ragged chunk, single
    tokens up to the LAST position, and greedy ids."""
    V, L, H, P, F, D = shape
    d = tmp_path_factory.mktemp("odd")
    kw = dict(n_vocab=V, n_layer=L, n_head=H, n_positions=P, d_ff=F, d_model=D, n_merges=2)
    if name in ("f32", "f16"):
        path = str(d / (name + ".bin"))
        pkg.write_synthetic(path, ftype=int(name == "f16"), **kw)
    else:
        f32 = str(d / "f32.bin")
        pkg.write_synthetic(f32, **kw)
        path = str(d / (name + ".bin"))
        pkg.quantize_file(f32, path, name)
    g = pkg.BiogptModel.load(path)
    o = oracle.OracleModel(path, n_threads=8)
    rng = np.random.default_rng(V + D)
    toks = [2] + [int(v) for v in rng.integers(0, V, P - 1)]
    worst, pos = 0.0, 0
    for n in (8, 3, 1, 1, 9, 1):
        lg, lo = g.eval(toks[pos:pos + n], pos), o.eval(toks[pos:pos + n], pos)
        worst = max(worst, float(np.abs(lg - lo).max()))
        pos += n
    while pos < P - 70:                 # long tables: ragged prompt chunks up to the last 70 positions
        lg, lo = g.eval(toks[pos:pos + 7], pos), o.eval(toks[pos:pos + 7], pos)
        worst = max(worst, float(np.abs(lg - lo).max()))
        pos += 7
    while pos < P:                      # single tokens to the last position of the table
        lg, lo = g.eval([toks[pos]], pos), o.eval([toks[pos]], pos)
        worst = max(worst, float(np.abs(lg - lo).max()))
        pos += 1
    ids, _ = g.generate_greedy(toks[:4], P)          # clamped to P - 4
    ref, _ = oracle.OracleModel(path, n_threads=8).generate_greedy(toks[:4], P)
    assert len(ids) == P - 4
    assert worst <= ATOL and list(ids) == list(ref), worst
    g.close()


@pytest.mark.parametrize("cols", [32, 128, 512])
@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0", "f32"])
def test_prompt_pass_full_shape(pkg, oracle, files, monkeypatch, name, cols):
    """BioGPT-base widths (specialised chain / generic float kernels): a 203-token prompt through
    biogpt_hip_eval_prompt (-b 8: 26 reference chunks, `cols` columns per pass) against the oracle fed chunk by chunk."""
    monkeypatch.setenv("BIOGPT_HIP_PROMPT_COLS", str(cols))     # 7 passes, 2 passes, the whole prompt in one pass
    path = files[name]
    g = pkg.BiogptModel.load(path)
    o = oracle.OracleModel(path, n_threads=16)
    rng = np.random.default_rng(5)
    hp = g.hparams
    toks = [2] + [int(v) for v in rng.integers(4, hp.n_vocab, 202)]
    lo = None
    for at in range(0, len(toks), 8):
        lo = o.eval(toks[at:at + 8], at)
    lg = g.eval_prompt(toks, 0, 8)
    d = float(np.abs(lg - lo).max())
    print("%s prompt pass: worst |diff| %.2e" % (name, d))
    assert d <= ATOL and int(lg.argmax()) == int(lo.argmax())
    ids, _ = g.generate_greedy(toks, 6, n_batch=8)          # the device loop ingests the prompt the same way
    ref, _ = oracle.OracleModel(path, n_threads=16).generate_greedy(toks, 6, n_batch=8)
    assert list(ids) == list(ref)
    g.close()


@pytest.mark.parametrize("name", ["q4_0", "q4_1", "q5_0", "q8_0"])
def test_batched_decode_on_the_matrix_cores(pkg, oracle, files, name):
    """52 sequences decoded together: enough columns for the int8-MFMA chain (16-wide tiles, row-tiled weight image).
    Every sequence's ids must equal the oracle's single-stream ids, like in the 8-column path."""
    g = pkg.BiogptModel.load(files[name])
    rng = np.random.default_rng(21)
    prompts = [[2] + [int(v) for v in rng.integers(4, KW["n_vocab"], int(rng.integers(2, 7)))] for _ in range(52)]
    ids, _ = g.generate_greedy_batch(prompts, 7)
    for s in (0, 7, 19, 51):
        ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompts[s], 7)
        assert list(ids[s]) == list(ref), (name, s)
    single = pkg.BiogptModel.load(files[name])
    for s in (3, 33):
        one, _ = single.generate_greedy(prompts[s], 7)
        assert list(ids[s]) == list(one)
    single.close()
    g.close()


@pytest.mark.parametrize("name", ["q4_0", "q5_1"])
def test_prompt_pass_full_context(pkg, oracle, files, name):
    """A 1024-token prompt (-b 8: 128 reference chunks) in two passes of 512 columns -- the matrix-core chain, the
    grouped-query attention at up to 1024 keys -- against the oracle fed chunk by chunk; every 64th chunk's KV rows too."""
    g = pkg.BiogptModel.load(files[name])
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(17)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 1023)]
    lo = None
    for at in range(0, 1024, 8):
        lo = o.eval(toks[at:at + 8], at)
    lg = g.eval_prompt(toks, 0, 8)
    d = float(np.abs(lg - lo).max())
    print("%s 1024-token prompt pass: worst |diff| %.2e" % (name, d))
    assert d <= ATOL and int(lg.argmax()) == int(lo.argmax())
    L, P, D = KW["n_layer"], KW["n_positions"], KW["d_model"]
    for which in (0, 1):
        kv = g.read_kv(which, 0, L * P * D).reshape(L, P, D)
        ref = o.kv(which)
        for pos in (0, 7, 8, 511, 512, 1000, 1023):
            assert np.abs(kv[:, pos] - ref[:, pos]).max() <= 1e-4, (which, pos)
    g.close()


@pytest.mark.parametrize("shape", [s for s in SHAPES if s[3] <= 300], ids=lambda s: "V%d_L%d_H%d_P%d_F%d_D%d" % s)
@pytest.mark.parametrize("name", ["f32", "q4_0", "q8_0"])
def test_prompt_pass_odd_shapes(pkg, oracle, tmp_path_factory, shape, name):
    """The pass form of prompt ingestion on the generic kernels (shapes the specialised chain does not cover, head sizes
    32 / 64 / 128, position tables that are not multiples of 64): the whole table as one prompt, -b 4 and -b 7."""
    V, L, H, P, F, D = shape
    d = tmp_path_factory.mktemp("oddp")
    kw = dict(n_vocab=V, n_layer=L, n_head=H, n_positions=P, d_ff=F, d_model=D, n_merges=2)
    if name == "f32":
        path = str(d / "f32.bin")
        pkg.write_synthetic(path, **kw)
    else:
        f32 = str(d / "f32.bin")
        pkg.write_synthetic(f32, **kw)
        path = str(d / (name + ".bin"))
        pkg.quantize_file(f32, path, name)
    rng = np.random.default_rng(P + D)
    toks = [2] + [int(v) for v in rng.integers(0, V, P - 1)]
    for nb in (4, 7):
        g = pkg.BiogptModel.load(path)
        o = oracle.OracleModel(path, n_threads=8)
        lo = None
        for at in range(0, P, nb):
            lo = o.eval(toks[at:at + nb], at)
        lg = g.eval_prompt(toks, 0, nb)
        assert float(np.abs(lg - lo).max()) <= ATOL and int(lg.argmax()) == int(lo.argmax()), (shape, name, nb)
        g.close()


@pytest.mark.parametrize("cols", [16, 512])
@pytest.mark.parametrize("name", ["q4_0", "q5_1"])
def test_batched_prompts_travel_together(pkg, oracle, files, monkeypatch, name, cols):
    """generate_greedy_batch ingests the prompts of all sequences in common passes (every token a column that knows its
    sequence, position and the end of its own n_batch-chunk): prompts of 1 .. 61 tokens, -b 4 (several ragged chunks per
    sequence), packed into passes of 16 columns (many passes, chunk-aligned cuts) or one pass.  Ids per sequence must
    equal the oracle's single-stream ids with the same -b."""
    monkeypatch.setenv("BIOGPT_HIP_PROMPT_COLS", str(cols))
    g = pkg.BiogptModel.load(files[name])
    rng = np.random.default_rng(31)
    lens = [1, 4, 9, 17, 26, 5, 61, 2, 33, 12]
    prompts = [[2] + [int(v) for v in rng.integers(4, KW["n_vocab"], n - 1)] for n in lens]
    ids, _ = g.generate_greedy_batch(prompts, 6, n_batch=4)
    for s in range(len(prompts)):
        ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompts[s], 6, n_batch=4)
        assert list(ids[s]) == list(ref), (name, cols, s, lens[s])
    g.close()


@pytest.mark.parametrize("name", ["f32", "f16"])
def test_float_weight_decode_kernels(pkg, oracle, files, tmp_path_factory, monkeypatch, name):
    """F32 / F16 files (biogpt.cpp:160-165) at BioGPT-base shapes: the single-token mat-vecs of csrc/kernels_fdecode.hip.h (whole matrix
    requested at once, one workgroup per compute unit) against the generic kernel they replace (BIOGPT_HIP_NO_FDEC=1: bit-identical logits
    and K / V rows) and the oracle (1e-3, north_star), single-token steps at short and long contexts + a greedy continuation."""
    path = files["f32"]
    if name == "f16":
        path = str(tmp_path_factory.mktemp("f16") / "f16.bin")
        pkg.write_synthetic(path, ftype=1, **KW)
    g = pkg.BiogptModel.load(path)
    monkeypatch.setenv("BIOGPT_HIP_NO_FDEC", "1")
    u = pkg.BiogptModel.load(path)
    monkeypatch.delenv("BIOGPT_HIP_NO_FDEC")
    o = oracle.OracleModel(path, n_threads=16)
    rng = np.random.default_rng(3)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 300)]
    checked = [0, 1, 8, 9, 63, 64, 255, 256, 299]
    n_past, worst = 0, 0.0
    while n_past <= checked[-1]:
        if n_past in checked:
            lg, lu, lo = g.eval([toks[n_past]], n_past), u.eval([toks[n_past]], n_past), o.eval([toks[n_past]], n_past)
            assert (lg == lu).all(), "%s: fdec != generic at n_past %d (max diff %g)" % (name, n_past, np.abs(lg - lu).max())
            kg = g.read_kv(0, ((KW["n_layer"] - 1) * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"])
            ku = u.read_kv(0, ((KW["n_layer"] - 1) * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"])
            assert (kg == ku).all()
            worst = max(worst, float(np.abs(lg - lo).max()))
            assert int(lg.argmax()) == int(lo.argmax())
            n_past += 1
        else:
            m = 1
            while (n_past + m) not in checked and m < 8:
                m += 1
            chunk = toks[n_past:n_past + m]
            g.eval_device(chunk, n_past); u.eval_device(chunk, n_past); o.eval(chunk, n_past)
            n_past += m
    print("%s: float-weight decode kernels worst |diff| vs oracle %.2e" % (name, worst))
    assert worst <= ATOL
    ids_g, _ = g.generate_greedy(toks[:5], 40, n_batch=8)
    ids_u, _ = u.generate_greedy(toks[:5], 40, n_batch=8)
    ref, _ = oracle.OracleModel(path, n_threads=16).generate_greedy(toks[:5], 40, n_batch=8)
    assert list(ids_g) == list(ids_u) == list(ref)
    g.close(); u.close()
