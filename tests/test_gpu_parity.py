"""Parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the same
inputs.  Bar (north_star): greedy arg-max token ids exact, fp32 logits within 1e-3.  Because the
mat-vec kernel adds the per-block terms in the reference's scalar order, the HIP logits are in practice
bit-identical to the oracle's; the tests assert the contractual tolerance and report exactness."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-3  # north_star: "within 1e-3 on fp32 logits"
ALL_TYPES = ["f32", "f16", "q4_0", "q4_1", "q5_0", "q5_1", "q8_0"]
PROMPT = [2, 17, 45, 300, 9, 128, 64, 255, 31, 7, 199, 3, 77, 12, 290, 41, 8, 8, 150]


@pytest.fixture(scope="module")
def loaded(pkg, oracle, tiny_models):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = (pkg.BiogptModel.load(tiny_models[name]), oracle.OracleModel(tiny_models[name], n_threads=4))
        return cache[name]
    yield get
    for g, _ in cache.values():
        g.close()


def _check(got, ref, what):
    d = float(np.abs(got - ref).max())
    assert np.isfinite(got).all(), what
    assert d <= ATOL, "%s: max |diff| %.3e > %.0e" % (what, d, ATOL)
    return d


@pytest.mark.parametrize("name", ALL_TYPES)
def test_decode_token_by_token(loaded, name):
    g, o = loaded(name)
    worst, exact = 0.0, 0
    for j, t in enumerate(PROMPT):
        lg, lo = g.eval([t], j), o.eval([t], j)
        worst = max(worst, _check(lg, lo, "%s step %d" % (name, j)))
        assert int(lg.argmax()) == int(lo.argmax())
        exact += int((lg == lo).all())
    print("%s: worst |diff| %.2e, %d/%d steps bit-identical" % (name, worst, exact, len(PROMPT)))


@pytest.mark.parametrize("name", ALL_TYPES)
@pytest.mark.parametrize("n_batch", [8, 5, 19])
def test_chunked_prompt_no_mask(loaded, name, n_batch):
    """Prompt ingestion in chunks of n_batch (main.cpp:129-137); every token of a chunk attends to the
    whole chunk (F1), so results depend on the chunking exactly as in the reference."""
    g, o = loaded(name)
    n_past = 0
    while n_past < len(PROMPT):
        c = PROMPT[n_past:n_past + n_batch]
        _check(g.eval(c, n_past), o.eval(c, n_past), "%s chunk@%d" % (name, n_past))
        n_past += len(c)


@pytest.mark.parametrize("name", ["f32", "q4_0", "q5_1"])
def test_all_rows_and_kv_cache(loaded, name):
    g, o = loaded(name)
    toks = PROMPT[:11]
    _check(g.eval_all(toks, 0), o.eval(toks, 0, all_rows=True), name + " all rows")
    more = PROMPT[11:14]
    _check(g.eval_all(more, 11), o.eval(more, 11, all_rows=True), name + " all rows, n_past=11")
    L, P, D = o.n_layer, o.n_positions, o.d_model
    for which in (0, 1):
        kv = g.read_kv(which, 0, L * P * D).reshape(L, P, D)
        ref = o.kv(which)
        assert np.abs(kv[:, :14] - ref[:, :14]).max() <= 1e-4    # flat [layer][pos][d_model] (biogpt.cpp:722-726)


@pytest.mark.parametrize("name", ["q4_0", "q8_0", "f16"])
def test_greedy_generation_matches_oracle(loaded, pkg, oracle, tiny_models, name):
    prompt = [2, 17, 45, 300]
    g = pkg.BiogptModel.load(tiny_models[name])
    ids, secs = g.generate_greedy(prompt, 40, n_batch=8)     # device-resident loop, hipGraph replay per token
    ref, _ = oracle.OracleModel(tiny_models[name], n_threads=4).generate_greedy(prompt, 40, n_batch=8)
    assert len(ids) == 40 and secs > 0
    # teacher-forced agreement: wherever the streams agree so far, the next id must agree unless the
    # oracle's own top-2 margin is inside the logit tolerance
    o = oracle.OracleModel(tiny_models[name], n_threads=4)
    lg = o.eval(prompt, 0)
    n_past = len(prompt)
    for k in range(40):
        top2 = np.sort(lg)[-2:]
        if int(ids[k]) != int(lg.argmax()):
            assert top2[1] - top2[0] <= 2 * ATOL, "greedy id differs at step %d with margin %.3e" % (k, top2[1] - top2[0])
        lg = o.eval([int(ids[k])], n_past)
        n_past += 1
    assert (ids == ref).all()
    # the eval-API loop (host arg-max) gives the same ids as the device-resident loop
    lg = g.eval(prompt, 0)
    n_past, mine = len(prompt), []
    for _ in range(40):
        t = int(lg.argmax()); mine.append(t)
        lg = g.eval([t], n_past); n_past += 1
    assert mine == list(ids)
    g.close()


def test_generation_clamps_to_context(loaded, pkg, tiny_models):
    g = pkg.BiogptModel.load(tiny_models["q4_0"])
    ids, _ = g.generate_greedy([2] * 60, 200, n_batch=8)   # n_positions = 64 -> 4 tokens (main.cpp:82)
    assert len(ids) == 4
    g.close()


def test_eval_argument_errors(loaded, pkg):
    g, _ = loaded("q4_0")
    with pytest.raises(pkg.BiogptError):
        g.eval([2, 3], 63)            # n_past + N > n_positions
    with pytest.raises(pkg.BiogptError):
        g.eval([320], 0)              # id out of range
    with pytest.raises(pkg.BiogptError):
        g.eval([], 0)
    assert np.isfinite(g.eval([2], 0)).all()   # context still usable


def test_loader_semantic_failures(pkg, tiny_models, tmp_path):
    from modelfile_py import read_model, write_model
    hp, vocab, merges, tensors = read_model(tiny_models["f32"])
    p = str(tmp_path / "m.bin")
    write_model(p, hp, vocab, merges, tensors[:-1])                       # biogpt.cpp:444-447
    with pytest.raises(pkg.BiogptError, match="not all tensors"):
        pkg.BiogptModel.load(p)
    bad = [dict(t) for t in tensors]
    bad[3] = dict(bad[3], name="biogpt.layers.0.bogus.weight")            # biogpt.cpp:394-397
    write_model(p, hp, vocab, merges, bad)
    with pytest.raises(pkg.BiogptError, match="unknown tensor"):
        pkg.BiogptModel.load(p)
    bad = [dict(t) for t in tensors]
    i = [k for k, t in enumerate(bad) if t["name"].endswith("fc1.weight")][0]
    bad[i] = dict(bad[i], ne=[bad[i]["ne"][1], bad[i]["ne"][0]])          # biogpt.cpp:406-410
    write_model(p, hp, vocab, merges, bad)
    with pytest.raises(pkg.BiogptError, match="wrong shape"):
        pkg.BiogptModel.load(p)
    write_model(p, hp, vocab, merges, [])                                 # biogpt.cpp:442-443: warning + loads
    m = pkg.BiogptModel.load(p)
    assert m.n_tensors == 0
    with pytest.raises(pkg.BiogptError, match="empty model"):
        m.eval([2], 0)
    m.close()


def test_vocab_round_trip(loaded, pkg, tiny_models):
    from modelfile_py import read_model
    g, _ = loaded("q4_0")
    _, vocab, merges, _ = read_model(tiny_models["q4_0"])
    assert g.vocab_token(0) == vocab[0] and g.vocab_token(319) == vocab[319]
    assert g.hparams.n_merges == len(merges)


def test_external_arena_and_attach(pkg, oracle, tiny_models):
    """The multi-GPU path on one device: load into a caller-owned arena, copy the arena, attach a second
    context to the copy (what a non-root rank does after the RCCL broadcast) -> identical logits."""
    import torch
    g0 = pkg.BiogptModel.load(tiny_models["q5_0"])
    hp = g0.hparams
    nbytes = pkg.arena_bytes_for(hp)
    a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    g1 = pkg.BiogptModel.load(tiny_models["q5_0"], arena=a.data_ptr(), arena_bytes=nbytes)
    torch.cuda.synchronize()
    b = a.clone()
    torch.cuda.synchronize()
    g2 = pkg.BiogptModel.attach(hp, 0, b.data_ptr(), nbytes)
    ref = oracle.OracleModel(tiny_models["q5_0"]).eval(PROMPT[:6], 0)
    for g in (g0, g1, g2):
        _check(g.eval(PROMPT[:6], 0), ref, "arena variant")
        g.close()


LONG_PROMPT = [2] + [(37 * i + 11) % 316 + 4 for i in range(52)]     # 53 tokens (tiny n_positions = 64)


@pytest.fixture(params=[16, 512])
def prompt_cols(request, monkeypatch):
    """columns per pass of the prompt path: several passes (16) and the whole 53-token prompt in one (512, the default)"""
    monkeypatch.setenv("BIOGPT_HIP_PROMPT_COLS", str(request.param))
    return request.param


@pytest.mark.parametrize("name", ALL_TYPES)
@pytest.mark.parametrize("n_batch", [8, 3, 13])
def test_prompt_pass_equals_chunk_by_chunk_evals(loaded, pkg, tiny_models, prompt_cols, name, n_batch):
    """biogpt_hip_eval_prompt: several reference chunks per pass through the layers, each column limited to the keys
    its own n_batch-chunk would have seen -> the same logits, the same KV rows and the same continuation as
    the reference's chunk-by-chunk prompt loop (oracle), for chunk sizes that do and do not divide the pass."""
    g, o = loaded(name)
    g.refresh_options()          # the switches are cached per context; this one was loaded by an earlier test
    n_past, lo = 0, None
    while n_past < len(LONG_PROMPT):
        c = LONG_PROMPT[n_past:n_past + n_batch]
        lo = o.eval(c, n_past)
        n_past += len(c)
    lg = g.eval_prompt(LONG_PROMPT, 0, n_batch)
    d = _check(lg, lo, "%s prompt pass, n_batch %d" % (name, n_batch))
    L, P, D = o.n_layer, o.n_positions, o.d_model
    for which in (0, 1):
        kv = g.read_kv(which, 0, L * P * D).reshape(L, P, D)
        assert np.abs(kv[:, :len(LONG_PROMPT)] - o.kv(which)[:, :len(LONG_PROMPT)]).max() <= 1e-4
    nxt = int(lg.argmax())
    _check(g.eval([nxt], len(LONG_PROMPT)), o.eval([nxt], len(LONG_PROMPT)), "%s decode after the prompt pass" % name)
    # continuing a prompt at n_past > 0 (second half as its own call) gives the same thing when the split is on a chunk boundary
    h = pkg.BiogptModel.load(tiny_models[name])
    cut = 2 * n_batch
    h.eval_prompt(LONG_PROMPT[:cut], 0, n_batch, want_logits=False)
    l2 = h.eval_prompt(LONG_PROMPT[cut:], cut, n_batch)
    assert (l2 == lg).all()
    h.close()
    assert d == 0.0 or d <= ATOL


# ---- the quantizer's inner loops on the device -------------------------------------------------------------------

@pytest.mark.parametrize("type_id,name", [(2, "q4_0"), (3, "q4_1"), (6, "q5_0"), (7, "q5_1"), (8, "q8_0")])
def test_device_quantizer_is_byte_identical_to_the_oracle_codec(pkg, oracle, type_id, name):
    """biogpt_hip_quantize_rows_device against the oracle's restatement of ggml_quantize_* (byte-exact bar): weight-like rows,
    wide-range rows, and the edge blocks the reference's formats are sensitive to (all zeros, one extreme value of either sign,
    equal magnitudes of both signs, ties at .5 codes, tiny values, a constant block)."""
    rng = np.random.default_rng(100 + type_id)
    k = 1024
    rows = [rng.normal(0.0, 0.02, (64, k)), rng.normal(0.0, 1.0, (64, k)) * np.exp(rng.normal(0.0, 3.0, (64, 1))),
            rng.integers(-8, 9, (8, k)).astype(np.float64) * 0.125]
    edge = np.zeros((8, k), dtype=np.float64)
    edge[1, 0] = 3.0; edge[1, 40] = -3.0                     # equal magnitudes: the first one decides the sign of a symmetric scale
    edge[2, :32] = -7.5; edge[2, 5] = 7.5
    edge[3, :] = 0.3                                         # constant block: asymmetric scale 0
    edge[4, ::2] = 1e-36; edge[4, 1::2] = -1e-36             # tiny but normal: 1 / d stays finite (with a denormal d it overflows and the
                                                             # reference's own float -> int conversion is undefined: not compared)
    edge[5, :32] = np.arange(32) * 0.5 - 8.0                 # codes landing on .5 boundaries
    edge[6, :32] = np.linspace(-1.0, 1.0, 32); edge[6, 7] = 65504.0
    edge[7, 3] = -1e-30
    rows.append(edge)
    a = np.concatenate(rows).astype(np.float32)
    got = pkg.quantize_rows_device(a, type_id)
    want = oracle.quantize(type_id, a.reshape(-1), k)
    assert got.size == want.size
    bad = np.nonzero(got != np.frombuffer(bytes(want), dtype=np.uint8))[0]
    assert bad.size == 0, "%s: %d bytes differ, first at %d" % (name, bad.size, int(bad[0]))
