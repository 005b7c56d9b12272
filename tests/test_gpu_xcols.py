"""biogpt_eval with 2 .. 8 tokens -- the chunks of the reference's prompt loop (main.cpp:129-137; no mask inside an eval, F1; biogpt.cpp:664-811) -- through the
column-per-XCD persistent launch (csrc/kernels_xcols.hip.h) against (a) the launch chain it replaces (BIOGPT_HIP_XCOLS=0) bit for bit, logits and appended
K / V rows, and (b) the oracle within the contract; all five block formats, chunk sizes 2 .. 8, every context variant (<= 64 / 128 / 256 / 512 keys) and the border
where the chain takes over (n_past + N > 512); float files keep the chain; the full 24-layer model; a disturbed launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-3
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
NIBBLE = ["q4_0", "q4_1", "q5_0", "q5_1"]
BLOCK = NIBBLE + ["q8_0"]


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("xcols")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, seed=77, **KW)
    out = {"f32": f32}
    for name in BLOCK:
        out[name] = str(d / (name + ".bin"))
        pkg.quantize_file(f32, out[name], name)
    return out


def _opts(g, monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g.refresh_options()
    for k in env:
        monkeypatch.delenv(k)


def _kv(g, n_past, n):
    last = (KW["n_layer"] - 1) * KW["n_positions"]
    return [g.read_kv(w, (last + n_past) * KW["d_model"], n * KW["d_model"]) for w in (0, 1)]


@pytest.mark.parametrize("name", BLOCK)
def test_chunk_launch_equals_the_launch_chain_and_the_oracle(pkg, oracle, files, monkeypatch, name):
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device (xpipe_state %d)" % g.xpipe_state())
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(61)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 539)]
    # borders at 64, 128, 256 keys crossed by chunks; then on to the 512-key variant (the second half of a head's old rows requested inside the attention stage) and across
    # ITS border at 512 keys, where the launch chain takes over
    sizes = [8, 2, 3, 8, 5, 7, 8, 4, 6, 8, 8, 8, 5, 8, 3, 8] + [8] * 12 + [7, 8, 2, 8, 8, 6, 8, 3, 5, 8, 8] + [8] * 6 + [3, 8, 5] + [8] * 20 + [7, 8, 6, 8, 8]
    n_past, worst, through = 0, 0.0, 0
    for n in sizes:
        if n_past + n > 530:
            break
        chunk = toks[n_past:n_past + n]
        _opts(g, monkeypatch, BIOGPT_HIP_XCOLS="1")
        before = g.chunk_launches()
        lx = g.eval(chunk, n_past)
        used = g.chunk_launches() - before
        assert used == (1 if n_past + n <= 512 else 0), "n_past %d n %d: %d chunk launches" % (n_past, n, used)
        through += used
        assert g.xpipe_state() == 1, "pipeline abandoned at n_past %d" % n_past
        kx = _kv(g, n_past, n)
        _opts(g, monkeypatch, BIOGPT_HIP_XCOLS="0")
        mid = g.chunk_launches()
        lc = g.eval(chunk, n_past)
        assert g.chunk_launches() == mid
        kc = _kv(g, n_past, n)
        lo = o.eval(chunk, n_past)
        assert (lx == lc).all(), "%s: n_past %d n %d: chunk launch != launch chain (max diff %g)" % (name, n_past, n, np.abs(lx - lc).max())
        assert (kx[0] == kc[0]).all() and (kx[1] == kc[1]).all(), (n_past, n)
        worst = max(worst, float(np.abs(lx - lo).max()))
        assert int(lx.argmax()) == int(lo.argmax())
        n_past += n
    assert n_past > 512 and through >= 60
    print("%s: %d chunk evals through the column-per-XCD launch, worst |diff| vs oracle %.2e" % (name, through, worst))
    assert worst <= ATOL
    g.close()


def test_float_files_keep_the_launch_chain(pkg, oracle, files):
    """F32 / F16 files (biogpt.cpp:160-165) have no pipelined launches: their evals of 2 .. 8 tokens stay on the generic chain (and are still right)."""
    g = pkg.BiogptModel.load(files["f32"])
    o = oracle.OracleModel(files["f32"], n_threads=16)
    chunk = [2, 100, 2000, 37, 4000, 5, 77, 901]
    lg, lo = g.eval(chunk, 0), o.eval(chunk, 0)
    assert g.chunk_launches() == 0
    assert np.abs(lg - lo).max() <= ATOL and int(lg.argmax()) == int(lo.argmax())
    g.close()


def test_asynchronous_chunks_then_a_token_and_generation(pkg, oracle, files):
    """The reference's loop shape: prompt chunks without reading their rows (biogpt_hip_eval_device), then single-token evals on top of the K / V rows the chunk
    launches appended (resident pipelined launch); and biogpt_hip_generate_greedy, whose 8-token prompt pass is one chunk launch."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    o = oracle.OracleModel(files["q4_0"], n_threads=16)
    rng = np.random.default_rng(67)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 60)]
    for at in range(0, 40, 8):
        g.eval_device(toks[at:at + 8], at); o.eval(toks[at:at + 8], at)
    g.eval_device(toks[40:43], 40); o.eval(toks[40:43], 40)
    assert g.chunk_launches() == 6
    for k in range(43, 48):
        lg, lo = g.eval([toks[k]], k), o.eval([toks[k]], k)
        assert np.abs(lg - lo).max() <= ATOL and int(lg.argmax()) == int(lo.argmax()), k
    assert g.xpipe_state() == 1
    before = g.chunk_launches()
    ids, _ = g.generate_greedy(toks[:8], 24, n_batch=8)
    assert g.chunk_launches() == before + 1
    ref, _ = oracle.OracleModel(files["q4_0"], n_threads=16).generate_greedy(toks[:8], 24, n_batch=8)
    assert list(ids) == list(ref)
    g.close()


def test_disturbed_chunk_launch_is_repeated_on_the_launch_chain(pkg, files, monkeypatch, capfd):
    """BIOGPT_HIP_XPIPE_FAULT=1 hands XCD 0 a 33rd ticket: the chunk launch drains with garbage, the call reports it, repeats itself on the chain and returns
    the right row; the context has left the pipeline."""
    ref = pkg.BiogptModel.load(files["q5_0"])
    if ref.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    chunk = [2, 11, 222, 3333, 44, 5]
    want = ref.eval(chunk, 0)
    ref.close()
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    g = pkg.BiogptModel.load(files["q5_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    got = g.eval(chunk, 0)
    assert g.chunk_launches() == 1 and g.xpipe_state() == -1
    assert (got == want).all()
    assert "pipelined decode step failed" in capfd.readouterr().err
    g.close()


def test_24_layers_prompt_in_chunks_of_8(pkg, oracle, tmp_path):
    """BioGPT-base depth (biogpt.h:25-35): a 48-token prompt as six chunk launches, rows and the last layer's K / V rows against the oracle."""
    kw = dict(KW, n_layer=24, n_vocab=8192, n_merges=100)
    f32, q = str(tmp_path / "f32.bin"), str(tmp_path / "q4_0.bin")
    pkg.write_synthetic(f32, seed=24, **kw)
    pkg.quantize_file(f32, q, "q4_0")
    g = pkg.BiogptModel.load(q)
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    o = oracle.OracleModel(q, n_threads=16)
    rng = np.random.default_rng(68)
    toks = [2] + [int(v) for v in rng.integers(4, kw["n_vocab"], 47)]
    worst = 0.0
    for at in range(0, 48, 8):
        lg, lo = g.eval(toks[at:at + 8], at), o.eval(toks[at:at + 8], at)
        worst = max(worst, float(np.abs(lg - lo).max()))
        assert int(lg.argmax()) == int(lo.argmax()), at
    assert g.chunk_launches() == 6 and g.xpipe_state() == 1
    for w in (0, 1):
        ref = o.kv(w)
        for l in (0, 11, 23):
            got = g.read_kv(w, l * kw["n_positions"] * kw["d_model"], 48 * kw["d_model"]).reshape(48, kw["d_model"])
            assert np.abs(got - ref[l, :48]).max() <= ATOL, (w, l)
    print("24 layers: worst |diff| vs oracle %.2e" % worst)
    assert worst <= ATOL
    g.close()


def test_24_layers_chunks_of_8_from_296_to_512_keys(pkg, oracle, tmp_path):
    """The second half of a 512-token prompt as the reference's unchanged loop issues it (main.cpp:129-137: biogpt_eval per 8 tokens) at BioGPT-base depth: 27 chunk
    launches of the 512-key variant on top of a 296-token prefix, every returned row and the K / V rows of layers 0 / 11 / 23 against the oracle."""
    kw = dict(KW, n_layer=24, n_vocab=8192, n_merges=100)
    f32, q = str(tmp_path / "f32.bin"), str(tmp_path / "q4_0.bin")
    pkg.write_synthetic(f32, seed=25, **kw)
    pkg.quantize_file(f32, q, "q4_0")
    g = pkg.BiogptModel.load(q)
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    o = oracle.OracleModel(q, n_threads=16)
    rng = np.random.default_rng(69)
    toks = [2] + [int(v) for v in rng.integers(4, kw["n_vocab"], 511)]
    g.eval_prompt(toks[:296], 0, 8)
    for at in range(0, 296, 8):
        o.eval(toks[at:at + 8], at)
    before, worst, exact = g.chunk_launches(), 0.0, 0
    for at in range(296, 512, 8):
        lg, lo = g.eval(toks[at:at + 8], at), o.eval(toks[at:at + 8], at)
        worst = max(worst, float(np.abs(lg - lo).max()))
        exact += int((lg == lo).all())
        assert int(lg.argmax()) == int(lo.argmax()), at
    assert g.chunk_launches() - before == 27 and g.xpipe_state() == 1
    for w in (0, 1):
        ref = o.kv(w)
        for l in (0, 11, 23):
            got = g.read_kv(w, (l * kw["n_positions"] + 296) * kw["d_model"], 216 * kw["d_model"]).reshape(216, kw["d_model"])
            assert np.abs(got - ref[l, 296:512]).max() <= ATOL, (w, l)
    print("24 layers, chunks at 296 .. 512 keys: worst |diff| vs oracle %.2e, %d/27 rows bit-identical" % (worst, exact))
    assert worst <= ATOL
    g.close()


def test_two_contexts_take_turns_with_chunk_evals(pkg, files, monkeypatch):
    """Two contexts of one process on one device alternate 8-token evals: the device's pipeline slot goes to whoever asks while the other's stream is idle
    (csrc/engine_xpipe.inc); whichever path an eval takes, rows and K / V rows are those of the launch chain."""
    monkeypatch.setenv("BIOGPT_HIP_XCOLS", "0")
    ref = pkg.BiogptModel.load(files["q4_1"])
    monkeypatch.delenv("BIOGPT_HIP_XCOLS")
    a, b = pkg.BiogptModel.load(files["q4_1"]), pkg.BiogptModel.load(files["q4_1"])
    if a.xpipe_state() < 0:
        pytest.skip("XCD pipeline not available on this device")
    rng = np.random.default_rng(71)
    ta = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 63)]
    tb = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 63)]
    want_a, want_b = [], []
    for at in range(0, 64, 8):
        want_a.append(ref.eval(ta[at:at + 8], at).copy())
    ka = _kv(ref, 0, 64)
    for at in range(0, 64, 8):
        want_b.append(ref.eval(tb[at:at + 8], at).copy())
    kb = _kv(ref, 0, 64)
    for k, at in enumerate(range(0, 64, 8)):
        ra = a.eval(ta[at:at + 8], at)
        if k % 2:
            b.eval_device(tb[at:at + 8], at)          # left in flight while the other context asks for the slot
            rb = None
        else:
            rb = b.eval(tb[at:at + 8], at)
        assert (ra == want_a[k]).all(), k
        if rb is not None:
            assert (rb == want_b[k]).all(), k
    b.synchronize()
    for w in (0, 1):
        assert (_kv(a, 0, 64)[w] == ka[w]).all() and (_kv(b, 0, 64)[w] == kb[w]).all()
    assert a.chunk_launches() + b.chunk_launches() >= 8      # most evals found the slot free
    assert a.xpipe_state() >= 0 and b.xpipe_state() >= 0
    ref.close(); a.close(); b.close()


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_batched_generation_steps_as_column_per_xcd_launches(pkg, oracle, files, monkeypatch, name):
    """biogpt_hip_generate_greedy_batch with 2 .. 8 sequences: every decode step is ONE launch with one sequence per XCD (streams mode of kernels_xcols.hip.h: own
    position, own K / V cache, no exchange between the XCDs), replayed from a graph per context bucket.  Ragged prompts, contexts that cross the 64 / 128-key variants and
    the 256-key border (beyond it the launch chain takes over inside the same call): ids equal the launch chain's (BIOGPT_HIP_XCOLS=0) and, sequence by sequence, the
    oracle's single-sequence generation."""
    rng = np.random.default_rng(73)
    cases = [((4, 9, 17, 5, 8, 30, 2, 11), 70), ((60, 61, 7), 80), ((5, 5), 262), ((120,) * 8, 16)]
    for lens, n_predict in cases:
        prompts = [[2] + [int(v) for v in rng.integers(4, KW["n_vocab"], n - 1)] for n in lens]
        g = pkg.BiogptModel.load(files[name])
        if g.xpipe_state() != 1:
            pytest.skip("XCD pipeline not available on this device")
        ids_x, _ = g.generate_greedy_batch(prompts, n_predict, n_batch=8)
        assert g.chunk_launches() > 0 and g.xpipe_state() == 1
        g.close()
        monkeypatch.setenv("BIOGPT_HIP_XCOLS", "0")
        g = pkg.BiogptModel.load(files[name])
        monkeypatch.delenv("BIOGPT_HIP_XCOLS")
        ids_c, _ = g.generate_greedy_batch(prompts, n_predict, n_batch=8)
        assert g.chunk_launches() == 0
        g.close()
        assert (np.asarray(ids_x) == np.asarray(ids_c)).all(), (name, lens)
        for s in (0, len(lens) - 1):
            ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompts[s], min(n_predict, 24), n_batch=8)
            assert list(np.asarray(ids_x)[s][:len(ref)]) == list(ref), (name, lens, s)


def test_disturbed_batched_generation_is_repeated_on_the_launch_chain(pkg, files, monkeypatch, capfd):
    ref = pkg.BiogptModel.load(files["q4_0"])
    if ref.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompts = [[2, 10 + s, 200, 3000 + s] for s in range(6)]
    want, _ = ref.generate_greedy_batch(prompts, 12, n_batch=8)
    ref.close()
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    got, _ = g.generate_greedy_batch(prompts, 12, n_batch=8)
    assert (np.asarray(got) == np.asarray(want)).all()
    assert g.xpipe_state() == -1
    assert "pipelined decode step failed" in capfd.readouterr().err
    g.close()


def test_chunk_launch_with_a_short_position_table(pkg, oracle, tmp_path, monkeypatch):
    """n_positions = 100 (biogpt.h:25-35 allows any >= the context): the 128-key variant runs with its loads bounded by the table's end (range-checked buffer loads
    for the K / V rows), 2 layers, a small vocabulary; chunks up to the last position; rows and K / V rows against the chain and the oracle."""
    kw = dict(KW, n_layer=2, n_positions=100, n_vocab=5000, n_merges=100)
    f32, q = str(tmp_path / "f32.bin"), str(tmp_path / "q5_0.bin")
    pkg.write_synthetic(f32, seed=100, **kw)
    pkg.quantize_file(f32, q, "q5_0")
    g = pkg.BiogptModel.load(q)
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    o = oracle.OracleModel(q, n_threads=16)
    rng = np.random.default_rng(72)
    toks = [2] + [int(v) for v in rng.integers(4, kw["n_vocab"], 99)]
    n_past = 0
    for n in (8, 8, 8, 8, 8, 8, 8, 8, 6, 8, 8, 7, 7):
        chunk = toks[n_past:n_past + n]
        _opts(g, monkeypatch, BIOGPT_HIP_XCOLS="1")
        before = g.chunk_launches()
        lx = g.eval(chunk, n_past)
        assert g.chunk_launches() == before + 1 and g.xpipe_state() == 1
        kx = [g.read_kv(w, (kw["n_positions"] + n_past) * kw["d_model"], n * kw["d_model"]) for w in (0, 1)]
        _opts(g, monkeypatch, BIOGPT_HIP_XCOLS="0")
        lc = g.eval(chunk, n_past)
        kc = [g.read_kv(w, (kw["n_positions"] + n_past) * kw["d_model"], n * kw["d_model"]) for w in (0, 1)]
        lo = o.eval(chunk, n_past)
        assert (lx == lc).all() and (kx[0] == kc[0]).all() and (kx[1] == kc[1]).all(), n_past
        assert np.abs(lx - lo).max() <= ATOL and int(lx.argmax()) == int(lo.argmax())
        n_past += n
    assert n_past == 100
    g.close()
