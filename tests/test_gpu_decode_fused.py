"""The single-token decode step of csrc/kernels_decode.hip.h (five short-chain launches per layer, embedding and sampler
folded into the first kernel of the next step) against (a) the oracle and (b) the first chain of kernels_fast.hip.h it
replaces (BIOGPT_HIP_NO_FUSED_DECODE=1), at BioGPT-base widths; the full 24-layer BioGPT-base configuration
(biogpt.h:25-35); device-side top-k; the C-ABI replicas; and bench.py's RCCL replica path on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATOL = 1e-3
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
QUANT = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"]


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("fused")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, **KW)
    out = {"f32": f32}
    for name in QUANT:
        out[name] = str(d / (name + ".bin"))
        pkg.quantize_file(f32, out[name], name)
    return out


def _unfused(pkg, path, monkeypatch):
    monkeypatch.setenv("BIOGPT_HIP_NO_FUSED_DECODE", "1")     # options are read when the context is created
    g = pkg.BiogptModel.load(path)
    monkeypatch.delenv("BIOGPT_HIP_NO_FUSED_DECODE")
    return g


@pytest.mark.parametrize("name", QUANT)
def test_fused_step_equals_unfused_chain_and_oracle(pkg, oracle, files, monkeypatch, name):
    """Single-token evals at positions around every context bucket of the fused kernels (64 / 128 / 192 / 256 keys) and
    just past them (257 keys: both contexts run the unfused chain): logits bit-identical between the fused step and the
    chain it replaces, and within the contract of the oracle (bit-identical in practice)."""
    gf = pkg.BiogptModel.load(files[name])
    gu = _unfused(pkg, files[name], monkeypatch)
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(17)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 259)]
    checked = {0, 1, 2, 62, 63, 64, 65, 127, 128, 191, 192, 200, 254, 255, 256, 257}
    n_past, worst, exact = 0, 0.0, 0
    # the prompt part in reference chunks, the checked positions one token at a time
    while n_past < len(toks):
        if n_past in checked:
            lf, lu, lo = gf.eval([toks[n_past]], n_past), gu.eval([toks[n_past]], n_past), o.eval([toks[n_past]], n_past)
            assert (lf == lu).all(), "%s: fused != unfused at n_past %d (max diff %g)" % (name, n_past, np.abs(lf - lu).max())
            worst = max(worst, float(np.abs(lf - lo).max()))
            exact += int((lf == lo).all())
            assert int(lf.argmax()) == int(lo.argmax())
            n_past += 1
        else:
            m = 1
            while n_past + m < len(toks) and (n_past + m) not in checked and m < 8:
                m += 1
            chunk = toks[n_past:n_past + m]
            gf.eval_device(chunk, n_past); gu.eval_device(chunk, n_past); o.eval(chunk, n_past)
            n_past += m
    print("%s: fused step worst |diff| vs oracle %.2e, %d/%d positions bit-identical" % (name, worst, exact, len(checked)))
    assert worst <= ATOL
    # the KV rows the fused kernel appended are the oracle's
    K = o.kv(0)
    for l in (0, KW["n_layer"] - 1):
        for pos in (0, 64, 255):
            got = gf.read_kv(0, (l * KW["n_positions"] + pos) * KW["d_model"], KW["d_model"])
            assert np.abs(got - K[l, pos]).max() <= ATOL
    gf.close(); gu.close()


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_generation_across_the_fused_range(pkg, oracle, files, monkeypatch, name):
    """Greedy generation that starts inside the fused range and leaves it (contexts 247 .. 270 keys): the sampler that
    lives in the next step's first kernel, the hand-over to the unfused graphs at 257 keys and the final sampler launch
    must give the oracle's ids -- and the unfused build's."""
    gf = pkg.BiogptModel.load(files[name])
    gu = _unfused(pkg, files[name], monkeypatch)
    rng = np.random.default_rng(23)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 245)]
    ids_f, _ = gf.generate_greedy(prompt, 24, n_batch=8)
    ids_u, _ = gu.generate_greedy(prompt, 24, n_batch=8)
    ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompt, 24, n_batch=8)
    assert list(ids_f) == list(ref) and list(ids_u) == list(ref)
    for n_predict in (1, 2, 9):                      # ends inside the fused range; n_predict = 1 has no decode step at all
        ids, _ = gf.generate_greedy(prompt[:5], n_predict, n_batch=8)
        ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompt[:5], n_predict, n_batch=8)
        assert list(ids) == list(ref), n_predict
    # eager (no graph) loop: the fused kernels with the token taken from the device state
    monkeypatch.setenv("BIOGPT_HIP_NO_GRAPH", "1")
    gf.refresh_options()
    monkeypatch.delenv("BIOGPT_HIP_NO_GRAPH")
    ids_e, _ = gf.generate_greedy(prompt[:7], 12, n_batch=8)
    ref, _ = oracle.OracleModel(files[name], n_threads=16).generate_greedy(prompt[:7], 12, n_batch=8)
    assert list(ids_e) == list(ref)
    gf.close(); gu.close()


@pytest.fixture(scope="module")
def base24_f32(pkg, tmp_path_factory):
    """The full BioGPT-base configuration (24 layers, biogpt.h:25-35), synthetic seeded weights (the bench's own seed)."""
    d = tmp_path_factory.mktemp("base24")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, seed=0x42494F47, **dict(KW, n_layer=24))
    yield f32
    if os.path.exists(f32):
        os.remove(f32)


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_biogpt_base_24_layers(pkg, oracle, base24_f32, tmp_path, name):
    """configs[1] (Q4_0) and configs[3] (Q5_1, Q8_0) at full depth: 32 teacher-forced single-token evals (logits vs the oracle) and the
    bench workload itself, the 200-token greedy continuation of a 4-token prompt (README.md:29 ids), ids == oracle -- on the XCD pipeline
    (48 half-layer units: every XCD takes six), which must still hold the path at the end."""
    path = str(tmp_path / (name + ".bin"))
    pkg.quantize_file(base24_f32, path, name)
    g = pkg.BiogptModel.load(path)
    o = oracle.OracleModel(path, n_threads=16)
    assert g.hparams.n_layer == 24
    had_pipeline = g.xpipe_state() == 1
    prompt = [2, 7548, 1171, 32924]
    lg, lo = g.eval(prompt, 0), o.eval(prompt, 0)
    worst, exact, n_past = float(np.abs(lg - lo).max()), 0, 4
    for _ in range(32):
        t = int(lo.argmax())
        assert int(lg.argmax()) == t
        lg, lo = g.eval([t], n_past), o.eval([t], n_past)
        worst = max(worst, float(np.abs(lg - lo).max()))
        exact += int((lg == lo).all())
        n_past += 1
    print("24 layers %s: worst |diff| %.2e, %d/32 steps bit-identical" % (name, worst, exact))
    assert worst <= ATOL
    ids, secs = g.generate_greedy(prompt, 200, n_batch=8)
    ref, _ = oracle.OracleModel(path, n_threads=16).generate_greedy(prompt, 200, n_batch=8)
    assert list(ids) == list(ref)
    print("24 layers %s: 200 greedy ids identical, %.0f tok/s" % (name, 200 / secs))
    if had_pipeline:
        assert g.xpipe_state() == 1, "the pipeline was abandoned during the run"
    g.close()
    os.remove(path)


def test_biogpt_base_24_layers_beyond_256_keys(pkg, oracle, base24_f32, tmp_path):
    """configs[1] at full depth AND long context (n_ctx = 1024 is in the metric's name): a 300-token prompt in reference chunks (biogpt_hip_eval_prompt),
    40 single-token biogpt_eval calls of a greedy caller -- served by the long-context launch in its resident form (kernels_xlong.hip.h: every XCD takes six
    units and the helper duty of all 24 layers) -- every row against the oracle; then the rest of the context as one chunk and the last 8 positions
    (n_past 1016 .. 1023: the 1024-key launch with its last helper range in use) one by one.  biogpt.cpp:729-764 at 24 layers."""
    path = str(tmp_path / "q4_0.bin")
    pkg.quantize_file(base24_f32, path, "q4_0")
    g = pkg.BiogptModel.load(path)
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    o = oracle.OracleModel(path, n_threads=16)
    rng = np.random.default_rng(31)
    ctx = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 1023)]
    lg = g.eval_prompt(ctx[:300], 0, 8)
    for at in range(0, 300, 8):
        lo = o.eval(ctx[at:min(300, at + 8)], at)
    assert float(np.abs(lg - lo).max()) <= ATOL and int(lg.argmax()) == int(lo.argmax())
    # the caller's loop, back to back (the resident launch leaves after 1 ms without a call: the oracle's rows come afterwards)
    rows, toks, n_past = [], [], 300
    for k in range(40):
        toks.append(int(lg.argmax()))
        lg = g.eval([toks[-1]], n_past + k)
        rows.append(lg)
    st = g.resident_stats()
    worst, exact = 0.0, 0
    for k in range(40):
        assert toks[k] == int(lo.argmax()), "step %d: the caller's arg-max is not the oracle's" % k
        lo = o.eval([toks[k]], n_past + k)
        worst = max(worst, float(np.abs(rows[k] - lo).max()))
        exact += int((rows[k] == lo).all())
    print("24 layers q4_0, 301 .. 340 keys through the resident long-context launch: worst |diff| %.2e, %d/40 rows bit-identical, speculation %s" % (worst, exact, st))
    assert worst <= ATOL and st["hits"] >= 20, st
    # K / V rows appended by the resident launch (layer 0 and the last layer) are the oracle's
    K = o.kv(0)
    for l in (0, 23):
        for pos in (299, 300, 339):
            got = g.read_kv(0, (l * KW["n_positions"] + pos) * KW["d_model"], KW["d_model"])
            assert np.abs(got - K[l, pos]).max() <= ATOL
    # fill the context up to 1016 keys (one chunk: no mask inside an eval, F1 -- the same on both sides), then the last 8 positions one token at a time
    n_past = 340
    g.eval_device(ctx[n_past:1016], n_past); o.eval(ctx[n_past:1016], n_past)
    worst2, exact2 = 0.0, 0
    for n_past in range(1016, 1024):
        lg, lo = g.eval([ctx[n_past]], n_past), o.eval([ctx[n_past]], n_past)
        worst2 = max(worst2, float(np.abs(lg - lo).max()))
        exact2 += int((lg == lo).all())
        assert int(lg.argmax()) == int(lo.argmax())
    print("24 layers q4_0, 1017 .. 1024 keys: worst |diff| %.2e, %d/8 rows bit-identical" % (worst2, exact2))
    assert worst2 <= ATOL
    assert g.xpipe_state() == 1, "the pipeline was abandoned during the run"
    g.close()
    os.remove(path)


def test_biogpt_base_24_layers_512_token_prompt_in_one_pass(pkg, oracle, base24_f32, tmp_path):
    """configs[2] at full depth, the shape bench.py's `prompt_pass` times: a 512-token prompt, -b 8, through biogpt_hip_eval_prompt as ONE pass of 512 columns (the
    matrix-core chain with the two-tile fc1 walk + attn_tile_kernel with its LDS ring) against the oracle fed the 64 chunks the reference's loop would issue
    (main.cpp:129-137): the returned row within 1e-3 (reported: exact), arg-max, and the K / V rows of layers 0 / 11 / 23.  biogpt.cpp:624-847."""
    path = str(tmp_path / "q4_0.bin")
    pkg.quantize_file(base24_f32, path, "q4_0")
    g = pkg.BiogptModel.load(path)
    o = oracle.OracleModel(path, n_threads=16)
    rng = np.random.default_rng(512)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 511)]
    lo = None
    for at in range(0, 512, 8):
        lo = o.eval(toks[at:at + 8], at)
    lg = g.eval_prompt(toks, 0, 8)
    d = float(np.abs(lg - lo).max())
    print("24 layers q4_0, 512-token prompt in one pass: worst |diff| %.2e%s" % (d, " (bit-identical)" if (lg == lo).all() else ""))
    assert d <= ATOL and int(lg.argmax()) == int(lo.argmax())
    P, D = KW["n_positions"], KW["d_model"]
    for which in (0, 1):
        ref = o.kv(which)
        for l in (0, 11, 23):
            got = g.read_kv(which, (l * P) * D, 512 * D).reshape(512, D)
            assert np.abs(got - ref[l, :512]).max() <= 1e-4, (which, l)
    # and the token after the prompt through the drop-in call: the first decode step on top of the pass's cache
    t = int(lo.argmax())
    lg2, lo2 = g.eval([t], 512), o.eval([t], 512)
    assert float(np.abs(lg2 - lo2).max()) <= ATOL and int(lg2.argmax()) == int(lo2.argmax())
    g.close()
    os.remove(path)


@pytest.mark.parametrize("name", ["q4_1", "q5_0", "f16"])
def test_biogpt_base_24_layers_other_formats(pkg, oracle, base24_f32, tmp_path, name):
    """The formats test_biogpt_base_24_layers leaves out, once at full depth: an 8-token chunk and 32 teacher-forced single-token evals against the oracle."""
    path = str(tmp_path / (name + ".bin"))
    if name == "f16":
        from modelfile_py import read_model, write_model      # convert.py --use-f16: the 2-D "*.weight" tensors as float16, ftype 1
        hp, vocab, merges, tensors = read_model(base24_f32)
        for t in tensors:
            if len(t["ne"]) == 2 and t["name"].endswith(".weight") and t["type"] == 0:
                t["raw"] = np.frombuffer(t["raw"], dtype=np.float32).astype(np.float16).tobytes()
                t["type"] = 1
        write_model(path, dict(hp, ftype=1), vocab, merges, tensors)
        del tensors
    else:
        pkg.quantize_file(base24_f32, path, name)
    g = pkg.BiogptModel.load(path)
    o = oracle.OracleModel(path, n_threads=16)
    assert g.hparams.n_layer == 24
    prompt = [2, 7548, 1171, 32924, 11, 4057, 29999, 5]
    lg, lo = g.eval(prompt, 0), o.eval(prompt, 0)
    worst, exact, n_past = float(np.abs(lg - lo).max()), 0, 8
    for _ in range(32):
        t = int(lo.argmax())
        assert int(lg.argmax()) == t
        lg, lo = g.eval([t], n_past), o.eval([t], n_past)
        worst = max(worst, float(np.abs(lg - lo).max()))
        exact += int((lg == lo).all())
        n_past += 1
    print("24 layers %s: worst |diff| %.2e, %d/32 steps bit-identical" % (name, worst, exact))
    assert worst <= ATOL
    g.close()
    os.remove(path)


def _last_json_line(text):
    lines = [ln for ln in text.strip().splitlines() if ln.strip()]
    assert lines and lines[-1].lstrip().startswith("{"), "the JSON line must be the last line on stdout:\n" + text[-600:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("launcher", ["env", "torchrun", "self"])
def test_bench_rccl_replica_path_on_one_gpu(oracle, tmp_path, launcher):
    """configs[4] plumbing on the GPU that is here: bench.py with the real `nccl` (= RCCL) backend -- broadcast_hparams ->
    biogpt_hip_load_into a torch-owned arena -> broadcast_arena -> timed region -> max_over_ranks -- replaces the single
    load at examples/main/main.cpp:38.  The JSON line must be the last stdout line and the ids of the last continuation
    must be the oracle's."""
    dump = str(tmp_path / "ids.json")
    env = dict(os.environ, BIOGPT_BENCH_DUMP_IDS=dump, BIOGPT_BENCH_DIR=str(tmp_path / "work"))   # bench.py sets HSA_ENABLE_IPC_MODE_LEGACY=0 itself (replicas.py says why)
    args = ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--n-layer", "4", "--n-predict", "40"]
    if launcher == "env":
        env.update(BIOGPT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        cmd = [sys.executable] + args
    elif launcher == "self":
        # `python bench.py --gpus N` with no launcher in the environment re-executes itself under torch.distributed.run (N > 1 always; N = 1 with the switch)
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        env.update(BIOGPT_BENCH_SELF_LAUNCH="1")
        cmd = [sys.executable] + args
    else:
        env.update(BIOGPT_BENCH_FORCE_DIST="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29532"] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert "replicas x1" in out["config"]["parallelism"]
    assert "broadcast" in r.stderr                      # the arena went through the collective
    rep = out["replicas"]                               # what makes an N > 1 record self-explanatory: ranks the communicator saw, device and rate per rank, the broadcast
    assert rep["ranks_seen"] == 1 and rep["backend"] == "nccl" and len(rep["devices"]) == 1 and rep["devices"][0]
    assert len(rep["per_rank_tokens_per_s"]) == 1 and rep["per_rank_tokens_per_s"][0] >= out["value"] * 0.99
    assert rep["broadcast_ms"] > 0 and rep["arena_bytes"] > 0
    if launcher == "self":
        assert "starting 1 rank(s)" in r.stderr and "torch.distributed.run" in r.stderr
        assert "rank 0 / 1 bound cuda:0" in r.stderr and "communicator size 1" in r.stderr and "backend nccl" in r.stderr
    rec = json.load(open(dump + ".rank0"))
    ref, _ = oracle.OracleModel(rec["model"], n_threads=16).generate_greedy(rec["prompt"], 40, n_batch=8)
    assert rec["ids"] == [int(v) for v in ref]


def test_bench_replicas_capi_on_one_gpu(tmp_path):
    """`bench.py --replicas capi`: the SINGLE-process form of SURVEY 8(e) (biogpt_hip_replicas_*: ncclCommInitAll, one broadcast, a host thread per device) behind the
    same JSON line, so that both forms can be compared on one lease.  One device here: the communicator has one rank."""
    env = dict(os.environ, BIOGPT_BENCH_DIR=str(tmp_path / "work"))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--replicas", "capi", "--steps", "2", "--warmup", "1", "--n-layer", "4", "--n-predict", "40"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and "ONE process" in out["config"]["parallelism"]
    assert out["replicas"]["ranks_seen"] == 1 and out["replicas"]["broadcast_ms"] >= 0 and out["replicas"]["arena_bytes"] > 0


def test_c_abi_replicas_single_process(pkg, oracle, files):
    """biogpt_hip_replicas_*: the C++-side multi-GPU entry (single process, ncclCommInitAll, one arena broadcast,
    one host thread per device).  On this one-GPU box the communicator has one rank -- the RCCL calls, the per-replica
    contexts, the prompt sharding g mod n and the shared vocabulary all run; ids == oracle for every prompt."""
    r = pkg.Replicas(files["q4_0"], [0], verbosity=1)
    assert r.count == 1 and r.broadcast_seconds >= 0.0
    rng = np.random.default_rng(77)
    prompts = [[2] + [int(v) for v in rng.integers(4, KW["n_vocab"], int(rng.integers(1, 9)))] for _ in range(3)]
    ids, secs = r.generate_greedy(prompts, 12, n_batch=8)
    assert secs > 0 and len(ids) == 3
    for g, p in enumerate(prompts):
        ref, _ = oracle.OracleModel(files["q4_0"], n_threads=16).generate_greedy(p, 12, n_batch=8)
        assert list(ids[g]) == list(ref), g
    assert r.vocab_of(0) is not None
    with pytest.raises(pkg.BiogptError):
        pkg.Replicas(files["q4_0"], [0, 0])           # a device listed twice
    with pytest.raises(pkg.BiogptError):
        pkg.Replicas(files["q4_0"], [0, 63])          # no such device
    r.close()


@pytest.mark.parametrize("k", [1, 7, 40, 64])
def test_device_top_k_equals_sorted_logits(pkg, files, k):
    """biogpt_hip_eval_topk: the k largest logits of the row (radix select on the device) == numpy on the full row that
    biogpt_hip_eval returns for the same token; a prompt chunk and single tokens (the graph-replayed eval path)."""
    g = pkg.BiogptModel.load(files["q5_0"])
    rng = np.random.default_rng(k)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 12)]
    n_past = 0
    for chunk in (toks[:8], toks[8:9], toks[9:10], toks[10:13]):
        full = g.eval(chunk, n_past)
        vals, ids = g.eval_topk(chunk, n_past, k)        # same tokens, same position: same cache rows, same logits
        order = np.lexsort((np.arange(full.size), -full))[:k]      # logit descending, ties: lower id first
        assert list(ids) == [int(i) for i in order]
        assert (vals == full[order]).all()
        n_past += len(chunk)
    with pytest.raises(pkg.BiogptError):
        g.eval_topk([2], 0, 65)
    g.close()


@pytest.mark.parametrize("name", ["q4_0", "q8_0"])
def test_opt_in_causal_mask(pkg, oracle, files, monkeypatch, name):
    """BIOGPT_HIP_CAUSAL=1 (the opt-in fix of the reference's missing intra-chunk mask, F1): prompt chunks, a pass of several
    chunks and single tokens against the oracle in causal mode."""
    monkeypatch.setenv("BIOGPT_HIP_CAUSAL", "1")
    g = pkg.BiogptModel.load(files[name])
    monkeypatch.delenv("BIOGPT_HIP_CAUSAL")
    o = oracle.OracleModel(files[name], n_threads=16)
    o.set_mode("ggml", n_threads=16, causal=1)
    rng = np.random.default_rng(41)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 30)]
    worst, n_past = 0.0, 0
    for n in (8, 5, 1, 1, 16):
        lg, lo = g.eval(toks[n_past:n_past + n], n_past), o.eval(toks[n_past:n_past + n], n_past)
        worst = max(worst, float(np.abs(lg - lo).max()))
        assert int(lg.argmax()) == int(lo.argmax())
        n_past += n
    # and it is NOT the default behaviour: the same chunk without the mask differs
    g2 = pkg.BiogptModel.load(files[name])
    assert np.abs(g2.eval(toks[:8], 0) - o.eval(toks[:8], 0)).max() > 1e-4
    print("%s causal: worst |diff| %.2e" % (name, worst))
    assert worst <= ATOL
    g.close(); g2.close()


# ---- XCD-pipelined decode step (csrc/kernels_xpipe.hip.h): ONE persistent launch for all layers -----------------------------

XPIPE_TYPES = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"]


def _with_xpipe(g, monkeypatch, on):
    monkeypatch.setenv("BIOGPT_HIP_XPIPE", "1" if on else "0")
    g.refresh_options()
    monkeypatch.delenv("BIOGPT_HIP_XPIPE")


@pytest.mark.parametrize("name", XPIPE_TYPES)
def test_xpipe_step_is_bit_identical_to_the_five_launch_layer_and_the_oracle(pkg, oracle, files, monkeypatch, name):
    """The same context, the same positions, with the pipeline on and off (options re-read in between): logits identical bit for
    bit around every context bucket of the pipeline (64 / 128 / 256 keys: 8 / 4 / 2 lanes per key) and within the contract of
    the oracle; the K / V rows the pipeline appended are the oracle's."""
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device (xpipe_state %d)" % g.xpipe_state())
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(29)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 256)]
    checked = [0, 1, 31, 62, 63, 64, 65, 100, 127, 128, 129, 200, 254, 255]
    n_past, worst = 0, 0.0
    while n_past <= checked[-1]:
        if n_past in checked:
            _with_xpipe(g, monkeypatch, True)
            assert g.xpipe_state() == 1
            lp = g.eval([toks[n_past]], n_past)
            kp = g.read_kv(0, ((KW["n_layer"] - 1) * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"])
            _with_xpipe(g, monkeypatch, False)
            assert g.xpipe_state() == 0
            lf = g.eval([toks[n_past]], n_past)
            kf = g.read_kv(0, ((KW["n_layer"] - 1) * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"])
            lo = o.eval([toks[n_past]], n_past)
            assert (lp == lf).all(), "%s: pipeline != five-launch layer at n_past %d (max diff %g)" % (name, n_past, np.abs(lp - lf).max())
            assert (kp == kf).all()
            worst = max(worst, float(np.abs(lp - lo).max()))
            assert int(lp.argmax()) == int(lo.argmax())
            n_past += 1
        else:
            m = 1
            while (n_past + m) not in checked and m < 8:
                m += 1
            chunk = toks[n_past:n_past + m]
            g.eval_device(chunk, n_past); g.synchronize(); o.eval(chunk, n_past)
            n_past += m
    assert worst <= ATOL
    K = o.kv(0)
    for pos in (0, 64, 255):
        got = g.read_kv(0, ((KW["n_layer"] - 1) * KW["n_positions"] + pos) * KW["d_model"], KW["d_model"])
        assert np.abs(got - K[KW["n_layer"] - 1, pos]).max() <= ATOL
    assert g.xpipe_state() == 0          # switched off above, not abandoned
    g.close()


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_xpipe_generation_equals_the_oracle(pkg, oracle, files, monkeypatch, name):
    """Greedy generation (device-resident sampler folded into the pipeline's first stage) through all three pipeline buckets
    and across the hand-over to the five-launch graphs at 257 keys: the oracle's ids, and the ids with the pipeline off."""
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    rng = np.random.default_rng(31)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 40)]
    n_predict = 230                          # contexts 41 .. 270 keys
    ids_p, _ = g.generate_greedy(prompt, n_predict=n_predict, n_batch=8)
    assert g.xpipe_state() == 1
    _with_xpipe(g, monkeypatch, False)
    ids_f, _ = g.generate_greedy(prompt, n_predict=n_predict, n_batch=8)
    assert list(ids_p) == list(ids_f)
    o = oracle.OracleModel(files[name], n_threads=16)
    ids_o, _ = o.generate_greedy(prompt, n_predict=40, n_batch=8)
    assert list(ids_p[:40]) == list(ids_o)
    g.close()


def test_xpipe_disturbed_launch_is_repeated_on_the_five_launch_layer(pkg, files, monkeypatch, capfd):
    """BIOGPT_HIP_XPIPE_FAULT=1 pre-loads one arrival ticket of XCD 0, so the first pipelined launch finds a 33rd workgroup
    there -- what a launch interleaved with another stream's workgroups looks like.  The launch must drain (bounded spins),
    the call must be repeated transparently and give the undisturbed results, and the context must stay off the pipeline."""
    ref = pkg.BiogptModel.load(files["q4_0"])
    if ref.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompt = [2, 100, 200, 300]
    want, _ = ref.generate_greedy(prompt, n_predict=24, n_batch=8)
    want_logits = ref.eval([7], 30)
    ref.close()
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    assert g.xpipe_state() == 1
    got, _ = g.generate_greedy(prompt, n_predict=24, n_batch=8)
    assert list(got) == list(want)
    assert g.xpipe_state() == -1
    assert "five-launch layer" in capfd.readouterr().err
    assert (g.eval([7], 30) == want_logits).all()
    g.close()
    # and through the single-token API
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    # (round 4: the 4-token prompt is itself a pipelined launch now -- the column-per-XCD chunk launch of kernels_xcols.hip.h -- so the fault hits the
    # asynchronous prompt eval: the synchronising call says from where to repeat, the repetition runs on the launch chain)
    g.eval_device(prompt, 0)
    with pytest.raises(pkg.BiogptError, match="from position 0 on must be repeated"):
        g.synchronize()
    assert g.xpipe_state() == -1
    g.eval_device(prompt, 0); g.synchronize()
    l0 = g.eval([7], 4)
    assert g.xpipe_state() == -1
    g.close()
    # ... and with the chunk launch off the fault hits the single-token eval, which is repeated transparently
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    monkeypatch.setenv("BIOGPT_HIP_XCOLS", "0")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    monkeypatch.delenv("BIOGPT_HIP_XCOLS")
    g.eval_device(prompt, 0); g.synchronize()
    assert (g.eval([7], 4) == l0).all()
    assert g.xpipe_state() == -1
    g.close()
    ref = pkg.BiogptModel.load(files["q4_0"])
    ref.eval_device(prompt, 0); ref.synchronize()
    assert (ref.eval([7], 4) == l0).all()
    ref.close()


def test_xpipe_survives_a_second_stream_generating_at_the_same_time(pkg, files):
    """Two contexts of one device generating concurrently from two host threads: whichever takes the device's pipeline slot first runs
    its call pipelined, the other runs the five-launch layer on its own stream for that call, so workgroups of both are dispatched
    interleaved.  Whatever happens to the pipelined launches (undisturbed, or drained and repeated on the five-launch layer) every call
    must return the undisturbed ids."""
    import threading
    a = pkg.BiogptModel.load(files["q4_0"])
    if a.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    b = pkg.BiogptModel.load(files["q4_0"])
    prompt_a, prompt_b = [2, 100, 200, 300], [2, 7, 8, 9, 10]
    want_a, _ = a.generate_greedy(prompt_a, n_predict=120, n_batch=8)
    want_b, _ = b.generate_greedy(prompt_b, n_predict=120, n_batch=8)
    assert a.xpipe_state() == 1 and b.xpipe_state() == 1          # the slot is handed back whenever its holder's stream has been synchronised
    errs = []

    def run(g, prompt, want, reps):
        try:
            for _ in range(reps):
                got, _ = g.generate_greedy(prompt, n_predict=120, n_batch=8)
                if list(got) != list(want):
                    errs.append("ids differ")
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    ta = threading.Thread(target=run, args=(a, prompt_a, want_a, 12))
    tb = threading.Thread(target=run, args=(b, prompt_b, want_b, 12))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    assert a.xpipe_state() in (1, -1)
    a.close(); b.close()


def test_xpipe_many_launches_stay_clean(pkg, files):
    """4000 tokens through the pipeline (multi-token launches and single-token evals mixed): no timeout, no fallback, same ids."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompt = [2, 11, 12, 13]
    want, _ = g.generate_greedy(prompt, n_predict=200, n_batch=8)
    for rep in range(19):
        got, _ = g.generate_greedy(prompt, n_predict=200, n_batch=8)
        assert list(got) == list(want)
        if rep % 5 == 0:
            lg = g.eval([int(want[0])], len(prompt))
            assert int(lg.argmax()) == int(want[1])
    assert g.xpipe_state() == 1
    g.close()


# ---- pipeline wrap-around: 10 layers = 20 half-layer units, every XCD takes 2-3 units (the "load the next unit while waiting" and
#      lm_head-ahead paths), all five block formats; a vocabulary that is not a multiple of 64 (partial last lm_head block) ----------

KW10 = dict(KW, n_layer=10, n_vocab=20011, n_merges=1000)


@pytest.fixture(scope="module")
def files10(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("fused10")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, seed=10, **KW10)
    out = {}
    for name in QUANT:
        out[name] = str(d / (name + ".bin"))
        pkg.quantize_file(f32, out[name], name)
    os.remove(f32)
    return out


@pytest.mark.parametrize("name", XPIPE_TYPES)
def test_xpipe_wraparound_step_and_generation(pkg, oracle, files10, monkeypatch, name):
    """10 layers: single-token steps with the pipeline on and off (bit-identical, oracle within the contract) around the buckets, then a
    greedy generation through every pipeline bucket in multi-token launches == pipeline off == oracle (first 48 ids); the pipeline must
    still be on at the end of every phase."""
    g = pkg.BiogptModel.load(files10[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device (xpipe_state %d)" % g.xpipe_state())
    o = oracle.OracleModel(files10[name], n_threads=16)
    rng = np.random.default_rng(101)
    toks = [2] + [int(v) for v in rng.integers(4, KW10["n_vocab"], 256)]
    checked = [0, 1, 63, 64, 100, 127, 128, 191, 192, 255]
    n_past, worst = 0, 0.0
    while n_past <= checked[-1]:
        if n_past in checked:
            _with_xpipe(g, monkeypatch, True)
            lp = g.eval([toks[n_past]], n_past)
            assert g.xpipe_state() == 1, "pipeline abandoned at n_past %d" % n_past
            _with_xpipe(g, monkeypatch, False)
            lf = g.eval([toks[n_past]], n_past)
            lo = o.eval([toks[n_past]], n_past)
            assert (lp == lf).all(), "%s: pipeline != five-launch layer at n_past %d (max diff %g)" % (name, n_past, np.abs(lp - lf).max())
            worst = max(worst, float(np.abs(lp - lo).max()))
            assert int(lp.argmax()) == int(lo.argmax())
            n_past += 1
        else:
            m = 1
            while (n_past + m) not in checked and m < 8:
                m += 1
            chunk = toks[n_past:n_past + m]
            g.eval_device(chunk, n_past); g.synchronize(); o.eval(chunk, n_past)
            n_past += m
    assert worst <= ATOL
    _with_xpipe(g, monkeypatch, True)
    prompt = toks[:9]
    ids_p, _ = g.generate_greedy(prompt, n_predict=262, n_batch=8)          # contexts 10 .. 271 keys
    assert g.xpipe_state() == 1
    _with_xpipe(g, monkeypatch, False)
    ids_f, _ = g.generate_greedy(prompt, n_predict=262, n_batch=8)
    assert list(ids_p) == list(ids_f)
    ids_o, _ = oracle.OracleModel(files10[name], n_threads=16).generate_greedy(prompt, n_predict=48, n_batch=8)
    assert list(ids_p[:48]) == list(ids_o)
    g.close()


def test_pipeline_slot_is_handed_back_and_replicas_reload(pkg, oracle, files):
    """One pipeline slot per device: a second context of the same device gets it as soon as the first one's call has synchronised (both
    generate pipelined, alternately, same ids as alone); biogpt_hip_replicas_load -> free -> load again gives the slot, the RCCL
    handle and the arenas back (ids == oracle both times)."""
    a = pkg.BiogptModel.load(files["q4_0"])
    if a.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    b = pkg.BiogptModel.load(files["q5_0"])
    pa, pb = [2, 100, 200, 300], [2, 7, 8, 9, 10]
    want_a, _ = a.generate_greedy(pa, n_predict=40, n_batch=8)
    want_b, _ = b.generate_greedy(pb, n_predict=40, n_batch=8)
    for _ in range(3):
        got_a, sa = a.generate_greedy(pa, n_predict=40, n_batch=8)
        assert a.xpipe_state() == 1 and b.xpipe_state() == 1
        got_b, sb = b.generate_greedy(pb, n_predict=40, n_batch=8)
        assert list(got_a) == list(want_a) and list(got_b) == list(want_b)
    # an un-synchronised pipelined eval of `a` keeps the slot: `b` runs that call on the five-launch layer, same logits
    lb_ref = b.eval([11], 5)               # (leaves b's resident launch on the device: b holds the slot until that launch has left)
    b.synchronize()
    a.eval_device([12], 4)
    assert b.xpipe_state() in (0, 1)
    assert (b.eval([11], 5) == lb_ref).all()
    a.synchronize(); b.synchronize()
    assert a.xpipe_state() == 1 and b.xpipe_state() == 1
    # a holder that is outside every call with nothing in flight (its resident launch left after its idle time) is relieved by the next context that asks
    la = a.eval([12], 4)
    import time
    time.sleep(0.01)
    ids_b2, _ = b.generate_greedy(pb, n_predict=40, n_batch=8)
    assert list(ids_b2) == list(want_b) and b.xpipe_state() == 1
    assert (a.eval([12], 4) == la).all()
    a.close(); b.close()
    for _ in range(2):
        r = pkg.Replicas(files["q4_0"], [0])
        ids, _ = r.generate_greedy([pa], 12, n_batch=8)
        ref, _ = oracle.OracleModel(files["q4_0"], n_threads=16).generate_greedy(pa, 12, n_batch=8)
        assert list(ids[0]) == list(ref)
        r.close()


def test_generation_replays_the_graph_it_captured_when_the_slot_frees_up_mid_call(pkg, oracle, files, monkeypatch):
    """generate_greedy with one token per launch (BIOGPT_HIP_XPIPE_MULTI=0) while ANOTHER context's asynchronous pipelined work is still in flight at the start:
    the call finds the slot taken, captures the five-launch graphs -- and must replay exactly those even when the other context goes idle mid-call and the slot
    could be taken over (round 3 asked again at every step and could name a graph that was never captured: rc -2).  ids == oracle, several times over."""
    a = pkg.BiogptModel.load(files["q4_0"])
    if a.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_MULTI", "0")
    b = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_MULTI")
    pb = [2, 7, 8, 9, 10]
    ref, _ = oracle.OracleModel(files["q4_0"], n_threads=16).generate_greedy(pb, 30, n_batch=8)
    rng = np.random.default_rng(5)
    for rep in range(4):
        for n_past in range(0, 12 + 4 * rep):          # a burst of asynchronous single-token evals: `a` holds the slot with work in flight ...
            a.eval_device([int(rng.integers(4, KW["n_vocab"]))], n_past)
        ids, _ = b.generate_greedy(pb, 30, n_batch=8)      # ... which ends somewhere inside this call
        assert list(ids) == list(ref), rep
        a.synchronize()
    assert b.xpipe_state() in (0, 1)
    a.close(); b.close()


@pytest.mark.parametrize("mode", ["0", "2"])
def test_hand_off_region_placement_does_not_change_results(pkg, files, monkeypatch, mode):
    """BIOGPT_HIP_HOP_PLACE: where the cross-XCD hand-off regions live (calibrated / first candidate / the slower candidate) is a matter of speed only: logits
    of a prompt + steps through the 256 / 257-key border and a generation are identical to the calibrated placement."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    monkeypatch.setenv("BIOGPT_HIP_HOP_PLACE", mode)
    u = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_HOP_PLACE")
    rng = np.random.default_rng(9)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 259)]
    for m in (g, u):
        m.eval_device(toks[:250], 0)
    for n_past in range(250, 260):
        assert (g.eval([toks[n_past]], n_past) == u.eval([toks[n_past]], n_past)).all(), n_past
    ig, _ = g.generate_greedy(toks[:5], 40, n_batch=8)
    iu, _ = u.generate_greedy(toks[:5], 40, n_batch=8)
    assert list(ig) == list(iu) and u.xpipe_state() == 1
    g.close(); u.close()


def test_matvec_sweep_over_the_models_own_matrices(pkg, oracle, base24_f32, tmp_path):
    """biogpt_hip_bench_sweep: the decode mat-vec on every block-quantized matrix of BioGPT-base (Q4_0: 196 MB) in one launch -- the measurement behind north_star's
    ">= 70 % of the HBM roofline" -- and on each shape alone (which = 2: q/k/v + out_proj, the K = 1024 D x D shapes = "the mat-vec at d_model = 1024" to the letter; 3: fc2,
    K = 4096; 4: fc1; 1: lm_head).  Rows are checked twice: against a host recompute inside the library (chk == 0) and, sampled over every matrix, against the ORACLE's scalar
    vec_dot (oracle/biogpt_oracle.c vec_dot_q4_0_q8_0 = biogpt.cpp:705-716,767,803's mat-vecs) fed the file's own row bytes and the launch's Q8 activation blocks: bit-identical."""
    import struct
    from modelfile_py import read_model
    path = str(tmp_path / "q4_0.bin")
    pkg.quantize_file(base24_f32, path, "q4_0")
    g = pkg.BiogptModel.load(path)
    secs, nbytes, chk = g.bench_sweep(reps=10)
    assert chk == 0.0, chk
    assert abs(nbytes - 195.96e6) < 0.5e6, nbytes        # 24 x (3072 + 1024 + 4096 + 1024 rows of 1024 or 4096) + 42384 x 1024 at 18 / 32 bytes + vectors
    assert nbytes / secs > 2.0e12, "the sweep moved %.0f GB/s" % (nbytes / secs / 1e9)      # (75 % of 8 TB/s measured; the bound only catches a broken launch)
    for which, want in ((1, 24.6e6), (2, 57.0e6), (3, 56.7e6), (4, 57.1e6)):
        s1, b1, c1 = g.bench_sweep(reps=10, which=which)
        assert c1 == 0.0 and abs(b1 - want) < 0.6e6, (which, b1, c1)
        assert b1 / s1 > 1.0e12, (which, b1 / s1)
    # ---- sampled rows against the oracle ----
    rows, xq, xd = g.bench_sweep_rows(which=0)
    hp, _, _, tensors = read_model(path)
    T = {t["name"]: t for t in tensors}
    D, F, V, L = KW["d_model"], KW["d_ff"], KW["n_vocab"], 24

    def q8_blocks(k):        # the launch's activation vector of width k as blk_q8_0 bytes (fp16 d + 32 int8)
        o = 0 if k == 1024 else 1024
        od = 0 if k == 1024 else 32
        out = bytearray()
        for b in range(k // 32):
            out += struct.pack("<e", float(xd[od + b])) + xq[o + 32 * b:o + 32 * b + 32].tobytes()
        return bytes(out)

    y = {1024: q8_blocks(1024), 4096: q8_blocks(4096)}
    rng = np.random.default_rng(97)
    off, checked = 0, 0
    for l in range(L + 1):
        mats = ([("biogpt.layers.%d.self_attn.%s_proj.weight" % (l, n), D, D) for n in "qkv"] + [("biogpt.layers.%d.self_attn.out_proj.weight" % l, D, D),
                ("biogpt.layers.%d.fc1.weight" % l, F, D), ("biogpt.layers.%d.fc2.weight" % l, D, F)]) if l < L else [("output_projection.weight", V, D)]
        for name, M, K in mats:
            if l in (0, 11, 23, L):
                raw = T[name]["raw"]
                rb = K // 32 * 18
                for r in [0, M - 1] + [int(v) for v in rng.integers(0, M, 2)]:
                    ref = oracle.vec_dot_q(oracle.TYPE_Q4_0, K, raw[r * rb:(r + 1) * rb], y[K])
                    assert rows[off + r] == np.float32(ref), (name, r, rows[off + r], ref)
                    checked += 1
            off += M
    assert off == rows.size and checked >= 4 * (3 * 6 + 1)
    g.close()


def test_lm_head_bench_with_cold_weights(pkg, files):
    """biogpt_hip_bench_matvec(12): the stand-alone lm_head with its weights taken from a different device copy every launch (not cache-resident) -- the same bytes,
    a sane time, and not faster than the cache-resident form by more than noise."""
    g = pkg.BiogptModel.load(files["q4_0"])
    s_warm, b_warm = g.bench_matvec(4, layer=0, reps=30)
    s_cold, b_cold = g.bench_matvec(12, layer=0, reps=28)
    assert b_warm == b_cold and 1e-6 < s_cold < 1e-3 and s_cold > 0.8 * s_warm
    print("lm_head: cache-resident %.2f us, cold %.2f us per launch" % (s_warm * 1e6, s_cold * 1e6))
    g.close()


# ---- the pipeline beyond 256 keys (csrc/kernels_xlong.hip.h): attention spread over the chip, 32 / 64 keys per helper workgroup ----------

@pytest.mark.parametrize("name", XPIPE_TYPES)
def test_xlong_step_is_bit_identical_to_the_five_launch_layer_and_the_oracle(pkg, oracle, files, monkeypatch, name):
    """Single-token steps at 257 .. 1024 keys with the pipeline on (key-range helpers on every XCD, scores / partial outputs handed over as
    granules) and off (five launches per layer + the three key-split attention launches): logits and the appended K / V rows identical bit
    for bit; the oracle within the contract.  Positions around the 256 / 512 / 1024 bucket borders and around helper-range borders."""
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device (xpipe_state %d)" % g.xpipe_state())
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(43)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 1023)]
    checked = [255, 256, 257, 287, 288, 320, 447, 511, 512, 513, 575, 576, 640, 831, 832, 1000, 1022, 1023]
    n_past, worst = 0, 0.0
    while n_past <= checked[-1]:
        if n_past in checked:
            _with_xpipe(g, monkeypatch, True)
            lp = g.eval([toks[n_past]], n_past)
            assert g.xpipe_state() == 1, "pipeline abandoned at n_past %d" % n_past
            kp = [g.read_kv(w, ((KW["n_layer"] - 1) * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"]) for w in (0, 1)]
            _with_xpipe(g, monkeypatch, False)
            lf = g.eval([toks[n_past]], n_past)
            kf = [g.read_kv(w, ((KW["n_layer"] - 1) * KW["n_positions"] + n_past) * KW["d_model"], KW["d_model"]) for w in (0, 1)]
            lo = o.eval([toks[n_past]], n_past)
            assert (lp == lf).all(), "%s: pipeline != five-launch layer at n_past %d (max diff %g)" % (name, n_past, np.abs(lp - lf).max())
            assert (kp[0] == kf[0]).all() and (kp[1] == kf[1]).all(), n_past
            worst = max(worst, float(np.abs(lp - lo).max()))
            assert int(lp.argmax()) == int(lo.argmax())
            n_past += 1
        else:
            m = 1
            while (n_past + m) not in checked and m < 8:
                m += 1
            chunk = toks[n_past:n_past + m]
            g.eval_device(chunk, n_past); g.synchronize(); o.eval(chunk, n_past)
            n_past += m
    print("%s: long-context pipeline worst |diff| vs oracle %.2e" % (name, worst))
    assert worst <= ATOL
    g.close()


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_xlong_generation_in_multi_token_launches(pkg, oracle, files10, monkeypatch, name):
    """10 layers, greedy generation from 250 to 1024 keys: the <= 256-key launch, then ONE launch per bucket (257 .. 512, 513 .. 1024) whose
    helpers append the K / V rows they will read back themselves for later tokens of the same launch -- ids identical with the pipeline off,
    the first 24 the oracle's, the pipeline still on at the end; then single-token evals on top of that cache."""
    g = pkg.BiogptModel.load(files10[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    rng = np.random.default_rng(47)
    prompt = [2] + [int(v) for v in rng.integers(4, KW10["n_vocab"], 249)]
    n_predict = KW10["n_positions"] - len(prompt)
    ids_p, _ = g.generate_greedy(prompt, n_predict=n_predict, n_batch=8)
    assert g.xpipe_state() == 1
    lp = g.eval([int(ids_p[-2])], KW10["n_positions"] - 1)            # the cache the long launches left, read by one more pipelined step
    _with_xpipe(g, monkeypatch, False)
    ids_f, _ = g.generate_greedy(prompt, n_predict=n_predict, n_batch=8)
    assert len(ids_p) == n_predict and list(ids_p) == list(ids_f)
    lf = g.eval([int(ids_f[-2])], KW10["n_positions"] - 1)
    assert (lp == lf).all()
    ids_o, _ = oracle.OracleModel(files10[name], n_threads=16).generate_greedy(prompt, n_predict=24, n_batch=8)
    assert list(ids_p[:24]) == list(ids_o)
    g.close()


@pytest.mark.parametrize("name", XPIPE_TYPES)
def test_two_workgroups_per_head_step_257_to_512_keys(pkg, oracle, files, monkeypatch, name):
    """257 .. 512 keys outside the resident launch (single-token graph replays of biogpt_hip_eval_device, generate_greedy's multi-token launches):
    dec_xpipe_kernel<.., KCAP = 512> -- workgroups h and 16 + h of the layer's XCD hold 256 keys' K / V rows each, compute half of the head's q / k / v rows
    each and exchange rows, scores and partial PV sums inside the XCD.  Logits and appended K / V rows bit for bit those of the key-range helpers
    (BIOGPT_HIP_XPIPE_DUAL=0, kernels_xlong.hip.h) and of the five-launch layer; the oracle within the contract.  Positions at both bucket borders, at the
    256-key border between the two workgroups, and where the upper workgroup holds 1 / 2 / 255 / 256 keys."""
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device (xpipe_state %d)" % g.xpipe_state())
    o = oracle.OracleModel(files[name], n_threads=16)
    rng = np.random.default_rng(53)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 511)]
    checked = [255, 256, 257, 258, 300, 383, 384, 447, 509, 510, 511]
    last = (KW["n_layer"] - 1) * KW["n_positions"]

    def step(n_past, env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g.refresh_options()
        for k in env:
            monkeypatch.delenv(k)
        g.eval_device([toks[n_past]], n_past)
        row = g.read_logits()
        kv = [g.read_kv(w, (last + n_past) * KW["d_model"], KW["d_model"]) for w in (0, 1)]
        return row, kv

    n_past, worst = 0, 0.0
    while n_past <= checked[-1]:
        if n_past in checked:
            ld, kd = step(n_past, {"BIOGPT_HIP_XPIPE": "1", "BIOGPT_HIP_XPIPE_DUAL": "1"})
            assert g.xpipe_state() == 1, "pipeline abandoned at n_past %d" % n_past
            lh, kh = step(n_past, {"BIOGPT_HIP_XPIPE": "1", "BIOGPT_HIP_XPIPE_DUAL": "0"})
            lf, kf = step(n_past, {"BIOGPT_HIP_XPIPE": "0"})
            lo = o.eval([toks[n_past]], n_past)
            assert (ld == lh).all() and (ld == lf).all(), "%s: n_past %d: two workgroups per head vs helpers %g, vs five launches %g" % (
                name, n_past, np.abs(ld - lh).max(), np.abs(ld - lf).max())
            for w in (0, 1):
                assert (kd[w] == kh[w]).all() and (kd[w] == kf[w]).all(), (n_past, w)
            worst = max(worst, float(np.abs(ld - lo).max()))
            assert int(ld.argmax()) == int(lo.argmax())
            n_past += 1
        else:
            m = 1
            while (n_past + m) not in checked and m < 8:
                m += 1
            chunk = toks[n_past:n_past + m]
            g.eval_device(chunk, n_past); g.synchronize(); o.eval(chunk, n_past)
            n_past += m
    print("%s: 257 .. 512 keys, two workgroups per head: worst |diff| vs oracle %.2e" % (name, worst))
    assert worst <= ATOL
    g.close()


def test_tripped_pipeline_with_evals_in_flight_says_where_to_resume(pkg, files, monkeypatch):
    """A pipelined launch that is disturbed while EARLIER asynchronous single-token evals are still in flight has spoiled their K / V rows too: the synchronising
    call must not silently repeat only itself -- it fails and names the position to resume from; resuming there gives the undisturbed logits."""
    ref = pkg.BiogptModel.load(files["q4_0"])
    if ref.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompt = [2, 100, 200, 300]
    ref.eval_device(prompt, 0); ref.eval_device([7], 4); want = ref.eval([8], 5)
    ref.close()
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    monkeypatch.setenv("BIOGPT_HIP_XCOLS", "0")          # the prompt chunk on the launch chain: the first pipelined launch is the single-token eval
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    monkeypatch.delenv("BIOGPT_HIP_XCOLS")
    g.eval_device(prompt, 0)
    g.eval_device([7], 4)                     # pipelined, not synchronised: drains with garbage rows
    with pytest.raises(pkg.BiogptError, match="n_past = 4"):
        g.eval([8], 5)
    assert g.xpipe_state() == -1
    g.eval_device([7], 4)                     # resume where told, now on the five-launch layer
    assert (g.eval([8], 5) == want).all()
    g.close()
    # round 4: the prompt chunk itself is a pipelined launch (kernels_xcols.hip.h) -- then it is the disturbed one, and everything from position 0 on must be repeated
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_FAULT", "1")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_FAULT")
    g.eval_device(prompt, 0)
    g.eval_device([7], 4)
    with pytest.raises(pkg.BiogptError, match="n_past = 0"):
        g.eval([8], 5)
    assert g.xpipe_state() == -1 and g.chunk_launches() == 1
    g.eval_device(prompt, 0); g.eval_device([7], 4)
    assert (g.eval([8], 5) == want).all()
    g.close()


def test_xlong_with_a_position_table_that_is_no_multiple_of_the_key_ranges(pkg, oracle, tmp_path, monkeypatch):
    """n_positions = 600 (biogpt.h:25-35 allows any): the 513 .. 600-key bucket runs the 64-key-range variant with its last ranges cut off at the table's end,
    the 257 .. 512 bucket the 32-key one; 2 layers (every XCD beyond the fourth owns no unit and is a pure helper + lm_head workgroup), vocabulary 5000."""
    kw = dict(KW, n_layer=2, n_positions=600, n_vocab=5000, n_merges=100)
    f32, q = str(tmp_path / "f32.bin"), str(tmp_path / "q5_0.bin")
    pkg.write_synthetic(f32, seed=600, **kw)
    pkg.quantize_file(f32, q, "q5_0")
    g = pkg.BiogptModel.load(q)
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    o = oracle.OracleModel(q, n_threads=16)
    rng = np.random.default_rng(6)
    toks = [2] + [int(v) for v in rng.integers(4, kw["n_vocab"], 599)]
    checked = [256, 300, 511, 512, 575, 576, 598, 599]
    n_past = 0
    mp = monkeypatch
    if True:
        while n_past <= checked[-1]:
            if n_past in checked:
                _with_xpipe(g, mp, True)
                lp = g.eval([toks[n_past]], n_past)
                assert g.xpipe_state() == 1
                _with_xpipe(g, mp, False)
                lf = g.eval([toks[n_past]], n_past)
                lo = o.eval([toks[n_past]], n_past)
                assert (lp == lf).all(), n_past
                assert np.abs(lp - lo).max() <= ATOL and int(lp.argmax()) == int(lo.argmax())
                n_past += 1
            else:
                m = 1
                while (n_past + m) not in checked and m < 8:
                    m += 1
                g.eval_device(toks[n_past:n_past + m], n_past); g.synchronize(); o.eval(toks[n_past:n_past + m], n_past)
                n_past += m
        _with_xpipe(g, mp, True)
        ids_p, _ = g.generate_greedy(toks[:250], 350, n_batch=8)
        assert g.xpipe_state() == 1
        _with_xpipe(g, mp, False)
        ids_f, _ = g.generate_greedy(toks[:250], 350, n_batch=8)
        assert len(ids_p) == 350 and list(ids_p) == list(ids_f)
    g.close()


@pytest.mark.parametrize("name", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_lm_head_stream_kernel_equals_the_block_kernel(pkg, files, monkeypatch, name):
    """kernels_lmhead.hip.h (three 64-row blocks per workgroup, all loads up front) against matvec_fast_kernel<PRO_LN, EPI_LOGITS> (BIOGPT_HIP_LM_STREAM=0), where the
    lm_head is a launch of its own: the five-launch decode layer (BIOGPT_HIP_XPIPE=0) -- logits rows, the top-5 selection (it reads the per-block partials) and greedy
    ids (the sampler reads them too) -- and the last row of a prompt chunk."""
    if name not in files:
        pytest.skip("format not in the fixture")
    monkeypatch.setenv("BIOGPT_HIP_XPIPE", "0")
    a = pkg.BiogptModel.load(files[name])
    monkeypatch.setenv("BIOGPT_HIP_LM_STREAM", "0")
    b = pkg.BiogptModel.load(files[name])
    monkeypatch.delenv("BIOGPT_HIP_LM_STREAM"); monkeypatch.delenv("BIOGPT_HIP_XPIPE")
    rng = np.random.default_rng(4)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 40)]
    assert (a.eval(toks[:8], 0) == b.eval(toks[:8], 0)).all()
    for n_past in range(8, 20):
        assert (a.eval([toks[n_past]], n_past) == b.eval([toks[n_past]], n_past)).all(), n_past
    va, ia = a.eval_topk(toks[20:23], 20, 5); vb, ib = b.eval_topk(toks[20:23], 20, 5)
    assert list(ia) == list(ib) and (va == vb).all()
    ga, _ = a.generate_greedy(toks[:6], 40, n_batch=8); gb, _ = b.generate_greedy(toks[:6], 40, n_batch=8)
    assert list(ga) == list(gb)
    a.close(); b.close()
