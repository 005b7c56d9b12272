"""Resident single-token evals (csrc/engine.hip resident_eval, kernels_xpipe.hip.h resident mode): biogpt_hip_eval with one token keeps the
pipelined launch on the device and feeds the next call's token through a pinned mailbox (replaces the per-call launch of biogpt_eval,
biogpt.cpp:812-847, in the caller's loop main.cpp:91-151).  Everything here compares with the same calls with BIOGPT_HIP_RESIDENT=0 and with
the oracle."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("resident")
    f32 = str(d / "f32.bin")
    pkg.write_synthetic(f32, seed=77, **KW)
    out = {}
    for name in ("q4_0", "q5_1", "q8_0"):
        out[name] = str(d / (name + ".bin"))
        pkg.quantize_file(f32, out[name], name)
    os.remove(f32)
    return out


def _plain(pkg, path, monkeypatch):
    monkeypatch.setenv("BIOGPT_HIP_RESIDENT", "0")
    g = pkg.BiogptModel.load(path)
    monkeypatch.delenv("BIOGPT_HIP_RESIDENT")
    return g


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_resident_eval_loop_equals_per_call_launches_and_the_oracle(pkg, oracle, files, monkeypatch, name):
    """The reference's loop: a prompt chunk, then one biogpt_eval per token with greedy sampling on the host, 300 tokens (through the 64 / 128 /
    192 / 256-key launches and past 256 keys, where the per-call path takes over): every logits row identical to the per-call launches', ids and
    the first 40 rows against the oracle."""
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    u = _plain(pkg, files[name], monkeypatch)
    o = oracle.OracleModel(files[name], n_threads=16)
    prompt = [2, 900, 17, 4211, 8]
    lg, lu, lo = g.eval(prompt, 0), u.eval(prompt, 0), o.eval(prompt, 0)
    assert (lg == lu).all()
    n_past = len(prompt)
    for k in range(300):
        tok = int(lg.argmax())
        assert tok == int(lu.argmax())
        lg, lu = g.eval([tok], n_past), u.eval([tok], n_past)
        assert (lg == lu).all(), "%s: resident row != per-call row at n_past %d (max diff %g)" % (name, n_past, np.abs(lg - lu).max())
        if k < 40:
            lo = o.eval([tok], n_past)
            assert np.abs(lg - lo).max() <= 1e-3 and int(lg.argmax()) == int(lo.argmax())
        n_past += 1
    assert g.xpipe_state() == 1
    g.close(); u.close()


def test_resident_launch_yields_to_every_other_call(pkg, oracle, files, monkeypatch):
    """Between single-token evals: K / V read-back, a device-side eval, a prompt chunk, re-evaluation of an earlier position, an idle pause longer
    than the launch waits (BIOGPT_HIP_RESIDENT_US), generate_greedy and a second context's work -- each must end the resident launch cleanly
    and every row must equal the per-call path's."""
    monkeypatch.setenv("BIOGPT_HIP_RESIDENT_US", "300")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_RESIDENT_US")
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    u = _plain(pkg, files["q4_0"], monkeypatch)
    rng = np.random.default_rng(5)
    toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 80)]

    def both(chunk, n_past):
        a, b = g.eval(chunk, n_past), u.eval(chunk, n_past)
        assert (a == b).all(), (len(chunk), n_past, float(np.abs(a - b).max()))
        return a

    both(toks[:8], 0)
    n_past = 8
    for step in range(60):
        both([toks[n_past]], n_past)
        n_past += 1
        if step == 5:
            ka = g.read_kv(0, 3 * KW["d_model"], KW["d_model"]); kb = u.read_kv(0, 3 * KW["d_model"], KW["d_model"])
            assert (ka == kb).all()
        if step == 9:
            g.eval_device([toks[n_past]], n_past); u.eval_device([toks[n_past]], n_past); n_past += 1
            g.synchronize(); u.synchronize()
        if step == 14:
            both(toks[n_past:n_past + 3], n_past); n_past += 3
        if step == 20:
            both([toks[10]], 10)                        # back to an earlier position, then on from where we were
        if step == 25:
            time.sleep(0.02)                            # the launch gives up waiting after 300 us
        if step == 30:
            time.sleep(0.0004)                          # right around the time-out: whichever way the race goes, the row must be right
        if step == 35:
            va, ia = g.eval_topk([toks[n_past]], n_past, 5); vb, ib = u.eval_topk([toks[n_past]], n_past, 5)
            assert list(ia) == list(ib) and (va == vb).all()
            n_past += 1
    # a second context working while the first one's resident launch is (or was just) on the device
    v = pkg.BiogptModel.load(files["q5_1"])
    both([toks[n_past]], n_past); n_past += 1
    ids_v, _ = v.generate_greedy([2, 5, 6], 20, n_batch=8)
    both([toks[n_past]], n_past); n_past += 1
    ref, _ = oracle.OracleModel(files["q5_1"], n_threads=16).generate_greedy([2, 5, 6], 20, n_batch=8)
    assert list(ids_v) == list(ref)
    ids_g, _ = g.generate_greedy(toks[:6], 30, n_batch=8)
    ids_u, _ = u.generate_greedy(toks[:6], 30, n_batch=8)
    assert list(ids_g) == list(ids_u)
    assert g.xpipe_state() == 1
    v.close(); g.close(); u.close()


def test_resident_race_with_the_idle_timeout(pkg, files, monkeypatch):
    """Pauses swept across the launch's idle limit (100 us): the next call either finds the launch still waiting or finds it gone -- 400 tokens,
    every row equal to the per-call path's (a stale or half-written row would show)."""
    monkeypatch.setenv("BIOGPT_HIP_RESIDENT_US", "100")
    g = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_RESIDENT_US")
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    u = _plain(pkg, files["q4_0"], monkeypatch)
    prompt = [2, 31, 41, 59]
    la, lb = g.eval(prompt, 0), u.eval(prompt, 0)
    n_past = 4
    for k in range(200):
        tok = int(la.argmax())
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < (k % 40) * 5e-6:      # 0 .. 195 us
            pass
        la, lb = g.eval([tok], n_past), u.eval([tok], n_past)
        assert (la == lb).all(), (k, n_past)
        n_past += 1
    g.close(); u.close()


def test_api_loop_through_the_resident_launch(pkg, oracle, files):
    """biogpt_hip_bench_api_loop (the C++ form of main.cpp's loop: biogpt_hip_eval per token, host arg-max) == the device-resident loop == oracle."""
    g = pkg.BiogptModel.load(files["q8_0"])
    prompt = [2, 11, 12, 13]
    ids, secs = g.bench_api_loop(prompt, 120, 0)
    dev, _ = g.generate_greedy(prompt, 120, n_batch=8)
    assert list(ids) == list(dev)
    ref, _ = oracle.OracleModel(files["q8_0"], n_threads=16).generate_greedy(prompt, 40, n_batch=8)
    assert list(ids[:40]) == list(ref)
    print("api loop: %.0f tok/s" % (120 / secs))
    g.close()


def _run(model, prompt, n, pick, hook=None):
    """prompt, then n single-token evals; pick(k, row, n_past) -> (token, n_past) of call k; returns the rows and the (token, n_past) of every call"""
    rows, calls = [model.eval(prompt, 0)], []
    n_past = len(prompt)
    for k in range(n):
        tok, n_past = pick(k, rows[-1], n_past)
        rows.append(model.eval([tok], n_past))
        calls.append((tok, n_past))
        n_past += 1
        if hook:
            hook(k, rows[-1])
    return rows, calls


def test_speculative_continuation_greedy_caller(pkg, files, monkeypatch):
    """A greedy caller (main.cpp:109-128 with top_k = 1) with the device to itself: after four calls that named the device's own arg-max the launch runs
    one position ahead of the caller.  250 tokens through the 64 / 128 / 192 / 256-key launches: every row equal to the per-call path's, nearly every
    call served by a pass that was already running, and -- the launch is stopped with such a pass in flight -- the device-side row read back afterwards
    is the LAST ASKED token's."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompt = [2, 77, 1234, 9]
    greedy = lambda k, row, n_past: (int(row.argmax()), n_past)

    def hook(k, row):
        if k in (20, 21, 90):      # odd and even sequence numbers: the device row after the launch was asked to leave mid-speculation
            assert (g.read_logits() == row).all(), k

    rows, calls = _run(g, prompt, 250, greedy, hook)
    st = g.resident_stats()
    print("speculation:", st)
    assert g.xpipe_state() == 1
    g.close()
    u = _plain(pkg, files["q4_0"], monkeypatch)
    rows_u, calls_u = _run(u, prompt, 250, greedy)
    assert calls == calls_u
    for k, (a, b) in enumerate(zip(rows, rows_u)):
        assert (a == b).all(), (k, float(np.abs(a - b).max()))
    u.close()
    assert st["misses"] == 0 and st["hits"] >= 200, st
    monkeypatch.setenv("BIOGPT_HIP_SPEC", "0")
    h = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_SPEC")
    rows_h, _ = _run(h, prompt, 12, greedy)
    assert h.resident_stats()["hits"] == 0 and all((a == b).all() for a, b in zip(rows_h, rows))
    h.close()


def test_speculative_continuation_wrong_guesses(pkg, files, monkeypatch):
    """A caller that follows the arg-max for a while and then does not (a sampled token, a step back, a repeated position): the pass that was started
    in vain must leave nothing behind -- rows and K / V rows equal to the per-call path's -- and every miss doubles the run of matching calls it takes
    to speculate again."""
    g = pkg.BiogptModel.load(files["q5_1"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompt = [2, 5, 6, 7, 8]

    def pick(k, row, n_past):
        tok = int(row.argmax())
        if k in (9, 30, 31, 75):
            tok = int(np.argsort(row)[-2])         # second best: not what the launch went ahead with
        if k == 50:
            n_past -= 3                            # step back: the position the running pass assumed is not the one asked for
        return tok, n_past

    need = [g.resident_stats()["need"]]
    rows, calls = _run(g, prompt, 140, pick, lambda k, row: need.append(g.resident_stats()["need"]))
    st = g.resident_stats()
    print("speculation:", st, sorted(set(need)))
    # ... and a long run ahead of the caller earns the trust back: 560 more greedy calls (through the 256-, 512- and 1024-key launches) halve the threshold twice
    row, n_more = rows[-1], calls[-1][1] + 1
    for k in range(560):
        row = g.eval([int(row.argmax())], n_more + k)
    st2 = g.resident_stats()
    print("after 560 greedy calls:", st2)
    assert st2["misses"] == st["misses"] and st2["need"] <= st["need"] // 4 and st2["hits"] >= st["hits"] + 500, (st, st2)
    D = KW["d_model"]
    n_past = calls[-1][1] + 1
    kv = [g.read_kv(which, 0, n_past * D) for which in (0, 1)]      # layer 0, every position written so far
    assert g.xpipe_state() == 1
    g.close()
    u = _plain(pkg, files["q5_1"], monkeypatch)
    rows_u, calls_u = _run(u, prompt, 140, pick)
    assert calls == calls_u
    for k, (a, b) in enumerate(zip(rows, rows_u)):
        assert (a == b).all(), (k, float(np.abs(a - b).max()))
    for which in (0, 1):
        assert (kv[which] == u.read_kv(which, 0, n_past * D)).all()
    u.close()
    assert st["misses"] >= 2 and st["hits"] >= 20 and need[-1] >= 4 * need[0], st


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_resident_launch_beyond_256_keys(pkg, files, monkeypatch, name):
    """The same loop where the attention is spread over the chip (kernels_xlong.hip.h, its resident instantiation): a 290-token prompt chunk, then single-token
    evals through 291 .. 530 keys (the 512-key launch, then the 1024-key one) -- greedy, so that the launch runs ahead of the caller -- with a deviation, a step back
    and an idle pause on the way; rows and K / V rows equal to the per-call launches'."""
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    rng = np.random.default_rng(21)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 289)]

    def pick(k, row, n_past):
        tok = int(row.argmax())
        if k in (40, 41):
            tok = int(np.argsort(row)[-2])
        if k == 100:
            n_past -= 5
        if k == 150:
            time.sleep(0.01)
        return tok, n_past

    rows, calls = _run(g, prompt, 240, pick)
    st = g.resident_stats()
    print("speculation:", st)
    D = KW["d_model"]
    n_past = calls[-1][1] + 1
    assert n_past > 512
    kv = [g.read_kv(which, 0, n_past * D) for which in (0, 1)]
    dev_row = g.read_logits()
    assert (dev_row == rows[-1]).all()
    assert g.xpipe_state() == 1
    g.close()
    u = _plain(pkg, files[name], monkeypatch)
    rows_u, calls_u = _run(u, prompt, 240, pick)
    assert calls == calls_u
    for k, (a, b) in enumerate(zip(rows, rows_u)):
        assert (a == b).all(), (k, calls[k - 1] if k else None, float(np.abs(a - b).max()))
    for which in (0, 1):
        assert (kv[which] == u.read_kv(which, 0, n_past * D)).all()
    u.close()
    assert st["hits"] >= 150 and st["misses"] >= 1, st


@pytest.mark.parametrize("k", [1, 5, 40, 64])
def test_topk_behind_the_resident_launch_uses_the_block_maxima(pkg, files, monkeypatch, k):
    """biogpt_hip_eval_topk in a loop of single-token calls (a caller that samples, biogpt.cpp:908-980 with top_k = 40): the resident launch leaves the maxima of the row's
    64-row blocks behind the pinned row and the host selects from the k blocks that can hold a candidate -- the same values, ids and order as the full scan of the row
    (BIOGPT_HIP_TOPK_BLOCKS=0) and as numpy on the row itself (value descending, equal values: lower id first); through the 256 -> 257-key border (the long-context launch)."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    monkeypatch.setenv("BIOGPT_HIP_TOPK_BLOCKS", "0")
    u = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_TOPK_BLOCKS")
    rng = np.random.default_rng(100 + k)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 245)]
    for m in (g, u):
        m.eval_device(prompt, 0)
    tok = 77
    for n_past in range(246, 266):
        vg, ig = g.eval_topk([tok], n_past, k)
        vu, iu = u.eval_topk([tok], n_past, k)
        assert list(ig) == list(iu) and (vg == vu).all(), n_past
        if n_past % 5 == 0:      # the row itself (the call re-evaluates the same position: same K / V row, same logits)
            full = g.eval([tok], n_past)
            order = np.lexsort((np.arange(full.size), -full))[:k]
            assert list(ig) == [int(i) for i in order] and (vg == full[order]).all()
        tok = int(ig[min(1, k - 1)])      # not the arg-max: the launch must wait for the caller's token, as it does for a sampling caller
    assert g.xpipe_state() == 1
    g.close(); u.close()


@pytest.mark.parametrize("u_env", [None, "BIOGPT_HIP_XPIPE"])
def test_two_contexts_interleave_single_token_evals_on_one_device(pkg, files, monkeypatch, u_env):
    """Two contexts of one process take turns with single-token biogpt_hip_eval calls on ONE device: the first holds the pipeline slot with its resident launch
    (which waits up to 1 ms for its caller's next token), the second runs every call on the five-launch layer behind it.  Round 3 replayed the second context's
    captured graph there and got the row of ITS PREVIOUS CALL back (tools/dbg_two_contexts.py: K / V rows right, logits one call stale); such calls now take eager
    launches.  Both contexts against a third one that evaluates the same tokens alone, without the pipeline."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    if u_env:
        monkeypatch.setenv(u_env, "0")
    u = pkg.BiogptModel.load(files["q4_0"])
    if u_env:
        monkeypatch.delenv(u_env)
    rng = np.random.default_rng(3)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 245)]
    g.eval_device(prompt, 0); u.eval_device(prompt, 0)
    toks, rows_g, rows_u, top_u = [77], [], [], []
    for n_past in range(246, 262):
        a = g.eval([toks[-1]], n_past)
        b = u.eval([toks[-1]], n_past) if n_past % 2 == 0 else None
        if b is None:
            vals, ids = u.eval_topk([toks[-1]], n_past, 5)
            top_u.append((vals, ids))
        rows_g.append(a); rows_u.append(b)
        toks.append(int(a.argmax()))
    g.close(); u.close()
    monkeypatch.setenv("BIOGPT_HIP_RESIDENT", "0"); monkeypatch.setenv("BIOGPT_HIP_XPIPE", "0")
    r = pkg.BiogptModel.load(files["q4_0"])
    monkeypatch.delenv("BIOGPT_HIP_RESIDENT"); monkeypatch.delenv("BIOGPT_HIP_XPIPE")
    r.eval_device(prompt, 0)
    k = 0
    for i, n_past in enumerate(range(246, 262)):
        t = r.eval([toks[i]], n_past)
        assert (rows_g[i] == t).all(), ("first context", n_past)
        if rows_u[i] is not None:
            assert (rows_u[i] == t).all(), ("second context", n_past)
        else:
            vals, ids = top_u[k]; k += 1
            order = np.lexsort((np.arange(t.size), -t))[:5]
            assert list(ids) == [int(v) for v in order] and (vals == t[order]).all(), ("second context, top-k", n_past)
    r.close()


def _load_env(pkg, path, monkeypatch, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    g = pkg.BiogptModel.load(path)
    for k in env:
        monkeypatch.delenv(k)
    return g


def test_a_stale_row_of_a_replayed_eval_is_detected_and_repaired(pkg, oracle, files, monkeypatch):
    """The contract of biogpt.cpp:840-844: the row an eval returns is the row of THIS eval.  A single-token eval off the pipeline is a replayed graph of the five-launch
    layer whose first node pulls {position, token, call number} from a pinned mailbox; the call number travels with the data (last layer's last kernel -> lm_head -> the
    row's copy to the host: kernels.hip.h SEQ_*) and the host compares it.  BIOGPT_HIP_FAULT_STALE=4 makes every fourth replay start from the PREVIOUS mailbox slot -- it
    evaluates the previous call's token again and returns that call's row, the symptom of profiles/two_contexts_r4.txt.  Every row must still be this call's (== eager
    launches of the same kernels, and the oracle), through biogpt_hip_eval, biogpt_hip_eval_topk and biogpt_hip_eval_device + biogpt_hip_read_logits."""
    f = _load_env(pkg, files["q4_0"], monkeypatch, BIOGPT_HIP_XPIPE=0, BIOGPT_HIP_RESIDENT=0, BIOGPT_HIP_FAULT_STALE=4)
    r = _load_env(pkg, files["q4_0"], monkeypatch, BIOGPT_HIP_XPIPE=0, BIOGPT_HIP_RESIDENT=0, BIOGPT_HIP_NO_GRAPH=1)
    o = oracle.OracleModel(files["q4_0"], n_threads=16)
    prompt = [2, 900, 17, 4211, 8, 31, 77]
    f.eval_device(prompt, 0); r.eval_device(prompt, 0); lo = o.eval(prompt, 0)
    tok, n_past = int(lo.argmax()), len(prompt)
    for k in range(18):
        want = r.eval([tok], n_past)
        if k % 3 == 0:
            got = f.eval([tok], n_past)
        elif k % 3 == 1:
            f.eval_device([tok], n_past)
            got = f.read_logits()
        else:
            vals, ids = f.eval_topk([tok], n_past, 5)
            order = np.lexsort((np.arange(want.size), -want))[:5]
            assert list(ids) == [int(v) for v in order] and (vals == want[order]).all(), ("top-k", k)
            got = want
        assert (got == want).all(), "call %d (position %d): the row is not this call's (max diff %g)" % (k, n_past, np.abs(got - want).max())
        lo = o.eval([tok], n_past)
        assert np.abs(want - lo).max() <= 1e-3 and int(want.argmax()) == int(lo.argmax())
        tok = int(want.argmax()); n_past += 1
    st = f.lineage_stats()
    assert st["graph_evals"] == 18 and st["stale_rows"] == 4, st      # replays 4, 8, 12, 16 (an eval, a device row, a top-k, an eval) were sent to the previous slot; each one found and repeated
    assert r.lineage_stats()["graph_evals"] == 0
    f.close(); r.close()


def test_two_contexts_with_the_graph_forced_beside_a_resident_launch(pkg, files, monkeypatch):
    """The arrangement of profiles/two_contexts_r4.txt with the workaround switched OFF (BIOGPT_HIP_GRAPH_CONTENDED=1: the second context replays its captured five-launch
    step although the first one holds the pipeline slot with a resident launch).  Round 3 / 4 got the previous call's row back there, silently; now a row that is not the
    call's is found by its lineage and the call repeated: every row equals a third context's (eager launches, alone on the device).  How many rows had to be repeated on
    this box is printed (0 = the runtime behaved; profiles/stale_row_r5.txt)."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    u = _load_env(pkg, files["q4_0"], monkeypatch, BIOGPT_HIP_GRAPH_CONTENDED=1)
    rng = np.random.default_rng(3)
    prompt = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], 245)]
    g.eval_device(prompt, 0); u.eval_device(prompt, 0)
    toks, rows_g, rows_u = [77], [], []
    for n_past in range(246, 276):
        a = g.eval([toks[-1]], n_past)
        rows_g.append(a); rows_u.append(u.eval([toks[-1]], n_past))
        toks.append(int(a.argmax()))
    st = u.lineage_stats()
    print("second context: %d replayed evals, %d rows repeated" % (st["graph_evals"], st["stale_rows"]))
    assert st["graph_evals"] > 0      # the graph path was taken (the point of the switch)
    g.close(); u.close()
    r = _load_env(pkg, files["q4_0"], monkeypatch, BIOGPT_HIP_RESIDENT=0, BIOGPT_HIP_XPIPE=0, BIOGPT_HIP_NO_GRAPH=1)
    r.eval_device(prompt, 0)
    for i, n_past in enumerate(range(246, 276)):
        t = r.eval([toks[i]], n_past)
        assert (rows_g[i] == t).all(), ("first context", n_past)
        assert (rows_u[i] == t).all(), ("second context", n_past)
    r.close()


def test_generate_beside_another_contexts_resident_launch(pkg, oracle, files, monkeypatch):
    """generate_greedy / generate_greedy_batch of a context that cannot take the pipeline slot replayed captured five-launch steps beside the holder's live
    resident launch -- a stale row there is a wrong arg-max fed forward.  They now take eager steps in that state (plain_graph_begin, decided under the slot's mutex):
    the ids equal the oracle's and a lone context's."""
    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    u = pkg.BiogptModel.load(files["q4_0"])
    prompt = [2, 900, 17, 4211, 8]
    lg = g.eval(prompt, 0)
    n_past = len(prompt)
    ids_u = None
    for k in range(6):      # the first context's resident launch is live (it waits up to 1 ms for its caller) while the second one generates
        lg = g.eval([int(lg.argmax())], n_past); n_past += 1
        if k == 2:
            ids_u, _ = u.generate_greedy(prompt, 40, n_batch=8)
            ids_b, _ = u.generate_greedy_batch([prompt, prompt[:3]], 12, n_batch=8)
    g.close(); u.close()
    lone = _load_env(pkg, files["q4_0"], monkeypatch, BIOGPT_HIP_RESIDENT=0, BIOGPT_HIP_XPIPE=0)
    ids_l, _ = lone.generate_greedy(prompt, 40, n_batch=8)
    lone.close()
    assert [int(t) for t in ids_u] == [int(t) for t in ids_l]
    ref, _ = oracle.OracleModel(files["q4_0"], n_threads=16).generate_greedy(prompt, 12)
    assert [int(t) for t in ids_u[:12]] == [int(t) for t in ref]
    assert [int(t) for t in ids_b[0][:12]] == [int(t) for t in ref]


@pytest.mark.parametrize("name", ["q4_0", "q5_1", "q8_0"])
def test_resident_instantiations_as_ordinary_multi_token_launches(pkg, oracle, files, monkeypatch, name):
    """BIOGPT_HIP_XPIPE_AS_RES=1 (measurement switch, xpipe_tu.hip) sends ORDINARY pipelined launches -- the device-resident greedy loop's multi-token launches --
    through the RES = true instantiations of dec_xpipe_kernel with resident = 0 (profiles/res_instantiation_ab_r4c.txt measures what that form costs the chain).
    Same arithmetic, so: the same 250 ids (64- / 128- / 192- / 256-key variants) as the ordinary instantiations, and the oracle's first 24.  A context of its own for
    each arm: launches are captured into graphs per context."""
    prompt = [2, 900, 17, 4211, 8]
    g = pkg.BiogptModel.load(files[name])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    ids_ord, _ = g.generate_greedy(prompt, 250, n_batch=8)
    g.close()
    monkeypatch.setenv("BIOGPT_HIP_XPIPE_AS_RES", "1")
    r = pkg.BiogptModel.load(files[name])
    ids_res, _ = r.generate_greedy(prompt, 250, n_batch=8)
    assert r.xpipe_state() == 1
    r.close()
    monkeypatch.delenv("BIOGPT_HIP_XPIPE_AS_RES")
    assert [int(t) for t in ids_ord] == [int(t) for t in ids_res]
    o = oracle.OracleModel(files[name], n_threads=16)
    lo = o.eval(prompt, 0)
    n_past = len(prompt)
    for k in range(min(24, len(ids_res))):
        tok = int(lo.argmax())
        assert tok == int(ids_res[k]), "%s: token %d: oracle %d, RES instantiation %d" % (name, k, tok, int(ids_res[k]))
        lo = o.eval([tok], n_past)
        n_past += 1


_SECOND_PROCESS = r"""
import sys, os, json, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import _pkg
pkg = _pkg.load()
m = pkg.BiogptModel.load(sys.argv[2], verbosity=0)
state = m.xpipe_state()
prompt = [2, 900, 17, 4211, 8]
t0 = time.time()
lg = m.eval(prompt, 0)
ids, n_past = [], len(prompt)
for k in range(24):
    tok = int(lg.argmax()); ids.append(tok)
    lg = m.eval([tok], n_past); n_past += 1
print(json.dumps({"state": state, "state_after": m.xpipe_state(), "ids": ids, "row": [float(v) for v in lg[:64]], "seconds": time.time() - t0}))
"""


def test_a_second_process_on_the_device_stays_off_the_pipeline(pkg, files, tmp_path):
    """Two PROCESSES on one device: persistent launches of both would each hold part of the compute units and wait for the rest.  The process
    that prepares a pipelined context first holds an advisory lock on the device's lock file (engine_xpipe.inc, xpipe_process_lock); a second process finds it taken,
    reports xpipe_state -1 and serves the reference's loop (main.cpp:91-151) on the five-launch layer -- same ids, same rows, no hand-off time-outs -- while the first
    keeps its resident launch going.  When the first process's contexts are gone, a new process gets the pipeline."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "second.py"
    script.write_text(_SECOND_PROCESS)

    def second():
        env = dict(os.environ); env.pop("BIOGPT_HIP_PROC_LOCK", None)
        r = subprocess.run([sys.executable, str(script), root, files["q4_0"]], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "hand-off" not in r.stderr and "timed out" not in r.stderr, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    g = pkg.BiogptModel.load(files["q4_0"])
    if g.xpipe_state() != 1:
        pytest.skip("XCD pipeline not available on this device")
    prompt = [2, 900, 17, 4211, 8]
    lg = g.eval(prompt, 0)
    ids, n_past = [], len(prompt)
    for k in range(12):
        tok = int(lg.argmax()); ids.append(tok)
        lg = g.eval([tok], n_past); n_past += 1
    other = second()                      # runs while this process's resident launch may still be on the device
    assert other["state"] == -1 and other["state_after"] == -1
    for k in range(12, 24):
        tok = int(lg.argmax()); ids.append(tok)
        lg = g.eval([tok], n_past); n_past += 1
    assert g.xpipe_state() == 1
    assert other["ids"] == ids
    assert np.array_equal(np.asarray(other["row"], dtype=np.float32), lg[:64])
    assert other["seconds"] < 20.0
    g.close()
    again = second()                      # the lock went with the last pipelined context of this process
    assert again["state"] == 1 and again["ids"] == ids
