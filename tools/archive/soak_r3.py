"""Round-3 soak (GPU box): for SECONDS (default 120) alternate (a) a 250 -> 1024-key greedy continuation (the long-context pipelined launches, K / V rows appended and re-read
inside multi-token launches), (b) the C++ api loop through the resident launch (200 tokens, ids), (c) 64 single-token biogpt_hip_eval calls whose full logits rows are
compared bit for bit with the rows of the first round (a row that reached the host before its completion word would show here), (d) the api loop beyond 256 keys
(a 300-token prompt, 300 tokens through the resident 512- and 1024-key launches, running one position ahead of the caller), (e) 48 single-token evals at 400+ keys that
leave the arg-max every 7th call (the pass the launch started in vain must leave nothing behind: rows compared with the first round's).  Any difference, fallback or error fails."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
q = os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-%s.bin" % (sys.argv[2] if len(sys.argv) > 2 else "q4_0"))
g = pkg.BiogptModel.load(q, verbosity=0)
assert g.xpipe_state() == 1, "pipeline not available"
rng = np.random.default_rng(99)
long_prompt = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 249)]
pr = [2, 100, 200, 300]
want_long = want_api = want_rows = want_api_long = want_dev = want_chunks = None
api_prompt = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 299)]
dev_prompt = [2] + [int(v) for v in rng.integers(4, g.hparams.n_vocab, 399)]
rounds = toks = 0
slowest = 0.0
t_end = time.time() + secs
while time.time() < t_end:
    t0 = time.time()
    ids, _ = g.generate_greedy(long_prompt, 1024 - 250, n_batch=8)
    if want_long is None: want_long = ids.copy()
    assert (ids == want_long).all(), "long-context ids differ in round %d" % rounds
    api, _ = g.bench_api_loop(pr, 200, 0)
    if want_api is None: want_api = api.copy()
    assert (api == want_api).all(), "api-loop ids differ in round %d" % rounds
    rows = []
    lg = g.eval(pr, 0); n_past = 4
    for k in range(64):
        t = int(lg.argmax())
        lg = g.eval([t], n_past); n_past += 1
        rows.append(lg.copy())
        if rounds % 2 and k % 16 == 7: time.sleep(0.0015)      # let the resident launch leave now and then
    rows = np.stack(rows)
    if want_rows is None: want_rows = rows
    assert (rows == want_rows).all(), "a logits row differs in round %d" % rounds
    apil, _ = g.bench_api_loop(api_prompt, 300, 0)
    if want_api_long is None: want_api_long = apil.copy()
    assert (apil == want_api_long).all(), "long api-loop ids differ in round %d" % rounds
    rows = []
    lg = g.eval(dev_prompt, 0); n_past = 400
    for k in range(48):
        t = int(lg.argmax()) if k % 7 != 6 else int(np.argsort(lg)[-2])
        lg = g.eval([t], n_past); n_past += 1
        rows.append(lg.copy())
    rows = np.stack(rows)
    if want_dev is None: want_dev = rows
    assert (rows == want_dev).all(), "a long-context logits row differs in round %d" % rounds
    # (f, round 4) the reference's prompt loop: a 200-token prompt in evals of 8 / 5 / 3 tokens (column-per-XCD chunk launches, kernels_xcols.hip.h), every row compared
    rows = []
    n_past = 0
    for k in range(36):
        n = (8, 5, 3)[k % 3] if k % 4 == 3 else 8
        if n_past + n > 200: break
        rows.append(g.eval(api_prompt[n_past:n_past + n], n_past).copy()); n_past += n
    rows = np.stack(rows)
    if want_chunks is None: want_chunks = rows
    assert (rows == want_chunks).all(), "a chunk eval's logits row differs in round %d" % rounds
    assert g.xpipe_state() == 1, "the pipeline was abandoned in round %d" % rounds
    rounds += 1; toks += len(ids) + len(api) + 64 + len(apil) + 48 + n_past
    slowest = max(slowest, time.time() - t0)
print("speculation:", g.resident_stats(), " chunk launches:", g.chunk_launches())
print("format", os.path.basename(q))
print("soak ok: %d rounds, %d tokens in %.0f s, slowest round %.3f s, pipeline state %d" % (rounds, toks, secs, slowest, g.xpipe_state()))
