"""Do two independent kernel chains overlap on one MI355X?  Two contexts (own stream, own scratch, own KV) ingest a
512-token prompt in chunks of 8 each; enqueue interleaved from one host thread, compare with one context alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import numpy as np
pkg = _pkg.load()
q = os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-q4_0.bin")
a = pkg.BiogptModel.load(q, verbosity=0)
b = pkg.BiogptModel.load(q, verbosity=0)
rng = np.random.default_rng(1)
toks = [int(t) for t in rng.integers(4, 42384, 512)]
chunks = [toks[i:i + 8] for i in range(0, 512, 8)]

def run(models):
    for m in models: m.synchronize()
    t0 = time.perf_counter()
    for i, c in enumerate(chunks):
        for m in models:
            m.eval_device(c, 8 * i)
    for m in models: m.synchronize()
    return time.perf_counter() - t0

for _ in range(2):
    t1 = run([a]); t2 = run([a, b])
    print("one chain: %.2f ms (%.0f tok/s)   two chains interleaved: %.2f ms (%.0f tok/s aggregate, x%.2f)" %
          (t1 * 1e3, 512 / t1, t2 * 1e3, 1024 / t2, 2 * t1 / t2))
