"""The api loop beyond 256 keys: python tools/api_loop_long.py [n_prompt] [n_predict]  -- device loop vs biogpt_hip_eval per token (host arg-max)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
m = pkg.BiogptModel.load(os.path.join(d, "synthetic-L24-q4_0.bin"), verbosity=0)
n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(11)
pr = [2] + [int(v) for v in rng.integers(4, m.n_vocab, n_prompt - 1)]
ids_d, s = m.generate_greedy(pr, n, n_batch=8); ids_d, s = m.generate_greedy(pr, n, n_batch=8)
print("contexts %d .. %d" % (n_prompt + 1, n_prompt + n))
print("device loop        %7.1f us/token (incl. the prompt pass)" % (s / n * 1e6))
for mode, name in ((3, "inplace + argmax8 "), (0, "eval + max_element")):
    m.bench_api_loop(pr, 8, mode)
    best, ids = None, None
    for _ in range(3):
        i, t = m.bench_api_loop(pr, n, mode)
        if best is None or t < best:
            best, ids = t, i
    print("%s %7.1f us/token   ids == device loop: %s   %s" % (name, best / n * 1e6, bool((np.asarray(ids) == np.asarray(ids_d)).all()), m.resident_stats()))
