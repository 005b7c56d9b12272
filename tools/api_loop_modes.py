"""The api loop's cost per token by mode (tools): python tools/api_loop_modes.py [n_predict]   (BIOGPT_HIP_RES_DBG / BIOGPT_HIP_RESIDENT vary the resident launch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
q = os.path.join(d, "synthetic-L24-q4_0.bin")
m = pkg.BiogptModel.load(q, verbosity=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pr = [2, 100, 200, 300]
_, s = m.generate_greedy(pr, n, n_batch=8); _, s = m.generate_greedy(pr, n, n_batch=8)
print("device loop        %7.1f us/token" % (s / n * 1e6))
modes = [int(v) for v in os.environ.get("API_LOOP_MODES", "4,3,0,1").split(",")]
for mode, name in [mn for mn in ((4, "eval_inplace only "), (3, "inplace + argmax8 "), (0, "eval + max_element"), (1, "eval_topk(40)     ")) if mn[0] in modes]:
    m.bench_api_loop(pr, 8, mode)
    best = min(m.bench_api_loop(pr, n, mode)[1] for _ in range(3))
    print("%s %7.1f us/token" % (name, best / n * 1e6))
