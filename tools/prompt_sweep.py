"""Time to ingest a prompt of n tokens (-b 8 semantics, biogpt_hip_eval_prompt), BioGPT-base Q4_0:
    python tools/prompt_sweep.py [n ...]      (BIOGPT_HIP_MFMA_MIN_COLS / BIOGPT_HIP_ATTN_GROUP_MIN select the kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
q = os.path.join(os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench"), "synthetic-L24-q4_0.bin")
m = pkg.BiogptModel.load(q, verbosity=0)
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48, 64, 96, 128, 256, 512]:
    toks = [2] + [int(v) for v in rng.integers(4, 42384, n - 1)]
    m.eval_prompt(toks, 0, 8, want_logits=False); m.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.eval_prompt(toks, 0, 8, want_logits=False)
    m.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("n=%4d  %7.3f ms  %9.0f tok/s" % (n, dt * 1e3, n / dt))
