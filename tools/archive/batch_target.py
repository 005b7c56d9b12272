"""Batched decode workload for rocprofv3 kernel traces:
    BIOGPT_HIP_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -- python tools/batch_target.py MODEL N_SEQS [N_PREDICT]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
import numpy as np
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1], verbosity=0)
S, n = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 24
rng = np.random.default_rng(0)
prompts = [[2] + [int(t) for t in rng.integers(4, 42384, 3)] for _ in range(S)]
ids, secs = g.generate_greedy_batch(prompts, n)
print("S=%d: %.1f tok/s, %.3f ms per step" % (S, S * (n - 1) / secs, secs / (n - 1) * 1e3))
