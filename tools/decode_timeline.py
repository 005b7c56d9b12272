"""Per-kernel / per-segment timing of one decode step on the GPU box (profiling aid, not part of the product).
usage: decode_timeline.py MODEL [n_past ...]     (BIOGPT_HIP_DBG=32 + BIOGPT_HIP_LIB=<profile-hook build> prints segment stamps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
for n_past in [int(a) for a in sys.argv[2:]] or [103]:
    print("n_past", n_past, "us/token", round(g.bench_decode(n_past, 30) * 1e6, 1), flush=True)
