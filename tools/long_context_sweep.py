"""Decode time per token vs context length (captured graph replay at fixed n_past), BioGPT-base Q4_0.
    python tools/long_context_sweep.py [n_past ...]   # BIOGPT_HIP_SPLIT_MIN=100000 disables the key-split attention
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg  # noqa: E402

pkg = _pkg.load()
d = os.environ.get("BIOGPT_BENCH_DIR", "/tmp/biogpt_amd_bench")
os.makedirs(d, exist_ok=True)
f32, q = os.path.join(d, "synthetic-L24-f32.bin"), os.path.join(d, "synthetic-L24-q4_0.bin")
if not os.path.exists(q):
    pkg.write_synthetic(f32)
    pkg.quantize_file(f32, q, "q4_0")
    os.remove(f32)
m = pkg.BiogptModel.load(q, verbosity=0)
m.generate_greedy([2, 5, 6, 7], 8)
hp = m.hparams if hasattr(m, "hparams") else None
print("split_min =", os.environ.get("BIOGPT_HIP_SPLIT_MIN", "default"))
points = [int(a) for a in sys.argv[1:]] or [63, 103, 255, 256, 300, 383, 511, 512, 700, 1023]
for n_past in points:
    t = m.bench_decode(n_past, reps=200)
    print("n_past %4d: %7.1f us/token  %7.1f tok/s" % (n_past, t * 1e6, 1.0 / t))
