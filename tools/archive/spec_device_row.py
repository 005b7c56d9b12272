import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
KW = dict(n_vocab=42384, n_layer=3, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
os.makedirs("/tmp/dbg", exist_ok=True)
f32 = "/tmp/dbg/f32.bin"; q = "/tmp/dbg/q4_0.bin"
pkg.write_synthetic(f32, seed=77, **KW)
pkg.quantize_file(f32, q, "q4_0")
g = pkg.BiogptModel.load(q)
row = g.eval([2, 77, 1234, 9], 0)
n_past = 4
every = int(os.environ.get("EVERY", "1"))
for k in range(40):
    tok = int(row.argmax())
    row = g.eval([tok], n_past); n_past += 1
    if k % every == every - 1:
        d = g.read_logits()
        print(k, g.resident_stats(), "device row == host row:", bool((d == row).all()))
