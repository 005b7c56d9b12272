import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
for n_past in (103, 255):
    print("n_past", n_past, "us/token", round(g.bench_decode(n_past, 20) * 1e6, 1), flush=True)
