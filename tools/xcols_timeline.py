"""Stage stamps of the column-per-XCD chunk launch (kernels_xcols.hip.h, XC_WALL: workgroups 0 and 16 of column 0, 100 MHz wall clock): where a layer's time goes.
    BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=<build with -DBIOGPT_HIP_PROFILE_HOOKS> python tools/xcols_timeline.py MODEL [n_past] [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
n_past = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rng = np.random.default_rng(1)
toks = [2] + [int(v) for v in rng.integers(4, 8000, n_past + n)]
for at in range(0, n_past, 8):
    g.eval_device(toks[at:min(at + 8, n_past)], at)
for rep in range(3):
    g.eval(toks[n_past:n_past + n], n_past)
L = g.hparams.n_layer
w = g.debug_stamps(0, L * 16).astype(np.int64).reshape(L, 16)
names = {0: "x in", 6: "LN+Q8 (A)", 1: "q/k/v rows out", 7: "q + new k/v rows in", 13: "scores+max", 14: "exp+sum", 15: "PV", 2: "att out", 8: "att seen", 3: "out_proj out", 9: "x1 in", 10: "LN+Q8 (D)", 11: "fc1+GELU", 4: "h out", 12: "h seen", 5: "fc2 out"}
order = [0, 6, 1, 7, 13, 14, 15, 2, 8, 3, 9, 10, 11, 4, 12, 5]
print("chunk of %d at n_past %d, %d layers: layer time (x in -> next x in), ticks of 10 ns" % (n, n_past, L))
per = [(w[l + 1, 0] - w[l, 0]) for l in range(L - 1)]
print("  per layer us:", " ".join("%.1f" % (p / 100.0) for p in per), " mean %.2f" % (np.mean(per) / 100.0))
for l in (1, L // 2, L - 2):
    if l < 1 or l >= L - 1: continue
    prev = w[l, 0]
    parts = []
    for k in order[1:]:
        parts.append("%s %.2f" % (names[k], (w[l, k] - prev) / 100.0)); prev = w[l, k]
    print("  layer %d:" % l, " | ".join(parts), "| next x in %.2f" % ((w[l + 1, 0] - prev) / 100.0))
print("  whole pass (layer 0 x in -> last fc2 out): %.1f us" % ((w[L - 1, 5] - w[0, 0]) / 100.0))
