"""What a token costs between the last layer's output and the next token's first LayerNorm inside a multi-token pipelined launch (round 4).
usage: BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=<profile-hook build> python tools/tail_timeline.py MODEL [n_predict]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ids, secs = g.generate_greedy([2, 100, 200, 300], n)
ids, secs = g.generate_greedy([2, 100, 200, 300], n)
print("generate: %.1f us per token" % (secs / n * 1e6))
w = g.debug_stamps(4096, 8 * 16).reshape(8, 16).astype(np.int64)
L = g.debug_stamps(0, 24 * 16).reshape(24, 16).astype(np.int64)
names = ["last layer's output published", "seen by lm_head workgroup 0", "final LayerNorm + Q8 done", "rows done (16 units per lane, in-order sums)", "partials published",
         "next token sampled on XCD 0 (partials seen, arg-max)", "its embedding done", "layer 0's LayerNorm + Q8 done"]
# banks hold the last 8 tokens of the launch; token t's stamps 0..4 in bank t % 8, token t + 1's stamps 5..7 in bank (t + 1) % 8
rows = []
for b in range(8):
    nb = (b + 1) % 8
    t = [w[b][0], w[b][1], w[b][2], w[b][3], w[b][4], w[nb][5], w[nb][6], w[nb][7]]
    if all(v > 0 for v in t) and all(0 <= t[i + 1] - t[i] < 100000 for i in range(7)):
        rows.append(np.diff(np.array(t)) * 0.01)
if rows:
    mean = np.mean(rows, axis=0)
    print("tail of a token inside a multi-token launch (us since the previous line, mean over %d tokens):" % len(rows))
    for k in range(1, 8):
        print("   %-52s %6.2f" % (names[k], mean[k - 1]))
    print("   tail = %.2f us" % mean.sum())
per = (L[1:, 5] - L[:-1, 5]) * 0.01
print("layer period (output published -> next layer's output published), last token of the launch: mean %.2f us, min %.2f, max %.2f" % (per.mean(), per.min(), per.max()))
