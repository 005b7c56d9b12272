"""Stage timeline of the float-weight persistent launch (kernels_fpipe.hip.h): BIOGPT_HIP_FPIPE_STAMPS=1, one single-token eval at a given context, then the s_memrealtime
stamps of workgroups 0 (attention), 128, 255 per layer.  usage: python tools/fpipe_timeline.py [f32|f16] [n_past]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BIOGPT_HIP_FPIPE_STAMPS"] = "1"
import _pkg
pkg = _pkg.load()
ft = sys.argv[1] if len(sys.argv) > 1 else "f32"
n_past = int(sys.argv[2]) if len(sys.argv) > 2 else 100
path = "/tmp/fpipe_tl_%s.bin" % ft
KW = dict(n_vocab=42384, n_layer=24, n_head=16, n_positions=1024, d_ff=4096, d_model=1024, n_merges=40000)
if not os.path.exists(path):
    pkg.write_synthetic(path, seed=5, **dict(KW, **({"ftype": 1} if ft == "f16" else {})))
g = pkg.BiogptModel.load(path)
rng = np.random.default_rng(3)
toks = [2] + [int(v) for v in rng.integers(4, KW["n_vocab"], n_past + 8)]
g.eval(toks[:n_past], 0)
for i in range(6):
    g.eval([toks[n_past + i]], n_past + i)
st = g.fpipe_stamps()
assert st is not None, "stamps inactive"
st = st.astype(np.int64)
t0 = st[0, 0, 0]
names = ["A.in", "A.out", "B.in", "C.in", "C.dot", "D.in", "D.out", "E.in", "E.dot", "rq.wo", "rq.qkv", "rq.w1", "rq.w2", "at.kv", "at.exp", "at.pv"]
pn = ["pA", "pB", "pC", "pD", "pE0", "geluD", "lnA", "lnD", "pE1"]
for w, wg in enumerate((0, 128, 255)):
    print("workgroup %d (us from launch start of wg 0)" % wg)
    print("  L  " + " ".join("%7s" % n for n in names) + " | " + " ".join("%7s" % n for n in pn))
    for L in (0, 1, 2, 3, 10, 11, 22, 23):
        r = st[w, L]
        print(" %2d  " % L + " ".join("%7.2f" % ((r[i] - t0) / 100.0) for i in range(16)) + " | " + " ".join("%7.2f" % ((r[16 + i] - t0) / 100.0) if r[16 + i] else "      -" for i in range(9)))
    d = (st[w, 1:24, 0] - st[w, 0:23, 0]) / 100.0
    print("  layer period: mean %.2f us, min %.2f, max %.2f; total %.1f us" % (d.mean(), d.min(), d.max(), (st[w, 23, 8] - st[w, 0, 0]) / 100.0))
    seg = np.zeros(9)
    for L in range(1, 23):
        r = st[w, L]
        nxt = st[w, L + 1, 0]
        pts = list(r[:9]) + [nxt]
        seg += np.diff(np.array(pts)) / 100.0
    print("  mean segment (layers 1..22): " + " ".join("%s %.2f" % (names[i] + ">", seg[i] / 22) for i in range(9)))
