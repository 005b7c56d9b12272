import sys, time
sys.path.insert(0, ".")
import _pkg
m = _pkg.load()
g = m.BiogptModel.load("/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin", verbosity=0)
pr = [2, 100, 200, 300]
for n in (1, 2, 61):
    g.generate_greedy(pr, n)
    ts = []
    for _ in range(20):
        _, s = g.generate_greedy(pr, n)
        ts.append(s)
    print("n_predict", n, "min %.1f us  median %.1f us" % (min(ts) * 1e6, sorted(ts)[10] * 1e6))
