import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
for rows in (1 << 17, 1 << 19, 1 << 20):
    for steps in (4, 8, 16, 32):
        s, b = g.bench_stream(rows, 10, steps)
        print("rows %8d steps %2d: %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)" % (rows, steps, s * 1e6, b / s / 1e9, b / s / 8e12 * 100), flush=True)
