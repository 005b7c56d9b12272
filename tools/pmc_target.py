"""Small eager (no hipGraph) workload for rocprofv3 --pmc passes: the mat-vec kernels back to back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
m = _pkg.load()
g = m.BiogptModel.load(sys.argv[1])
for which in (0, 1, 2, 3, 4):
    s, b = g.bench_matvec(which, 0, 48)
    print(which, round(s * 1e6, 2), "us", b, "bytes", flush=True)
