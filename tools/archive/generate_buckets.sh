M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 python - <<PY
import os, sys, time; sys.path.insert(0, '.')
import _pkg
m=_pkg.load()
g=m.BiogptModel.load("$M")
prompt=[2, 100, 200, 300]
for n in (61, 125, 189, 200):
    g.generate_greedy(prompt, n_predict=n, n_batch=8)
    best=1e9; bi=1e9
    for rep in range(5):
        t0=time.perf_counter(); ids,secs=g.generate_greedy(prompt, n_predict=n, n_batch=8); t1=time.perf_counter()
        best=min(best,t1-t0); bi=min(bi,secs)
    print("n_predict %3d: wall %.3f ms, inside the library %.3f ms" % (n, best*1e3, bi*1e3), flush=True)
PY
