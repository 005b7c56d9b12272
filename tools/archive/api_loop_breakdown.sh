M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import sys; sys.path.insert(0, '.')
import _pkg, numpy as np
m=_pkg.load(); g=m.BiogptModel.load("$M")
pr=[2,100,200,300]
for md in (0,1,2): g.bench_api_loop(pr, 8, md)
for rep in range(2):
    r={md: g.bench_api_loop(pr,200,md)[1] for md in (0,1,2)}
    d,sd=g.generate_greedy(pr,200)
    print("per token us: full-row %.1f  topk %.1f  eval+sync only %.1f  device loop %.1f" % (r[0]/200*1e6, r[1]/200*1e6, r[2]/200*1e6, sd/200*1e6))
PY
