# soak: every block-quantized type, 60 x 200-token continuations + single-token evals in between; ids must stay identical, no fallback
for FT in q4_0 q4_1 q5_0 q5_1 q8_0; do
M=/tmp/biogpt_amd_bench/synthetic-L24-$FT.bin
[ -f $M ] || python bench.py --ftype $FT --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 python - <<PY
import sys; sys.path.insert(0, '.')
import numpy as np, _pkg
m=_pkg.load(); g=m.BiogptModel.load("$M")
pr=[2,100,200,300]
want,_=g.generate_greedy(pr,200)
bad=0
for rep in range(60):
    got,_=g.generate_greedy(pr,200)
    bad+=int(list(got)!=list(want))
    if rep%10==0:
        lg=g.eval([int(want[0])], len(pr)); bad+=int(int(lg.argmax())!=int(want[1]))
print("$FT", "mismatches", bad, "xpipe_state", g.xpipe_state(), flush=True)
PY
done
