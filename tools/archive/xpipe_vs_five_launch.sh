# XCD-pipelined decode step vs the five-launch layer: ids, logits, time (same process, options refreshed in between)
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 300 python - <<PY
import os, sys; sys.path.insert(0, '.')
import numpy as np
import _pkg
m=_pkg.load()
g=m.BiogptModel.load("$M")
prompt=[2, 100, 200, 300]
out={}
for xp in ("0","1"):
    os.environ["BIOGPT_HIP_XPIPE"]=xp; g.refresh_options()
    ids,secs=g.generate_greedy(prompt, n_predict=200, n_batch=8)
    lg=g.eval([5], n_past=150)
    out[xp]=(list(ids), np.array(lg), secs)
    t=[g.bench_decode(n, 40)*1e6 for n in (40, 103, 200)]
    print("XPIPE", xp, "generate 200: %.1f tok/s |" % (200/secs), "us/token at 41/104/201 keys:", " ".join("%.1f" % v for v in t), flush=True)
print("ids equal:", out["0"][0]==out["1"][0], out["1"][0][:12])
if out["0"][1] is not None: print("logits max abs diff:", float(np.max(np.abs(out["0"][1]-out["1"][1]))))
PY
