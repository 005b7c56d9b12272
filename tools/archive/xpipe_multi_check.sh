M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 python - <<PY
import os, sys; sys.path.insert(0, '.')
import numpy as np
import _pkg
m=_pkg.load()
prompt=[2, 100, 200, 300]
out={}
for name,env in (("multi",{}),("single",{"BIOGPT_HIP_XPIPE_MULTI":"0"}),("five",{"BIOGPT_HIP_XPIPE":"0"})):
    for k in list(os.environ):
        if k.startswith("BIOGPT_HIP_XPIPE"): os.environ.pop(k)
    os.environ.update(env)
    g=m.BiogptModel.load("$M")
    g.generate_greedy(prompt, n_predict=16, n_batch=8)
    best=0
    for rep in range(4):
        ids,secs=g.generate_greedy(prompt, n_predict=200, n_batch=8); best=max(best,200/secs)
    lg=g.eval([5], 150)
    out[name]=(list(ids), lg)
    print(name, "%.1f tok/s" % best, "state", g.xpipe_state(), flush=True)
    g.close()
print("ids multi==single", out["multi"][0]==out["single"][0], "multi==five", out["multi"][0]==out["five"][0], "logits", float(np.abs(out["multi"][1]-out["five"][1]).max()))
PY
