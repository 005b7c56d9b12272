# same-process A/B of pipeline options: us/token at 41 / 104 / 201 keys
FT=${FT:-q4_0}; M=/tmp/biogpt_amd_bench/synthetic-L24-$FT.bin
[ -f $M ] || python bench.py --ftype $FT --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 python - <<PY
import os, sys; sys.path.insert(0, '.')
import _pkg
m=_pkg.load()
import json
cfgs=json.loads(os.environ.get("XP_CFGS", '[{}]'))
gs=[]
res={}
for rep in range(3):
    for i,c in enumerate(cfgs):
        for k in list(os.environ):
            if k.startswith("BIOGPT_HIP_XPIPE"): os.environ.pop(k)
        os.environ.update(c)
        g=m.BiogptModel.load("$M")     # table slices / LDS size are fixed at load time
        res.setdefault(i,[]).append([g.bench_decode(n, 60)*1e6 for n in (40,103,160,200)])
        g.close()
for i,c in enumerate(cfgs):
    print(c, " | ".join(" ".join("%.1f" % v for v in r) for r in res[i]))
PY
