M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import os, sys; sys.path.insert(0, '.')
import _pkg
m=_pkg.load()
cfgs=[{}, {"BIOGPT_HIP_ATTN_WAVES":"8"}, {"BIOGPT_HIP_QKV_WAVES":"16","BIOGPT_HIP_FC1_WAVES":"16"}]
g=m.BiogptModel.load("$M")
res={i:[] for i in range(len(cfgs))}
for rep in range(4):
    for i,c in enumerate(cfgs):
        for k in ("BIOGPT_HIP_FC2_WAVES","BIOGPT_HIP_OPROJ_WAVES","BIOGPT_HIP_FC1_BLOCKS","BIOGPT_HIP_QKV_WAVES","BIOGPT_HIP_FC1_WAVES","BIOGPT_HIP_ATTN_WAVES"): os.environ.pop(k, None)
        os.environ.update(c); g.refresh_options()
        res[i].append(g.bench_decode(103, 40)*1e6); res.setdefault(('T40',i),[]).append(g.bench_decode(40, 40)*1e6)
for i,c in enumerate(cfgs): print(c, " ".join("%.1f" % v for v in res[i]), "min %.1f" % min(res[i]), "| 41 keys:", " ".join("%.1f" % v for v in res[("T40",i)]))
PY
