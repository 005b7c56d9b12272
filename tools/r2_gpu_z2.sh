M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so BIOGPT_HIP_DBG=128 timeout 300 python - <<PY
import os, sys; sys.path.insert(0, '.')
import _pkg
m=_pkg.load()
g=m.BiogptModel.load("$M")
for n in (103,):
    print(n, "%.1f us/token" % (g.bench_decode(n, 40)*1e6), flush=True)
PY
