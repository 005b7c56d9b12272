# A/B of two builds of the library (default vs $ALT): us/token at 41 / 104 / 161 / 201 keys, interleaved; FT = weight type
FT=${FT:-q4_0}; M=/tmp/biogpt_amd_bench/synthetic-L24-$FT.bin
[ -f $M ] || python bench.py --ftype $FT --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2; do
for lib in biogpt.cpp_amd/libbiogpt_hip.so $ALT; do
BIOGPT_HIP_LIB=$PWD/$lib timeout 300 python - <<PY
import os, sys; sys.path.insert(0, '.')
import _pkg
m=_pkg.load()
g=m.BiogptModel.load("$M")
ids=None
best=0
for _ in range(3):
    ids,secs=g.generate_greedy([2,100,200,300], 200); best=max(best,200/secs)
print("$lib", " ".join("%.1f" % (g.bench_decode(n, 60)*1e6) for n in (40,103,160,200)), "| generate 200: %.1f tok/s" % best, "ids", int(sum(int(v)*(i+1) for i,v in enumerate(ids)) % 1000003), flush=True)
PY
done; done
