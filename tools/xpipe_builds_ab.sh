# A/B of two builds of the library (default vs $ALT): us/token at 41 / 104 / 161 / 201 keys, interleaved
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2 3; do
for lib in biogpt.cpp_amd/libbiogpt_hip.so $ALT; do
BIOGPT_HIP_LIB=$PWD/$lib timeout 300 python - <<PY
import os, sys; sys.path.insert(0, '.')
import _pkg
m=_pkg.load()
g=m.BiogptModel.load("$M")
print("$lib", " ".join("%.1f" % (g.bench_decode(n, 60)*1e6) for n in (40,103,160,200)), "| generate 200: %.1f tok/s" % max(200/g.generate_greedy([2,100,200,300], 200)[1] for _ in range(3)), flush=True)
PY
done; done
