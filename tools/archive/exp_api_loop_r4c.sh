# round 4, last session: where the resident launch's 15 us per token go (api loop 258.9 against the device loop's 242.5 us per token)
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
[ -f /tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
(
for r in 1 2; do
echo "== ordinary instantiations"; API_LOOP_MODES=0 timeout 200 python tools/api_loop_modes.py
echo "== BIOGPT_HIP_XPIPE_AS_RES=1 (device loop through the RES instantiations, resident = 0)"; BIOGPT_HIP_XPIPE_AS_RES=1 API_LOOP_MODES=0 timeout 200 python tools/api_loop_modes.py
done
) > $OUT/api_loop_exp2.txt 2>&1
grep -v loading $OUT/api_loop_exp2.txt
