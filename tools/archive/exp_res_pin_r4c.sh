# round 4, last session: the resident instantiations with their parameter block pinned at entry (XP_PIN_PARAMS), per bucket and through the api loop
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
(
echo "== RES instantiations as ordinary launches (BIOGPT_HIP_XPIPE_AS_RES=1)"; BIOGPT_HIP_XPIPE_AS_RES=1 timeout 600 python tools/bucket_ab.py $LIBS
for l in $LIBS; do echo "== api loop, $l"; BIOGPT_HIP_LIB=$PWD/$l API_LOOP_MODES=0,4,1 timeout 200 python tools/api_loop_modes.py; done
) 2>&1 | grep -v loading > $OUT/res_pin${TAG}.txt
cat $OUT/res_pin${TAG}.txt
