# round 4, last session: two (XP_CROSS_PIPE=1, the product) against three (=2) poll passes in flight on the cross-XCD sweeps; ordinary and RES instantiations, per bucket; api loop
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
(
echo "== ordinary instantiations"; BIOGPT_HIP_XPIPE_AS_RES=0 timeout 600 python tools/bucket_ab.py $LIBS
echo "== RES instantiations as ordinary launches (BIOGPT_HIP_XPIPE_AS_RES=1)"; BIOGPT_HIP_XPIPE_AS_RES=1 timeout 600 python tools/bucket_ab.py $LIBS
for r in 1 2; do for l in $LIBS; do echo "== api loop, $l"; BIOGPT_HIP_LIB=$PWD/$l API_LOOP_MODES=0 timeout 200 python tools/api_loop_modes.py; done; done
) 2>&1 | grep -v loading > $OUT/cross_pipe${TAG}.txt
cat $OUT/cross_pipe${TAG}.txt
