# round 4, last session: the RES instantiations against the ordinary ones, per context bucket (multi-token launches; BIOGPT_HIP_XPIPE_AS_RES=1 routes them through RES)
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
LIBS=${LIBS:-biogpt.cpp_amd/libbiogpt_hip.so}
(
for a in 0 1; do echo "== BIOGPT_HIP_XPIPE_AS_RES=$a"; BIOGPT_HIP_XPIPE_AS_RES=$a timeout 300 python tools/bucket_ab.py $LIBS; done
) > $OUT/res_buckets${TAG}.txt 2>&1
cat $OUT/res_buckets${TAG}.txt
