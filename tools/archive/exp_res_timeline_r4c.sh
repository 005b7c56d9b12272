# round 4, last session: stage stamps of the ordinary and of the RES instantiation of dec_xpipe_kernel (profiling build; BIOGPT_HIP_XPIPE_AS_RES=1 routes ordinary launches through RES)
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
export BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so
(
for a in 0 1; do
echo "==== BIOGPT_HIP_XPIPE_AS_RES=$a"
BIOGPT_HIP_XPIPE_AS_RES=$a timeout 120 python tools/decode_timeline.py $M 40 103
BIOGPT_HIP_XPIPE_AS_RES=$a timeout 120 python tools/tail_timeline.py $M 40
done
) 2>&1 | grep -v loading > $OUT/res_timeline.txt
cat $OUT/res_timeline.txt
