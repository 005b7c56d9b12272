# round 4, last session: which piece of the resident form costs the chain its 8 .. 19 us per token -- RES instantiations compiled with -DXP_RES_AB=n (kernels_xpipe.hip.h),
# run as ORDINARY multi-token launches (BIOGPT_HIP_XPIPE_AS_RES=1), per context bucket (tools/bucket_ab.py).  LIBS = the variant libraries.
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
(
echo "== ordinary instantiations"; BIOGPT_HIP_XPIPE_AS_RES=0 timeout 300 python tools/bucket_ab.py biogpt.cpp_amd/libbiogpt_hip.so | head -1
echo "== RES instantiations as ordinary launches"; BIOGPT_HIP_XPIPE_AS_RES=1 timeout 600 python tools/bucket_ab.py $LIBS
) > $OUT/res_ab${TAG}.txt 2>&1
cat $OUT/res_ab${TAG}.txt
