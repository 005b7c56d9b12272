# round 4, last session: the RES instantiations against the ordinary ones for the other block formats (per bucket, multi-token launches)
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
for t in q8_0 q5_1; do BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --ftype $t --steps 1 --warmup 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1; done
ls /tmp/biogpt_amd_bench/ > $OUT/res_types.txt
(for t in q8_0 q5_1; do for a in 0 1; do echo "== $t BIOGPT_HIP_XPIPE_AS_RES=$a"; BUCKET_AB_FTYPE=$t BIOGPT_HIP_XPIPE_AS_RES=$a timeout 300 python tools/bucket_ab.py biogpt.cpp_amd/libbiogpt_hip.so; done; done) >> $OUT/res_types.txt 2>&1
cat $OUT/res_types.txt
