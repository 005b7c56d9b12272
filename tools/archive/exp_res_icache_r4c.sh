# round 4, last session: instruction-cache counters of the ordinary and the RES instantiation (64-key variant, multi-token launches)
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT; R=$PWD
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || BIOGPT_BENCH_SKIP_TYPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_quick.json 2>$OUT/bench_quick.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|INST_CACHE|SQ_IFETCH|SQ_WAIT_INST|SQ_INSTS_SALU|SQ_BUSY_CYC" | head -40 > $OUT/icache_counters_avail.txt
(
for c in SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_WAIT_INST_ANY; do
for a in 0 1; do
  rm -rf /tmp/pi_$c$a; BIOGPT_HIP_XPIPE_AS_RES=$a timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pi_$c$a -o p -- python $R/tools/pmc_bucket_target.py > /tmp/pi.log 2>&1
  DB=$(find /tmp/pi_$c$a -name "*.db" | head -1); echo "== $c AS_RES=$a"; python $R/tools/pmc_summary.py $DB 2>&1 | grep -E "dec_xpipe" 
done; done
) > $OUT/res_icache.txt 2>&1
cat $OUT/icache_counters_avail.txt | cut -c1-200 | head -20; cat $OUT/res_icache.txt | cut -c1-260
