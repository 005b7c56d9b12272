M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
export R=$PWD OUT=$PWD/gpurun_out/final_r2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python $R/tools/pmc_target.py $M > /tmp/pmc_$c.out 2>&1
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r2.txt 2>&1
  grep -E "^11|xpipe|five-launch" /tmp/pmc_$c.out | head -3
done
grep -h "xpipe" $OUT/pmc_FETCH_SIZE_r2.txt $OUT/pmc_WRITE_SIZE_r2.txt
