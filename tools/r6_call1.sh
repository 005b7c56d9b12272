set -x
OUT=$PWD/gpurun_out/r6a; mkdir -p $OUT
timeout 120 tools/microbench24 > $OUT/microbench24.txt 2>&1; cat $OUT/microbench24.txt
timeout 300 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prefill_base.json 2> /dev/null; tail -c 1500 $OUT/bench_prefill_base.json
timeout 1500 bash tools/pmc_prefill.sh $OUT > $OUT/pmc_prefill.log 2>&1
tail -5 $OUT/pmc_prefill.log
