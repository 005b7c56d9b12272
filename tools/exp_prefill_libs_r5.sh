# Round 5: per-kernel times of a 512-token prompt pass for several builds of the library (BIOGPT_HIP_LIB).  usage (GPU box): bash tools/exp_prefill_libs_r5.sh name ...
R=$PWD; OUT=$PWD/gpurun_out/prefill_libs; mkdir -p $OUT
python bench.py --workload prefill --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1      # writes the model files
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  lib=$R/biogpt.cpp_amd/libbiogpt_hip_$n.so; [ "$n" = base ] && lib=$R/biogpt.cpp_amd/libbiogpt_hip.so
  rm -rf /tmp/prof_$n
  BIOGPT_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o pre -- python $R/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_$n.json 2> /tmp/prof_$n.err
  find /tmp/prof_$n -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$n.csv \;
  echo "== $n: $(python -c "import json;d=json.loads(open('$OUT/bench_$n.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])") ms per pass under the profiler"
  grep "matmul_mfma\|attn_tile\|lnq" $OUT/stats_$n.csv | awk -F, '{printf "   %-60s %8.2f us\n", substr($1,1,60), $4/1000}'
done
