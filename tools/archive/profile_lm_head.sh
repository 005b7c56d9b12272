#!/bin/bash
# rocprofv3 kernel durations of the stand-alone lm_head launch, both kernels (tools/lm_head_ab.py child)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/bench.py --steps 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1 < /dev/null
for v in 1 0; do
  rm -rf /tmp/prof_lm$v
  BIOGPT_HIP_LM_STREAM=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lm$v -o lm -- python $R/tools/lm_head_ab.py child > /dev/null 2>&1 < /dev/null
  find /tmp/prof_lm$v -name "*kernel_stats.csv" -exec cp {} $OUT/lm_head_kernel_stats_$v.csv \;
done
ls -la $OUT/lm_head_kernel_stats_*.csv
