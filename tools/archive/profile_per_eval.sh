#!/bin/bash
# rocprofv3 kernel stats of the reference's prompt loop through the drop-in API: one biogpt_eval per 8 tokens (bench.py --workload prefill, BIOGPT_BENCH_CHUNK_CALLS=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 python $R/bench.py --steps 1 --no-cpu-baseline > /dev/null 2>&1
rm -rf /tmp/prof_pe
BIOGPT_BENCH_CHUNK_CALLS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pe -o pe -- python $R/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline > $OUT/per_eval_under_rocprof.json 2> /tmp/pe.err < /dev/null
find /tmp/prof_pe -name "*kernel_stats.csv" -exec cp {} $OUT/per_eval_kernel_stats${1}.csv \;
ls /tmp/prof_pe/* | head
