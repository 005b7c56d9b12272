set -x
mkdir -p gpurun_out/final
python bench.py --steps 5 --warmup 1 > gpurun_out/final/bench_r1_n1.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/final/rocprofv3_kernel_stats_r1.csv \;
cd $GRAFT_REPO_ROOT
for t in q5_1 q8_0 q4_1 q5_0 f16; do python bench.py --ftype $t --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_r1_$t.json 2>/dev/null; done
python bench.py --workload prefill --no-cpu-baseline > gpurun_out/final/bench_r1_prefill_q4_0.json 2>/dev/null
BIOGPT_BENCH_CHUNK_CALLS=1 python bench.py --workload prefill --no-cpu-baseline > gpurun_out/final/bench_r1_prefill_q4_0_per_eval.json 2>/dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -o pre -- python $GRAFT_REPO_ROOT/bench.py --workload prefill --no-cpu-baseline > /dev/null 2>&1; find /tmp/prof_pre -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/final/rocprofv3_kernel_stats_r1_prefill.csv \;)
BIOGPT_HIP_PREFILL_MFMA=1 python bench.py --workload prefill --no-cpu-baseline > gpurun_out/final/bench_r1_prefill_q4_0_mfma.json 2>/dev/null
ls -la gpurun_out/final
