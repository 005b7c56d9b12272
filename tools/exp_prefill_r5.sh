# Round 5, configs[2]: the software-pipelined matmul_mfma_kernel.  Parity first (the prompt-pass tests), then the pass time and the per-kernel times.
# usage (GPU box): bash tools/exp_prefill_r5.sh <tag>     -> gpurun_out/prefill_r5_<tag>/
set -x
TAG=${1:-a}
OUT=$PWD/gpurun_out/prefill_r5_$TAG; mkdir -p $OUT; export R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "prompt or prefill or mfma or chunk or batch" > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt
timeout 300 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prefill.json 2> $OUT/bench_prefill.err; cat $OUT/bench_prefill.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o pre -- python $R/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.json 2> /tmp/prof_p.err
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r5_prefill.csv \;
head -14 $OUT/rocprofv3_kernel_stats_r5_prefill.csv | cut -c1-220
