# Round-6 evidence on the GPU box (one gpurun call): the full GPU test suite, the driver-shaped bench line, kernel-trace stats of the same commands, the prefill line and its
# kernel stats, the headline's own multi-token launches, the all-matrices mat-vec launch, the long-context sweep, the API loop, the SQ-counter passes over a prompt pass, the
# microbenchmarks.  -> gpurun_out/final4_r6/
set -x
OUT=$PWD/gpurun_out/final4_r6; mkdir -p $OUT; export R=$PWD
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
timeout 900 python bench.py --steps 20 --warmup 2 > $OUT/bench_r6_n1.json 2> $OUT/bench_r6_n1.err; tail -c 400 $OUT/bench_r6_n1.json
timeout 300 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_r6_prefill_q4_0.json 2> /dev/null
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
cd /tmp && export TMPDIR=/tmp
BIOGPT_HIP_XPIPE_MULTI=0 BIOGPT_HIP_RESIDENT=0 BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o pre -- python $R/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2> /tmp/prof_p.err
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6_prefill.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o head -- python $R/tools/pmc_target.py $M headline > $OUT/headline_launches.txt 2> /tmp/prof_h.err
find /tmp/prof_h -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6_headline.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o sw -- python $R/tools/pmc_target.py $M sweep > $OUT/sweep_under_rocprof.txt 2> /tmp/prof_s.err
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6_sweep.csv \;
cd $R
timeout 200 python tools/api_loop_modes.py 2>&1 | grep -v loading > $OUT/api_loop_modes_r6.txt
timeout 300 python tools/long_context_sweep.py 63 103 255 256 300 511 512 700 1023 2>&1 | grep -v loading > $OUT/long_context_sweep_r6.txt
timeout 300 python tools/soak_two_contexts_r5.py 600 300 2>&1 | grep -v "loading\|hand-off" > $OUT/soak_two_contexts_r6.txt

for t in f32 f16; do
timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_$t.json 2> /dev/null
BIOGPT_HIP_FPIPE=0 timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_${t}_five_launches.json 2> /dev/null
timeout 300 python tools/fpipe_timeline.py $t 100 > $OUT/fpipe_timeline_r6_$t.txt 2>&1
done
timeout 300 python tools/fpipe_timeline.py f32 200 > $OUT/fpipe_timeline_r6_f32_200keys.txt 2>&1
cd /tmp
BIOGPT_BENCH_SKIP_TYPES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -o f32 -- python $R/bench.py --ftype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_f32_under_rocprof.json 2> /tmp/prof_f32.err
find /tmp/prof_f32 -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6_f32.csv \;
cd $R

ls -la $OUT
