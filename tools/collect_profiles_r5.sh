# Round-5 evidence on the GPU box (one gpurun call): the full GPU test suite, the driver-shaped bench line, kernel-trace stats of the same commands, the prefill line and its
# kernel stats, the all-matrices mat-vec launch under rocprofv3, the long-context sweep, the API loop, the two microbenchmarks with per-workgroup stamps.  -> gpurun_out/final_r5/
set -x
OUT=$PWD/gpurun_out/final_r5; mkdir -p $OUT; export R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
timeout 600 python bench.py --steps 20 --warmup 2 > $OUT/bench_r5_n1.json 2> $OUT/bench_r5_n1.err; tail -c 600 $OUT/bench_r5_n1.json
timeout 300 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_r5_prefill_q4_0.json 2> /dev/null
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
cd /tmp && export TMPDIR=/tmp
BIOGPT_HIP_XPIPE_MULTI=0 BIOGPT_HIP_RESIDENT=0 BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r5.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o pre -- python $R/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2> /tmp/prof_p.err
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r5_prefill.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o sw -- python $R/tools/pmc_target.py $M sweep > $OUT/sweep_under_rocprof.txt 2> /tmp/prof_s.err
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r5_sweep.csv \;
cd $R
timeout 200 python tools/api_loop_modes.py 2>&1 | grep -v loading > $OUT/api_loop_modes_r5.txt
timeout 300 python tools/long_context_sweep.py 63 103 255 256 300 511 512 700 1023 2>&1 | grep -v loading > $OUT/long_context_sweep_r5.txt
# (built here, against the headers as they are: a binary left over from an earlier revision of the kernel reads stamps the kernel no longer writes)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMFMA_STAMPS -Ibiogpt.cpp_amd/csrc -Iinclude -o /tmp/microbench22 tools/microbench22.hip && timeout 100 /tmp/microbench22 > $OUT/microbench22_mfma_stamps_r5.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DATTN_STAMPS -Ibiogpt.cpp_amd/csrc -Iinclude -o /tmp/microbench23 tools/microbench23.hip && timeout 100 /tmp/microbench23 512 > $OUT/microbench23_attn_stamps_r5.txt 2>&1
timeout 300 python tools/soak_two_contexts_r5.py 600 300 2>&1 | grep -v "loading\|hand-off" > $OUT/soak_two_contexts_r5.txt
ls -la $OUT
