# Round-4 (last session) evidence on the GPU box: kernel-trace stats of the final library (same commands as tools/collect_profiles_r4b.sh); lands in gpurun_out/final_r4c/
set -x
OUT=$PWD/gpurun_out/final_r4c; mkdir -p $OUT; export R=$PWD
cd /tmp && export TMPDIR=/tmp
BIOGPT_HIP_XPIPE_MULTI=0 BIOGPT_HIP_RESIDENT=0 BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4c.csv \;
BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof_multi.json 2> /tmp/prof_m.err
find /tmp/prof_m -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4c_multi.csv \;
cd $R; timeout 200 python tools/api_loop_modes.py 2>&1 | grep -v loading > $OUT/api_loop_modes_r4c.txt
python tools/long_context_sweep.py 63 103 255 256 300 511 512 700 1023 2>&1 | grep -v loading > $OUT/long_context_sweep_r4c.txt
ls -la $OUT
