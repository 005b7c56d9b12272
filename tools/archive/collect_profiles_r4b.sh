# Round-4 (second half) evidence collection on the GPU box: everything lands in gpurun_out/final_r4b/ (copied into profiles/ afterwards).
set -x
OUT=$PWD/gpurun_out/final_r4b
mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_r4b_n1.json 2> $OUT/bench.err < /dev/null
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
export R=$PWD
(timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $OUT/pytest_gpu_r4b.txt
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats: single-token launches only, and the headline as it runs (multi-token launches, resident api loop)
BIOGPT_HIP_XPIPE_MULTI=0 BIOGPT_HIP_RESIDENT=0 BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4b.csv \;
BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof_multi.json 2> /tmp/prof_m.err
find /tmp/prof_m -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4b_multi.csv \;
# the chunk launch (kernels_xcols.hip.h): kernel stats of the reference's prompt loop, and its fetched bytes (every XCD streams all weights)
rm -rf /tmp/prof_x; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o x -- python $R/tools/dbg_xcols.py q4_0 > /dev/null 2>&1
find /tmp/prof_x -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4b_chunk_evals.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcx_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcx_$c -o p -- python $R/tools/pmc_target.py $M chunk > /dev/null 2>&1
  DB=$(find /tmp/pmcx_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r4b_chunk.txt 2>&1
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python $R/tools/pmc_target.py $M > /dev/null 2>&1
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r4b.txt 2>&1
  rm -rf /tmp/pmcd_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcd_$c -o p -- python $R/tools/pmc_target.py $M dual > /dev/null 2>&1
  DB=$(find /tmp/pmcd_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r4b_T400.txt 2>&1
done
cd $R
for t in q5_1 q8_0 q4_1 q5_0; do BIOGPT_BENCH_SKIP_TYPES=1 timeout 600 python bench.py --ftype $t --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r4b_$t.json 2>/dev/null; done
BIOGPT_BENCH_CHUNK_CALLS=1 python bench.py --workload prefill --no-cpu-baseline > $OUT/bench_r4b_prefill_q4_0_per_eval.json 2>/dev/null
for n in 40 103 300 400; do BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M $n; done > $OUT/xpipe_timeline_r4b.txt 2>&1
for np in 40 200; do BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so timeout 120 python tools/xcols_timeline.py $M $np 8; done 2>&1 | grep -v loading > $OUT/xcols_timeline_r4b.txt
python tools/long_context_sweep.py 63 103 255 256 300 511 512 700 1023 > $OUT/long_context_sweep_r4b.txt 2>&1
timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_modes_r4b.txt 2>&1 < /dev/null
timeout 200 python tools/api_loop_long.py 300 200 > $OUT/api_loop_long_r4b.txt 2>&1; timeout 200 python tools/api_loop_long.py 700 200 >> $OUT/api_loop_long_r4b.txt 2>&1
bash tools/ref_cli_timing.sh > $OUT/ref_cli_timing_r4b.txt 2>&1
timeout 400 python tools/soak_r3.py 150 > $OUT/soak_r4b.txt 2>&1
timeout 300 python tools/dbg_streams.py 2>&1 | grep -v loading > $OUT/xcols_streams_r4b.txt
for b in 20 21; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/microbench$b tools/microbench$b.hip 2> /dev/null; done
timeout 100 ./tools/microbench20 > $OUT/microbench20_xcd_stream_r4b.txt 2>&1
timeout 100 ./tools/microbench21 > $OUT/microbench21_handoff_beside_stream_r4b.txt 2>&1
ls -la $OUT
