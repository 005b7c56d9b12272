# Round-4 evidence collection on the GPU box: everything lands in gpurun_out/final_r4/ (copied into profiles/ afterwards).
set -x
OUT=$PWD/gpurun_out/final_r4
mkdir -p $OUT
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_r4_n1.json 2> $OUT/bench.err < /dev/null
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
export R=$PWD
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats: single-token launches only (every dec_xpipe_kernel / dec_xlong_kernel call is ONE token: the launch roofline.us_per_launch and token_roofline are about),
# and the headline as it runs (multi-token launches, resident api loop)
BIOGPT_HIP_XPIPE_MULTI=0 BIOGPT_HIP_RESIDENT=0 BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4.csv \;
BIOGPT_BENCH_SKIP_TYPES=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof_multi.json 2> /tmp/prof_m.err
find /tmp/prof_m -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4_multi.csv \;
rm -rf /tmp/prof_p; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o p -- python $R/tools/pmc_target.py $M prefill > /dev/null 2>&1
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r4_prefill.csv \;
# PMC passes, each on its own (no trace options beside --kernel-trace)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python $R/tools/pmc_target.py $M > /dev/null 2>&1
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r4.txt 2>&1
  rm -rf /tmp/pmcl_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcl_$c -o p -- python $R/tools/pmc_target.py $M long > /dev/null 2>&1
  DB=$(find /tmp/pmcl_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r4_T1024.txt 2>&1
done
rm -rf /tmp/pmc_mfma; timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p -- python $R/tools/pmc_target.py $M prefill > /dev/null 2>&1
DB=$(find /tmp/pmc_mfma -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB 2>&1 | grep -v "^ *(" > $OUT/pmc_mfma_chain_r4.txt
cd $R
for t in q5_1 q8_0 q4_1 q5_0 f32 f16; do BIOGPT_BENCH_SKIP_TYPES=1 timeout 600 python bench.py --ftype $t --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r4_$t.json 2>/dev/null; done
python bench.py --workload prefill --no-cpu-baseline > $OUT/bench_r4_prefill_q4_0.json 2>/dev/null
BIOGPT_BENCH_CHUNK_CALLS=1 python bench.py --workload prefill --no-cpu-baseline > $OUT/bench_r4_prefill_q4_0_per_eval.json 2>/dev/null
for n in 40 103 200 300 1023; do BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M $n; done > $OUT/xpipe_timeline_r4.txt 2>&1
BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/tail_timeline.py $M 40 > $OUT/tail_timeline_r4.txt 2>&1
python tools/long_context_sweep.py 63 103 255 256 300 511 512 700 1023 > $OUT/long_context_sweep_r4.txt 2>&1
timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_modes_r4.txt 2>&1 < /dev/null
BIOGPT_HIP_SPEC=0 timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_modes_r4_waiting_for_every_token.txt 2>&1 < /dev/null
API_LOOP_MODES=0 BIOGPT_HIP_RES_DBG=32 timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_device_clock_r4.txt 2>&1 < /dev/null
timeout 200 python tools/api_loop_long.py 300 200 > $OUT/api_loop_long_r4.txt 2>&1; timeout 200 python tools/api_loop_long.py 700 200 >> $OUT/api_loop_long_r4.txt 2>&1
bash tools/ref_cli_timing.sh > $OUT/ref_cli_timing_r4.txt 2>&1
timeout 400 python tools/soak_r3.py 180 > $OUT/soak_r4.txt 2>&1
ls -la $OUT
