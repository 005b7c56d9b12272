# Round-2 evidence collection on the GPU box: everything lands in gpurun_out/final_r2/ (copied into profiles/ afterwards).
set -x
OUT=$PWD/gpurun_out/final_r2
mkdir -p $OUT
python bench.py --steps 5 --warmup 1 > $OUT/bench_r2_n1.json 2> $OUT/bench.err
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
export R=$PWD
cd /tmp && export TMPDIR=/tmp
# kernel-trace stats twice: with single-token launches only (BIOGPT_HIP_XPIPE_MULTI=0: every dec_xpipe_kernel call is ONE token, the launch
# roofline.us_per_launch is about) and as the headline runs (the 200-token continuation = three multi-token launches of 60 / 64 / 72 tokens)
BIOGPT_HIP_XPIPE_MULTI=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/prof.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r2.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o dec -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof_multi.json 2> /tmp/prof_m.err
find /tmp/prof_m -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r2_multi.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -o pre -- python $R/bench.py --workload prefill --no-cpu-baseline > /dev/null 2>&1
find /tmp/prof_pre -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r2_prefill.csv \;
# PMC passes, each on its own (no trace options beside --kernel-trace): HBM-side bytes of the decode kernels, MFMA counters of a prompt pass
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p -- python $R/tools/pmc_target.py $M > /dev/null 2>&1
  DB=$(find /tmp/pmc_$c -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_${c}_r2.txt 2>&1
done
rm -rf /tmp/pmc_mfma; timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_mfma -o p -- python $R/tools/pmc_target.py $M prefill > /dev/null 2>&1
DB=$(find /tmp/pmc_mfma -name "*.db" | head -1); python $R/tools/pmc_summary.py $DB > $OUT/pmc_mfma_chain_r2.txt 2>&1
cd $R
for t in q5_1 q8_0 q4_1 q5_0 f16; do python bench.py --ftype $t --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_r2_$t.json 2>/dev/null; done
python bench.py --ftype f32 --steps 2 --warmup 1 --cpu-seconds 12 > $OUT/bench_r2_f32.json 2>/dev/null
python bench.py --workload prefill --no-cpu-baseline > $OUT/bench_r2_prefill_q4_0.json 2>/dev/null
BIOGPT_BENCH_CHUNK_CALLS=1 python bench.py --workload prefill --no-cpu-baseline > $OUT/bench_r2_prefill_q4_0_per_eval.json 2>/dev/null
BIOGPT_HIP_XPIPE=0 BIOGPT_HIP_DBG=96 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M 103 > $OUT/decode_timeline_r2.txt 2>&1
for n in 40 103 200; do BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M $n; done > $OUT/xpipe_timeline_r2.txt 2>&1
[ -x tools/microbench11 ] && timeout 120 tools/microbench11 > $OUT/microbench11_xcd_pipeline_r2.txt 2>&1
ls -la $OUT
