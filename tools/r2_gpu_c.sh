set -x
mkdir -p gpurun_out/r2c
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1   # writes /tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
BIOGPT_HIP_DBG=32 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M 103 255 > gpurun_out/r2c/stamps.txt 2>&1
cat gpurun_out/r2c/stamps.txt
export R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python $R/tools/decode_timeline.py $M 103 > $R/gpurun_out/r2c/rocprof_run.txt 2>&1
find /tmp/prof_dec -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r2c/kernel_stats_fused.csv \;
head -12 $R/gpurun_out/r2c/kernel_stats_fused.csv
