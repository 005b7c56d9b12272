mkdir -p gpurun_out/r4m
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
python tools/long_context_sweep.py 40 > /dev/null 2>&1
for n in 40 103 200 300 1023; do BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M $n; done > gpurun_out/r4m/xpipe_timeline_r4.txt 2>&1
grep -v "wg " gpurun_out/r4m/xpipe_timeline_r4.txt | tail -75
