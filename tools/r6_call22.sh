OUT=$PWD/gpurun_out/r6xl; mkdir -p $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1   # writes the model file
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so BIOGPT_HIP_DBG=128 timeout 300 python tools/decode_timeline.py $M 1023 600 300 103 > $OUT/xlong_timeline.txt 2>&1; cat $OUT/xlong_timeline.txt | cut -c1-200
