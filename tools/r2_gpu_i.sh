M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
BIOGPT_HIP_DBG=96 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M 103 2>&1 | grep -v "loading model"
