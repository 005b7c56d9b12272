set -x
mkdir -p gpurun_out/r4f
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
python tools/long_context_sweep.py 40 > /dev/null 2>&1
for n in 40 103 200; do BIOGPT_HIP_DBG=128 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M $n; done > gpurun_out/r4f/xpipe_timeline.txt 2>&1
head -60 gpurun_out/r4f/xpipe_timeline.txt
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py -m gpu -x -q -k "not bench_rccl and not 24_layers" > gpurun_out/r4f/pytest.txt 2>&1
tail -3 gpurun_out/r4f/pytest.txt
