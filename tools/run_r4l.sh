mkdir -p gpurun_out/r4l
D=biogpt.cpp_amd
timeout 900 python tools/ab_quick.py --reps 3 --points 40,103,200 $D/libbiogpt_hip_nolate.so $D/libbiogpt_hip.so > gpurun_out/r4l/ab.txt 2>&1
tail -4 gpurun_out/r4l/ab.txt
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py -m gpu -x -q -k "not bench_rccl and not 24_layers" > gpurun_out/r4l/pytest.txt 2>&1
grep -E "passed|failed|rror" gpurun_out/r4l/pytest.txt | tail -3
for L in libbiogpt_hip_nolate.so libbiogpt_hip.so; do BIOGPT_HIP_LIB=$PWD/$D/$L timeout 200 python tools/api_loop_modes.py 2>&1 | grep -v loading; done
