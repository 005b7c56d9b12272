set -x
mkdir -p gpurun_out/r4e
D=biogpt.cpp_amd
timeout 900 python tools/ab_quick.py --reps 3 --points 40,103,200,300,1023 $D/libbiogpt_hip_prev.so $D/libbiogpt_hip.so $D/libbiogpt_hip_lpipe.so > gpurun_out/r4e/ab.txt 2>&1
cat gpurun_out/r4e/ab.txt
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py -m gpu -x -q -k "not bench_rccl and not 24_layers" 2>&1 | tail -4
