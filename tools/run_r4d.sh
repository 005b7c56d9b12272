set -x
mkdir -p gpurun_out/r4d
D=biogpt.cpp_amd
timeout 900 python tools/ab_quick.py --reps 3 --points 40,103,200,300,1023 $D/libbiogpt_hip_prev.so $D/libbiogpt_hip.so $D/libbiogpt_hip_pipens.so > gpurun_out/r4d/ab.txt 2>&1
cat gpurun_out/r4d/ab.txt
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py -m gpu -x -q -k "not bench_rccl" 2>&1 | tail -8
