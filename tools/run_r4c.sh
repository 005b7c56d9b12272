set -x
mkdir -p gpurun_out/r4c
D=biogpt.cpp_amd
timeout 900 python tools/ab_quick.py --reps 2 --points 40,103,200 $D/libbiogpt_hip.so $D/libbiogpt_hip_noexp.so $D/libbiogpt_hip_pipe.so $D/libbiogpt_hip_nosleep.so $D/libbiogpt_hip_pipens.so > gpurun_out/r4c/ab.txt 2>&1
cat gpurun_out/r4c/ab.txt
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py -m gpu -x -q -k "xpipe_step or generation_across or resident" 2>&1 | tail -5
