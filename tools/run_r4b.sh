set -x
mkdir -p gpurun_out/r4b
L=biogpt.cpp_amd/libbiogpt_hip.so
timeout 900 python tools/ab_quick.py --reps 2 --points 40,103,200,300,1023 $L:BIOGPT_HIP_HOP_PLACE=2 $L:BIOGPT_HIP_HOP_PLACE=0 $L:BIOGPT_HIP_HOP_PLACE=1,BIOGPT_HIP_VERBOSE=1 > gpurun_out/r4b/ab_hop_place.txt 2>&1
BIOGPT_HIP_VERBOSE=1 python tools/long_context_sweep.py 103 2>&1 | tail -3 >> gpurun_out/r4b/ab_hop_place.txt
cat gpurun_out/r4b/ab_hop_place.txt
timeout 900 python -m pytest tests/test_gpu_decode_fused.py -m gpu -x -q -k "xpipe_step or generation_across or xlong" 2>&1 | tail -5
