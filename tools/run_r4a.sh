set -x
mkdir -p gpurun_out/r4a
tools/microbench18 32 > gpurun_out/r4a/microbench18.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py -m gpu -x -q -k "bench_rccl or beyond_256 or xpipe_step or generation_across" 2>&1 | tail -15 > gpurun_out/r4a/pytest_subset.txt
timeout 600 python tools/ab_quick.py --reps 2 --points 103,200,300,1023 biogpt.cpp_amd/libbiogpt_hip_nt.so biogpt.cpp_amd/libbiogpt_hip.so > gpurun_out/r4a/ab_sc1.txt 2>&1
cat gpurun_out/r4a/pytest_subset.txt gpurun_out/r4a/ab_sc1.txt
