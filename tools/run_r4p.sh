mkdir -p gpurun_out/r4p
timeout 60 tools/microbench19 3 > gpurun_out/r4p/microbench19.txt 2>&1; cat gpurun_out/r4p/microbench19.txt
D=biogpt.cpp_amd
timeout 900 python tools/ab_quick.py --reps 3 --points 40,103,200 $D/libbiogpt_hip_c2.so $D/libbiogpt_hip.so > gpurun_out/r4p/ab.txt 2>&1
tail -3 gpurun_out/r4p/ab.txt
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py -m gpu -x -q -k "not bench_rccl" > gpurun_out/r4p/pytest.txt 2>&1
grep -E "passed|failed|rror" gpurun_out/r4p/pytest.txt | tail -3
