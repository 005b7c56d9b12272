mkdir -p gpurun_out/r4n
D=biogpt.cpp_amd
timeout 900 python tools/ab_quick.py --reps 2 --points 256,257,300,511,512,700,1023 $D/libbiogpt_hip_c1.so $D/libbiogpt_hip.so > gpurun_out/r4n/ab.txt 2>&1
tail -4 gpurun_out/r4n/ab.txt
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_resident.py tests/test_gpu_fullsize.py -m gpu -x -q -k "xlong or beyond_256 or long_context or full_context" > gpurun_out/r4n/pytest.txt 2>&1
grep -E "passed|failed|rror" gpurun_out/r4n/pytest.txt | tail -3
