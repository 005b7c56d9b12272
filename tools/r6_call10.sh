OUT=$PWD/gpurun_out/r6i; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py -x -q -k "xlong or beyond_256 or two_workgroups" > $OUT/tests_xl.txt 2>&1; tail -3 $OUT/tests_xl.txt
timeout 300 python tools/long_context_sweep.py 63 103 255 256 300 511 512 700 1023 2>&1 | grep -v loading > $OUT/long_context_sweep_r6.txt; cat $OUT/long_context_sweep_r6.txt
