OUT=$PWD/gpurun_out/r6fault; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fpipe.py -x -q -s > $OUT/tests_fpipe.txt 2>&1; tail -15 $OUT/tests_fpipe.txt | cut -c1-300
