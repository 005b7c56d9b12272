timeout 1500 python -m pytest tests/test_gpu_decode_fused.py tests/test_gpu_fullsize.py -x -q -k "fused or full_context or long_context or generation" 2>&1 | grep -E "passed|failed|rror" | tail -5
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
for cfg in "" "BIOGPT_HIP_NO_FUSED_DECODE=1"; do
env $cfg python tools/decode_timeline.py $M 103 255 300 511 700 1023 2>&1 | grep -v "loading model" | tr '\n' ' '; echo " [$cfg]"
done
