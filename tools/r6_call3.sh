set -x
OUT=$PWD/gpurun_out/r6b; mkdir -p $OUT
timeout 100 tools/microbench23 512 > $OUT/microbench23.txt 2>&1; head -40 $OUT/microbench23.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "prompt_pass or travel" > $OUT/tests_prompt.txt 2>&1; tail -5 $OUT/tests_prompt.txt
timeout 300 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prefill.json 2> /dev/null; python -c "
import json,sys; d=json.load(open('$OUT/bench_prefill.json')); print(d['value'], d['ms_per_step'])"
