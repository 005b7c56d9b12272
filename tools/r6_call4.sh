set -x
OUT=$PWD/gpurun_out/r6c; mkdir -p $OUT; R=$PWD
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "prompt_pass or travel or matrix_cores" > $OUT/tests_prompt.txt 2>&1; tail -5 $OUT/tests_prompt.txt
timeout 300 python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prefill.json 2> /dev/null; python -c "
import json,sys; d=json.load(open('$OUT/bench_prefill.json')); print(d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o pre -- python $R/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2> /tmp/prof_p.err
find /tmp/prof_p -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_prefill.csv \;
head -12 $OUT/kernel_stats_prefill.csv | cut -c1-160
