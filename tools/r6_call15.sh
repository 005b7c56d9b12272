OUT=$PWD/gpurun_out/r6y; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fpipe.py -x -q > $OUT/tests_fpipe.txt 2>&1; tail -3 $OUT/tests_fpipe.txt | cut -c1-300
for lead in 24; do
for t in f32 f16; do BIOGPT_BENCH_SKIP_TYPES=1 BIOGPT_HIP_FPIPE_LEAD=$lead timeout 600 python bench.py --ftype $t --steps 4 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_${t}_$lead.json 2> $OUT/bench_${t}_$lead.err; python - <<PY
import json
d=json.load(open('gpurun_out/r6y/bench_${t}_$lead.json')); print('lead $lead $t', d['value'], d['ms_per_step'])
PY
done; done
BIOGPT_HIP_FPIPE_LEAD=24 timeout 300 python tools/fpipe_timeline.py f32 100 > $OUT/tl_f32.txt 2>&1; grep -B13 "workgroup 128" $OUT/tl_f32.txt | cut -c1-250
