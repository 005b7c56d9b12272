OUT=$PWD/gpurun_out/r6s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fpipe.py -x -q > $OUT/tests_fpipe.txt 2>&1; tail -5 $OUT/tests_fpipe.txt | cut -c1-300
timeout 300 python tools/fpipe_timeline.py f32 100 > $OUT/tl_f32.txt 2>&1; grep -B13 "workgroup 128" $OUT/tl_f32.txt | cut -c1-250
timeout 300 python tools/fpipe_timeline.py f16 100 > $OUT/tl_f16.txt 2>&1; grep -B13 "workgroup 128" $OUT/tl_f16.txt | cut -c1-250
for t in f32 f16; do timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_$t.json 2> $OUT/bench_$t.err; python - <<PY
import json
d=json.load(open('gpurun_out/r6s/bench_$t.json')); print('$t', d['value'], d['ms_per_step'], d.get('token_roofline'))
PY
done
