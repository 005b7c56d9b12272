OUT=$PWD/gpurun_out/r6j; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fpipe.py -x -q -s > $OUT/tests_fpipe.txt 2>&1; tail -25 $OUT/tests_fpipe.txt | cut -c1-300
for t in f32 f16; do timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_$t.json 2> $OUT/bench_$t.err; python - <<PY
import json
d=json.load(open('gpurun_out/r6j/bench_$t.json')); print('$t', d['value'], d['ms_per_step'], d.get('token_roofline'))
PY
done
BIOGPT_HIP_FPIPE=0 timeout 600 python bench.py --ftype f32 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_f32_off.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r6j/bench_f32_off.json')); print('f32 five-launch', d['value'], d['ms_per_step'])"
