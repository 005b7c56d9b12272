OUT=$PWD/gpurun_out/r6final_check; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -2
timeout 900 python bench.py --no-cpu-baseline --no-pmc > $OUT/bench_default.json 2> /dev/null; python - <<PY
import json
d=json.load(open('gpurun_out/r6final_check/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['decode_f32']['tokens_per_s'])
PY
for t in f32 f16; do timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_$t.json 2> /dev/null; python - <<PY
import json
d=json.load(open('gpurun_out/r6final_check/bench_r6_$t.json')); r=d['roofline']; print('$t', d['value'], d['token_roofline']['T=104']['us_per_token'], r['us_per_launch'], r['frac'])
PY
done
