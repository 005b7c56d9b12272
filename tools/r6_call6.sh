OUT=$PWD/gpurun_out/r6e; mkdir -p $OUT
( time timeout 900 python bench.py --steps 20 --warmup 2 > $OUT/bench_n1.json 2> $OUT/bench_n1.err ) 2> $OUT/bench_time.txt
tail -3 $OUT/bench_time.txt; tail -3 $OUT/bench_n1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6e/bench_n1.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('cpu_baseline'), indent=1)[:3000])
print(d.get('cpu_baseline_error'))
print({k: d['prompt_pass'].get(k) for k in ('ms_per_pass','tokens_per_s')} if 'prompt_pass' in d else None)
PY
