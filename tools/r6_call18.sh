OUT=$PWD/gpurun_out/r6as; mkdir -p $OUT
timeout 900 python bench.py --no-cpu-baseline --no-pmc > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<PY
import json
d=json.load(open('gpurun_out/r6as/bench_default.json')); print('default', d['value'], d['ms_per_step'], d.get('us_per_token_in_launch'), d['roofline']['us_per_launch'], {k:v.get('us_per_token') for k,v in d['token_roofline'].items()}, d.get('api_loop',{}).get('frac_of_device_loop'), d.get('api_loop',{}).get('eval_only_frac'), d.get('prompt_chunk_evals'), d.get('decode_q5_1',{}).get('tokens_per_s'), d.get('decode_q8_0',{}).get('tokens_per_s'))
PY
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -3
