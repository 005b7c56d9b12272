OUT=$PWD/gpurun_out/r6h; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_resident.py -x -q > $OUT/tests_res.txt 2>&1; tail -3 $OUT/tests_res.txt
timeout 900 python -m pytest tests/test_gpu_decode_fused.py -x -q -k "xpipe or 24_layers" > $OUT/tests_xp.txt 2>&1; tail -3 $OUT/tests_xp.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench$i.json 2> /dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r6h/bench$i.json'))
print(d['value'], d['ms_per_step'], {k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if a in ('tokens_per_s','frac_of_device_loop','ids_match_device_loop','eval_only_tokens_per_s')}) for k,v in d.items() if k.startswith('api_loop')}, d['token_roofline']['T=1024']['us_per_token'], d['token_roofline']['T=512']['us_per_token'])
PY
done
