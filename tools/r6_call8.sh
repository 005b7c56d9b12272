OUT=$PWD/gpurun_out/r6g; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_xcols.py -x -q -s > $OUT/tests_xcols.txt 2>&1; grep -E "chunk evals|24 layers|passed|failed|Error|assert" $OUT/tests_xcols.txt | tail -14
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench.json 2> /dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6g/bench.json'))
print(d['value'], json.dumps(d.get('prompt_chunk_evals'))[:900])
PY
