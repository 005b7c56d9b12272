set -x
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_decode_fused.py tests/test_sampler_oracle.py tests/test_compat.py -m gpu -x -q -k "topk or top_k or graph_it_captured or placement or cold_weights or sampled or sampler or cli" > gpurun_out/r4i/pytest.txt 2>&1
grep -E "passed|failed|Error|error" gpurun_out/r4i/pytest.txt | tail -5
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4i/bench.json').read().strip().splitlines()[-1])
print(d['value'])
for k in ('api_loop','api_loop_inplace','api_loop_topk','api_loop_long'): print(k, {kk:v for kk,v in d[k].items() if kk!='note' and kk!='speculation'})
print(d['prompt_pass'])
print(d['roofline']['other_kernels'].get('lm_head'), d['roofline']['other_kernels'].get('lm_head_cold'))
print(d.get('roofline_error'))
PY
