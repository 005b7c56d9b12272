mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_sampler_oracle.py -x -q > gpurun_out/r2n/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2n/pytest.txt
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r2n/bench_q4_0.json 2> gpurun_out/r2n/bench_q4_0.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2n/bench_q4_0.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
print(json.dumps(d.get("roofline"), indent=1)[:3000])
print("token", d.get("token_roofline")); print("cpu", d.get("cpu_baseline")); print("api", d.get("api_loop")); print("err", d.get("roofline_error"), d.get("cpu_baseline_error"))
PY
