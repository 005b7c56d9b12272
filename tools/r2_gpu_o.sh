mkdir -p gpurun_out/r2o
timeout 900 python -m pytest tests/test_gpu_decode_fused.py tests/test_sampler_oracle.py tests/test_compat.py -x -q -m gpu > gpurun_out/r2o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2o/pytest.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2o/bench_q4_0.json 2> gpurun_out/r2o/bench_q4_0.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2o/bench_q4_0.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"])
r=d.get("roofline") or {}
print("fc1", r.get("us_per_launch"), r.get("frac")); print({k:(v["us"],v["frac"]) for k,v in (r.get("other_kernels") or {}).items()})
print("api", d.get("api_loop")); print("api_topk", d.get("api_loop_topk")); print("err", d.get("roofline_error"))
PY
