set -x
mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py -x -q -s > gpurun_out/r2b/pytest_fused.txt 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r2b/pytest_fused.txt
for cfg in "" "BIOGPT_HIP_NO_FUSED_DECODE=1" "BIOGPT_HIP_FC1_BLOCKS=2" "BIOGPT_HIP_FC2_WAVES=4" "BIOGPT_HIP_FC2_WAVES=8"; do
  tag=$(echo "$cfg" | tr -c 'A-Za-z0-9\n' '_'); [ -z "$tag" ] && tag=default
  env $cfg timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2b/bench_$tag.json 2> gpurun_out/r2b/bench_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2b/bench_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["value"], d["ms_per_step"], d.get("token_roofline",{}).get("T=104"), d.get("roofline_error"))
PY
done
