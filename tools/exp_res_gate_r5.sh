# Round 5: the resident form of the pipelined launch with token-granular exits (waves end where they wait) against the publishing-tag form.
# usage (GPU box): bash tools/exp_res_gate_r5.sh <tag>   -> gpurun_out/res_gate_<tag>/
set -x
TAG=${1:-a}
OUT=$PWD/gpurun_out/res_gate_$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "resident or two_contexts or api or spec or stale or beside" > $OUT/tests.txt 2>&1; tail -5 $OUT/tests.txt
python -c "import bench" 2>/dev/null
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "api_loop", d["api_loop"]["tokens_per_s"], d["api_loop"]["frac_of_device_loop"], d["api_loop"]["speculation"])
for k in ("api_loop_inplace","api_loop_long","api_loop_topk"): print(k, json.dumps(d.get(k))[:300])
PY
timeout 200 python tools/api_loop_modes.py 2>&1 | grep -v loading > $OUT/api_loop_modes.txt; cat $OUT/api_loop_modes.txt
