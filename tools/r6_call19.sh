OUT=$PWD/gpurun_out/r6lead; mkdir -p $OUT
for lead in 16 20 24 28 32 36; do
for t in f32 f16; do BIOGPT_BENCH_SKIP_TYPES=1 BIOGPT_HIP_FPIPE_LEAD=$lead timeout 600 python bench.py --ftype $t --steps 4 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_${t}_$lead.json 2> /dev/null; python - <<PY
import json
d=json.load(open('gpurun_out/r6lead/bench_${t}_$lead.json')); print('lead $lead $t', d['value'], d['token_roofline']['T=104']['us_per_token'])
PY
done; done
