OUT=$PWD/gpurun_out/final3_r6; mkdir -p $OUT
for t in f32 f16; do
timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_$t.json 2> $OUT/bench_$t.err
BIOGPT_HIP_FPIPE=0 timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_${t}_five_launches.json 2> /dev/null
python - <<PY
import json
for n in ('','_five_launches'):
    d=json.load(open('gpurun_out/final3_r6/bench_r6_$t%s.json' % n)); r=d.get('roofline',{}); print('$t'+n, d['value'], d['ms_per_step'], d.get('token_roofline',{}).get('T=104',{}).get('us_per_token'), r.get('kernel','')[:40], r.get('us_per_launch'), r.get('frac'))
PY
timeout 300 python tools/fpipe_timeline.py $t 100 > $OUT/fpipe_timeline_r6_$t.txt 2>&1
done
timeout 300 python tools/fpipe_timeline.py f32 200 > $OUT/fpipe_timeline_r6_f32_200keys.txt 2>&1
grep "layer period\|mean segment" $OUT/fpipe_timeline_r6_f32.txt $OUT/fpipe_timeline_r6_f16.txt $OUT/fpipe_timeline_r6_f32_200keys.txt | cut -c1-260
