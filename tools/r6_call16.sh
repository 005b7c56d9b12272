OUT=$PWD/gpurun_out/final2_r6; mkdir -p $OUT
R=$PWD
timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<PY
import json
d=json.load(open('gpurun_out/final2_r6/bench_default.json')); print('default', d['value'], d['ms_per_step'], d.get('decode_f32'), d.get('roofline'))
PY
for t in f32 f16; do
timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_$t.json 2> $OUT/bench_$t.err
BIOGPT_HIP_FPIPE=0 timeout 600 python bench.py --ftype $t --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_r6_${t}_five_launches.json 2> /dev/null
python - <<PY
import json
for n in ('','_five_launches'):
    d=json.load(open('gpurun_out/final2_r6/bench_r6_$t%s.json' % n)); print('$t'+n, d['value'], d['ms_per_step'], d.get('token_roofline',{}).get('T=104'))
PY
timeout 300 python tools/fpipe_timeline.py $t 100 > $OUT/fpipe_timeline_r6_$t.txt 2>&1
timeout 300 python tools/fpipe_timeline.py $t 200 > $OUT/fpipe_timeline_r6_${t}_200keys.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
BIOGPT_BENCH_SKIP_TYPES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -o f32 -- python $R/bench.py --ftype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_f32_under_rocprof.json 2> /tmp/prof_f32.err
find /tmp/prof_f32 -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6_f32.csv \;
head -6 $OUT/rocprofv3_kernel_stats_r6_f32.csv | cut -c1-200
