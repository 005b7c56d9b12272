OUT=$PWD/gpurun_out/r6q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in f32 f16; do
BIOGPT_BENCH_SKIP_TYPES=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$t -o trace -- python $GRAFT_REPO_ROOT/bench.py --ftype $t --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_prof_$t.json 2> $OUT/bench_prof_$t.err
f=$(find $OUT/prof_$t -name '*kernel_stats.csv' | head -1); echo "== $t $f"; head -12 "$f" | cut -c1-200
done
