OUT=$PWD/gpurun_out/r6f; mkdir -p $OUT; R=$PWD
timeout 1500 python -m pytest tests/test_gpu_decode_fused.py -x -q -s -k "512_token_prompt or other_formats or matvec_sweep" > $OUT/tests_new.txt 2>&1; grep -E "24 layers|passed|failed|Error|assert" $OUT/tests_new.txt | tail -12
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "P90 or P201" > $OUT/tests_p90.txt 2>&1; tail -3 $OUT/tests_p90.txt
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o head -- python $R/tools/pmc_target.py $M headline > $OUT/headline_under_rocprof.txt 2> /tmp/prof_h.err
find /tmp/prof_h -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_r6_headline.csv \;
find /tmp/prof_h -name "*kernel_trace.csv" -exec cp {} $OUT/headline_kernel_trace.csv \;
cat $OUT/headline_under_rocprof.txt | grep -v loading; head -6 $OUT/rocprofv3_kernel_stats_r6_headline.csv | cut -c1-200
python3 - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/root/repo/gpurun_out/r6f/headline_kernel_trace.csv')) if 'dec_xpipe' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in rows]
print(len(d), [round(x,1) for x in d[-8:]])
PY
