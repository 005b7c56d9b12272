set -x
OUT=$PWD/gpurun_out/r6a; mkdir -p $OUT; R=$PWD
timeout 120 tools/microbench24 > $OUT/microbench24.txt 2>&1; cat $OUT/microbench24.txt
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
n=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  n=$((n + 1)); rm -rf /tmp/pmcq_$n
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcq_$n -o p -- python $R/tools/pmc_target.py $M prefill > /dev/null 2> /tmp/pmcq_$n.err
  f=$(find /tmp/pmcq_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY' >> $OUT/pmc_prefill_valu_types.txt
import csv, sys, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r'\(.*', '', r['Kernel_Name'])[:64]
    acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    if len(v) >= 24 and ('mfma' in k or 'attn' in k or 'lnq' in k): print("%-64s %-28s mean per launch %14.0f  (%d launches)" % (k, c, sum(v) / len(v), len(v)))
PY
  else echo "set '$set': no counter file; $(tail -2 /tmp/pmcq_$n.err | tr '\n' ' ')" >> $OUT/pmc_prefill_valu_types.txt; fi
done
cat $OUT/pmc_prefill_valu_types.txt
