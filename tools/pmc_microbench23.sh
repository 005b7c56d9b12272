# PMC passes over tools/microbench23 (attn_tile_kernel alone): where do the SIMDs' cycles go?   usage: bash tools/pmc_microbench23.sh  -> gpurun_out/pmc_mb23/
OUT=$PWD/gpurun_out/pmc_mb23; mkdir -p $OUT; R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- $R/tools/microbench23 512 > /dev/null 2> /tmp/pmc_$tag.err
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $OUT/summary.txt
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_tile' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items(): print("%-28s mean per launch %16.0f  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
done
cat $OUT/summary.txt
