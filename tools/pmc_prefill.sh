# SQ-counter passes over one 512-column prompt pass (tools/pmc_target.py <model> prefill): per kernel of the pass (matmul_mfma_kernel<...>, attn_tile_kernel, lnq_kernel, embed ...)
# the mean per launch of every counter.  One rocprofv3 run per counter set (<= 8 SQ counters a pass; never together with --stats / a trace domain other than the kernel trace).
#   usage: bash tools/pmc_prefill.sh [outdir]   -> <outdir>/pmc_prefill_summary.txt
OUT=${1:-$PWD/gpurun_out/pmc_prefill}; mkdir -p $OUT; R=$PWD
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2> /dev/null | grep -o "SQ_INSTS_VALU_MFMA[A-Z0-9_]*\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_MFMA\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_INSTS_VALU_[A-Z0-9_]*" | sort -u > $OUT/counters_listed.txt
: > $OUT/pmc_prefill_summary.txt
n=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  n=$((n + 1)); rm -rf /tmp/pmcp_$n
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcp_$n -o p -- python $R/tools/pmc_target.py $M prefill > /dev/null 2> /tmp/pmcp_$n.err
  f=$(find /tmp/pmcp_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY' >> $OUT/pmc_prefill_summary.txt
import csv, sys, collections, re
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    k = re.sub(r'\(.*', '', k)[:64]
    acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(acc.items()):
    if len(v) >= 24: print("%-64s %-28s mean per launch %14.0f  (%d launches)" % (k, c, sum(v) / len(v), len(v)))
PY
  else echo "set '$set': no counter file; $(tail -2 /tmp/pmcp_$n.err | tr '\n' ' ')" >> $OUT/pmc_prefill_summary.txt; fi
done
cat $OUT/pmc_prefill_summary.txt | head -150
