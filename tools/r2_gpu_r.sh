mkdir -p gpurun_out/r2r
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "prompt or batched or full_context" > gpurun_out/r2r/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2r/pytest.txt
for cfg in "BIOGPT_HIP_MFMA_NT2_MIN=64" "BIOGPT_HIP_MFMA_NT2_MIN=0"; do
for i in 1 2; do
env $cfg python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pre -o pre -- python $GRAFT_REPO_ROOT/bench.py --workload prefill --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find /tmp/prof_pre -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r2r/kernel_stats_prefill.csv \;
head -9 $GRAFT_REPO_ROOT/gpurun_out/r2r/kernel_stats_prefill.csv | cut -c1-150
