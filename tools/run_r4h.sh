set -x
mkdir -p gpurun_out/r4h
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "prompt_pass or matrix_cores or batched_prompts" > gpurun_out/r4h/pytest.txt 2>&1
grep -E "passed|failed|Error" gpurun_out/r4h/pytest.txt | tail -3
for L in libbiogpt_hip_prev.so libbiogpt_hip.so libbiogpt_hip_prev.so libbiogpt_hip.so; do
  BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/$L python bench.py --workload prefill --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])"
done > gpurun_out/r4h/prefill_ab.txt 2>&1
cat gpurun_out/r4h/prefill_ab.txt
cd /tmp && export TMPDIR=/tmp
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
for L in libbiogpt_hip_prev.so libbiogpt_hip.so; do
rm -rf /tmp/prof_$L; BIOGPT_HIP_LIB=$GRAFT_REPO_ROOT/biogpt.cpp_amd/$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$L -o p -- python $GRAFT_REPO_ROOT/tools/pmc_target.py $M prefill > /dev/null 2>&1
find /tmp/prof_$L -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/r4h/kernel_stats_prefill_$L.csv \;
done
head -12 $GRAFT_REPO_ROOT/gpurun_out/r4h/kernel_stats_prefill_libbiogpt_hip_prev.so.csv | cut -c1-150
head -12 $GRAFT_REPO_ROOT/gpurun_out/r4h/kernel_stats_prefill_libbiogpt_hip.so.csv | cut -c1-150
