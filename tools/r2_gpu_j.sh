mkdir -p gpurun_out/r2j
timeout 900 python -m pytest tests/test_gpu_decode_fused.py -x -q > gpurun_out/r2j/pytest_fused.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2j/pytest_fused.txt
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
[ -f $M ] || python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
BIOGPT_HIP_DBG=96 BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_prof.so python tools/decode_timeline.py $M 103 2>&1 | grep -v "loading model"
for i in 1 2; do
python tools/decode_timeline.py $M 103 255 2>&1 | grep -v "loading model"
BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/libbiogpt_hip_plain.so python tools/decode_timeline.py $M 103 255 2>&1 | grep -v "loading model" | sed 's/^/plain stores: /'
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2j/bench_default.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r2j/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('token_roofline',{}).get('T=104'))"
