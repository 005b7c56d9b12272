for i in 1 2 3; do
for lib in libbiogpt_hip.so libbiogpt_hip_alt.so; do
BIOGPT_HIP_LIB=$PWD/biogpt.cpp_amd/$lib python bench.py --workload prefill --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done
done
