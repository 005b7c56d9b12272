# Refresh of the api-loop evidence (resident launch one position ahead of the caller): gpurun_out/final_r3b/
OUT=$PWD/gpurun_out/final_r3b
mkdir -p $OUT
timeout 900 python bench.py --steps 5 --warmup 1 > $OUT/bench_r3_n1.json 2> $OUT/bench.err < /dev/null
timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_modes_r3.txt 2>&1 < /dev/null
BIOGPT_HIP_SPEC=0 timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_modes_r3_waiting_for_every_token.txt 2>&1 < /dev/null
BIOGPT_HIP_RESIDENT=0 timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_modes_r3_per_call_launches.txt 2>&1 < /dev/null
API_LOOP_MODES=0 BIOGPT_HIP_RES_DBG=32 timeout 200 python tools/api_loop_modes.py > $OUT/api_loop_device_clock_r3.txt 2>&1 < /dev/null
ls -la $OUT
