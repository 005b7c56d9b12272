mkdir -p gpurun_out/r4k
python tools/long_context_sweep.py 40 > /dev/null 2>&1
timeout 200 python tools/api_loop_modes.py > gpurun_out/r4k/api_loop_modes.txt 2>&1 < /dev/null
BIOGPT_HIP_SPEC=0 timeout 200 python tools/api_loop_modes.py > gpurun_out/r4k/api_loop_modes_nospec.txt 2>&1 < /dev/null
API_LOOP_MODES=0 BIOGPT_HIP_RES_DBG=32 timeout 200 python tools/api_loop_modes.py > gpurun_out/r4k/api_loop_device_clock.txt 2>&1 < /dev/null
cat gpurun_out/r4k/api_loop_modes.txt gpurun_out/r4k/api_loop_modes_nospec.txt; tail -20 gpurun_out/r4k/api_loop_device_clock.txt
