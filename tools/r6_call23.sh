OUT=$PWD/gpurun_out/r6xllead; mkdir -p $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc > /dev/null 2>&1
M=/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin
for lead in 0 4 8 12 16 20 24; do echo "lead $lead"; BIOGPT_HIP_XL_LEAD=$lead timeout 300 python tools/long_context_sweep.py 512 700 1023 2>&1 | grep n_past; done
