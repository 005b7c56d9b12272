set -x
mkdir -p gpurun_out/r4g
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4g/pytest_all.txt 2>&1
tail -5 gpurun_out/r4g/pytest_all.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
tail -c 3000 gpurun_out/r4g/bench.json
