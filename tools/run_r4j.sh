set -x
mkdir -p gpurun_out/r4j
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4j/pytest_all.txt 2>&1
grep -E "passed|failed|rror" gpurun_out/r4j/pytest_all.txt | tail -5
