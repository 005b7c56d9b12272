#!/bin/bash
# The reference's UNMODIFIED examples/main/main.cpp (built against include/compat by `make -C oracle ref_cli`, binary in oracle/_ref/) on the bench's 24-layer Q4_0 file:
# its own timing lines (main.cpp:155-162) with the default sampling flags (top_k 40, top_p 0.9, temp 0.9) and with --top_k 1.
R=${GRAFT_REPO_ROOT:-$PWD}
M=${1:-/tmp/biogpt_amd_bench/synthetic-L24-q4_0.bin}
export BIOGPT_DATA_DIR=$R/tests/golden/tokenizer_data
for flags in "" "--top_k 1" ; do
  for rep in 1 2; do
    echo "== biogpt_ref_cli -n 200 -b 8 -s 3 $flags (run $rep)"
    timeout 120 $R/oracle/_ref/biogpt_ref_cli -m $M -n 200 -b 8 -s 3 $flags -p "the patient was treated with" 2>&1 < /dev/null | grep -E "time|tokens in prompt" | sed 's/^main: *//'
  done
done
