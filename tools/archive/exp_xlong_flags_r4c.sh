# round 4, last session: compiler settings for the long-context translation units (xlong_32_tu, xlong_64_tu), decode time per token at 512 .. 1023 keys; LIBS = variant libraries
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
(for r in 1 2; do for l in $LIBS; do echo "== $l"; BIOGPT_HIP_LIB=$PWD/$l timeout 200 python tools/long_context_sweep.py 512 700 1023 | grep n_past; done; done) > $OUT/xlong_flags.txt 2>&1
cat $OUT/xlong_flags.txt
