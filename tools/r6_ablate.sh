# the matrix-core loop without one of its ingredients (tools/microbench22 built with -DMFMA_ABLATE=k): which one does a compute unit wait for?
OUT=$PWD/gpurun_out/r6d; mkdir -p $OUT; : > $OUT/mfma_ablation.txt
for a in 0 1 2 3 4 5; do echo "== MFMA_ABLATE=$a (0 nothing removed, 1 weight-scale LDS reads, 2 B-operand LDS reads, 3 A-operand global loads, 4 the MFMAs, 5 the block arithmetic)" >> $OUT/mfma_ablation.txt; MB_ONLY=1 timeout 60 tools/microbench22_abl$a 2>&1 | cut -c1-230 >> $OUT/mfma_ablation.txt; done
cat $OUT/mfma_ablation.txt
