OUT=$PWD/gpurun_out/r6d; mkdir -p $OUT
timeout 100 tools/microbench22 > $OUT/microbench22.txt 2>&1
MB_GAPS=2 timeout 60 tools/microbench22 > $OUT/microbench22_gaps_fc1.txt 2>&1
MB_GAPS=3 timeout 60 tools/microbench22 > $OUT/microbench22_gaps_fc1_walk.txt 2>&1
cut -c1-420 $OUT/microbench22.txt | head -8; cat $OUT/microbench22_gaps_fc1.txt | cut -c1-300 | head -4; cat $OUT/microbench22_gaps_fc1_walk.txt | cut -c1-300 | head -4
