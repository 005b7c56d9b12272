OUT=$PWD/gpurun_out/r6k; mkdir -p $OUT
timeout 300 python tools/fpipe_timeline.py f32 100 > $OUT/tl_f32.txt 2>&1; tail -50 $OUT/tl_f32.txt | cut -c1-220
timeout 300 python tools/fpipe_timeline.py f16 100 > $OUT/tl_f16.txt 2>&1; tail -40 $OUT/tl_f16.txt | cut -c1-220
