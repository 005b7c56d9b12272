#!/bin/bash
# Experiment build of the library (round 4 A/B runs): tools/build_variant.sh NAME "FLAGS" [TU ...]
# recompiles only the named translation units (default: xpipe_tu) with FLAGS (+ -DBIOGPT_HIP_ONLY_Q4_0: a quarter of the compile time) and links them with the
# product build's other objects into biogpt.cpp_amd/libbiogpt_hip_NAME.so.  The flags must not change the layout of bgk::XpParams / XpLayer (engine.o is reused).
set -e
NAME=$1; FLAGS=$2; shift 2
TUS=${@:-xpipe_tu}
cd "$(dirname "$0")/../biogpt.cpp_amd/csrc"
make -j8 > /dev/null
mkdir -p obj_$NAME
OBJS=""
for o in obj/*.o; do
  b=$(basename $o .o)
  if [[ " $TUS " == *" $b "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable -DBIOGPT_HIP_ONLY_Q4_0 $FLAGS -c -o obj_$NAME/$b.o $b.hip &
    OBJS="$OBJS obj_$NAME/$b.o"
  else
    OBJS="$OBJS $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbiogpt_hip_$NAME.so $OBJS -lpthread -ldl
ls -la ../libbiogpt_hip_$NAME.so
