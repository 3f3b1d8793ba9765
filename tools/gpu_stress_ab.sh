#!/bin/bash
# A/B proof for the assign_kernel DMA race: round 3's library against the current build, under second-stream HBM traffic at
# low_dim 64 and 128.  Output: gpurun_out/stress_ab.txt (the run of round 4: profiles/r4_stress_ab.txt).
# The old library is not kept in the tree; rebuild it with
#   mkdir -p /tmp/r3src tools/ab && git archive 22c84f5 falcon_amd/csrc include | tar -x -C /tmp/r3src &&
#   make -C /tmp/r3src/falcon_amd/csrc -j8 && cp /tmp/r3src/falcon_amd/libfalcon_hip.so tools/ab/libfalcon_hip_r3.so
# (without it the script runs the current build only).
cd $GRAFT_REPO_ROOT
O=gpurun_out/stress_ab.txt
: > $O
for d in 64 128; do
  for lib in tools/ab/libfalcon_hip_r3.so falcon_amd/libfalcon_hip.so; do
    [ -f $lib ] || continue
    timeout 900 python tools/stress_build.py ${REPS_OLD:-300} --d $d --hammer --lib $lib 2>&1 | tail -14 >> $O
  done
done
timeout 1500 python tools/stress_build.py ${REPS_NEW:-2000} --d 128 --hammer --keyed 2>&1 | tail -5 >> $O
timeout 900 python tools/stress_build.py ${REPS_NEW:-2000} --d 64 --hammer 2>&1 | tail -5 >> $O
cat $O
