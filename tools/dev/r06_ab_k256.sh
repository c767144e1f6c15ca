#!/bin/bash
# round 6: the K = 256 problems at large M (C4: FFN linear1 / its dgrad at M = 24,800, layer3 conv3 at M = 64,000) are LDS-DMA-bound on 64x64
# tiles (bytes staged per flop); wider tiles on the 4-wave 2-stage kernel?  developer-build knobs, same box
set -u
root=$GRAFT_REPO_ROOT
cd $root
export SEDT_DEV=1 SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so
o=gpurun_out/r06_ab_k256.txt
: > $o
run() { env "$@" python tools/dev/ab_step.py --config $CFG --replays $REP --tag "$*" >> $o 2>/dev/null; }
for CFG in c4 c2 c3; do
  REP=60; [ $CFG = c2 ] && REP=150; [ $CFG = c3 ] && REP=150
  run X=0
  run SEDT_IGEMM_BN128_MINK=256
  run SEDT_IGEMM_BN128_MINK=256 SEDT_IGEMM3_W4_BELOW_K=512
  run SEDT_IGEMM_BN128_MINK=256 SEDT_IGEMM3_W4_BELOW_K=512 SEDT_IGEMM3_STAGES=2
  run X=0
done
cat $o
