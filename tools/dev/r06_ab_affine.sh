#!/bin/bash
# scale / bias of the 8-wave LDS-DMA GEMMs read before the K loop (build B) instead of in the epilogue (build A): same-box A/B + phase stamps
set -u
root=$GRAFT_REPO_ROOT
cd $root
export SEDT_DEV=1
o=gpurun_out/r06_ab_affine.txt
: > $o
for i in 1 2 3; do
  SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev_a.so python tools/dev/ab_step.py --config c2 --replays 200 --tag affine=late >> $o 2>/dev/null
  SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so python tools/dev/ab_step.py --config c2 --replays 200 --tag affine=early >> $o 2>/dev/null
done
for i in 1 2; do
  SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev_a.so python tools/dev/ab_step.py --config c4 --replays 60 --tag affine=late >> $o 2>/dev/null
  SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so python tools/dev/ab_step.py --config c4 --replays 60 --tag affine=early >> $o 2>/dev/null
done
cat $o
SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so timeout 300 python tools/dev/r06_phase_ts.py 2>&1 | grep "res   hot"
