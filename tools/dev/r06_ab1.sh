#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06_ab1.txt
: > $o
for i in 1 2; do
  python tools/dev/ab_step.py --config c2 --dtype bf16x3 --replays 100 --set ops.X3_WGROUP=0 >> $o 2>/dev/null
  python tools/dev/ab_step.py --config c2 --dtype bf16x3 --replays 100 --set ops.X3_WGROUP=3 >> $o 2>/dev/null
  python tools/dev/ab_step.py --config c2 --dtype bf16x3 --replays 100 --set ops.X3_WGROUP=5 >> $o 2>/dev/null
  python tools/dev/ab_step.py --config c2 --dtype bf16x3 --replays 100 --set ops.X3_WGROUP=7 >> $o 2>/dev/null
done
python tools/dev/ab_step.py --config c4 --replays 60 >> $o 2>/dev/null
python tools/dev/ab_step.py --config c2 --replays 200 >> $o 2>/dev/null
cat $o
