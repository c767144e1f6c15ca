#!/bin/bash
# round 6: the weight gradients of a transformer stack's layers as ONE launch pair (ops.defer_layer_wgrads) - tests, then same-box A/B per configuration
set -u
cd $GRAFT_REPO_ROOT
o=gpurun_out/r06_ab_defer.txt
python -m pytest tests/test_trajectory_gpu.py tests/test_steps_gpu.py tests/test_mixup_steps_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_pooling_gpu.py -q -m gpu -x 2>&1 | tail -3
: > $o
for i in 1 2; do
  for c in c2 c3 c5 c4; do
    python tools/dev/ab_step.py --config $c --replays 100 --set ops.DEFER_LAYER_WGRADS=False --tag per-layer >> $o 2>/dev/null
    python tools/dev/ab_step.py --config $c --replays 100 --tag per-stack >> $o 2>/dev/null
  done
done
cat $o
