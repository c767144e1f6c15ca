#!/bin/bash
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_probe5
mkdir -p $out
cd $root
python -m pytest tests/test_steps_gpu.py tests/test_mixup_steps_gpu.py tests/test_pooling_gpu.py tests/test_gradient_parity_gpu.py tests/test_headline_parity_gpu.py -q -m gpu -x > $out/tests.log 2>&1
tail -4 $out/tests.log
python tools/glue_ops.py c5 aten > $out/glue_c5.txt 2>&1
sed -n 5,30p $out/glue_c5.txt | cut -c1-200
python tools/dev/ab_step.py --config c5 --replays 100 2>/dev/null
python tools/dev/ab_step.py --config c2 --replays 200 2>/dev/null
