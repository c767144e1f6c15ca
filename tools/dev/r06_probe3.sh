#!/bin/bash
# round 6: the C4 clean-up (dec_in kernel, avgpool8, f32 epilogue for feature_align, host targets) and the grouped bf16x3 weight images:
# tests, then same-box A/B of the two x3 settings and the C4 / x3 / default bench lines
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_probe3
mkdir -p $out
cd $root
python -m pytest tests/test_x3_gpu.py tests/test_trajectory_gpu.py tests/test_model_gpu.py tests/test_steps_gpu.py tests/test_bench_size_parity_gpu.py tests/test_ops_gpu.py tests/test_gradient_parity_gpu.py tests/test_parity_depth_gpu.py -q -m gpu -x -s > $out/tests.log 2>&1
tail -4 $out/tests.log
B="--steps 30 --warmup 5 --settle 30 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks"
python bench.py --config c4 $B > $out/c4.json 2> $out/c4.err
python bench.py --config c2 --dtype bf16x3 $B > $out/x3.json 2> $out/x3.err
python tools/glue_ops.py c4 aten > $out/glue_c4.txt 2>&1
python - <<'PY'
import json, os
o = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r06_probe3')
for n in ('c4', 'x3'):
    try:
        d = json.loads(open(f'{o}/{n}.json').read().strip().split('\n')[-1])
        print(n, d['ms_per_step'], d['value'])
    except Exception as e:
        print(n, 'failed', e, open(f'{o}/{n}.err').read()[-800:])
PY
