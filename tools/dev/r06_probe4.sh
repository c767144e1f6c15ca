#!/bin/bash
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_probe4
mkdir -p $out
cd $root
python -m pytest tests/test_steps_gpu.py tests/test_model_gpu.py tests/test_criterion_variants_gpu.py tests/test_criterion_gpu.py tests/test_gradient_parity_gpu.py tests/test_bench_size_parity_gpu.py tests/test_x3_gpu.py tests/test_dp_gpu.py -q -m gpu -x > $out/tests.log 2>&1
tail -4 $out/tests.log
python tools/glue_ops.py c4 aten > $out/glue_c4.txt 2>&1
sed -n 5,20p $out/glue_c4.txt | cut -c1-200
