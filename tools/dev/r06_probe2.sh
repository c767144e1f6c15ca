#!/bin/bash
# round 6: (1) the new / changed tests, (2) what a C4 / C5 step contains outside the library, (3) a clean per-step kernel table of the bf16x3 step
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_probe2
mkdir -p $out
cd $root
python -m pytest tests/test_activation_postnorm_gpu.py tests/test_trajectory_gpu.py tests/test_bench_size_parity_gpu.py tests/test_model_gpu.py tests/test_steps_gpu.py tests/test_gradient_parity_gpu.py -q -m gpu -s > $out/tests.log 2>&1
tail -5 $out/tests.log
python tools/glue_ops.py c4 aten > $out/glue_c4.txt 2>&1
python tools/glue_ops.py c5 aten > $out/glue_c5.txt 2>&1
python tools/glue_ops.py c3 aten > $out/glue_c3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace_x3 -- python3 $root/bench.py --config c2 --dtype bf16x3 --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --model-only-off 2> /dev/null > $out/trace_x3.log || rocprofv3 --kernel-trace --output-format csv -d $out/trace_x3 -- python3 $root/bench.py --config c2 --dtype bf16x3 --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks > $out/trace_x3.log 2>&1
python3 $root/tools/step_sequence.py $out/trace_x3 $out/step_sequence_x3.txt > $out/seq_x3.log 2>&1
rm -rf $out/trace_x3
ls -la $out
