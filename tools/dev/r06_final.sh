#!/bin/bash
# end of round 6: the tests touching the last kernel change, then the round's measurement record on the final build, then the driver-style line
set -u
root=$GRAFT_REPO_ROOT
cd $root
python -m pytest tests/test_model_gpu.py tests/test_gradient_parity_gpu.py tests/test_steps_gpu.py tests/test_bench_size_parity_gpu.py -q -m gpu -x 2>&1 | tail -3
bash tools/dev/r06_profile.sh > gpurun_out/r06_profile.log 2>&1
tail -12 gpurun_out/r06_profile.log
bash tools/dev/r06_sequences.sh > gpurun_out/r06_sequences.log 2>&1
tail -6 gpurun_out/r06_sequences.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench9.json 2> gpurun_out/r06_bench9.err
tail -c 300 gpurun_out/r06_bench9.json
