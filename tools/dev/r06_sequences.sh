#!/bin/bash
# kernel traces only -> ordered step sequences (a step inside the timed loop) for every configuration
set -u
root=$GRAFT_REPO_ROOT
cd $root
S=gpurun_out/r06_summaries
mkdir -p $S
for spec in "c2 bf16 1 c2" "c3 bf16 1 c3" "c4 bf16 2 c4" "c5 bf16 2 c5" "c2 bf16x3 1 c2x3"; do
  set -- $spec
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tr_$4 -- python3 $root/bench.py --config $1 --dtype $2 --steps 6 --warmup 2 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family > $root/gpurun_out/tr_$4.log 2>&1 )
  python3 tools/step_sequence.py gpurun_out/tr_$4 $S/r06_step_sequence_$4.txt $3
  rm -rf gpurun_out/tr_$4
done
grep "#" $S/r06_step_sequence_*.txt
