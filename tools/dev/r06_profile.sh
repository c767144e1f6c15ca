#!/bin/bash
# round 6 measurement record: kernel traces + stats of every BASELINE configuration and the bf16x3 mode, the three PMC passes over C2, C4 and C5,
# summarised on the box (per-kernel step tables, ordered step sequences) so that only the summaries travel back
set -u
root=$GRAFT_REPO_ROOT
cd $root
bash tools/profile_round.sh r06 c2 c3 c4 c5 > gpurun_out/prof_r06.log 2>&1
PMC_CFG=c4 bash tools/profile_round.sh r06c4 c4 > gpurun_out/prof_r06c4.log 2>&1
PMC_CFG=c5 bash tools/profile_round.sh r06c5 c5 > gpurun_out/prof_r06c5.log 2>&1
P=gpurun_out/prof_r06
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$P/trace_c2x3 -- python3 $root/bench.py --config c2 --dtype bf16x3 --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family > $root/$P/trace_c2x3.log 2>&1 )
S=gpurun_out/r06_summaries
mkdir -p $S
python3 tools/pmc_step_summary.py $P $S/r06_step_c2.csv $S/r06_pmc_c2.json c2 1 > $S/sum_c2.log 2>&1
python3 tools/pmc_step_summary.py gpurun_out/prof_r06c4 $S/r06_step_c4.csv $S/r06_pmc_c4.json c4 2 > $S/sum_c4.log 2>&1
python3 tools/pmc_step_summary.py gpurun_out/prof_r06c5 $S/r06_step_c5.csv $S/r06_pmc_c5.json c5 2 > $S/sum_c5.log 2>&1
python3 tools/step_sequence.py $P/trace_c2 $S/r06_step_sequence_c2.txt 1
python3 tools/step_sequence.py $P/trace_c3 $S/r06_step_sequence_c3.txt 1
python3 tools/step_sequence.py $P/trace_c4 $S/r06_step_sequence_c4.txt 2
python3 tools/step_sequence.py $P/trace_c5 $S/r06_step_sequence_c5.txt 2
python3 tools/step_sequence.py $P/trace_c2x3 $S/r06_step_sequence_c2x3.txt 1
for c in c2 c3 c4 c5 c2x3; do
  f=$(find $P/trace_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $S/r06_bench_${c}_kernel_stats.csv
done
cat $S/sum_c2.log $S/sum_c4.log $S/sum_c5.log
tail -1 $S/r06_step_sequence_*.txt
# the raw traces stay on the box (too large to be worth merging): drop them
rm -rf gpurun_out/prof_r06/trace_* gpurun_out/prof_r06/pmc_* gpurun_out/prof_r06c4/trace_* gpurun_out/prof_r06c4/pmc_* gpurun_out/prof_r06c5/trace_* gpurun_out/prof_r06c5/pmc_*
ls -la $S
