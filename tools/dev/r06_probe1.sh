#!/bin/bash
# round 6, first GPU call: (1) L2 hit-rate / fabric read-request PMC passes over the C2 step (where are the weight-gradient
# re-reads served?), (2) kernel traces of C3 / C4 / C5 for the per-step sequence (item 7), (3) the default bench line on this box.
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_probe1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_hit -- $B --config c2 > $out/pmc_hit.log 2>&1 || echo "pmc hit failed"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $out/pmc_rd -- $B --config c2 > $out/pmc_rd.log 2>&1 || echo "pmc rd failed"
rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $out/pmc_req -- $B --config c2 > $out/pmc_req.log 2>&1 || echo "pmc req failed"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d $out/pmc_dram -- $B --config c2 > $out/pmc_dram.log 2>&1 || echo "pmc dram failed"
python3 $root/tools/pmc_cache_summary.py $out/cache_c2.csv $out/pmc_hit $out/pmc_rd $out/pmc_req $out/pmc_dram > $out/cache_c2.log 2>&1
for c in c3 c4 c5; do
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$c -- python3 $root/bench.py --config $c --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families > $out/trace_$c.log 2>&1 || echo "trace $c failed"
  python3 $root/tools/step_sequence.py $out/trace_$c $out/step_sequence_$c.txt $([ $c = c3 ] && echo 1 || echo 2) > $out/seq_$c.log 2>&1
done
# keep the merge small: drop the raw traces / counter dumps, keep summaries
rm -rf $out/pmc_hit $out/pmc_rd $out/pmc_req $out/pmc_dram $out/trace_c3 $out/trace_c4 $out/trace_c5
cd $root && python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 600 $out/bench_default.json
ls -la $out
