#!/bin/bash
# round 6: per-kernel instruction-cache and address-translation counters over one C2 step (what is cold at a launch inside a step besides the weights)
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_pmc_cold
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $out/p1 -- $B --config c2 > $out/p1.log 2>&1 || echo "p1 failed"
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum --kernel-trace --output-format csv -d $out/p2 -- $B --config c2 > $out/p2.log 2>&1 || echo "p2 failed"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES --kernel-trace --output-format csv -d $out/p3 -- $B --config c2 > $out/p3.log 2>&1 || echo "p3 failed"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/p4 -- $B --config c2 > $out/p4.log 2>&1 || echo "p4 failed"
python3 $root/tools/pmc_cache_summary.py $out/cold_c2.csv $out/p1 $out/p2 $out/p3 $out/p4 > $out/cold_c2.log 2>&1
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
head -45 $out/cold_c2.csv | cut -c1-230
tail -3 $out/*.log | cut -c1-200
