#!/bin/bash
# round 6: what do the GEMM-family launches of the C2 step wait for?  LDS-side counters per kernel over one steady-state step
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_pmc_lds
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --config c2 --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/pmc_a -- $B > $out/pmc_a.log 2>&1 || echo "pmc a failed"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $out/pmc_b -- $B > $out/pmc_b.log 2>&1 || echo "pmc b failed"
python3 $root/tools/pmc_cache_summary.py $out/lds_c2.csv $out/pmc_a $out/pmc_b > $out/lds.log 2>&1
rm -rf $out/pmc_a $out/pmc_b
head -30 $out/lds_c2.csv
