#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh <tag> [configs...]
# rocprofv3 kernel-trace + stats of `bench.py --config <cfg>` for every config given (default c2), and - for c2 - three
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ matrix-core + instruction counters) as MI355X_MICROARCH.md prescribes.
# Everything lands under gpurun_out/prof_<tag>/; tools/pmc_step_summary.py turns it into the tables committed in profiles/.
set -u
tag=$1; shift
cfgs=${@:-c2}
pmc=${PMC_CFG:-c2}      # the configuration the three PMC passes run (PMC_CFG=c3 bash tools/profile_round.sh tag c3)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
mkdir -p $out
python3 -c "import sys; sys.path.insert(0, '$root'); from sound_event_detection_transformer_amd import _build; print(_build.source_stamp())" > $out/build_stamp.txt
cd /tmp && export TMPDIR=/tmp
for c in $cfgs; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$c -- python3 $root/bench.py --config $c --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family > $out/trace_$c.log 2>&1 || echo "trace $c failed"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --config $pmc --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family > $out/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --config $pmc --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family > $out/pmc_write.log 2>&1 || echo "pmc write failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --config $pmc --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks --no-gemm-family > $out/pmc_sq.log 2>&1 || echo "pmc sq failed"
ls $out
