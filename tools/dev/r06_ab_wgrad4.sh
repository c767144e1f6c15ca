#!/bin/bash
# round 6, VERDICT r05 item 1: does the weight-gradient launches' fabric traffic cost time?  The developer build's SEDT_WGRAD4_CBMAJOR=1
# re-orders the tiles of the 3x3 problems so that one XCD works on the nine taps of ONE channel block against ONE dY tile (less traffic,
# the same arithmetic): step time (interleaved, same box), FETCH_SIZE of the wgrad rows in both settings, and the gradient tests.
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r06_ab_wgrad4
mkdir -p $out
cd $root
export SEDT_DEV=1 SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so
SEDT_WGRAD4_CBMAJOR=1 python -m pytest tests/test_gradient_parity_gpu.py tests/test_headline_parity_gpu.py -q -m gpu -x -k "bf16 or b64" > $out/tests_cb1.log 2>&1
tail -2 $out/tests_cb1.log
: > $out/ab.txt
for i in 1 2 3; do
  SEDT_WGRAD4_CBMAJOR=0 python tools/dev/ab_step.py --config c2 --replays 200 --tag cbmajor=0 >> $out/ab.txt 2>/dev/null
  SEDT_WGRAD4_CBMAJOR=1 python tools/dev/ab_step.py --config c2 --replays 200 --tag cbmajor=1 >> $out/ab.txt 2>/dev/null
done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --config c2 --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernels --no-other-configs --no-families --no-clocks"
for v in 0 1; do
  export SEDT_WGRAD4_CBMAJOR=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$v -- $B > $out/pmc_fetch_$v.log 2>&1 || echo "pmc $v failed"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_hit_$v -- $B > $out/pmc_hit_$v.log 2>&1 || echo "pmc hit $v failed"
  python3 $root/tools/pmc_cache_summary.py $out/cache_$v.csv $out/pmc_fetch_$v $out/pmc_hit_$v > $out/cache_$v.log 2>&1
  rm -rf $out/pmc_fetch_$v $out/pmc_hit_$v
  grep -i "wgrad" $out/cache_$v.csv
done
