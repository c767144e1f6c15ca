#!/bin/bash
# same-box A/B of the wgrad4 variants on the C2 step (SEDT_WGRAD4_STAGES / SEDT_WGRAD4_BM / SEDT_SPLITK_TARGET_WIDE / SEDT_WGRAD4_WIDE_MIN / SEDT_WGRAD4_BIAS)
set -o pipefail
out=gpurun_out/ab_wgrad4.log
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --no-other-configs --no-cpu-baseline --no-kernels --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out
}
for rep in 1 2; do
  run SEDT_X=0
  run SEDT_WGRAD4_BIAS=1
  run SEDT_WGRAD4_BIAS=1 SEDT_SPLITK_TARGET_WIDE=48
  run SEDT_WGRAD4_BIAS=1 SEDT_SPLITK_TARGET_WIDE=40
  run SEDT_WGRAD4_BIAS=1 SEDT_SPLITK_TARGET_WIDE=32
done
