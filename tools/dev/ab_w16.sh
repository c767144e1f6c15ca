#!/bin/bash
# same-box A/B: 16-wave (two K halves per workgroup) form of the 64x128 ping-pong GEMM - which problems take it
out=gpurun_out/ab_w16.log
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --no-other-configs --no-cpu-baseline --no-kernels --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $out
}
for rep in 1 2; do
  run SEDT_X=0
  run SEDT_IGEMM3_W16_TILES=640
  run SEDT_IGEMM3_W16_TILES=640 SEDT_IGEMM_BM128_MINK=100000
  run SEDT_IGEMM3_W16_TILES=1100
  run SEDT_IGEMM_BM128_MINK=100000
done
