# developer A/B (round 5): tile rules on C4 (B = 200 clips + 2000 patches: M is 8x C2's, so more shapes have enough 128x128 tiles)
export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
run() {
  echo -n "$1: "
  env $1 python bench.py --config c4 --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
}
run X=0
run SEDT_IGEMM_BM128_MINK=1024
run SEDT_IGEMM_BM128_MINK=512
run SEDT_IGEMM_BM128_MINK=256
run X=0
run SEDT_IGEMM_BN128_MINK=256
run X=0
