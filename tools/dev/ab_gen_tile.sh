# developer A/B: tile of the generic (f32 / bf16x3) GEMM.  SEDT_GEN_TILE = 128064 (128 x 64) | 128128, SEDT_GEN_TILE_MIN = minimum number of tiles
export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
for v in "0 0" "128064 512" "128064 256" "128128 512" "128128 256" "0 0"; do set -- $v
  for d in ${DTYPES:-bf16x3}; do
  echo -n "tile $1 min $2 $d: "
  SEDT_GEN_TILE=$1 SEDT_GEN_TILE_MIN=$2 python bench.py --dtype $d --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
