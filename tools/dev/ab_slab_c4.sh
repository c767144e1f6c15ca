export SEDT_DEV=1
for v in "1 1" "0 0" "1 0" "1 1" "0 0"; do set -- $v
  echo "cfg c4 SLAB_ENC=$1 SLAB_ENC_BWD=$2"
  SEDT_SLAB_ENC=$1 SEDT_SLAB_ENC_BWD=$2 python bench.py --config c4 --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', d['ms_per_step'], 'clips/s', d['value'])"
done
