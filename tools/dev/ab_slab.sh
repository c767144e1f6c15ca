# same-box A/B of the C2 / C3 step over the slab switches (developer switches: SEDT_DEV=1)
export SEDT_DEV=1
for cfg in ${CFGS:-c2 c3}; do
for v in ${VARIANTS:-"1 1 0" "1 1 1" "1 1 0" "1 1 1"}; do set -- $v
  echo "cfg $cfg SLAB_ENC=$1 SLAB_ENC_BWD=$2 SLAB_DEC=$3"
  SEDT_SLAB_ENC=$1 SEDT_SLAB_ENC_BWD=$2 SEDT_SLAB_DEC=$3 python bench.py --config $cfg --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', d['ms_per_step'], 'clips/s', d['value'])"
done; done
