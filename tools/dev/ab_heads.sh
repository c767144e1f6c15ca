# same-box A/B of the one-launch heads (SEDT_SLAB_HEADS, developer switch) on the configurations with many head rows
export SEDT_DEV=1
for cfg in ${CFGS:-c4 c3}; do
for v in 1 0 1 0; do
  echo -n "cfg $cfg SLAB_HEADS=$v: "
  SEDT_SLAB_HEADS=$v python bench.py --config $cfg --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 30 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
