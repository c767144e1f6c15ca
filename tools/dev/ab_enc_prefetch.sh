# developer A/B: the encoder qkv launch touching the next launch's weights (SEDT_ENC_PREFETCH)
export SEDT_DEV=1
for v in 0 1 0 1; do
  echo -n "ENC_PREFETCH=$v: "
  SEDT_ENC_PREFETCH=$v python bench.py --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
