# developer A/B: the reduce launches closing a decoder / encoder layer's backward touching the next encoder layer's backward weights
export SEDT_DEV=1
for v in 0 1 0 1; do
  echo -n "RED_PREFETCH=$v: "
  SEDT_RED_PREFETCH=$v python bench.py --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
