# developer A/B: lower bound of the strip count for the fused layer3 Bottleneck (C3 / C5's teacher have 128 strips)
export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
for cfg in c3 c5; do for v in 192 128 192 128; do
  echo -n "cfg $cfg BNECK3_MIN=$v: "
  SEDT_BNECK3_MIN=$v python bench.py --config $cfg --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
