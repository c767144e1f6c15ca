# same-box A/B of the training step with / without the fused layer1 Bottlenecks (developer switch: SEDT_DEV=1 SEDT_BNECK=0 off | 1 layer1's identity blocks | 2 + layer2's | 3 + layer1's block 0 | 4 + layer2's block 0 | 5 + layer3's identity blocks)
export SEDT_DEV=1
for cfg in ${CFGS:-c2 c4}; do
for v in ${VARIANTS:-4 5 4 5}; do
  echo "cfg $cfg BNECK=$v"
  SEDT_BNECK=$v python bench.py --config $cfg --no-cpu-baseline --no-kernels --no-other-configs --no-families --steps ${STEPS:-40} --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ms/step', d['ms_per_step'], 'clips/s', d['value'])"
done; done
