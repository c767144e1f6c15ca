for cfg in "384 128" "256 128" "512 128" "384 64" "384 192" "384 256"; do
set -- $cfg
for c in c3 c5; do
echo "SPLITK_TARGET=$1 WIDE=$2 $c: $(SEDT_SPLITK_TARGET=$1 SEDT_SPLITK_TARGET_WIDE=$2 python bench.py --config $c --no-cpu-baseline --no-kernels 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')" >> gpurun_out/t49.log
done; done
