for t in 250 400 600 1000 100000; do
for c in c3 c5; do
echo "BN128_MINTILES=$t $c: $(SEDT_IGEMM_BN128_MINTILES=$t python bench.py --config $c --no-cpu-baseline --no-kernels 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')" >> gpurun_out/t46.log
done; done
