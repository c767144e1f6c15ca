for k in 512 1024 2048 100000; do
for c in c3 c5; do
echo "BN128_MINK=$k $c: $(SEDT_IGEMM_BN128_MINK=$k python bench.py --config $c --no-cpu-baseline --no-kernels 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')" >> gpurun_out/t17.log
done; done
