for k in 2048 1024 512; do for t in 256 512 1024; do
echo "K>=$k tiles>=$t" >> gpurun_out/c4tile.log
SEDT_IGEMM_BM128_MINK=$k SEDT_IGEMM_BM128_MINTILES=$t python bench.py --config c4 --no-cpu-baseline --no-kernels 2>/dev/null | cut -c1-190 >> gpurun_out/c4tile.log
done; done
