cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --dtype bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernels --no-other-configs --no-families > $GRAFT_REPO_ROOT/gpurun_out/prof_x3.log 2>&1
cp $(ls $GRAFT_REPO_ROOT/gpurun_out/prof_x3/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/x3_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -name "*.csv" -size +4M -delete
