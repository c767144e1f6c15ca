export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
for d in ${DBGS:-0 1 2 3 17}; do SEDT_SLAB_DBG=$d python tools/dev/time_slab.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${TAG:-slab}_time.txt 2>&1
unset SEDT_DEV SEDT_LIB_AB
python -m pytest tests/test_slab_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/${TAG:-slab}_test.log
