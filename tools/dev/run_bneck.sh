python -m pytest tests/test_bneck_gpu.py -q -x 2>&1 | tail -25 > gpurun_out/${TAG:-bneck}_test.log
export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
for l in ${LAYERS:-1 2}; do LAYER=$l python tools/dev/time_bneck.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${TAG:-bneck}_time.txt 2>&1
