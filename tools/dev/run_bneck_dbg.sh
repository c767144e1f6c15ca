export SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so
for d in ${DBGS:-0 7 24 31 1 2 4 8 16}; do SEDT_BNECK_DBG=$d LAYER=${LAYER:-1} python tools/dev/time_bneck.py 2>&1 | grep "fused" | sed "s/^/dbg $d /"; done > gpurun_out/${TAG:-bneck}_dbg.txt 2>&1
