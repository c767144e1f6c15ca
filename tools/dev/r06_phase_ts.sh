#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
export SEDT_DEV=1 SEDT_LIB_AB=$GRAFT_REPO_ROOT/build/dev/libsedt_hip_dev.so SEDT_IGEMM_BREG=${BREG:-0}
timeout 600 python tools/dev/r06_phase_ts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_phase_ts.txt
