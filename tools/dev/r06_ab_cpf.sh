#!/bin/bash
# own-code prefetch of the LDS-DMA GEMMs (csrc/igemm3.hip, SEDT_IGEMM_CPF): same-box A/B + in-step phase stamps
set -u
root=$GRAFT_REPO_ROOT
cd $root
export SEDT_DEV=1 SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so
o=gpurun_out/r06_ab_cpf.txt
: > $o
SEDT_IGEMM_CPF=1 timeout 900 python -m pytest tests/test_headline_parity_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -1
for i in 1 2 3; do
  for b in 0 1; do SEDT_IGEMM_CPF=$b python tools/dev/ab_step.py --config c2 --replays 200 --tag cpf=$b >> $o 2>/dev/null; done
done
for c in c3 c5; do
  for i in 1 2; do
    for b in 0 1; do SEDT_IGEMM_CPF=$b python tools/dev/ab_step.py --config $c --replays 80 --tag cpf=$b >> $o 2>/dev/null; done
  done
done
cat $o
for b in 0 1; do echo "== phase stamps inside the C2 step, cpf=$b"; SEDT_IGEMM_CPF=$b timeout 300 python tools/dev/r06_phase_ts_step.py 2>&1 | grep "in step" | tee -a $o; done
