#!/bin/bash
# round 6, review item 6a: the 256x128 forward tile with late-read epilogue operands for layer4's N = 2048, K <= 1024 launches (developer build switch)
set -u
root=$GRAFT_REPO_ROOT
cd $root
export SEDT_DEV=1 SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so
o=gpurun_out/r06_ab_bm256.txt
SEDT_IGEMM_BM256=1 python -m pytest tests/test_headline_parity_gpu.py tests/test_ops_gpu.py tests/test_bneck_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^\[slab\|amdgpu.ids" | tail -12
SEDT_IGEMM_BM256=1 python - <<'PY'
import torch
from sound_event_detection_transformer_amd import ops, lib as L
g = torch.Generator().manual_seed(1)
for (M, N, K) in ((8192, 2048, 512), (8192, 2048, 1024), (24832, 2048, 512)):
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()
    sc, bi = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    with L.launch_log() as log:
        y = ops.linear(L.BF16, x, w, scale=sc, bias=bi, res=res, ldr=N, act=L.ACT_RELU, act_post_res=1)
    ref = torch.relu((x.float() @ w.float().t()) * sc + bi + res.float())
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    print(M, N, K, {k: v for k, v in log.items() if k.startswith('igemm:')}, 'max rel err vs torch f32', f'{err:.2e}')
    assert err < 1e-2
PY
: > $o
for i in 1 2 3; do
  SEDT_IGEMM_BM256=0 python tools/dev/ab_step.py --config c2 --replays 200 --tag bm256=0 >> $o 2>/dev/null
  SEDT_IGEMM_BM256=1 python tools/dev/ab_step.py --config c2 --replays 200 --tag bm256=1 >> $o 2>/dev/null
done
for i in 1 2; do
  SEDT_IGEMM_BM256=0 python tools/dev/ab_step.py --config c4 --replays 60 --tag bm256=0 >> $o 2>/dev/null
  SEDT_IGEMM_BM256=1 python tools/dev/ab_step.py --config c4 --replays 60 --tag bm256=1 >> $o 2>/dev/null
done
cat $o
