#!/bin/bash
# round 6, review item 6b: the weight operand of the 64x128 / 128x128 ping-pong GEMMs through registers (fragment-major image, igemm3_br_kernel)
set -u
root=$GRAFT_REPO_ROOT
cd $root
export SEDT_DEV=1 SEDT_LIB_AB=$root/build/dev/libsedt_hip_dev.so
o=gpurun_out/r06_ab_breg.txt
SEDT_IGEMM_BREG=1 timeout 600 python - <<'PY'
import ctypes as C
import torch
from sound_event_detection_transformer_amd import ops, lib as L
g = torch.Generator().manual_seed(1)
def frag(w):
    N, K = w.shape
    return w.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
for (M, N, K, tile) in ((8192, 512, 2048, (0, 0)), (8192, 2048, 512, (0, 0)), (8192, 256, 1024, (0, 0)), (8000, 1024, 576, (0, 0)), (8192, 512, 4608, (128, 128)),
                        (3968, 2048, 1024, (64, 128)), (8192, 512, 128, (64, 128))):
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()
    sc, bi = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    fr = frag(w)
    outs = []
    names = []
    for use in (False, True):
        y = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        a = ops.igemm_args(M, N, K, x, K, w, K, y, N, scale=sc, bias=bi, res=res, ldr=N, act=L.ACT_RELU, act_post_res=1, tile=tile)
        if use:
            a.bfrag = fr.data_ptr()
        buf = C.create_string_buffer(160)
        L.load().sedt_igemm_describe(C.byref(a), L.BF16, 0, buf, 160)
        names.append(buf.value.decode())
        L.check(L.load().sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
        outs.append(y)
    torch.cuda.synchronize()
    ref = torch.relu((x.float() @ w.float().t()) * sc + bi + res.float())
    err = ((outs[1].float() - ref).abs().max() / ref.abs().max()).item()
    print(M, N, K, names, 'bit-equal', torch.equal(outs[0], outs[1]), 'max rel err vs torch f32', f'{err:.2e}', flush=True)
    assert err < 1e-2
PY
SEDT_IGEMM_BREG=1 timeout 1200 python -m pytest tests/test_headline_parity_gpu.py tests/test_bneck_gpu.py -q -m gpu -x 2>&1 | grep -v "^\[slab\|amdgpu.ids" | tail -5
SEDT_IGEMM_BREG=1 timeout 600 python tools/glue_ops.py c2 2>&1 | grep "GEMM kernel instances"
: > $o
for i in 1 2 3; do
  SEDT_IGEMM_BREG=0 python tools/dev/ab_step.py --config c2 --replays 200 --tag breg=0 >> $o 2>/dev/null
  SEDT_IGEMM_BREG=1 python tools/dev/ab_step.py --config c2 --replays 200 --tag breg=1 >> $o 2>/dev/null
done
cat $o
