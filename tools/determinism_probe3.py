import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import runtime, ops
runtime.set_compute_dtype('bf16')
torch.manual_seed(0)
B = 64
g = ops.ConvGeom(125, 16, 256, 64, 1, 1, 0, 1)
g3 = ops.ConvGeom(125, 16, 64, 256, 1, 1, 0, 1)
M = B * 125 * 16
xx = torch.randn(M, 256, device='cuda').bfloat16()
x64 = torch.randn(M, 64, device='cuda').bfloat16()
w = torch.randn(64, 256, 1, 1, device='cuda') / 16
w3 = torch.randn(256, 64, 1, 1, device='cuda') / 8
wf, wb = ops.pack_conv(1, w)
wf3, wb3 = ops.pack_conv(1, w3)
gy = torch.randn(M, 64, device='cuda').bfloat16()
res = torch.randn(M, 256, device='cuda').bfloat16()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def count(fn):
    first = fn().clone(); torch.cuda.synchronize()
    bad = 0
    for i in range(N):
        o = fn(); torch.cuda.synchronize()
        bad += int((o != first).sum() > 0)
    return bad
out = torch.zeros(M, 256, device='cuda', dtype=torch.bfloat16)
print('dgrad mask        :', count(lambda: ops.conv_dgrad(1, gy, B, g, wb, mask=xx, ldm=256, out=out)), 'of', N)
print('dgrad plain       :', count(lambda: ops.conv_dgrad(1, gy, B, g, wb, out=out)), 'of', N)
print('fwd K=64 res+relu :', count(lambda: ops.conv_fwd(1, x64, B, g3, wf3, act=1, res=res, ldr=256, out=out)), 'of', N)
print('fwd K=64 plain    :', count(lambda: ops.conv_fwd(1, x64, B, g3, wf3, out=out)), 'of', N)
o64 = torch.zeros(M, 64, device='cuda', dtype=torch.bfloat16)
print('fwd K=256 N=64    :', count(lambda: ops.conv_fwd(1, xx, B, g, wf, act=1, out=o64)), 'of', N)
