#!/usr/bin/env python3
"""Which op of the forward is not bit-reproducible?  Repeats single kernels on fixed inputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import runtime, ops
runtime.set_compute_dtype('bf16')
torch.manual_seed(0)
B = 64

def rep(name, fn, n=4):
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    d = max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:])
    print(f'{name:48s} identical={same} maxdiff={d:.3e}', flush=True)

# stem
x = torch.randn(B, 1, 500, 64, device='cuda')
rep('stem_im2col', lambda: ops.stem_im2col(1, x, B, 500, 64)[0])
shapes = [('l1.conv1 1x1 256->64', 125, 16, 256, 64, 1, 1, 0, 1), ('l1.conv2 3x3 64', 125, 16, 64, 64, 3, 1, 1, 1),
          ('l1.conv3 1x1 64->256', 125, 16, 64, 256, 1, 1, 0, 1), ('l2.conv2 3x3 128', 63, 8, 128, 128, 3, 1, 1, 1),
          ('l3.conv2 3x3 256', 32, 4, 256, 256, 3, 1, 1, 1), ('l3.conv3 256->1024', 32, 4, 256, 1024, 1, 1, 0, 1),
          ('l4.conv1 2048->512', 32, 4, 2048, 512, 1, 1, 0, 1), ('l4.conv2 3x3 512 d2', 32, 4, 512, 512, 3, 1, 2, 2),
          ('l4.conv3 512->2048', 32, 4, 512, 2048, 1, 1, 0, 1), ('ffn1 256->2048', 128, 1, 256, 2048, 1, 1, 0, 1),
          ('ffn2 2048->256', 128, 1, 2048, 256, 1, 1, 0, 1)]
for name, Hi, Wi, Ci, Co, k, s, pd, dl in shapes:
    g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
    xx = torch.randn(B * Hi * Wi, Ci, device='cuda').bfloat16()
    w = torch.randn(Co, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5
    wf, wb = ops.pack_conv(1, w)
    gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda').bfloat16()
    res = torch.randn(B * g.Ho * g.Wo, Co, device='cuda').bfloat16()
    rep(name + ' fwd', lambda: ops.conv_fwd(1, xx, B, g, wf, act=1, res=res, ldr=Co))
    rep(name + ' dgrad', lambda: ops.conv_dgrad(1, gy, B, g, wb, mask=xx, ldm=Ci))
    rep(name + ' wgrad', lambda: ops.wgrad(1, gy, xx, B, g))
