#!/usr/bin/env python3
"""Run one GEMM shape of the step repeatedly (for rocprofv3 --pmc).  usage: one_gemm.py <name> [fwd|dgrad|wgrad] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import ops
SH = {'l4.conv2': (32, 4, 512, 512, 3, 1, 2, 2), 'l4.conv1': (32, 4, 2048, 512, 1, 1, 0, 1), 'l3.conv2': (32, 4, 256, 256, 3, 1, 1, 1),
      'ffn1': (128, 1, 256, 2048, 1, 1, 0, 1), 'l2.conv2': (63, 8, 128, 128, 3, 1, 1, 1), 'l1.conv3': (125, 16, 64, 256, 1, 1, 0, 1)}
name = sys.argv[1]; kind = sys.argv[2] if len(sys.argv) > 2 else 'fwd'; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
Hi, Wi, Ci, Co, k, s, pd, dl = SH[name]
B = 64
g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
x = torch.randn(B * Hi * Wi, Ci, device='cuda').bfloat16()
w = torch.randn(Co, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5
wf, wb = ops.pack_conv(1, w)
gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda').bfloat16()
y = torch.empty(B * g.Ho * g.Wo, Co, device='cuda', dtype=torch.bfloat16); dx = torch.empty_like(x)
for _ in range(iters):
    if kind == 'fwd': ops.conv_fwd(1, x, B, g, wf, out=y)
    elif kind == 'dgrad': ops.conv_dgrad(1, gy, B, g, wb, out=dx)
    else: ops.wgrad(1, gy, x, B, g)
torch.cuda.synchronize()
