#!/usr/bin/env python3
"""A/B of one conv shape against the plain GEMM of the same (M, N, K), per tile: is the time in the gather or in the GEMM skeleton?
usage: shape_probe.py Hi Wi Ci Co k dil   (env SEDT_IGEMM3_STAGES etc. apply)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import ops

Hi, Wi, Ci, Co, k, dl = [int(v) for v in sys.argv[1:7]]
B = 64
g = ops.ConvGeom(Hi, Wi, Ci, Co, k, 1, dl * (k // 2), dl)
M, K = B * Hi * Wi, Ci * k * k
x = torch.randn(M, Ci, device='cuda').bfloat16()
w = torch.randn(Co, Ci, k, k, device='cuda') / K ** 0.5
wf, wb = ops.pack_conv(1, w)
y = torch.empty(M, Co, device='cuda', dtype=torch.bfloat16)
xp = torch.randn(M, K, device='cuda').bfloat16()
wp = torch.randn(Co, K, device='cuda').bfloat16()


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


fl = 2.0 * M * Co * K
for tile in ((64, 64), (128, 64), (64, 128), (128, 128)):
    if Co % tile[1]:
        continue
    a = t(lambda: ops.conv_fwd(1, x, B, g, wf, out=y, tile=tile))
    b = t(lambda: ops.linear(1, xp, wp, out=y, tile=tile))
    print(f'tile {tile}: conv {a:.1f} us ({fl / a / 1e6:.0f} TF)   plain GEMM {b:.1f} us ({fl / b / 1e6:.0f} TF)', flush=True)
