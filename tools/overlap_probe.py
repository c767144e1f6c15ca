#!/usr/bin/env python3
"""Do a dgrad GEMM and a weight-gradient GEMM of layer4 overlap when issued on two streams?  (sizing the benefit of
co-scheduling wgrad tiles into under-filled dgrad launches)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import ops
B = 64
def mk(Hi, Wi, Ci, Co, k, s, pd, dl):
    g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
    x = torch.randn(B * Hi * Wi, Ci, device='cuda').bfloat16()
    w = torch.randn(Co, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5
    wf, wb = ops.pack_conv(1, w)
    gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda').bfloat16()
    dx = torch.empty_like(x)
    return g, x, wb, gy, dx
cases = {'l4.conv2': (32, 4, 512, 512, 3, 1, 2, 2), 'l4.conv1': (32, 4, 2048, 512, 1, 1, 0, 1), 'l3.conv2': (32, 4, 256, 256, 3, 1, 1, 1)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, sh in cases.items():
    g, x, wb, gy, dx = mk(*sh)
    g2, x2, wb2, gy2, dx2 = mk(*sh)
    def dgrad(): ops.conv_dgrad(1, gy, B, g, wb, out=dx)
    def wgrad(): ops.wgrad(1, gy2, x2, B, g2)
    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    def both():
        with torch.cuda.stream(s1): dgrad()
        with torch.cuda.stream(s2): wgrad()
    def serial():
        dgrad(); wgrad()
    print(name, 'dgrad %.1f us  wgrad(+reduce) %.1f us  serial %.1f us  two streams %.1f us' % (timeit(dgrad), timeit(wgrad), timeit(serial), timeit(both)))
