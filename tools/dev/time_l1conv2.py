"""layer1 / layer2 3x3 convolutions (the N = 64 / 128 implicit GEMMs) in isolation, captured graphs"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, lib as L
dt = L.BF16
g = torch.Generator().manual_seed(1)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3
B = int(os.environ.get("B", "64"))
for name, Hi, Wi, C in (('l1.conv2 3x3 64->64  M=128000', 125, 16, 64),):
    gm = ops.ConvGeom(Hi, Wi, C, C, 3, 1, 1, 1)
    x = torch.randn(B * Hi * Wi, C, generator=g).to('cuda', torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda()
    sc, bi = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda()
    wf, wb = ops.pack_conv(dt, w, sc)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    flops = 2.0 * x.shape[0] * C * C * 9
    line = f'{name}: {flops/1e9:.1f} GF |'
    tf = timeit(lambda: ops.conv_fwd(dt, x, B, gm, wf, out=y, scale=sc, bias=bi, act=L.ACT_RELU))
    td = timeit(lambda: ops.conv_dgrad(dt, x, B, gm, wb, out=dx, mask=y, ldm=C))
    line += f' direct: fwd {tf:.1f} us ({flops/tf/1e6:.0f} TF/s) dgrad {td:.1f} |'
    for tile in ((64, 64),):
        if tile[1] > C: continue
        try:
            tf = timeit(lambda: ops.conv_fwd(dt, x, B, gm, wf, out=y, scale=sc, bias=bi, act=L.ACT_RELU, tile=tile))
            td = timeit(lambda: ops.conv_dgrad(dt, x, B, gm, wb, out=dx, mask=y, ldm=C, tile=tile))
            line += f' {tile}: fwd {tf:.1f} us ({flops/tf/1e6:.0f} TF/s) dgrad {td:.1f} |'
        except Exception as e:
            line += f' {tile}: {str(e)[:40]} |'
    print(line, flush=True)
