"""round 6 probe: are the LDS-DMA GEMMs slowed by power-of-two row strides (L2 channel hot-spotting)?  The same problems with the
activation operands' rows padded by `pad` elements (the kernels take lda / ldb / ldc): forward linear (layer4 conv1 shape and an FFN shape),
and the weight gradient of the layer4 conv1 / conv3 shapes.  Graph-captured timing, 20 launches per replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, lib as L     # noqa: E402

dev = torch.device('cuda')


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def strided(rows, cols, pad):
    buf = torch.randn(rows, cols + pad, device=dev).bfloat16()
    return buf[:, :cols]


for (M, N, K, what) in ((8192, 512, 2048, 'layer4 conv1 fwd'), (8192, 2048, 512, 'layer4 conv3 fwd'), (8192, 2048, 256, 'FFN linear1'),
                        (8192, 256, 2048, 'FFN linear2'), (32768, 256, 1024, 'layer3-sized 1x1')):
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    row = []
    for pad in (0, 8, 64, 136):
        x = strided(M, K, pad)
        outb = torch.empty(M, N + pad, device=dev, dtype=torch.bfloat16)
        out = outb[:, :N]
        t = timeit(lambda: ops.linear(L.BF16, x, w, out))
        row.append(f'pad {pad}: {t:6.1f} us')
    print(f'fwd  {what:22s} M={M} N={N} K={K}: ' + ', '.join(row), flush=True)
for (Mo, No, Kp, what) in ((512, 2048, 8192, 'layer4 conv1 wgrad'), (2048, 512, 8192, 'layer4 conv3 wgrad'), (256, 1024, 32768, 'layer3 conv1 wgrad'),
                           (2048, 256, 8192, 'FFN linear1 wgrad')):
    row = []
    for pad in (0, 8, 64, 136):
        dy = strided(Kp, Mo, pad)
        x = strided(Kp, No, pad)
        t = timeit(lambda: ops.linear_wgrad(L.BF16, dy, x), reps=10)
        row.append(f'pad {pad}: {t:6.1f} us')
    print(f'wgrad {what:21s} Cout={Mo} Cin={No} pixels={Kp}: ' + ', '.join(row), flush=True)
