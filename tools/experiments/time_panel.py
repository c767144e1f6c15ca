"""row-panel GEMM (csrc/igemm_panel.hip) against the tiled kernel on the step's short-K shapes; usage: python tools/dev/time_panel.py"""
import math
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
import torch
from sound_event_detection_transformer_amd import ops
from sound_event_detection_transformer_amd.lib import BF16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for M, N, K in [(8192, 2048, 256), (8192, 1024, 256), (8192, 512, 256), (32256, 512, 128)]:
    x = torch.randn(M, K, device='cuda').bfloat16()
    w = (torch.randn(N, K, device='cuda') / math.sqrt(K)).bfloat16()
    res = torch.randn(M, N, device='cuda').bfloat16()
    b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    bits = torch.zeros(M, N // 8, device='cuda', dtype=torch.uint8)
    for name, ep in [('plain', {}), ('bias+relu+drop', dict(bias=b, act=ops.ACT_RELU, drop_p=0.1, seed=3)),
                     ('scale+bias+res+relu+bits', dict(scale=b, bias=b, res=res, ldr=N, act=ops.ACT_RELU, act_post_res=1, bits_out=bits)),
                     ('res+maskbits', dict(res=res, ldr=N, mask=bits, ldm=bits.stride(0), mask_bits=True))]:
        tp = timeit(lambda: ops.linear(BF16, x, w, out=out, **ep))
        tt = timeit(lambda: ops.linear(BF16, x, w, out=out, tile=(64, 64), **ep))
        t2 = timeit(lambda: ops.linear(BF16, x, w, out=out, tile=(64, 128), **ep))
        fl = 2.0 * M * N * K
        print(f'{M}x{N}x{K} {name:26s} default {tp:6.1f} us ({fl / tp / 1e6:5.0f} TF/s)   64x64 {tt:6.1f}   64x128 {t2:6.1f}', flush=True)
