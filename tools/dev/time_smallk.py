import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, lib as L
dt = L.BF16
g = torch.Generator().manual_seed(1)
rnd = lambda *s, sc=1.0, d=torch.bfloat16: (torch.randn(*s, generator=g) * sc).to('cuda', d)
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
cases = [('l1.conv3 fwd  M128000 K64  N256 +res+relu', 128000, 64, 256, 'res'),
         ('l1.conv1 dgr  M128000 K64  N256 +res+mask', 128000, 64, 256, 'resmask'),
         ('l1.conv1 fwd  M128000 K256 N64  relu', 128000, 256, 64, 'relu'),
         ('l1.conv3 dgr  M128000 K256 N64  mask', 128000, 256, 64, 'mask'),
         ('l2.conv3 fwd  M32256  K128 N512 +res+relu', 32256, 128, 512, 'res'),
         ('l2.conv1 fwd  M32256  K512 N128 relu', 32256, 512, 128, 'relu'),
         ('l3.conv3 fwd  M8192   K256 N1024 +res+relu', 8192, 256, 1024, 'res'),
         ('enc out-proj  M8192   K256 N256 +res', 8192, 256, 256, 'res0')]
for name, M, K, N, ep in cases:
    x = rnd(M, K); w = rnd(N, K, sc=0.05); sc = rnd(N, d=torch.float32); bi = rnd(N, d=torch.float32)
    res = rnd(M, N); mask = rnd(M, N); out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    kw = dict(scale=sc, bias=bi)
    if ep == 'res': kw.update(res=res, ldr=N, act=L.ACT_RELU, act_post_res=1)
    if ep == 'res0': kw.update(res=res, ldr=N)
    if ep == 'resmask': kw = dict(res=res, ldr=N, mask=mask, ldm=N)
    if ep == 'relu': kw.update(act=L.ACT_RELU)
    if ep == 'mask': kw = dict(mask=mask, ldm=N)
    byt = (M * K + M * N + (M * N if 'res' in ep else 0) + (M * N if 'mask' in ep else 0)) * 2
    line = f'{name:46s} ideal {byt/6e6:6.1f}us@6TB/s |'
    for tile in ((0, 0), (64, 64), (64, 128), (64, 256), (128, 128)):
        if tile[1] > N: continue
        try:
            t = timeit(lambda: ops.igemm(dt, M, N, K, x, K, w, K, out, N, tile=tile, **kw))
            line += f' {tile}: {t:6.1f}'
        except Exception as e:
            line += f' {tile}: ERR'
    print(line, flush=True)
