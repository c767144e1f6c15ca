"""old (im2col -> GEMM -> max-pool / max-pool backward -> wgrad) against the one-launch stem kernels (csrc/stem.hip), captured graphs"""
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
for B, H in ((64, 500), (32, 496), (2000, 128)):
    W = 64
    x = torch.randn(B, 1, H, W, generator=g).cuda()
    w0, b0 = torch.randn(3, 1, 1, 1, generator=g).cuda(), torch.randn(3, generator=g).cuda()
    w1 = (torch.randn(64, 3, 7, 7, generator=g) / 12).cuda()
    sc, bi = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda() * 0.1
    wcat = ops.stem_prep(dt, w0, b0, w1)
    def old_fwd():
        col, Ho, Wo = ops.stem_im2col(dt, x, B, H, W)
        s1 = ops.linear(dt, col, wcat, scale=sc, bias=bi, act=L.ACT_RELU)
        return ops.maxpool_fwd(dt, s1, B, Ho, Wo, 64) + (col,)
    pool, idx, Hp, Wp, col = old_fwd()
    Ho = (H - 1) // 2 + 1
    gy = torch.randn(B * Hp * Wp, 64, generator=g).to('cuda', torch.bfloat16)
    def old_bwd():
        gs = ops.maxpool_bwd(dt, gy, idx, None, B, Ho, 32, 64, y=pool)
        return ops.wgrad(dt, gs, col, gs.shape[0], ops.ConvGeom(1, 1, 128, 64), rowscale=sc)
    t = [timeit(old_fwd), timeit(lambda: ops.stem_pool_fwd(x, wcat, sc, bi, B, H, W)),
         timeit(lambda: ops.stem_pool_fwd(x, wcat, sc, bi, B, H, W, want_idx=False)),
         timeit(old_bwd), timeit(lambda: ops.stem_pool_wgrad(x, gy, idx, pool, sc, B, H, W))]
    print(f'B={B} H={H}: fwd old {t[0]:.1f} us, one-launch {t[1]:.1f} (no idx {t[2]:.1f}); bwd old {t[3]:.1f}, one-launch+reduce {t[4]:.1f}', flush=True)
    lib = L.load()
    ns = lib.sedt_stem_pool_wgrad_slabs(B, H)
    slab = torch.empty((ns, 64, 128), device='cuda', dtype=torch.float32)
    Gm = torch.empty((64, 128), device='cuda', dtype=torch.float32)
    p = ops._p
    tk = timeit(lambda: L.check(lib.sedt_stem_pool_wgrad(p(x), p(gy), p(idx), p(pool), p(slab), ns, B, H, W, L.stream_ptr()), 'k'))
    tr = timeit(lambda: L.check(lib.sedt_wgrad_reduce_bias(p(slab), ns, 64, 1, 128, p(sc), p(Gm), None, None, L.stream_ptr()), 'r'))
    print(f'   wgrad kernel alone {tk:.1f} us ({ns} slabs), reduce alone {tr:.1f} us', flush=True)
