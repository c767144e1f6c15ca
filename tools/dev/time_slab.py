"""Isolated, graph-captured timings of the encoder slab kernels (csrc/enc_slab.hip) and of the per-op launches they replace.
Phase ablation of sedt_encoder_attn_ffn_fwd (developer build only: SEDT_DEV_BUILD=1 python -m sound_event_detection_transformer_amd._build;
SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so SEDT_SLAB_DBG=<bits> python tools/dev/time_slab.py):
bit 0 no attention, 1 no FFN, 2 no by-product stores, 3 linear1 only, 4 no K/V staging"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, packing, lib as L      # noqa: E402

dev = torch.device('cuda')
dt = L.BF16
B, S, E, H, FF = int(os.environ.get('B', 64)), 128, 256, 8, 2048
M = B * S
g = torch.Generator().manual_seed(1)


def rnd(*shape, scale=1.0, dtype_=torch.bfloat16):
    return (torch.randn(*shape, generator=g) * scale).to(device=dev, dtype=dtype_)


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for _ in range(reps):
            fn()
    g_.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g_.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


x, pos = rnd(M, E), rnd(M, E, scale=0.5)
gam, bet, gam2, bet2 = (rnd(E, dtype_=torch.float32) for _ in range(4))
masters = [torch.nn.Parameter(rnd(n, k, scale=s, dtype_=torch.float32)) for n, k, s in
           ((3 * E, E, 0.06), (E, E, 0.06), (FF, E, 0.06), (E, FF, 0.02))]
b_in, b_o, b1, b2 = rnd(3 * E, dtype_=torch.float32), rnd(E, dtype_=torch.float32), rnd(FF, dtype_=torch.float32), rnd(E, dtype_=torch.float32)
plan = packing.PackPlan(dt, dev, [], masters, (), masters)
plan.run()
torch.cuda.synchronize()
fr = [plan.frag_table[m.data_ptr()] for m in masters]
print('dbg', os.environ.get('SEDT_SLAB_DBG', '0'), 'B', B)
print('enc_qkv_fwd train      %7.2f us' % timeit(lambda: ops.encoder_qkv_fwd(x, pos, gam, bet, fr[0][0], b_in, B, S, train=True)))
print('enc_qkv_fwd no-grad    %7.2f us' % timeit(lambda: ops.encoder_qkv_fwd(x, pos, gam, bet, fr[0][0], b_in, B, S, train=False)))
qk, v, by1 = ops.encoder_qkv_fwd(x, pos, gam, bet, fr[0][0], b_in, B, S, train=True)
for p_ in (0.1, 0.0):
    for tr in (True, False):
        t = timeit(lambda: ops.encoder_attn_ffn_fwd(x, qk, v, None, fr[1][0], b_o, gam2, bet2, fr[2][0], b1, fr[3][0], b2, B, S, FF, p_,
                                                    (7, 3, 5, 6), None, train=tr))
        print('enc_attn_ffn_fwd p=%.1f train=%d %7.2f us' % (p_, tr, t))
x2, by2 = ops.encoder_attn_ffn_fwd(x, qk, v, None, fr[1][0], b_o, gam2, bet2, fr[2][0], b1, fr[3][0], b2, B, S, FF, 0.1, (7, 3, 5, 6), None,
                                   train=True)
ctx, lse, x1, m2, r2, x1n, h = by2
gx2 = rnd(M, E)
if hasattr(ops, 'encoder_ffn_bwd'):
    t = timeit(lambda: ops.encoder_ffn_bwd(gx2, h, x1, m2, r2, gam2, fr[3][1], fr[2][1], fr[1][1], B, S, 0.1, (6, 3), None))
    print('enc_ffn_bwd            %7.2f us' % t)
    dqk, dv = rnd(M, 2 * E), rnd(M, E)
    t = timeit(lambda: ops.encoder_qkv_bwd(dqk, dv, x, by1[2], by1[3], gam, gx2, fr[0][1], B, S))
    print('enc_qkv_bwd            %7.2f us' % t)
# the per-op launches
w1b, w2b = masters[2].detach().bfloat16(), masters[3].detach().bfloat16()
t = timeit(lambda: ops.linear(dt, x1n, w1b, bias=b1, act=L.ACT_RELU, drop_p=0.1, seed=5))
print('per-op linear1         %7.2f us' % t)
t = timeit(lambda: ops.linear(dt, h, w2b, bias=b2, drop_p=0.1, seed=6, res=x1, ldr=x1.stride(0)))
print('per-op linear2         %7.2f us' % t)
t = timeit(lambda: ops.attention_fwd(dt, qk[:, :E], qk[:, E:], v, B, H, S, S, None, None, 0.1, 7, None))
print('per-op attention core  %7.2f us' % t)
