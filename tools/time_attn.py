import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import ops
def tm(fn, it=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
B, H = 64, 8
for Lq, Lk in ((128, 128), (11, 128), (11, 11)):
    q = torch.randn(B * Lq, 256, device='cuda').bfloat16(); k = torch.randn(B * Lk, 256, device='cuda').bfloat16(); v = torch.randn_like(k)
    do = torch.randn_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(k)
    o, lse = ops.attention_fwd(1, q, k, v, B, H, Lq, Lk, drop_p=0.1, seed=1)
    print(Lq, Lk, 'fwd us', round(tm(lambda: ops.attention_fwd(1, q, k, v, B, H, Lq, Lk, drop_p=0.1, seed=1)), 1),
          'bwd us', round(tm(lambda: ops.attention_bwd(1, q, k, v, o, do, lse, B, H, Lq, Lk, dq, dk, dv, drop_p=0.1, seed=1)), 1))
