import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, lib as L
dt = L.BF16
g = torch.Generator().manual_seed(3)
for (B, H, Lq, Lk, pad) in ((2, 8, 128, 128, 0), (2, 8, 11, 128, 0), (2, 8, 11, 11, 0), (2, 8, 124, 124, 0), (2, 8, 128, 128, 23), (2, 8, 21, 124, 0)):
    E = H * 32
    q = (torch.randn(B * Lq, E, generator=g)).to('cuda', torch.bfloat16)
    k = (torch.randn(B * Lk, E, generator=g)).to('cuda', torch.bfloat16)
    v = (torch.randn(B * Lk, E, generator=g)).to('cuda', torch.bfloat16)
    kpm = None
    if pad:
        kpm = torch.zeros(B, Lk, dtype=torch.uint8, device='cuda'); kpm[:, Lk - pad:] = 1
    o, lse = ops.attention_fwd(dt, q, k, v, B, H, Lq, Lk, kpm, None, 0.0, 7, None)
    o2, lse2 = ops.attention_fwd(dt, q, k, v, B, H, Lq, Lk, kpm, None, 0.0, 7, None)
    qf = q.float().view(B, Lq, H, 32).permute(0, 2, 1, 3); kf = k.float().view(B, Lk, H, 32).permute(0, 2, 1, 3); vf = v.float().view(B, Lk, H, 32).permute(0, 2, 1, 3)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(32)
    if pad: s[:, :, :, Lk - pad:] = float('-inf')
    ref_lse = torch.logsumexp(s, -1)
    ref_o = (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3).reshape(B * Lq, E)
    print((B, H, Lq, Lk, pad), 'o err', (o.float() - ref_o).abs().max().item(), 'lse err', (lse.view(B, H, Lq) - ref_lse).abs().max().item(), 'repeat equal', torch.equal(o, o2), torch.equal(lse, lse2), flush=True)
