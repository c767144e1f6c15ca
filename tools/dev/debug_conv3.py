import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, lib as L
dt = L.BF16
g = torch.Generator().manual_seed(11)
for B, H in ((2, 125), (64, 125), (3, 32)):
    C, W = 64, 16
    gm = ops.ConvGeom(H, W, C, C, 3, 1, 1, 1)
    x = torch.randn(B * H * W, C, generator=g).to('cuda', torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, generator=g) / 24).cuda()
    sc = (torch.rand(C, generator=g) + 0.5).cuda(); bi = (torch.randn(C, generator=g) * 0.1).cuda()
    wf, wb = ops.pack_conv(dt, w, sc)
    msk = torch.randn(B * H * W, C, generator=g).to('cuda', torch.bfloat16)
    outs = {}
    for mode in (True, False):
        ops.CONV3_DIRECT = mode
        y = ops.conv_fwd(dt, x, B, gm, wf, scale=sc, bias=bi, act=L.ACT_RELU)
        dx = ops.conv_dgrad(dt, x, B, gm, wb, mask=msk, ldm=C)
        y2 = ops.conv_fwd(dt, x, B, gm, wf, scale=sc, bias=bi, act=L.ACT_RELU)
        outs[mode] = (y.float(), dx.float(), torch.equal(y, y2))
    ops.CONV3_DIRECT = True
    for k, name in ((0, 'fwd'), (1, 'dgrad')):
        a, b = outs[True][k], outs[False][k]
        d = (a - b).abs()
        bad = (d > 0.05 * b.abs().max()).nonzero()
        print((B, H), name, 'max diff', d.max().item(), 'scale', b.abs().max().item(), 'n bad', len(bad), 'repeat-equal', outs[True][2])
        if len(bad):
            rows = bad[:, 0]
            pix = rows % (H * W)
            print('   bad rows (h, w):', sorted(set((int(p) // W, int(p) % W) for p in pix[:40]))[:20], 'channels', sorted(set(int(c) for c in bad[:40, 1]))[:16])
