#!/usr/bin/env python3
"""Time the igemm shapes of one C2 training step under each tile configuration (runs on the GPU box)."""
import json
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import ops  # noqa: E402

BF16 = 1


def time_call(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    B = 64
    res = []
    # (name, kind, params)
    convs = [
        ('l1.conv1 1x1 256->64', 125, 16, 256, 64, 1, 1, 0, 1), ('l1.conv2 3x3 64', 125, 16, 64, 64, 3, 1, 1, 1),
        ('l1.conv3 1x1 64->256', 125, 16, 64, 256, 1, 1, 0, 1), ('l2.conv1 1x1 512->128', 63, 8, 512, 128, 1, 1, 0, 1),
        ('l2.conv2 3x3 128', 63, 8, 128, 128, 3, 1, 1, 1), ('l2.conv3 1x1 128->512', 63, 8, 128, 512, 1, 1, 0, 1),
        ('l3.conv1 1x1 1024->256', 32, 4, 1024, 256, 1, 1, 0, 1), ('l3.conv2 3x3 256', 32, 4, 256, 256, 3, 1, 1, 1),
        ('l3.conv3 1x1 256->1024', 32, 4, 256, 1024, 1, 1, 0, 1), ('l4.conv1 1x1 2048->512', 32, 4, 2048, 512, 1, 1, 0, 1),
        ('l4.conv2 3x3 512 d2', 32, 4, 512, 512, 3, 1, 2, 2), ('l4.conv3 1x1 512->2048', 32, 4, 512, 2048, 1, 1, 0, 1),
        ('ffn1 256->2048', 128, 1, 256, 2048, 1, 1, 0, 1), ('ffn2 2048->256', 128, 1, 2048, 256, 1, 1, 0, 1),
        ('proj 256->256', 128, 1, 256, 256, 1, 1, 0, 1),
    ]
    for name, Hi, Wi, Ci, Co, k, s, pd, dl in convs:
        g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
        x = torch.randn(B * Hi * Wi, Ci, device='cuda').bfloat16()
        w = torch.randn(Co, Ci, k, k, device='cuda') / (Ci * k * k) ** 0.5
        wf, wb = ops.pack_conv(BF16, w)
        gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda').bfloat16()
        y = torch.empty(B * g.Ho * g.Wo, Co, device='cuda', dtype=torch.bfloat16)
        dx = torch.empty_like(x)
        flops = 2.0 * B * g.Ho * g.Wo * Co * Ci * k * k
        row = {'name': name, 'gflop': flops / 1e9}
        for tile in (((128, 128), (128, 64), (64, 128), (64, 64)) if not os.environ.get('SKIP_FWD') else ()):
            t = time_call(lambda: ops.conv_fwd(BF16, x, B, g, wf, out=y, act=1, res=y, ldr=Co, tile=tile))
            row[f'fwd{tile}'] = (round(t, 1), round(flops / t / 1e6, 0))
            t = time_call(lambda: ops.conv_dgrad(BF16, gy, B, g, wb, out=dx, mask=x, ldm=Ci, tile=tile))
            row[f'dgrad{tile}'] = (round(t, 1), round(flops / t / 1e6, 0))
            t = time_call(lambda: ops.conv_fwd(BF16, x, B, g, wf, out=y, tile=tile))
            row[f'plain{tile}'] = (round(t, 1), round(flops / t / 1e6, 0))
        if os.environ.get('SKIP_WGRAD'):
            res.append(row); print(json.dumps(row), flush=True); continue
        # wgrad with split-K as chosen by the library, per tile
        lib = ops.L.load()
        Mo, No, Kp = Co, g.taps * Ci, B * g.Ho * g.Wo
        conv = None if g.plain else ops._geom_tuple(g)
        for tile in ((64, 64),):
            for skm in (0.5, 1, 2):
                sk = max(1, int(lib.sedt_igemm_splitk(Mo, No, Kp, BF16) * skm))
                slab = torch.empty((sk, Mo, No), device='cuda', dtype=torch.float32)
                t = time_call(lambda: ops.igemm(BF16, Mo, No, Kp, gy, Co, x, Ci, slab, No, trans=1, conv=conv, out_f32=1,
                                                splitk=sk, slab=slab, tile=tile))
                row[f'wgrad{tile}sk{sk}'] = (round(t, 1), round(flops / t / 1e6, 0))
        res.append(row)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
