"""match_targets / set_criterion / skinny head kernels timed back to back (warm instruction cache, captured graph) against what
they cost inside the step (each runs once per replay, between unrelated kernels).  usage: python tools/dev/time_criterion.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
from sound_event_detection_transformer_amd import ops, runtime, lib as L      # noqa: E402
from sound_event_detection_transformer_amd.sedt import build_model, default_args, TargetTables   # noqa: E402
from sound_event_detection_transformer_amd.utilities.synthetic import synthetic_targets           # noqa: E402

dev = torch.device('cuda')
runtime.set_compute_dtype('bf16')
_, crit, _ = build_model(default_args())
crit.to(dev)
Lh, B, Q, C1 = 3, 64, 10, 11
g = torch.Generator().manual_seed(0)
logits = (torch.randn(Lh, B, Q + 1, C1, generator=g) * 2).to(dev)
boxes = (torch.rand(Lh, B, Q + 1, 2, generator=g) * 0.5 + 0.2).to(dev)
at = torch.rand(B, 10, generator=g).to(dev)
tables = TargetTables(B, B, B, dev).load(synthetic_targets(B, 1, 10))
out = {'pred_logits': logits[-1, :, 1:], 'pred_boxes': boxes[-1, :, 1:], 'at': at, '_stacked': (logits, boxes), '_q0': 1,
       'aux_outputs': [{'pred_logits': logits[i, :, 1:], 'pred_boxes': boxes[i, :, 1:]} for i in range(Lh - 1)]}
pack = torch.empty(crit.dense_numel((Lh, B, Q, B, 10, B)), device=dev)
x = torch.randn(Lh * B * (Q + 1), 256, device=dev).bfloat16()
wc, bc = torch.randn(11, 256, device=dev), torch.randn(11, device=dev)
big = torch.randn(64 * 1024 * 1024, device=dev)


def timeit(fn, reps, spoil=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
            if spoil:
                big.mul_(1.0)                      # 256 MB through L2 and an unrelated kernel's code between two calls
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / reps * 1e3


spoil_us = None
for name, fn in (('match_targets', lambda: crit.prepare_device(out, tables, pack=pack)),
                 ('set_criterion', lambda: crit.compute(out, crit.dense_views(pack, (Lh, B, Q, B, 10, B)))),
                 ('skinny_fwd 2112x256->11', lambda: ops.skinny_linear_fwd(L.BF16, x, wc, bc, 0, True))):
    warm = timeit(fn, 20)
    if spoil_us is None:
        spoil_us = timeit(lambda: None, 20, spoil=True)
    cold = timeit(fn, 20, spoil=True) - spoil_us
    print(f'{name:28s} back to back {warm:6.1f} us   between unrelated work {cold:6.1f} us')
