"""find reads of uninitialised memory: poison the caching allocator's free blocks with NaN, run an eager step, look for NaN"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from sound_event_detection_transformer_amd import runtime, sedt
from sound_event_detection_transformer_amd.engine import train_step, build_optimizer
from oracle import sedt_oracle as O
from oracle.criterion_oracle import synthetic_targets
runtime.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
B, ns, T, E, Q = [int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (4, 2, 496, 6, 20))]


def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    keep = []
    for k in range(9, 28):
        for _ in range(6 if k < 24 else 2):
            keep.append(torch.full(((1 << k) // 2,), float('nan'), dtype=torch.bfloat16, device='cuda'))
    torch.cuda.synchronize()
    del keep


model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=E, num_queries=Q, dropout=0.0))
model.load_state_dict(O.seeded_state_dict(model.state_dict(), 5))
model.cuda().train(); crit.cuda()
opt = build_optimizer(model)
x = torch.randn(B, 1, T, 64, generator=torch.Generator().manual_seed(1)).cuda()
t = synthetic_targets(B, 2, 10)
for tt in t[ns:]:
    tt['boxes'] = torch.zeros(0, 2)
t = [{k: v.cuda() for k, v in tt.items()} for tt in t]
wm = slice(ns, B) if ns < B else None
for it in range(3):
    poison()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out = model(x)
        ld, _ = crit(out, t, wm, slice(ns))
        total = crit.last_total
        first_bad = []
        total.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.requires_grad and not torch.isfinite(p.grad).all()]
    print('iter', it, 'loss', float(total), 'bad grads', len(bad), bad[:6], bad[-3:])
    opt.zero_grad(set_to_none=True)
