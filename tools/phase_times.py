#!/usr/bin/env python3
"""Wall-time breakdown of one C2 training step by phase (syncs between phases; GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import runtime
from sound_event_detection_transformer_amd.sedt import build_model, default_args
from sound_event_detection_transformer_amd.engine import build_optimizer
from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict
from bench import synthetic_batch

dev = torch.device('cuda', 0)
runtime.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
model, crit, _ = build_model(default_args(dropout=0.1))
model.load_state_dict(seeded_state_dict(model.state_dict(), 2020))
model.to(dev).train(); crit.to(dev)
opt = build_optimizer(model)
B = 64
x, targets = synthetic_batch(B, 500, 2020, dev)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append(time.perf_counter() - t0); return time.perf_counter()
for it in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    o = model(x); t = tick('fwd', t)
    ld, _ = crit(o, targets, None, slice(B)); t = tick('criterion', t)
    loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict); v = loss.item(); t = tick('loss_sum', t)
    loss.backward(); t = tick('bwd', t)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1); t = tick('clip', t)
    opt.step(); t = tick('adamw', t)
    opt.zero_grad(set_to_none=True); t = tick('zero', t)
for k, v in T.items():
    print(f'{k:10s} {1e3 * sum(v[3:]) / len(v[3:]):8.3f} ms')
# python-side issue time of fwd (no sync): how long the host needs to enqueue the forward
torch.cuda.synchronize(); t0 = time.perf_counter(); o = model(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'fwd enqueue {1e3*(t1-t0):.3f} ms, total {1e3*(t2-t0):.3f} ms')
