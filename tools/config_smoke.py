#!/usr/bin/env python3
"""Train-step smoke of other BASELINE configurations (C3: DCASE geometry, E=6, Q=20, 16 strong + 16 weak clips) through
the eager and the graphed step: finite losses, no crashes.  usage: config_smoke.py [enc dec queries batch strong]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_event_detection_transformer_amd import runtime                                                   # noqa: E402
from sound_event_detection_transformer_amd.sedt import build_model, default_args                            # noqa: E402
from sound_event_detection_transformer_amd.engine import GraphedTrainStep, train_step, build_optimizer      # noqa: E402
from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict, synthetic_targets  # noqa: E402

E, D, Q, B, ns = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (6, 6, 20, 32, 16)
dm = (sys.argv[6] != 'host') if len(sys.argv) > 6 else True
T = int(sys.argv[7]) if len(sys.argv) > 7 else 496
runtime.set_compute_dtype('bf16')
dev = torch.device('cuda')
model, crit, _ = build_model(default_args(enc_layers=E, dec_layers=D, num_queries=Q, dec_at=True, dropout=0.1))
model.load_state_dict(seeded_state_dict(model.state_dict(), 1))
model.to(dev).train()
crit.to(dev)
opt = build_optimizer(model)


def batch(seed):
    x = torch.randn(B, 1, T, 64, generator=torch.Generator().manual_seed(seed)).to(dev)
    t = synthetic_targets(B, seed + 1, crit.num_classes)
    for tt in t[ns:]:
        tt['boxes'] = torch.zeros(0, 2)
    return x, [{k: v.to(dev) for k, v in tt.items()} for tt in t]


wm = slice(ns, B) if ns < B else None
x, t = batch(5)
l, _ = train_step(model, crit, opt, x, t, wm, slice(ns), max_norm=0.1)
print(f'E={E} D={D} Q={Q} B={B} ns={ns}: eager loss', float(l), flush=True)
g = GraphedTrainStep(model, crit, opt, x, t, wm, slice(ns), max_norm=0.1, device_matching=dm)
for i in range(3):
    x, t = batch(10 + i)
    l, _ = g(x, t, check_finite=True)
    print('graph loss', float(l), flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    g(x, t)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('ms/step %.3f  clips/s %.0f' % (dt / 20 * 1e3, B * 20 / dt), flush=True)
