"""whole model, criterion with a FIXED assignment, per-parameter gradient norms: direct 3x3 kernel vs implicit GEMM"""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from sound_event_detection_transformer_amd import runtime, sedt, ops
from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict, synthetic_targets
model, crit, _ = sedt.build_model(sedt.default_args(dropout=0.0))
model.load_state_dict(seeded_state_dict(model.state_dict(), 2020))
model.cuda().train(); crit.cuda()
B = 2
x = torch.randn(B, 1, 500, 64, generator=torch.Generator().manual_seed(7)).cuda()
tg = [{k: v.cuda() for k, v in t.items()} for t in synthetic_targets(B, 99, 10)]
runtime.set_compute_dtype('f32')
with torch.no_grad():
    dense, _ = crit.prepare(model(x), tg, None, slice(B))
res = {}
for mode in ('f32', 'direct', 'igemm', 'direct2'):
    runtime.set_compute_dtype('f32' if mode == 'f32' else 'bf16')
    ops.CONV3_DIRECT = mode.startswith('direct')
    model.zero_grad(set_to_none=True)
    crit.compute(model(x), dense)
    total = crit.last_total
    total.backward()
    torch.cuda.synchronize()
    res[mode] = (total.item(), {n: p.grad.norm().item() for n, p in model.named_parameters() if p.grad is not None})
names = list(res['f32'][1])
import numpy as np
for mode in ('direct', 'igemm', 'direct2'):
    r = np.array([res[mode][1][n] / (res['f32'][1][n] + 1e-30) for n in names])
    print(mode, 'loss', res[mode][0], 'f32 loss', res['f32'][0], 'ratio of grad norms to f32: median', np.median(r), 'min', r.min(), 'max', r.max())
    worst = np.argsort(-np.abs(r - 1))[:6]
    print('   worst:', [(names[i], round(float(r[i]), 3)) for i in worst])
    for key in ('backbone.0.body.conv0.weight', 'backbone.0.body.layer2.0.conv1.weight', 'backbone.0.body.layer4.2.conv3.weight', 'transformer.encoder.layers.0.linear1.weight', 'class_embed.weight'):
        print('     ', key, round(res[mode][1][key] / res['f32'][1][key], 4))
