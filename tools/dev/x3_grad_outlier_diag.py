"""Developer diagnostic (round 5): in the bf16x3 mode one weight-gradient tensor of tests/test_gradient_parity_gpu.py (urban geometry)
sits at 5.5e-3 of its largest entry from the oracle (f32 mode: 2.3e-3).  Where is the difference - spread over the tensor (a precision
problem of the kernel) or concentrated in a few input channels (one ReLU decision of the layer below falling the other way)?"""
import sys

import torch

sys.path.insert(0, '.')
from oracle import sedt_oracle as O                                                     # noqa: E402
from oracle.criterion_oracle import build_oracle_criterion, synthetic_targets           # noqa: E402
from sound_event_detection_transformer_amd import runtime, sedt                         # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'backbone.0.body.layer4.2.conv1.weight'
B, T, E, Q, D = 2, 500, 3, 10, 3
x = torch.randn(B, 1, T, 64, generator=torch.Generator().manual_seed(21))
targets = synthetic_targets(B, 22, 10)
oracle = O.build_oracle_model(10, Q, E, D, True, True, True, dropout=0.0).train()
sd = O.seeded_state_dict(oracle.state_dict(), 23)
oracle.load_state_dict(sd)
crit_o = build_oracle_criterion(10, D, True, True)
ld, _ = crit_o(oracle(x), targets, None, slice(B))
sum(ld[k] * crit_o.weight_dict[k] for k in ld if k in crit_o.weight_dict).backward()
ref = dict(oracle.named_parameters())[name].grad.double()
for mode in ('f32', 'bf16x3'):
    runtime.set_compute_dtype(mode)
    model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=E, dec_layers=D, num_queries=Q, dropout=0.0))
    model.load_state_dict(sd)
    model.cuda().train()
    crit.cuda()
    crit(model(x.cuda()), [{k: v.cuda() for k, v in t.items()} for t in targets], None, slice(B))
    crit.last_total.backward()
    g = dict(model.named_parameters())[name].grad.double().cpu()
    err = (g - ref).abs() / ref.abs().max()
    flat = err.flatten()
    top = flat.topk(8)
    print(f'[{mode}] {name}: max-rel {flat.max():.3e}, elements above 1e-3: {(flat > 1e-3).sum().item()} of {flat.numel()}, '
          f'above 1e-4: {(flat > 1e-4).sum().item()}, median {flat.median():.2e}')
    idx = [tuple(int(v) for v in torch.unravel_index(i, err.shape)) for i in top.indices]
    print('   top elements (co, ci, kh, kw):', idx)
    per_ci = err.amax(dim=(0, 2, 3))
    print('   input channels holding errors above 1e-3:', (per_ci > 1e-3).nonzero().flatten().tolist()[:20])
runtime.set_compute_dtype('f32')
