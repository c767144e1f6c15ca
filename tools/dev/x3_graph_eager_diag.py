"""diagnostic: in the bf16x3 mode, is an eager step bit-repeatable, and where does a captured step first differ from it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import sedt_oracle as O
from oracle.criterion_oracle import synthetic_targets
from sound_event_detection_transformer_amd import runtime, sedt, ops
from sound_event_detection_transformer_amd.engine import train_step, build_optimizer, GraphedTrainStep


def cu(ts):
    return [{k: v.cuda() for k, v in t.items()} for t in ts]


def run(mode, fast, nsteps=2, accum=1):
    ops.X3_FAST = fast
    runtime.set_compute_dtype('bf16x3')
    B = 2
    batches = [(torch.randn(B, 1, 500, 64, generator=torch.Generator().manual_seed(40 + i)).cuda(), cu(synthetic_targets(B, 50 + i, 10))) for i in range(nsteps)]
    model, crit, _ = sedt.build_model(sedt.default_args(dropout=0.0))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 5))
    model.cuda().train(); crit.cuda()
    opt = build_optimizer(model)
    losses = []
    if mode == 'graph':
        st = GraphedTrainStep(model, crit, opt, batches[0][0], batches[0][1], None, slice(B), warmup=1, accum_steps=accum)
    grads = None
    for i, (x, t) in enumerate(batches):
        if mode == 'graph':
            l, _ = st(x, t)
        else:
            l, _ = train_step(model, crit, opt, x, t, None, slice(B), do_step=(accum == 1 or i % accum == accum - 1))
        losses.append(float(l))
    torch.cuda.synchronize()
    return losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}


def cmp(a, b, tag):
    worst = max(((a[1][k] - b[1][k]).abs().max().item() / (b[1][k].abs().max().item() + 1e-12), k) for k in a[1])
    print(tag, 'losses', a[0], b[0], 'worst param rel', worst)


for fast in (True, False):
    for accum in (1, 2):
        n = 2 * accum
        e1, e2 = run('eager', fast, n, accum), run('eager', fast, n, accum)
        g1 = run('graph', fast, n, accum)
        cmp(e1, e2, f'fast={fast} accum={accum} eager vs eager')
        cmp(g1, e1, f'fast={fast} accum={accum} graph vs eager')
