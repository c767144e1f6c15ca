"""round 6 diagnostic: the C4-size SP-SEDT gradient test in f32 and bf16 - per-tensor cosines against the oracle on the sampled clips"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
import test_bench_size_parity_gpu as T
from sound_event_detection_transformer_amd import lib, ops, runtime, sedt
torch.set_num_threads(16)
B, P = int(os.environ.get('B', 200)), 10
pick = [0, B // 3, B - 1]
gen = torch.Generator().manual_seed(77)
x = torch.randn(B, 1, 496, 64, generator=gen)
patches = torch.randn(B, P, 1, 128, 64, generator=gen)
qm = (torch.rand(20, B, 1, generator=gen) > 0.1).float()
mask = torch.zeros(B, 496, 64, dtype=torch.bool)
oracle, model = T._sp_pair(sedt, 4040)
ro = oracle((x[pick], mask[pick]), patches[pick], query_mask=qm[:, pick])
T._sp_loss(ro, slice(None)).backward()
po = dict(oracle.named_parameters())
for mode in ('f32', 'bf16'):
    runtime.set_compute_dtype(mode)
    model.zero_grad(set_to_none=True)
    o = model((x.cuda(), mask.cuda()), patches.cuda(), query_mask=qm.cuda())
    T._sp_loss(o, pick).backward()
    torch.cuda.synchronize()
    rows = []
    for n, p in model.named_parameters():
        if p.grad is None or po[n].grad is None:
            continue
        a, b = p.grad.double().flatten().cpu(), po[n].grad.double().flatten()
        rows.append((float((a * b).sum() / (a.norm() * b.norm() + 1e-300)), n, float(a.norm()), float(b.norm())))
    rows.sort()
    print(mode, 'B', B, 'lowest cosines:')
    for r in rows[:8]:
        print('   %.6f %-60s |hip| %.4e |ref| %.4e' % r)
runtime.set_compute_dtype('f32')
