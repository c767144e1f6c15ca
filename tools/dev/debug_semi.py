import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from sound_event_detection_transformer_amd import runtime, sedt
from sound_event_detection_transformer_amd.engine import GraphedSemiStep, semi_train_step
import test_steps_gpu as T
runtime.set_compute_dtype('bf16')
masks = dict(mask_strong=slice(2), mask_weak=slice(2, 4), mask_label=slice(4), mask_unlabel=slice(4, 8))
thr = torch.full((10,), 0.115).cuda()
batches = [T._rand_semi_batch(500 + i, 2, 2, 4) for i in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else 'nozero'
model, crit, ema, opt = T._semi_model(sedt)
if mode == 'nozero':
    opt.zero_grad = lambda set_to_none=True: None
    def pre_hook():
        for p in model.parameters():
            p.grad = None
stepper = GraphedSemiStep(model, ema, crit, opt, batches[0][0], batches[0][1], batches[0][2], classwise_threshold=thr, **masks)
xt, xs, tg = batches[0]
total, sup, unsup = stepper(xt, xs, tg)
torch.cuda.synchronize()
print('total', float(total), 'sumsq', opt._sumsq.item())
bad = [(n, tuple(p.shape)) for n, p in model.named_parameters() if p.requires_grad and (p.grad is None or not torch.isfinite(p.grad).all())]
print('bad grads', len(bad), bad[:12])
print('none grads', sum(1 for p in model.parameters() if p.requires_grad and p.grad is None))
good = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is not None and torch.isfinite(p.grad).all()]
print('good grads', len(good), good[:40])
for n, p in list(model.named_parameters()):
    if p.requires_grad and p.grad is not None and not torch.isfinite(p.grad).all():
        f = (~torch.isfinite(p.grad)).float().mean().item()
        print(n, tuple(p.shape), 'nonfinite frac %.4f' % f, 'nan' if torch.isnan(p.grad).any() else 'inf')
        if n.endswith('norm2.bias'):
            break
