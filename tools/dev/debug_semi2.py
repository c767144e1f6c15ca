import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from sound_event_detection_transformer_amd import runtime, sedt, ops
from sound_event_detection_transformer_amd.engine import GraphedSemiStep, semi_train_step, pseudo_label_tables
import test_steps_gpu as T
runtime.set_compute_dtype('bf16')
masks = dict(mask_strong=slice(2), mask_weak=slice(2, 4), mask_label=slice(4), mask_unlabel=slice(4, 8))
thr = torch.full((10,), 0.115).cuda()
xt, xs, tg = T._rand_semi_batch(500, 2, 2, 4)
model, crit, ema, opt = T._semi_model(sedt)
from sound_event_detection_transformer_amd.sedt import TargetTables


def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    keep = []
    for k in range(9, 28):
        for _ in range(6 if k < 24 else 2):
            keep.append(torch.full(((1 << k) // 2,), float('nan'), dtype=torch.bfloat16, device='cuda'))
    torch.cuda.synchronize()
    del keep


tab_l = TargetTables(4, 2, 4, xt.device, max_targets=32).load(tg[:4])
tab_u = TargetTables(4, 4, 4, xt.device, max_targets=20)
counter = torch.zeros(10, dtype=torch.int32, device='cuda')
which = sys.argv[1]
for it in range(3):
    poison()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        total = 0
        if which in ('both', 'lab'):
            out_l = model(xt[:4])
            crit.compute(out_l, crit.prepare_device(out_l, tab_l))
            total = total + crit.last_total
        if which in ('both', 'unl', 'unl_noteacher'):
            if which != 'unl_noteacher':
                ema.apply_shadow()
                with torch.no_grad():
                    tea = model(xt[4:])
                ema.restore()
            else:
                with torch.no_grad():
                    tea = model(xt[4:])
            pseudo_label_tables(tea, thr, 10.0, tab_u, counter)
            out_s = model(xs[4:])
            crit.compute(out_s, crit.prepare_device(out_s, tab_u))
            total = total + crit.last_total
        total.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.requires_grad and not torch.isfinite(p.grad).all()]
    print(which, 'iter', it, 'loss', float(total), 'bad grads', len(bad), bad[:4], bad[-2:])
    opt.zero_grad(set_to_none=True)
