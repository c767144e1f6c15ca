"""graph vs eager mean-teacher step with mix-up: per-step losses, pseudo-label counts and mixing decisions (diagnostic)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden'))
from sound_event_detection_transformer_amd import runtime, sedt
from sound_event_detection_transformer_amd.engine import semi_train_step, GraphedSemiStep
import test_mixup_steps_gpu as T

dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
runtime.set_compute_dtype(dt)
ns, nw, nu = 5, 5, 6
masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
thr = torch.full((10,), 0.115).cuda()
batches = [T._rand_semi_batch(600 + i, ns, nw, nu) for i in range(4)]
for mode in ('eager', 'graph'):
    model, crit, ema, opt = T._mix_semi_model(sedt, 2023, perturb=False)
    with torch.no_grad():
        for n in ema.shadow:
            ema.shadow[n].mul_(1.01)
    if mode == 'graph':
        stepper = GraphedSemiStep(model, ema, crit, opt, batches[0][0], batches[0][1], batches[0][2], classwise_threshold=thr, mix_up_ratio=0.6, **masks)
    np.random.seed(3)
    for xt, xs, tg in batches:
        if mode == 'eager':
            sup, unsup, total, pseudo = semi_train_step(model, ema, crit, opt, xt, xs, T._cuda_targets(tg), classwise_threshold=thr, mix_up_ratio=0.6, **masks)
            print(mode, float(total), 'sup', float(sup['loss_ce']), float(sup['loss_weak']), 'unsup', float(unsup['loss_ce']), float(unsup['loss_bbox']),
                  'nlab', [len(t['labels']) for t in pseudo], 'ratio', ['ratio' in t for t in pseudo])
        else:
            total, sup, unsup = stepper(xt, xs, tg)
            torch.cuda.synchronize()
            lo = stepper.tab_u.as_dict()['lab_off'].cpu().numpy()
            lp = stepper.tab_p.as_dict()['lab_off'].cpu().numpy()
            print(mode, float(total), 'sup', float(sup['loss_ce']), float(sup['loss_weak']), 'unsup', float(unsup['loss_ce']), float(unsup['loss_bbox']),
                  'nlab', np.diff(lo).tolist(), 'pseudo', np.diff(lp).tolist(), 'modes', stepper.jobs_u.cpu().numpy().view(np.int32).reshape(-1, 4)[:, 2].tolist(),
                  'split', stepper.tab_l.cur_ns, stepper.tab_l.cur_n_lab)
