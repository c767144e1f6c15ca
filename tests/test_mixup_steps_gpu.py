"""GPU: mix-up INSIDE the training steps (reference engine.py:50-53, 128-133, 150-153; train_sedt.py / train_ss_sedt.py
--mix_up_ratio 0.6): the eager steps against fixture G15 (one engine.semi_train / engine.train iteration of the reference itself,
np.random seeded), the captured steps against the eager ones, the device label merge against the host one, and the strong | weak
split as table data against fixed-split tables.

Tolerances: f32 parity mode - total loss 1e-3, gradient norms 2e-3, AdamW deltas 2e-2 (as G3/G7/G12); label work exact."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, x3_skips_gradient_elements

sys.path.insert(0, GOLDEN)
import inputs as GI                                                                    # noqa: E402
from oracle import sedt_oracle as O                                                    # noqa: E402
from oracle import semi_oracle as S                                                    # noqa: E402
from oracle.criterion_oracle import synthetic_targets                                  # noqa: E402

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def _rows(a):
    return [r[r >= 0] for r in a]


@pytest.fixture(scope='module')
def pkg():
    from sound_event_detection_transformer_amd import runtime, sedt
    assert torch.cuda.is_available()
    return runtime, sedt


def _cuda_targets(targets):
    return [{k: v.cuda() for k, v in t.items()} for t in targets]


def _check_rows(g, key, targets, rtol=1e-6):
    np.testing.assert_array_equal([len(t['labels']) for t in targets], g[f'{key}_nlabels'])
    np.testing.assert_array_equal([len(t['boxes']) for t in targets], g[f'{key}_nboxes'])
    for b, t in enumerate(targets):
        np.testing.assert_array_equal(t['labels'].cpu().numpy(), _rows(g[f'{key}_labels'])[b])
        np.testing.assert_allclose(t['boxes'].reshape(-1, 2)[:, 0].cpu().numpy(), _rows(g[f'{key}_centre'])[b], rtol=rtol)
        np.testing.assert_allclose(t['boxes'].reshape(-1, 2)[:, 1].cpu().numpy(), _rows(g[f'{key}_length'])[b], rtol=rtol)
        r = t['ratio'].cpu().numpy() if 'ratio' in t else np.zeros(0, np.float32)
        np.testing.assert_allclose(r, _rows(g[f'{key}_ratio'])[b], rtol=1e-6)


def _mix_semi_model(sedt, seed, dropout=0.0, decay=0.9, perturb=True):
    from sound_event_detection_transformer_amd.engine import build_optimizer
    from sound_event_detection_transformer_amd.utilities.utils import EMA
    model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=6, num_queries=20, dropout=dropout))
    sd = O.seeded_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    model.cuda().train()
    crit.cuda()
    ema = EMA(model, decay)
    ema.register()
    if perturb:                                           # fixture G12 / G15's teacher: the student + seeded noise, in the
        om = O.build_oracle_model(10, 20, 6, 3, True, True, True, dropout=0.0)      # reference's parameter order
        om.load_state_dict(O.seeded_state_dict(om.state_dict(), seed))
        oe = S.EMA(om, decay)
        oe.register()
        gen = torch.Generator().manual_seed(5)
        for n in oe.shadow:
            oe.shadow[n] = oe.shadow[n] + 0.02 * oe.shadow[n].abs().mean() * torch.randn(oe.shadow[n].shape, generator=gen)
        for n in ema.shadow:
            ema.shadow[n].copy_(oe.shadow[n])
    return model, crit, ema, build_optimizer(model)


def test_g15_mean_teacher_step_with_mixup_f32(pkg, golden_dir):
    """one semi_train iteration WITH mix-up on the HIP path == the reference's (fixture G15): the mixed pseudo targets, the total
    loss, every gradient norm; then the complete iteration: AdamW and EMA deltas"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import semi_train_step
    runtime.set_compute_dtype('f32')
    g = np.load(os.path.join(golden_dir, 'g15_mixup_steps.npz'))
    c = GI.SEMI_MIX
    ns, nw, nu = c['n_strong'], c['n_weak'], c['n_unl']
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    thr = torch.full((10,), c['thr']).cuda()
    x_t, x_s, targets = GI.semi_mix_batch()
    for mode in ('grads', 'step'):
        model, crit, ema, opt = _mix_semi_model(sedt, c['seed_w'])
        before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
        shadow0 = {n: v.clone() for n, v in ema.shadow.items()}
        np.random.seed(c['np_seed'])
        sup, unsup, total, pseudo = semi_train_step(model, ema, crit, opt, x_t.cuda(), x_s.cuda(), _cuda_targets(targets),
                                                    classwise_threshold=thr, do_step=(mode == 'step'), do_ema=(mode == 'step'),
                                                    mix_up_ratio=c['ratio'], **masks)
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        params = dict(model.named_parameters())
        if mode == 'grads':
            _check_rows(g, 'semi_lu', pseudo, rtol=1e-3)              # pseudo boxes come from the f32-mode teacher: 1e-3
            assert abs(total.item() - float(g['semi_total'])) < 1e-3 * abs(float(g['semi_total']))
            assert names == [str(n) for n in g['semi_gradnames']]
            x3_skips_gradient_elements()
            gn = np.array([params[n].grad.norm().item() for n in names], np.float32)
            bad = [(n, a, b) for n, a, b in zip(names, gn, g['semi_gradnorm']) if abs(a - b) > 2e-3 * b + 1e-6]
            assert not bad, bad[:10]
        else:
            assert abs(total.item() - float(g['semi_step_total'])) < 1e-3 * abs(float(g['semi_step_total']))
            delta = np.array([(params[n].detach() - before[n]).norm().item() for n in names], np.float32)
            np.testing.assert_allclose(delta, g['semi_step_delta'], rtol=2e-2, atol=1e-7)
            ed = np.array([(ema.shadow[n] - shadow0[n]).norm().item() for n in names], np.float32)
            np.testing.assert_allclose(ed, g['semi_ema_delta'], rtol=2e-3, atol=1e-7)


def _sup_model(sedt, seed, dropout=0.0):
    from sound_event_detection_transformer_amd.engine import build_optimizer
    model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=3, num_queries=20, dropout=dropout))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), seed))
    model.cuda().train()
    crit.cuda()
    return model, crit, build_optimizer(model)


def test_g15_supervised_step_with_mixup_f32(pkg, golden_dir):
    """engine.train's body with mix_up_ratio = 0.6 (a weak clip moves into the strong part: split 5|5 -> 6|4): mixed batch, total
    loss and every gradient norm against the reference"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import train_step
    from sound_event_detection_transformer_amd.utilities.mixup import mixup_data
    runtime.set_compute_dtype('f32')
    g = np.load(os.path.join(golden_dir, 'g15_mixup_steps.npz'))
    c = GI.SUP_MIX
    ns, nw = c['n_strong'], c['n_weak']
    x, targets = GI.sup_mix_batch()
    np.random.seed(c['np_seed'])
    xm, ym, ms, mw = mixup_data(x.cuda(), _cuda_targets(targets), slice(ns), slice(ns, ns + nw), c['ratio'], alpha=1)
    np.testing.assert_array_equal([ms.stop, mw.start, mw.stop], g['sup_md_split'])
    _check_rows(g, 'sup_md', ym)
    for i in range(ns + nw):
        t = xm[i].float().flatten().cpu()
        idx = torch.linspace(0, t.numel() - 1, 16).long()
        np.testing.assert_allclose(t[idx].numpy(), g['sup_md_x_digest'][i][2:], rtol=1e-5, atol=1e-6)
    model, crit, opt = _sup_model(sedt, c['seed_w'])
    np.random.seed(c['np_seed'])
    total, _ = train_step(model, crit, opt, x.cuda(), _cuda_targets(targets), slice(ns, ns + nw), slice(ns), mix_up_ratio=c['ratio'],
                          do_step=False)
    assert abs(total.item() - float(g['sup_total'])) < 1e-3 * abs(float(g['sup_total']))
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == [str(n) for n in g['sup_gradnames']]
    x3_skips_gradient_elements()
    gn = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
    # conv0.weight is ONE scalar (Conv2d(1, 1, 1)): its "norm" is the absolute value of a single cancelling sum over all 10 x 496
    # x 64 input positions, 3e-3 off in f32 where every real tensor is within 2e-3
    bad = [(n, a, b) for n, a, b in zip(names, gn, g['sup_gradnorm']) if abs(a - b) > (5e-3 if n.endswith('conv0.weight') else 2e-3) * b + 1e-6]
    assert not bad, bad[:10]


# ------------------------------------------------------------------------------------------------ device pieces
def _sparse(B, seed, n_strong):
    t = GI.sparse_targets(B, seed)
    for tt in t[n_strong:]:
        tt['boxes'] = torch.zeros(0, 2)
    return t


@pytest.mark.parametrize('case', ['plain', 'crowded', 'weak_first'])
def test_device_label_merge_matches_host_mixup_label_unlabel(case):
    """sedt_mixup_targets (label half of mixup_label_unlabel on the device) == utilities.mixup.plan_mixup_label_unlabel on the
    same tables: merged labels / boxes / ratios, the per-clip decision (mixed, labelled clip, pseudo clip) and the features"""
    from sound_event_detection_transformer_amd import ops
    from sound_event_detection_transformer_amd.sedt import TargetTables
    from sound_event_detection_transformer_amd.utilities.mixup import plan_mixup_label_unlabel, job_table, lam_pair
    dev = torch.device('cuda')
    n_l, n_u = 12, 10
    ns = {'plain': 8, 'crowded': 8, 'weak_first': 3}[case]
    if case == 'crowded':                                  # more than max_events together -> the pseudo target (or the labelled one)
        y1 = synthetic_targets(n_l, 31, 10)
        for t in y1[ns:]:
            t['boxes'] = torch.zeros(0, 2)
        y2 = synthetic_targets(n_u, 32, 10)
        y2[1]['labels'], y2[1]['boxes'] = torch.zeros(0, dtype=torch.int64), torch.zeros(0, 2)
        max_events = 9
    else:
        y1, y2 = _sparse(n_l, 41, ns), _sparse(n_u, 42, n_u)
        y2[2]['labels'], y2[2]['boxes'] = torch.zeros(0, dtype=torch.int64), torch.zeros(0, 2)
        # a pseudo event of the class of a labelled one, overlapping it in time
        y2[0]['labels'] = torch.cat([y2[0]['labels'], y1[0]['labels'][:1]])
        y2[0]['boxes'] = torch.cat([y2[0]['boxes'], y1[0]['boxes'][:1] + torch.tensor([0.01, 0.0])])
        if case == 'weak_first':                           # box j of the merged list carries label j of the merged LABEL list:
            y1[4]['labels'] = torch.tensor([7, 7])         # two overlapping pseudo events of DIFFERENT classes are read as two
            y2[4]['labels'] = torch.tensor([1, 2])         # events of the weak clip's class 7 -> abandoned (mixup.py:84-93)
            y2[4]['boxes'] = torch.tensor([[0.5, 0.2], [0.52, 0.2]])
        max_events = 20
    for k, t in enumerate(y1):                             # some labelled targets already carry ratios from mixup_data
        if k % 3 == 0:
            t['ratio'] = torch.full((len(t['labels']),), 0.25 + 0.05 * k)
    lam = 0.37
    jobs_h, lab_h = plan_mixup_label_unlabel(y1, y2, lam, n_l, 0.5, max_events)
    tab1 = TargetTables(n_l, ns, n_l, dev, with_ratio=True, dynamic_split=True).load(y1)
    tab2 = TargetTables(n_u, n_u, n_u, dev, max_targets=20).load(y2)
    tabo = TargetTables(n_u, n_u, n_u, dev, with_ratio=True)
    jobs_d = torch.zeros(16 * n_u, dtype=torch.uint8, device=dev)
    lam_d = torch.from_numpy(lam_pair(lam)).to(dev)
    ops.mixup_targets(tab1, tab2, lam_d, int(n_l * 0.5), tabo, jobs_d, max_events)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(jobs_d.cpu().numpy().view(np.int32).reshape(-1, 4)[:, 2], [j[2] for j in jobs_h])
    got_modes = set(j[2] for j in jobs_h[:int(n_l * 0.5)])
    assert {'plain': {0, 1}, 'crowded': {1, 2}, 'weak_first': {0, 1}}[case] <= got_modes, got_modes
    d = tabo.as_dict()
    lo, bo = d['lab_off'].cpu().numpy(), d['box_off'].cpu().numpy()
    for i, t in enumerate(lab_h):
        np.testing.assert_array_equal(d['lab_cat'][lo[i]:lo[i + 1]].cpu().numpy(), t['labels'].numpy())
        np.testing.assert_array_equal(d['box_cat'][bo[i]:bo[i + 1]].cpu().numpy(), t['boxes'].reshape(-1, 2).numpy())
        want = t['ratio'].float().numpy() if 'ratio' in t else np.ones(len(t['labels']), np.float32)
        np.testing.assert_array_equal(d['ratio_cat'][lo[i]:lo[i + 1]].cpu().numpy(), want)
    # and the features the device records produce == the host records'
    g = torch.Generator().manual_seed(7)
    x1, x2 = torch.randn(n_l, 1, 16, 8, generator=g).cuda(), torch.randn(n_u, 1, 16, 8, generator=g).cuda()
    assert torch.equal(ops.mixup(x1, x2, jobs_d), ops.mixup(x1, x2, job_table(jobs_h).cuda()))


def test_split_as_table_data_equals_fixed_split_tables(pkg):
    """TargetTables(dynamic_split=True) + the {ns, n_lab} device words give the same losses and gradients as tables / dense
    buffers built for that split - for two different splits through the SAME buffers"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.sedt import TargetTables
    runtime.set_compute_dtype('f32')
    dev = torch.device('cuda')
    B, Q, L = 10, 20, 3
    _, crit, _ = sedt.build_model(sedt.default_args(enc_layers=3, num_queries=20))
    crit.cuda()
    g = torch.Generator().manual_seed(3)
    dyn = TargetTables(B, 5, 10, dev, with_ratio=True, dynamic_split=True)
    for ns, n_lab in ((5, 10), (7, 9), (3, 3)):
        tg = _sparse(B, 50 + ns, ns)
        for t in tg[n_lab:]:
            t['labels'] = torch.zeros(0, dtype=torch.int64)
        res = []
        for tables in (TargetTables(B, ns, n_lab, dev, with_ratio=True).load(tg), dyn.load(tg, ns=ns, n_lab=n_lab)):
            logits = (torch.randn(L, B, Q, 11, generator=torch.Generator().manual_seed(9)) * 2).cuda().requires_grad_(True)
            boxes = (torch.rand(L, B, Q, 2, generator=torch.Generator().manual_seed(10)) * 0.5 + 0.2).cuda().requires_grad_(True)
            at = torch.rand(B, 10, generator=torch.Generator().manual_seed(11)).cuda().requires_grad_(True)
            out = {'pred_logits': logits[-1], 'pred_boxes': boxes[-1], 'at': at, '_stacked': (logits, boxes),
                   'aux_outputs': [{'pred_logits': logits[i], 'pred_boxes': boxes[i]} for i in range(L - 1)]}
            ld = crit.compute(out, crit.prepare_device(out, tables))
            crit.last_total.backward()
            res.append(({k: float(v) for k, v in ld.items()}, logits.grad.clone(), boxes.grad.clone(), at.grad.clone()))
        assert res[0][0].keys() == res[1][0].keys()
        for k in res[0][0]:
            assert res[0][0][k] == pytest.approx(res[1][0][k], rel=1e-6, abs=1e-7), (ns, n_lab, k)
        for a, b in zip(res[0][1:], res[1][1:]):
            assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ captured steps
def _rand_semi_batch(seed, n_strong, n_weak, n_unl, T=496, cuda_targets=False):
    gen = torch.Generator().manual_seed(seed)
    B = n_strong + n_weak + n_unl
    x_t = torch.randn(B, 1, T, 64, generator=gen)
    x_s = x_t.clone()
    x_s[n_strong + n_weak:] += 0.1 * torch.randn(n_unl, 1, T, 64, generator=gen)
    t = GI.sparse_targets(B, seed + 1)
    for tt in t[n_strong:]:
        tt['boxes'] = torch.zeros(0, 2)
    for tt in t[n_strong + n_weak:]:
        tt['labels'] = torch.zeros(0, dtype=torch.int64)
    return x_t.cuda(), x_s.cuda(), (_cuda_targets(t) if cuda_targets else t)


def test_graphed_semi_step_with_mixup_matches_eager(pkg):
    """GraphedSemiStep(mix_up_ratio=0.6): both mix-ups inside ONE graph (labelled: host plan + feature kernel; unlabelled: label
    merge + feature kernel on the device, fed by the pseudo labels the graph produces) == the eager semi_train_step with the
    host mix-ups, on changing batches whose strong | weak split changes with the draws.  f32 mode: the comparison is about the
    LOGIC (draw order, label merge, split, feature mixing); in bf16 the graph's fused student forward (16 clips in one pass) and
    the eager step's two passes round differently and the criterion's kinks turn that into 1-2 % loss differences after a few
    steps (measured: tools/dev/semi_mix_diag.py), which would hide a real mistake"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import semi_train_step, GraphedSemiStep
    runtime.set_compute_dtype('f32')
    ns, nw, nu = 5, 5, 6
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    thr = torch.full((10,), 0.115).cuda()
    batches = [_rand_semi_batch(600 + i, ns, nw, nu) for i in range(4)]
    res, splits = {}, []
    for mode in ('eager', 'graph'):
        model, crit, ema, opt = _mix_semi_model(sedt, 2023, perturb=False)
        with torch.no_grad():
            for n in ema.shadow:
                ema.shadow[n].mul_(1.01)
        if mode == 'graph':
            stepper = GraphedSemiStep(model, ema, crit, opt, batches[0][0], batches[0][1], batches[0][2], classwise_threshold=thr,
                                      mix_up_ratio=0.6, **masks)
        np.random.seed(3)
        losses = []
        for xt, xs, tg in batches:
            if mode == 'eager':
                _, _, total, pseudo = semi_train_step(model, ema, crit, opt, xt, xs, _cuda_targets(tg), classwise_threshold=thr,
                                                      mix_up_ratio=0.6, **masks)
            else:
                total, _, _ = stepper(xt, xs, tg)
                splits.append((stepper.tab_l.cur_ns, stepper.tab_l.cur_n_lab))
            losses.append(float(total))
        res[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()},
                     {k: v.detach().float().cpu().clone() for k, v in ema.shadow.items()})
    x3 = runtime.compute_mode() == 'bf16x3'      # (the --x3 run: LDS-DMA kernels, whose summation order follows the row count - the fused
    runtime.set_compute_dtype('f32')             # student forward of the graph and the eager step's two passes differ in the last bit)
    assert len(set(splits)) > 1, splits                                   # the split really was data
    np.testing.assert_allclose(res['graph'][0], res['eager'][0], rtol=2e-3 if x3 else 1e-4)
    for k in res['eager'][1]:
        assert rel(res['graph'][1][k], res['eager'][1][k]) < (1e-2 if x3 else 2e-3), k
    for k in res['eager'][2]:
        assert rel(res['graph'][2][k], res['eager'][2][k]) < (1e-2 if x3 else 2e-3), k


def test_graphed_train_step_with_mixup_matches_eager(pkg):
    """GraphedTrainStep(mix_up_ratio=0.6) (engine.py:50-53 inside the captured step) == eager train_step(mix_up_ratio=0.6)"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import train_step, GraphedTrainStep
    runtime.set_compute_dtype('bf16')
    ns, nw = 5, 5
    B = ns + nw
    batches = []
    for i in range(4):
        x = torch.randn(B, 1, 496, 64, generator=torch.Generator().manual_seed(800 + i)).cuda()
        batches.append((x, _sparse(B, 810 + i, ns)))
    res, splits = {}, []
    for mode in ('eager', 'graph'):
        model, crit, opt = _sup_model(sedt, 2024)
        if mode == 'graph':
            stepper = GraphedTrainStep(model, crit, opt, batches[0][0], batches[0][1], slice(ns, B), slice(ns), mix_up_ratio=0.6,
                                       warmup=2)
        np.random.seed(5)
        losses = []
        for x, tg in batches:
            if mode == 'eager':
                l, _ = train_step(model, crit, opt, x, _cuda_targets(tg), slice(ns, B), slice(ns), mix_up_ratio=0.6)
            else:
                l, _ = stepper(x, tg)
                splits.append(stepper.tables.cur_ns)
            losses.append(float(l))
        res[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
    runtime.set_compute_dtype('f32')
    assert len(set(splits)) > 1, splits
    np.testing.assert_allclose(res['graph'][0], res['eager'][0], rtol=2e-3)
    for k in res['eager'][1]:
        assert rel(res['graph'][1][k], res['eager'][1][k]) < 2e-3, k


def test_graphed_train_step_takes_the_eager_step_when_mixup_shrinks_the_batch(pkg):
    """mixup_data without a weak mask (train_sedt.py --mix_up_ratio on URBAN-SED) drops a merged pair of clips that both have no
    events (utilities/mixup.py:104-122): the batch shrinks and the captured shapes do not hold.  The stepper then runs that batch
    through engine.train_step with the np.random stream rewound - same result as the eager loop; batches that keep their size
    are replayed.  Also: a stepper leaves no guard / seed word behind on the shared optimizer."""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import train_step, GraphedTrainStep
    runtime.set_compute_dtype('bf16')
    B = 8
    batches = []
    for i in range(4):
        x = torch.randn(B, 1, 496, 64, generator=torch.Generator().manual_seed(900 + i)).cuda()
        tg = GI.sparse_targets(B, 910 + i)
        if i % 2 == 1:                                    # six clips without events: some mixed pair is empty | empty
            for t in tg[:6]:
                t['labels'], t['boxes'] = torch.zeros(0, dtype=torch.int64), torch.zeros(0, 2)
        batches.append((x, tg))
    res, fallbacks = {}, []
    for mode in ('eager', 'graph'):
        model, crit, opt = _sup_model(sedt, 2025)
        if mode == 'graph':
            stepper = GraphedTrainStep(model, crit, opt, batches[0][0], batches[0][1], None, slice(B), mix_up_ratio=0.6, warmup=2)
            assert opt.guard is None and opt.seed_word is None
            inner = stepper._eager_batch
            stepper._eager_batch = lambda *a, **k: (fallbacks.append(1), inner(*a, **k))[1]
        np.random.seed(7)
        losses = []
        for x, tg in batches:
            if mode == 'eager':
                l, _ = train_step(model, crit, opt, x, _cuda_targets(tg), None, slice(B), mix_up_ratio=0.6)
            else:
                l, _ = stepper(x, tg)
            losses.append(float(l))
        res[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
    runtime.set_compute_dtype('f32')
    assert 1 <= len(fallbacks) <= 2, fallbacks
    np.testing.assert_allclose(res['graph'][0], res['eager'][0], rtol=2e-3)
    for k in res['eager'][1]:
        assert rel(res['graph'][1][k], res['eager'][1][k]) < 2e-3, k


def test_mixup_without_a_weak_mask_is_refused_under_the_flat_gradient_schedule(pkg):
    """a batch that mix-up shrinks cannot bypass the captured data-parallel / accumulation schedule, and finding that out on one rank
    in the middle of an epoch would hang the others in the collective: the combination is refused at construction; an empty
    weak mask makes mixup_data keep the batch size, and that construction goes through"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import GraphedTrainStep
    runtime.set_compute_dtype('bf16')
    try:
        B = 4
        x = torch.randn(B, 1, 496, 64, generator=torch.Generator().manual_seed(1)).cuda()
        tg = GI.sparse_targets(B, 2)
        model, crit, opt = _sup_model(sedt, 2026)
        with pytest.raises(ValueError, match='mask_weak'):
            GraphedTrainStep(model, crit, opt, x, tg, None, slice(B), mix_up_ratio=0.6, accum_steps=2, warmup=1)
        stepper = GraphedTrainStep(model, crit, opt, x, tg, slice(B, B), slice(B), mix_up_ratio=0.6, accum_steps=2, warmup=1)
        np.random.seed(3)
        for _ in range(2):
            stepper(x, tg)
        torch.cuda.synchronize()
        assert int(stepper.nonfinite.item()) == 0
    finally:
        runtime.set_compute_dtype('f32')


def test_c5_full_size_semi_step_with_mixup_properties(pkg):
    """BASELINE config C5 as its recipe runs it (train_ss_sedt.py --mix_up_ratio 0.6 --freq_mask --time_mask): 16 synthetic + 16 weak
    + 32 unlabelled clips, E=6, Q=20, bf16, dropout on.  Per step BOTH views are produced on the device from raw mel amplitudes by
    the reference's transform chain (sedt_box_transform; utilities/BoxTransforms.py:363-427, 454-490: teacher = log + pad + FreqMask
    + normalise, student = the noisy copy + TimeMask too - TimeMask skips view 0, BoxTransforms.py:24-26), then both mix-ups run
    inside the captured step.  Two independently captured steppers fed the same raw batches under the same np.random seed end
    bit-identical; losses finite; the strong | weak split moves between replays; mixed unlabelled clips carry ratios; parameters and
    the EMA teacher move; the student view of a clip equals the CPU oracle's chain on the same drawn parameters."""
    runtime, sedt = pkg
    from oracle import transforms_oracle as TO
    from sound_event_detection_transformer_amd.engine import GraphedSemiStep
    from sound_event_detection_transformer_amd.utilities.synthetic import synthetic_semi_raw, semi_view_transforms, SEMI_SCALER
    runtime.set_compute_dtype('bf16')
    masks = dict(mask_strong=slice(16), mask_weak=slice(16, 32), mask_label=slice(32), mask_unlabel=slice(32, 64))
    thr = torch.full((10,), 0.1).cuda()
    raws = []
    for i in range(3):
        rt, rs = synthetic_semi_raw(32, 32, 496, 1900 + i)
        raws.append((rt.cuda(), rs.cuda(), _rand_semi_batch(1900 + i, 16, 16, 32)[2]))
    tf_t, tf_s = semi_view_transforms(496, 'cuda')
    finals, curves, splits = [], [], []
    for run in range(2):
        runtime.manual_seed(777)
        np.random.seed(11)
        model, crit, ema, opt = _mix_semi_model(sedt, 2023, dropout=0.1, decay=0.9996, perturb=False)
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        xt, xs = tf_t(raws[0][0]), tf_s(raws[0][1])
        stepper = GraphedSemiStep(model, ema, crit, opt, xt, xs, raws[0][2], classwise_threshold=thr, mix_up_ratio=0.6, **masks)
        losses, masked_rows, masked_bands = [], 0, 0
        for it in range(5):
            rt, rs, tg = raws[it % 3]
            p_t = np.stack([tf_t.draw(496) for _ in range(64)])
            p_s = np.stack([tf_s.draw(496) for _ in range(64)])
            tf_t(rt, params=p_t, out=xt)
            tf_s(rs, params=p_s, out=xs)
            if run == 0:
                masked_rows += int((p_s['tm_t'] > 0).sum())
                masked_bands += int(p_s['fm_on'].sum()) + int(p_t['fm_on'].sum())
                assert int(p_t['tm_t'].sum()) == 0                       # the teacher view is never time-masked
                if it == 0:                                              # composition check of one masked student clip vs the oracle
                    b = int(np.argmax((p_s['tm_t'] > 0) & (p_s['fm_on'] > 0))) if ((p_s['tm_t'] > 0) & (p_s['fm_on'] > 0)).any() \
                        else int(np.argmax(p_s['fm_on'] > 0))
                    q = p_s[b]
                    ref = TO.box_transform(rs[b].cpu().numpy().astype(np.float32), 496, np.full(64, SEMI_SCALER[0]),
                                           np.full(64, SEMI_SCALER[1]),
                                           (q['tm_t'] > 0, q['tm_t'] / 496 + 1e-9, q['tm_t0'] / 496 + 1e-9),
                                           (bool(q['fm_on']), q['fm_f'] / 64 + 1e-9, q['fm_f0'] / 64 + 1e-9), None)
                    assert rel(xs[b, 0], np.asarray(ref).reshape(496, 64)) < 5e-5
            total, sup, unsup = stepper(xt, xs, tg, check_finite=True)
            losses.append(float(total))
            if run == 0:
                splits.append((stepper.tab_l.cur_ns, stepper.tab_l.cur_n_lab))
        torch.cuda.synchronize()
        if run == 0:
            assert masked_rows > 0 and masked_bands > 0, (masked_rows, masked_bands)
            modes = stepper.jobs_u.cpu().numpy().view(np.int32).reshape(-1, 4)[:, 2]
            assert set(modes[16:].tolist()) == {2} and len(modes) == 32            # clips beyond mix_num keep their pseudo target
            ratios = stepper.tab_u.ratio_cat[:int(stepper.tab_u.off[32])].cpu()
            assert (modes[:16] == 0).any() and ((ratios - 1.0).abs() > 1e-6).any()      # some clips really were mixed
            assert any(not torch.equal(v, sd0[k]) for k, v in model.state_dict().items() if v.dtype.is_floating_point)
        curves.append(losses)
        finals.append(({k: v.detach().clone() for k, v in model.state_dict().items()}, {k: v.clone() for k, v in ema.shadow.items()}))
    runtime.set_compute_dtype('f32')
    assert np.isfinite(curves).all() and curves[0] == curves[1]
    assert len(set(splits)) > 1 and all(s[1] == 32 for s in splits), splits
    for k in finals[0][0]:
        assert torch.equal(finals[0][0][k], finals[1][0][k]), k
    for k in finals[0][1]:
        assert torch.equal(finals[0][1][k], finals[1][1][k]), k


def test_cached_constants_follow_shape_and_dtype_changes(pkg):
    """round 3 keeps three per-shape constants instead of recomputing them every forward (the decoder's zero input, the resized
    all-False padding mask, its position encoding): a forward must give the same result whatever ran before it - other batch
    sizes, the other compute dtype, an explicit NestedTensor with real padding in between"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.utilities.utils import NestedTensor
    model, _, _ = sedt.build_model(sedt.default_args(dropout=0.0))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 17))
    model.cuda().eval()
    xs = {b: torch.randn(b, 1, 500, 64, generator=torch.Generator().manual_seed(b)).cuda() for b in (2, 3)}

    def run(b, dt):
        runtime.set_compute_dtype(dt)
        with torch.no_grad():
            o = model(xs[b])
        return {k: o[k].float().clone() for k in ('pred_logits', 'pred_boxes', 'at')}
    first = {(b, dt): run(b, dt) for dt in ('f32', 'bf16') for b in (2, 3)}
    # a padded batch in between: its mask is NOT one of the cached all-False masks and must not poison the caches
    m = torch.zeros(2, 500, 64, dtype=torch.bool).cuda()
    m[1, 300:] = True
    runtime.set_compute_dtype('f32')
    with torch.no_grad():
        padded = model(NestedTensor(xs[2], m))
    assert (padded['pred_logits'][1] - first[(2, 'f32')]['pred_logits'][1]).abs().max() > 1e-4       # padding changed clip 1
    assert rel(padded['pred_logits'][0], first[(2, 'f32')]['pred_logits'][0]) < 1e-5                   # ... and only clip 1
    for dt in ('bf16', 'f32'):
        for b in (3, 2):
            again = run(b, dt)
            for k, v in again.items():
                assert torch.equal(v, first[(b, dt)][k]), (b, dt, k)
    runtime.set_compute_dtype('f32')


def test_target_tables_host_and_device_targets_agree():
    """TargetTables.load: host-resident targets travel as ONE pinned blob, device-resident ones table by table - same tables"""
    from sound_event_detection_transformer_amd.sedt import TargetTables
    dev = torch.device('cuda')
    B = 7
    tg = synthetic_targets(B, 5, 10)
    for t in tg[4:]:
        t['boxes'] = torch.zeros(0, 2)
    tg[1]['ratio'] = torch.rand(len(tg[1]['labels']))
    a = TargetTables(B, 4, 6, dev, with_ratio=True, dynamic_split=True).load(tg, ns=4, n_lab=6)
    b = TargetTables(B, 4, 6, dev, with_ratio=True, dynamic_split=True).load(_cuda_targets(tg), ns=4, n_lab=6)
    torch.cuda.synchronize()
    nl = int(a.off[B])
    nb = int(a.off[B + 1 + 4])
    assert torch.equal(a.off, b.off) and torch.equal(a.split.cpu(), torch.tensor([4, 6], dtype=torch.int32))
    assert torch.equal(a.lab_cat[:nl], b.lab_cat[:nl]) and torch.equal(a.box_cat[:nb], b.box_cat[:nb])
    assert torch.equal(a.ratio_cat[:nl], b.ratio_cat[:nl])
    # a second load with fewer events leaves no stale offsets behind
    tg2 = synthetic_targets(B, 6, 10)
    a.load(tg2, ns=7, n_lab=7)
    b.load(_cuda_targets(tg2), ns=7, n_lab=7)
    torch.cuda.synchronize()
    assert torch.equal(a.off, b.off) and torch.equal(a.lab_cat[:int(a.off[B])], b.lab_cat[:int(b.off[B])])
