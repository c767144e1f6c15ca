"""GPU: the --pooling variants of SEDT (reference sedt.py:47-61, 96-119) and loss_weak_p (sedt.py:182-185) on the HIP path:
csrc/pool_at.hip + the at_p branch of the fused criterion kernel.

* kernel level: sedt_pool_at / sedt_pool_at_bwd against a plain torch f32 restatement differentiated by autograd (CPU), all four
  modes, with / without the audio-tag query row, including the corner cases the gradients depend on (ties of the max, a clipped
  weighted sum, attention weights below the clamp);
* criterion level: the fused kernel's loss_weak_p and d/d at_p against the oracle criterion;
* model level: the reference's own outputs, losses and gradients (fixture G16, f32 mode, north_star tolerance 1e-3), the same
  in bf16 at the bf16 mode's stated bound, and a graphed training step against the eager one.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, x3_skips_gradient_elements

sys.path.insert(0, GOLDEN)
import inputs as GI                                                                    # noqa: E402
from oracle import sedt_oracle as O                                                    # noqa: E402
from oracle.criterion_oracle import build_oracle_criterion, synthetic_targets          # noqa: E402

pytestmark = pytest.mark.gpu

MODES = ('max', 'avg', 'attn', 'weighted_sum')


def rel(got, ref):
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


@pytest.fixture(scope='module')
def pkg():
    from sound_event_detection_transformer_amd import runtime, sedt
    assert torch.cuda.is_available()
    return runtime, sedt


def _pool_ref(logits, boxes, attn, mode, q0):
    """sedt.py:96-119 with torch ops (f32, CPU)"""
    y = F.softmax(logits[:, q0:], -1)[:, :, :-1]
    if mode == 'weighted_sum':
        return (y * boxes[:, q0:, 1][:, :, None]).sum(1).clip(0, 1)
    if mode == 'attn':
        sof = torch.clamp(F.softmax(attn, -1), min=1e-7, max=1)
        return (sof * y).sum(1) / sof.sum(1)
    if mode == 'max':
        return torch.nn.AdaptiveMaxPool2d((1, None))(y).squeeze(1)
    return torch.nn.AdaptiveAvgPool2d((1, None))(y).squeeze(1)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('q0,Q,B', [(1, 10, 5), (0, 20, 3), (1, 63, 2)])
def test_pool_at_kernels_against_torch(mode, q0, Q, B):
    from sound_event_detection_transformer_amd import ops
    C = 10
    gen = torch.Generator().manual_seed(1000 + 7 * Q + q0)
    logits = torch.randn(B, q0 + Q, C + 1, generator=gen) * 2.0
    boxes = torch.rand(B, q0 + Q, 2, generator=gen) * 0.3
    attn = torch.randn(B, Q, C, generator=gen) * 3.0
    # corner cases: two queries with identical logits that hold the maximum of class 2 (max: the FIRST gets the gradient) ...
    logits[0, q0 + 3] = logits[0, q0 + 1]
    logits[0, q0 + 1, 2] = logits[0, q0 + 3, 2] = 9.0
    # ... a clip whose weighted sum exceeds 1 for class 4 (clip: no gradient there) ...
    logits[1, q0:, 4] = 8.0
    boxes[1, q0:, 1] = 0.9
    # ... and attention logits so peaked that the other classes fall under the 1e-7 clamp (no gradient through the clamp)
    attn[0, 0, :] = -30.0
    attn[0, 0, 5] = 30.0
    g = torch.randn(B, C, generator=gen)
    lr, br, ar = logits.clone().requires_grad_(True), boxes.clone().requires_grad_(True), attn.clone().requires_grad_(True)
    ref = _pool_ref(lr, br, ar, mode, q0)
    ref.backward(g)
    dl, db, da = logits.cuda(), boxes.cuda(), attn.cuda()
    out = ops.pool_at(dl, db if mode == 'weighted_sum' else None, da if mode == 'attn' else None, mode, q0, Q)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=2e-6, atol=1e-7)
    gl, gb, ga = ops.pool_at_bwd(dl, db if mode == 'weighted_sum' else None, da if mode == 'attn' else None, mode, q0, Q, g.cuda())
    torch.testing.assert_close(gl.cpu(), lr.grad, rtol=2e-5, atol=2e-7)
    if q0:
        assert gl[:, 0].abs().max().item() == 0.0                 # the audio-tag query takes no part
    if mode == 'weighted_sum':
        torch.testing.assert_close(gb.cpu(), br.grad, rtol=2e-5, atol=2e-7)
        assert gb[1, q0:, 1].abs().max().item() > 0 and float(out[1, 4]) == 1.0
    else:
        assert gb is None
    if mode == 'attn':
        torch.testing.assert_close(ga.cpu(), ar.grad, rtol=2e-5, atol=2e-7)
    else:
        assert ga is None
    if mode == 'max':                                            # tie: all of class 2's gradient of clip 0 went through query 1
        assert lr.grad[0, q0 + 3].abs().max().item() == 0.0 or Q == 0
        assert gl[0, q0 + 3].abs().max().item() == 0.0


def test_pool_at_rejects_bad_arguments():
    from sound_event_detection_transformer_amd import ops
    lg = torch.randn(2, 11, 11).cuda()
    with pytest.raises(AssertionError):
        ops.pool_at(lg, None, None, 'weighted_sum', 1, 10)          # boxes missing
    with pytest.raises(RuntimeError, match='query window'):
        ops.pool_at(lg, None, None, 'max', 2, 10)                   # window beyond the rows
    with pytest.raises(RuntimeError, match='4096'):
        ops.pool_at(torch.randn(1, 64, 65).cuda(), None, None, 'avg', 0, 64)


def _stacked_gpu(outputs):
    la = torch.stack([a['pred_logits'] for a in outputs['aux_outputs']] + [outputs['pred_logits']]).cuda().requires_grad_(True)
    ba = torch.stack([a['pred_boxes'] for a in outputs['aux_outputs']] + [outputs['pred_boxes']]).cuda().requires_grad_(True)
    at = outputs['at'].cuda().requires_grad_(True)
    at_p = outputs['at_p'].cuda().requires_grad_(True)
    o = {'pred_logits': la[-1], 'pred_boxes': ba[-1], 'at': at, 'at_p': at_p,
         'aux_outputs': [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(la[:-1], ba[:-1])], '_stacked': (la, ba)}
    return o, la, at, at_p


@pytest.mark.parametrize('split', ['weak', 'nomask', 'device_weak', 'device_nomask'])
def test_loss_weak_p_fused_kernel_against_oracle(pkg, split):
    """loss_weak_p and its gradient from the fused criterion launch: host matching (prepare) and device matching (TargetTables)"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.sedt.sedt import TargetTables
    crit = sedt.build_model(sedt.default_args(pooling='avg', weak_loss_p_coef=0.7))[1].cuda()
    oc = build_oracle_criterion(pooling='avg', weak_loss_p_coef=0.7)
    outputs, targets, B, Q = GI.g9_inputs()
    outputs['at_p'] = torch.rand(B, 10, generator=torch.Generator().manual_seed(58)) * 0.98 + 0.01
    outputs['at_p'][5, 3] = 0.0                                  # BCE's log clamp (-100) and the gradient's 1e-12 floor
    ns = 4 if 'weak' in split else B
    wm = slice(ns, B) if 'weak' in split else None
    tg = [dict(t) for t in targets]
    for t in tg[ns:]:
        t['boxes'] = torch.zeros(0, 2)
    o_ref = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) else v) for k, v in outputs.items()}
    ld_ref, _ = oc(o_ref, tg, wm, slice(ns))
    tot_ref = sum(ld_ref[k] * oc.weight_dict[k] for k in ld_ref if k in oc.weight_dict)
    tot_ref.backward()
    o, la, at, at_p = _stacked_gpu(outputs)
    if split.startswith('device'):
        tables = TargetTables(B, ns, B, torch.device('cuda'), max_targets=16, weak_mask_none=wm is None).load(tg)
        dense = crit.prepare_device(o, tables)
        ld = crit.compute(o, dense)
    else:
        ld, _ = crit(o, [{k: v.cuda() for k, v in t.items()} for t in tg], wm, slice(ns))
    assert set(ld) == set(ld_ref)
    for k in ld_ref:
        assert abs(ld[k].item() - ld_ref[k].item()) <= 2e-5 * max(1.0, abs(ld_ref[k].item())), k
    crit.last_total.backward()
    assert abs(crit.last_total.item() - tot_ref.item()) <= 2e-5 * abs(tot_ref.item())
    torch.testing.assert_close(at_p.grad.cpu(), o_ref['at_p'].grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(at.grad.cpu(), o_ref['at'].grad, rtol=1e-4, atol=1e-6)
    r0 = ns if wm is not None else 0
    assert at_p.grad[:r0].abs().max().item() == 0.0 if r0 else True
    # a single entry of the loss dict differentiates on its own too (a caller's own weighting)
    o2, _, _, at_p2 = _stacked_gpu(outputs)
    ld2, _ = crit(o2, [{k: v.cuda() for k, v in t.items()} for t in tg], wm, slice(ns))
    (3.0 * ld2['loss_weak_p']).backward()
    torch.testing.assert_close(at_p2.grad.cpu(), o_ref['at_p'].grad * (3.0 / 0.7), rtol=1e-4, atol=1e-6)


def _build(sedt, pooling, seed, **kw):
    model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=3, num_queries=10, dropout=0.0, pooling=pooling, weak_loss_p_coef=0.7, **kw))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), seed))
    return model.cuda(), crit.cuda()


@pytest.mark.parametrize('mode', MODES + ('max_nomask',))
def test_g16_pooling_model_f32(pkg, golden_dir, mode):
    """the reference's at_p, losses (incl. loss_weak_p) and gradients for every --pooling variant; f32 mode, 1e-3"""
    runtime, sedt = pkg
    runtime.set_compute_dtype('f32')
    g = np.load(os.path.join(golden_dir, 'g16_pooling.npz'))
    c = GI.POOL
    ns, B = c['n_strong'], c['n_strong'] + c['n_weak']
    nomask = mode.endswith('_nomask')
    pooling = mode.split('_nomask')[0]
    model, crit = _build(sedt, pooling, c['seed_w'] + (0 if nomask else c['modes'].index(mode)))
    sd = model.state_dict()
    assert ('attn_dense_softmax.weight' in sd) == (pooling == 'attn')
    x, targets = GI.pool_batch()
    if nomask:
        x, targets = x[:ns], targets[:ns]
    model.eval()
    with torch.no_grad():
        o = model(x.cuda())
    assert o['at_p'].shape == g[f'{mode}_eval_at_p'].shape
    assert rel(o['at_p'], g[f'{mode}_eval_at_p']) < 1e-3 and rel(o['at'], g[f'{mode}_eval_at']) < 1e-3
    model.train()
    o = model(x.cuda())
    ld, _ = crit(o, targets, None if nomask else slice(ns, B), slice(ns))
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    total.backward()
    assert rel(o['at_p'].detach(), g[f'{mode}_train_at_p']) < 1e-3
    assert abs(total.item() - float(g[f'{mode}_train_total'])) < 1e-3 * abs(float(g[f'{mode}_train_total']))
    assert abs(crit.last_total.item() - total.item()) < 1e-5 * abs(total.item())
    assert set(ld) == {k[len(mode) + 12:] for k in g.files if k.startswith(f'{mode}_train_loss_')}
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'{mode}_train_loss_{k}'])) < 1e-3 * max(1.0, abs(v.item())), k
    x3_skips_gradient_elements()
    params = dict(model.named_parameters())
    names = [str(n) for n in g[f'{mode}_train_gradnames']]
    assert names == [n for n, p in model.named_parameters() if p.requires_grad]
    gn = np.array([params[n].grad.norm().item() for n in names], dtype=np.float32)
    bad = [(n, a, b) for n, a, b in zip(names, gn, g[f'{mode}_train_gradnorm']) if abs(a - b) > 2e-3 * b + 1e-6]
    assert not bad, bad[:10]
    for key in g.files:
        if key.startswith(f'{mode}_train_grad::'):
            n = key.split('::')[1]
            r = torch.from_numpy(g[key])
            gr = params[n].grad.detach().float().cpu().flatten()
            idx = torch.linspace(0, gr.numel() - 1, 32).long()
            got = torch.cat([gr.mean()[None], gr.abs().mean()[None], gr[idx]])
            assert rel(got[1:], r[1:]) < 2e-3, n


@pytest.mark.parametrize('mode', MODES)
def test_pooling_model_bf16_and_graphed_step(pkg, golden_dir, mode):
    """bf16 mode: at_p / loss_weak_p within the bf16 mode's stated bound of the reference's values (5e-2, as the other outputs); the
    whole training step with a --pooling model captured as one HIP graph follows the eager step"""
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import train_step, build_optimizer, GraphedTrainStep
    runtime.set_compute_dtype('bf16')
    try:
        g = np.load(os.path.join(golden_dir, 'g16_pooling.npz'))
        c = GI.POOL
        ns, B = c['n_strong'], c['n_strong'] + c['n_weak']
        x, targets = GI.pool_batch()
        x = x.cuda()
        t = [{k: v.cuda() for k, v in tt.items()} for tt in targets]
        model, crit = _build(sedt, mode, c['seed_w'] + c['modes'].index(mode))
        model.eval()
        with torch.no_grad():
            o = model(x)
        assert rel(o['at_p'], g[f'{mode}_eval_at_p']) < 5e-2
        res = {}
        for how in ('graph', 'eager'):
            model, crit = _build(sedt, mode, c['seed_w'] + c['modes'].index(mode))
            model.train()
            opt = build_optimizer(model)
            if how == 'graph':
                stepper = GraphedTrainStep(model, crit, opt, x, t, slice(ns, B), slice(ns), warmup=1)
            losses = []
            for i in range(3):
                if how == 'graph':
                    out = stepper(x, t)
                else:
                    opt.zero_grad(set_to_none=True)
                    out = train_step(model, crit, opt, x, t, slice(ns, B), slice(ns))
                losses.append(float(out[0].detach()))
                assert 'loss_weak_p' in out[1]
            res[how] = (losses, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
        assert abs(res['graph'][0][0] - float(g[f'{mode}_train_total'])) < 5e-2 * float(g[f'{mode}_train_total'])
        for a, b in zip(res['graph'][0], res['eager'][0]):
            assert abs(a - b) < 5e-3 * abs(b), (res['graph'][0], res['eager'][0])
        for k in res['eager'][1]:
            assert rel(res['graph'][1][k], res['eager'][1][k]) < 5e-3, k
    finally:
        runtime.set_compute_dtype('f32')


def test_mean_teacher_step_with_pooling_against_oracle(pkg):
    """the semi-supervised iteration (reference engine.py:117-181) of a --pooling model: the labelled loss takes loss_weak_p over
    the weak clips, the pseudo-label loss over all unlabelled clips (criterion called with weak_mask None, engine.py:159); f32
    mode against the oracle's iteration on the CPU (total 1e-3, every gradient norm 2e-3), then the one-graph step against the
    eager one"""
    from collections import Counter
    from oracle import semi_oracle as S
    runtime, sedt = pkg
    from sound_event_detection_transformer_amd.engine import semi_train_step, GraphedSemiStep, build_optimizer
    from sound_event_detection_transformer_amd.utilities.utils import EMA
    runtime.set_compute_dtype('f32')
    c = GI.SEMI
    ns, nw, nu = c['n_strong'], c['n_weak'], c['n_unl']
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    x_t, x_s, targets = GI.semi_batch()
    thr = torch.full((10,), c['thr'])
    # ---- oracle
    om = O.build_oracle_model(10, 20, 6, 3, True, True, True, dropout=0.0, pooling='max')
    om.load_state_dict(O.seeded_state_dict(om.state_dict(), c['seed_w']))
    om.train()
    oc = build_oracle_criterion(10, 3, True, True, pooling='max', weak_loss_p_coef=0.7)
    oema = S.EMA(om, 0.9)
    oema.register()
    gen = torch.Generator().manual_seed(5)
    for n in oema.shadow:
        oema.shadow[n] = oema.shadow[n] + 0.02 * oema.shadow[n].abs().mean() * torch.randn(oema.shadow[n].shape, generator=gen)
    shadow = {n: v.clone() for n, v in oema.shadow.items()}
    sup_r, unsup_r, total_r, pseudo_r = S.semi_step(om, oema, oc, None, x_t, x_s, targets, classwise_threshold=thr, do_step=False,
                                                    counter=Counter(), **masks)
    assert 'loss_weak_p' in sup_r and 'loss_weak_p' in unsup_r
    assert sum(len(t['labels']) for t in pseudo_r) > 0
    gn_r = {n: p.grad.norm().item() for n, p in om.named_parameters() if p.requires_grad}

    def hip_model():
        model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=6, num_queries=20, dropout=0.0, pooling='max', weak_loss_p_coef=0.7))
        model.load_state_dict(O.seeded_state_dict(model.state_dict(), c['seed_w']))
        model.cuda().train()
        ema = EMA(model, 0.9)
        ema.register()
        for n in ema.shadow:
            ema.shadow[n].copy_(shadow[n])
        return model, crit.cuda(), ema, build_optimizer(model)
    cuda_t = [{k: v.cuda() for k, v in t.items()} for t in targets]
    model, crit, ema, opt = hip_model()
    sup, unsup, total, pseudo = semi_train_step(model, ema, crit, opt, x_t.cuda(), x_s.cuda(), cuda_t, classwise_threshold=thr.cuda(),
                                                counter=Counter(), do_step=False, do_ema=False, **masks)
    assert [len(t['labels']) for t in pseudo] == [len(t['labels']) for t in pseudo_r]
    assert abs(total.item() - total_r.item()) < 1e-3 * abs(total_r.item())
    for k in ('loss_weak_p', 'loss_weak'):
        assert abs(sup[k].item() - sup_r[k].item()) < 1e-3 * max(1.0, abs(sup_r[k].item())), k
        assert abs(unsup[k].item() - unsup_r[k].item()) < 1e-3 * max(1.0, abs(unsup_r[k].item())), k
    tol = 2e-2 if runtime.compute_mode() == 'bf16x3' else 2e-3      # (--x3: conftest.x3_skips_gradient_elements says why; the half below still runs)
    bad = [(n, p.grad.norm().item(), gn_r[n]) for n, p in model.named_parameters()
           if p.requires_grad and abs(p.grad.norm().item() - gn_r[n]) > tol * gn_r[n] + 1e-6]
    assert not bad, bad[:10]
    # ---- one graph == eager (bf16, two batches)
    runtime.set_compute_dtype('bf16')
    try:
        res = {}
        for how in ('eager', 'graph'):
            model, crit, ema, opt = hip_model()
            if how == 'graph':
                stepper = GraphedSemiStep(model, ema, crit, opt, x_t.cuda(), x_s.cuda(), cuda_t, classwise_threshold=thr.cuda(), **masks)
            tot = []
            for i in range(2):
                if how == 'eager':
                    tot.append(float(semi_train_step(model, ema, crit, opt, x_t.cuda(), x_s.cuda(), cuda_t,
                                                     classwise_threshold=thr.cuda(), **masks)[2].detach()))
                else:
                    tot.append(float(stepper(x_t.cuda(), x_s.cuda(), cuda_t)[0].detach()))
            res[how] = (tot, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
        np.testing.assert_allclose(res['graph'][0], res['eager'][0], rtol=2e-3)
        for k in res['eager'][1]:
            assert rel(res['graph'][1][k], res['eager'][1][k]) < 5e-3, k
    finally:
        runtime.set_compute_dtype('f32')


@pytest.mark.parametrize('mode', ['max', 'avg', 'attn'])
def test_pooling_without_the_audio_tag_query_against_oracle(pkg, mode):
    """dec_at=False (sedt.py:108-119): all queries are event queries, at_p is an output only (no 'weak' loss consumes it), a single
    clip squeezes to [C] as the reference's .squeeze() does; f32 mode against the oracle"""
    runtime, sedt = pkg
    runtime.set_compute_dtype('f32')
    oracle = O.build_oracle_model(10, 10, 3, 3, False, True, True, dropout=0.0, pooling=mode).eval()
    sd = O.seeded_state_dict(oracle.state_dict(), 1700)
    oracle.load_state_dict(sd)
    model, crit, _ = sedt.build_model(sedt.default_args(enc_layers=3, num_queries=10, dropout=0.0, dec_at=False, pooling=mode))
    model.load_state_dict(sd)
    model.cuda().eval()
    assert 'weak' not in crit.losses and 'loss_weak_p' in crit.weight_dict          # sedt/__init__.py:41-45
    for B in (3, 1):
        x = torch.randn(B, 1, 248, 64, generator=torch.Generator().manual_seed(1701 + B))
        with torch.no_grad():
            ref = oracle(x)
            got = model(x.cuda())
        assert 'at' not in got and got['at_p'].shape == ref['at_p'].shape
        assert rel(got['at_p'], ref['at_p']) < 1e-3 and rel(got['pred_logits'], ref['pred_logits']) < 1e-3
    # a training step: the criterion leaves at_p alone (no gradient reaches the attention-pooling layer, as in the reference)
    model.train()
    x = torch.randn(2, 1, 248, 64, generator=torch.Generator().manual_seed(1705)).cuda()
    ld, _ = crit(model(x), [{k: v.cuda() for k, v in t.items()} for t in synthetic_targets(2, 1706, 10)], None, slice(2))
    assert 'loss_weak_p' not in ld and 'loss_weak' not in ld
    crit.last_total.backward()
    if mode == 'attn':
        assert model.attn_dense_softmax.weight.grad is None
