"""GPU: the criterion variants (fine_tune re-matching, normalize, focal loss, positional mixup ratios), PostProcess, the
pseudo-label kernel and the SP-SEDT feature loss on the HIP path - against the REFERENCE's results (fixtures G9-G11) and,
for gradients, against the oracle differentiated by autograd on the CPU.  Index work (matching, labels, orders, counters)
must be exact; loss values 2e-5 relative; gradients 1e-5 absolute + 1e-4 relative."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import inputs as GI                                                                    # noqa: E402
from oracle.criterion_oracle import build_oracle_criterion, PostProcess as OraclePost  # noqa: E402

pytestmark = pytest.mark.gpu

G9_CASES = {'ft': (True, False, False, 1.0), 'ft_eps3': (True, False, False, 3.0), 'ft_norm_eps3': (True, True, False, 3.0),
            'fl': (False, False, True, 1.0), 'fl_ft_eps3': (True, False, True, 3.0)}


def _rows(a):
    return [r[r >= 0] for r in a]


def _crit(eps=1.0):
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    c = build_model(default_args())[1].cuda()
    c.matcher.epsilon = eps
    return c


def _stacked_gpu(outputs, grad=True):
    la = torch.stack([a['pred_logits'] for a in outputs['aux_outputs']] + [outputs['pred_logits']]).cuda().requires_grad_(grad)
    ba = torch.stack([a['pred_boxes'] for a in outputs['aux_outputs']] + [outputs['pred_boxes']]).cuda().requires_grad_(grad)
    at = outputs['at'].cuda().requires_grad_(grad)
    o = {'pred_logits': la[-1], 'pred_boxes': ba[-1], 'at': at,
         'aux_outputs': [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(la[:-1], ba[:-1])], '_stacked': (la, ba)}
    return o, la, ba, at


def _cuda_targets(targets):
    return [{k: v.cuda() for k, v in t.items()} for t in targets]


def _check_losses(ld, g, prefix):
    assert set(ld) == {k[len(prefix):] for k in g.files if k.startswith(prefix)}
    for k, v in ld.items():
        assert abs(v.item() - float(g[prefix + k])) <= 2e-5 * max(1.0, abs(v.item())), (k, v.item(), float(g[prefix + k]))


@pytest.mark.parametrize('name', list(G9_CASES))
def test_g9_host_matching_fused_losses(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'g9_criterion_variants.npz'))
    ft, norm, fl, eps = G9_CASES[name]
    crit = _crit(eps)
    outputs, targets, B, Q = GI.g9_inputs()
    o, *_ = _stacked_gpu(outputs)
    ld, idx = crit(o, _cuda_targets(targets), None, slice(B), ft, norm, fl, ft_rand=_rows(g[f'{name}_rand']) if ft else None)
    for b, (i, j) in enumerate(idx):
        np.testing.assert_array_equal(i.numpy(), _rows(g[f'{name}_src'])[b])
        np.testing.assert_array_equal(j.numpy(), _rows(g[f'{name}_tgt'])[b])
    _check_losses(ld, g, f'{name}_loss_')


@pytest.mark.parametrize('name', list(G9_CASES))
def test_g9_device_matching(golden_dir, name):
    """the same variants with the matching solved on the device (what a graphed step runs)"""
    from sound_event_detection_transformer_amd.sedt import TargetTables
    g = np.load(os.path.join(golden_dir, 'g9_criterion_variants.npz'))
    ft, norm, fl, eps = G9_CASES[name]
    crit = _crit(eps)
    outputs, targets, B, Q = GI.g9_inputs()
    o, *_ = _stacked_gpu(outputs)
    tables = TargetTables(B, B, B, torch.device('cuda'), max_targets=16).load(_cuda_targets(targets))
    rand = None
    if ft:
        r = np.zeros((B, Q), np.float32)
        for b, row in enumerate(_rows(g[f'{name}_rand'])):
            r[b, :len(row)] = row
        rand = torch.from_numpy(r).cuda()
    assign = torch.full((3, B, Q), -7, dtype=torch.int32, device='cuda')
    dense = crit.prepare_device(o, tables, assign=assign, normalize=norm, fine_tune=ft, fl=fl, ft_rand=rand)
    a = assign[0].cpu().numpy()
    for b in range(B):
        src, tgt = _rows(g[f'{name}_src'])[b].astype(np.int64), _rows(g[f'{name}_tgt'])[b].astype(np.int64)
        want = -np.ones(Q, np.int64)
        want[src] = tgt
        np.testing.assert_array_equal(a[b], want)
    ld = crit.compute(o, dense, fl)
    _check_losses(ld, g, f'{name}_loss_')


def test_fine_tune_hash_rng_changes_per_replay_and_keeps_rate():
    """without injected uniforms the device draws them from a counter hash of (seed, *seed_ptr): fresh per bump, and the
    share of added queries stays near alpha * n_gt / Q"""
    from sound_event_detection_transformer_amd.sedt import TargetTables
    from sound_event_detection_transformer_amd import runtime
    crit = _crit(50.0)                                  # every query is "close": all non-Hungarian queries are candidates
    outputs, targets, B, Q = GI.g9_inputs()
    o, *_ = _stacked_gpu(outputs, grad=False)
    tables = TargetTables(B, B, B, torch.device('cuda'), max_targets=16).load(_cuda_targets(targets))
    seen, rates = set(), []
    for _ in range(6):
        runtime.bump_seed(torch.device('cuda'))
        assign = torch.zeros((3, B, Q), dtype=torch.int32, device='cuda')
        crit.prepare_device(o, tables, assign=assign, fine_tune=True)
        a = assign[0].cpu().numpy()
        seen.add(a.tobytes())
        for b, t in enumerate(targets):
            n = len(t['boxes'])
            if n < Q:
                rates.append((((a[b] >= 0).sum() - n) / (Q - n), n / Q))
    assert len(seen) >= 4
    got, want = np.mean([r[0] for r in rates]), np.mean([r[1] for r in rates])
    assert abs(got - want) < 0.2, (got, want)


def test_g9_focal_weak_split_and_ratio(golden_dir):
    from sound_event_detection_transformer_amd.sedt import TargetTables
    g = np.load(os.path.join(golden_dir, 'g9_criterion_variants.npz'))
    crit = _crit()
    outputs, targets, B, Q = GI.g9_inputs()
    t2 = [dict(t) for t in targets]
    for t in t2[4:]:
        t['boxes'] = torch.zeros(0, 2)
    o, *_ = _stacked_gpu(outputs)
    ld, _ = crit(o, _cuda_targets(t2), slice(4, 6), slice(4), False, False, True)
    _check_losses(ld, g, 'fl_ws_loss_')
    t3 = [dict(t) for t in targets]
    for t, r in zip(t3, _rows(g['ratio_values'])):
        if len(r):
            t['ratio'] = torch.from_numpy(r.copy())
    o, *_ = _stacked_gpu(outputs)
    ld, _ = crit(o, _cuda_targets(t3), None, slice(B))
    _check_losses(ld, g, 'ratio_loss_')
    # device matching with the positional ratios
    tables = TargetTables(B, B, B, torch.device('cuda'), max_targets=16, with_ratio=True).load(_cuda_targets(t3))
    o, *_ = _stacked_gpu(outputs)
    ld = crit.compute(o, crit.prepare_device(o, tables))
    _check_losses(ld, g, 'ratio_loss_')


@pytest.mark.parametrize('case', ['fl', 'fl_weak_strong', 'fl_ft'])
def test_focal_gradients_match_oracle(golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'g9_criterion_variants.npz'))
    outputs, targets, B, Q = GI.g9_inputs()
    ns, wm, ft, eps, rows = B, None, False, 1.0, None
    if case == 'fl_weak_strong':
        ns, wm = 4, slice(4, 6)
        for t in targets[4:]:
            t['boxes'] = torch.zeros(0, 2)
    if case == 'fl_ft':
        ft, eps, rows = True, 3.0, _rows(g['fl_ft_eps3_rand'])
    crit, oc = _crit(eps), build_oracle_criterion(epsilon=eps)
    oo = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) else
              [{kk: vv.clone().requires_grad_(True) for kk, vv in a.items()} for a in v]) for k, v in outputs.items()}
    if ft:
        it = iter(rows)
        oc.matcher.rand = lambda n: torch.from_numpy(np.asarray(next(it)[:n], np.float32))
    ldo, _ = oc(oo, targets, wm, slice(ns), ft, False, True)
    to = sum(ldo[k] * oc.weight_dict[k] for k in ldo if k in oc.weight_dict)
    to.backward()
    o, la, ba, at = _stacked_gpu(outputs)
    crit(o, _cuda_targets(targets), wm, slice(ns), ft, False, True, ft_rand=rows)
    assert abs(crit.last_total.item() - to.item()) <= 2e-5 * abs(to.item())
    crit.last_total.backward()

    def gz(t):
        return t.grad if t.grad is not None else torch.zeros_like(t)
    ref_l = torch.stack([gz(a['pred_logits']) for a in oo['aux_outputs']] + [gz(oo['pred_logits'])])
    ref_b = torch.stack([gz(a['pred_boxes']) for a in oo['aux_outputs']] + [gz(oo['pred_boxes'])])
    for got, ref in ((la.grad, ref_l), (ba.grad, ref_b), (at.grad, gz(oo['at']))):
        got = got.cpu()
        assert (got - ref).abs().max().item() < 1e-5 + 1e-4 * ref.abs().max().item()


def test_nonfinite_flag_is_set_only_for_bad_totals():
    """device path (what a graphed step runs): NaN model outputs must neither stall the on-device matching nor go unnoticed"""
    from sound_event_detection_transformer_amd.sedt import TargetTables
    crit = _crit()
    crit.nonfinite = torch.zeros(1, dtype=torch.int32, device='cuda')
    outputs, targets, B, Q = GI.g9_inputs()
    tables = TargetTables(B, B, B, torch.device('cuda'), max_targets=16).load(_cuda_targets(targets))
    o, *_ = _stacked_gpu(outputs)
    crit.compute(o, crit.prepare_device(o, tables))
    assert crit.nonfinite.item() == 0 and torch.isfinite(crit.last_total).item()
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in outputs.items()}
    bad['pred_logits'][0, 0, 0] = float('nan')
    bad['pred_boxes'][1, 2, 1] = float('inf')
    o, *_ = _stacked_gpu(bad)
    crit.compute(o, crit.prepare_device(o, tables))
    torch.cuda.synchronize()
    assert crit.nonfinite.item() == 1 and not torch.isfinite(crit.last_total).item()
    with pytest.raises(RuntimeError):                 # host matching refuses such costs, like scipy in the reference
        crit(o, _cuda_targets(targets), None, slice(B))


# ------------------------------------------------------------------------------------------------ PostProcess (G10)
@pytest.mark.parametrize('name,kw', [('none', dict(audio_tags=None)), ('m1', dict(at_m=1)), ('m2', dict(at_m=2)), ('m3', dict(at_m=3)),
                                     ('m2_t03', dict(at_m=2, threshold=0.3)), ('semi', dict(at_m=1, is_semi=True, threshold=None))])
def test_g10_postprocess_kernel(golden_dir, name, kw):
    from sound_event_detection_transformer_amd.sedt import PostProcess
    g = np.load(os.path.join(golden_dir, 'g10_postprocess.npz'))
    outputs, tags, sizes = GI.g10_inputs()
    kw = dict(kw)
    kw.setdefault('audio_tags', tags)
    if kw['audio_tags'] is not None:
        kw['audio_tags'] = kw['audio_tags'].cuda()
    r = PostProcess()({k: v.cuda() for k, v in outputs.items()}, sizes.cuda(), **kw)
    assert len(r) == 5 and set(r[0]) == {'scores', 'labels', 'boxes'} and r[0]['labels'].dtype == torch.int64
    np.testing.assert_array_equal(np.stack([x['labels'].cpu().numpy() for x in r]), g[f'{name}_labels'])
    np.testing.assert_allclose(np.stack([x['scores'].cpu().numpy() for x in r]), g[f'{name}_scores'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(np.stack([x['boxes'].cpu().numpy() for x in r]), g[f'{name}_boxes'], rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------ pseudo labels (G11)
@pytest.mark.parametrize('name,nms', [('nms', True), ('raw', False)])
def test_g11_pseudo_label_kernel(golden_dir, name, nms):
    from sound_event_detection_transformer_amd import ops
    from sound_event_detection_transformer_amd.sedt import TargetTables
    g = np.load(os.path.join(golden_dir, 'g11_pseudo_labels.npz'))
    tea, thr, B = GI.g11_inputs()
    tables = TargetTables(B, B, B, torch.device('cuda'), max_targets=20)
    counter = torch.zeros(10, dtype=torch.int32, device='cuda')
    for rep in range(2):                                                # the counter accumulates over calls
        ops.pseudo_labels(tea['pred_logits'].cuda(), tea['pred_boxes'].cuda(), tea['at'].cuda(), thr.cuda(), 0.2 / 10.0,
                          tables.as_dict(), counter, del_overlap=nms)
    off = tables.off[:B + 1].cpu().numpy()
    np.testing.assert_array_equal(np.diff(off), g[f'{name}_count'])
    np.testing.assert_array_equal(tables.off[B + 1:2 * B + 2].cpu().numpy(), off)
    lab, box = tables.lab_cat.cpu().numpy(), tables.box_cat.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(lab[off[b]:off[b + 1]], _rows(g[f'{name}_labels'])[b].astype(np.int64))
        np.testing.assert_array_equal(box[off[b]:off[b + 1], 0], _rows(g[f'{name}_centre'])[b])       # copied values: exact
        np.testing.assert_array_equal(box[off[b]:off[b + 1], 1], _rows(g[f'{name}_length'])[b])
    np.testing.assert_array_equal(counter.cpu().numpy(), 2 * g[f'{name}_counter'])


def test_pseudo_labels_feed_device_matching():
    """the tables the pseudo-label kernel writes are what match_targets reads: losses equal the host path on the same events"""
    from sound_event_detection_transformer_amd import ops
    from sound_event_detection_transformer_amd.sedt import TargetTables
    tea, thr, B = GI.g11_inputs()
    tables = TargetTables(B, B, B, torch.device('cuda'), max_targets=20)
    ops.pseudo_labels(tea['pred_logits'].cuda(), tea['pred_boxes'].cuda(), tea['at'].cuda(), thr.cuda(), 0.02, tables.as_dict())
    off = tables.off[:B + 1].cpu().numpy()
    targets = [{'labels': tables.lab_cat[off[b]:off[b + 1]].clone(), 'boxes': tables.box_cat[off[b]:off[b + 1]].clone()} for b in range(B)]
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    crit = build_model(default_args(num_queries=20))[1].cuda()
    gen = torch.Generator().manual_seed(4)
    la = torch.randn(3, B, 20, 11, generator=gen).cuda()
    ba = (torch.rand(3, B, 20, 2, generator=gen) * 0.8 + 0.1).cuda()
    o = {'pred_logits': la[-1], 'pred_boxes': ba[-1], 'at': torch.rand(B, 10, generator=gen).cuda(),
         'aux_outputs': [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(la[:-1], ba[:-1])], '_stacked': (la, ba)}
    ld_h, _ = crit(o, targets, None, slice(B))
    th = crit.last_total.item()
    crit.compute(o, crit.prepare_device(o, tables))
    assert abs(crit.last_total.item() - th) <= 1e-6 * abs(th)


# ------------------------------------------------------------------------------------------------ SP-SEDT feature loss
@pytest.mark.parametrize('B,Q,P,F', [(3, 20, 10, 2048), (2, 10, 10, 256)])
def test_feature_loss_and_gradient(B, Q, P, F):
    import torch.nn.functional as Fnn
    from sound_event_detection_transformer_amd.sedt.sedt import SetCriterion, _FeatureLossFn
    L = 3
    gen = torch.Generator().manual_seed(9)
    pred = torch.randn(L, B, Q, F, generator=gen)
    gt = torch.randn(B * P, F, generator=gen)
    wbox = (torch.rand(L, B, Q, generator=gen) > 0.5).float()
    tidx = torch.randint(0, P, (L, B, Q), generator=gen).float()
    nb = wbox[0].sum().clamp(min=1).view(1)
    layer_of = [L - 1] + list(range(L - 1))
    wv = torch.tensor([1.0, 0.5, 2.0])
    # CPU reference (sedt.py:263-283 semantics on the dense targets)
    pc = pred.clone().requires_grad_(True)
    losses = []
    for d in range(L):
        p = pc[layer_of[d]]
        tgt = gt.view(B, P, F)[torch.arange(B)[:, None], tidx[d].long()]
        mse = (Fnn.normalize(p, dim=-1) - Fnn.normalize(tgt, dim=-1)).square().sum(-1)
        losses.append((mse * wbox[d]).sum() / nb[0])
    ref = torch.stack(losses)
    (ref * wv).sum().backward()
    dense = {'wbox': wbox.cuda(), 'tidx': tidx.cuda(), 'ns': B, 'L': L}
    pg = pred.cuda().requires_grad_(True)
    out = _FeatureLossFn.apply(pg, gt.cuda(), dense, layer_of, nb.cuda(), wv.cuda())
    assert torch.allclose(out[:L].cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    assert abs(out[L].item() - (ref * wv).sum().item()) <= 1e-5 * abs((ref * wv).sum().item())
    out[L].backward()
    assert (pg.grad.cpu() - pc.grad).abs().max().item() <= 1e-6 + 1e-4 * pc.grad.abs().max().item()
