"""GPU: the one-launch SetCriterion kernel (csrc/criterion.hip) against the reference's losses (golden G5) and, for the
gradients, against the oracle criterion differentiated by autograd on the CPU.  Tolerance: 1e-5 relative on every loss
value (f32 sums of ~2000 terms), 1e-5 absolute + 1e-4 relative on gradients."""
import os

import numpy as np
import pytest
import torch

from oracle.criterion_oracle import build_oracle_criterion, synthetic_targets

pytestmark = pytest.mark.gpu


def _crit():
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    return build_model(default_args())[1].cuda()


def _fixed(B=6, Q=10, seed=55, tseed=56):
    gen = torch.Generator().manual_seed(seed)
    outputs = {'pred_logits': torch.randn(B, Q, 11, generator=gen), 'pred_boxes': torch.rand(B, Q, 2, generator=gen) * 0.8 + 0.1,
               'at': torch.rand(B, 10, generator=gen),
               'aux_outputs': [{'pred_logits': torch.randn(B, Q, 11, generator=gen),
                                'pred_boxes': torch.rand(B, Q, 2, generator=gen) * 0.8 + 0.1} for _ in range(2)]}
    return outputs, synthetic_targets(B, tseed, 10), B


def _stacked_gpu(outputs):
    """what SEDT.forward hands over: the heads applied to all decoder layers at once, main layer last"""
    la = torch.stack([a['pred_logits'] for a in outputs['aux_outputs']] + [outputs['pred_logits']]).cuda().requires_grad_(True)
    ba = torch.stack([a['pred_boxes'] for a in outputs['aux_outputs']] + [outputs['pred_boxes']]).cuda().requires_grad_(True)
    at = outputs['at'].cuda().requires_grad_(True)
    o = {'pred_logits': la[-1], 'pred_boxes': ba[-1], 'at': at,
         'aux_outputs': [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(la[:-1], ba[:-1])], '_stacked': (la, ba)}
    return o, la, ba, at


def _cuda_targets(targets):
    return [{k: v.cuda() for k, v in t.items()} for t in targets]


def test_fused_losses_match_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_criterion.npz'))
    crit = _crit()
    outputs, targets, B = _fixed()
    o, *_ = _stacked_gpu(outputs)
    ld, _ = crit(o, _cuda_targets(targets), None, slice(B))
    assert set(ld) == {k[5:] for k in g.files if k.startswith('loss_')}
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'loss_{k}'])) < 1e-5 * max(1.0, abs(v.item())), k
    # weak + strong split: clips 4,5 carry tags only
    t2 = [dict(t) for t in targets]
    for t in t2[4:]:
        t['boxes'] = torch.zeros(0, 2)
    o, *_ = _stacked_gpu(outputs)
    ld, _ = crit(o, _cuda_targets(t2), slice(4, 6), slice(4))
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'ws_loss_{k}'])) < 1e-5 * max(1.0, abs(v.item())), k


@pytest.mark.parametrize('case', ['strong', 'weak_strong', 'entries', 'big'])
def test_fused_gradients_match_oracle(case):
    crit, oc = _crit(), build_oracle_criterion()
    if case == 'big':
        outputs, targets, B = _fixed(B=64, Q=10, seed=3, tseed=4)
    else:
        outputs, targets, B = _fixed()
    ns = B
    wm = None
    if case == 'weak_strong':
        ns, wm = 4, slice(4, 6)
        for t in targets[4:]:
            t['boxes'] = torch.zeros(0, 2)
    # oracle on the CPU
    oo = {k: (v.clone().requires_grad_(True) if torch.is_tensor(v) else
              [{kk: vv.clone().requires_grad_(True) for kk, vv in a.items()} for a in v]) for k, v in outputs.items()}
    ldo, _ = oc(oo, targets, wm, slice(ns))
    # product: fused kernels
    o, la, ba, at = _stacked_gpu(outputs)
    ld, _ = crit(o, _cuda_targets(targets), wm, slice(ns))
    if case == 'entries':
        # a caller's own weighting of single entries (not the weight_dict total) must differentiate too
        pick = {'loss_ce': 0.3, 'loss_giou_1': 1.7, 'loss_bbox_0': 0.9, 'loss_weak': 0.2}
        sum(ldo[k] * w for k, w in pick.items()).backward()
        sum(ld[k] * w for k, w in pick.items()).backward()
    else:
        to = sum(ldo[k] * oc.weight_dict[k] for k in ldo if k in oc.weight_dict)
        to.backward()
        tp = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        assert abs(tp.item() - to.item()) < 1e-5 * abs(to.item())
        assert abs(crit.last_total.item() - to.item()) < 1e-5 * abs(to.item())
        tp.backward()
    def gz(t):
        return t.grad if t.grad is not None else torch.zeros_like(t)
    ref_l = torch.stack([gz(a['pred_logits']) for a in oo['aux_outputs']] + [gz(oo['pred_logits'])])
    ref_b = torch.stack([gz(a['pred_boxes']) for a in oo['aux_outputs']] + [gz(oo['pred_boxes'])])
    for got, ref in ((la.grad, ref_l), (ba.grad, ref_b), (at.grad, gz(oo['at']))):
        got = got.cpu()
        assert (got - ref).abs().max().item() < 1e-5 + 1e-4 * ref.abs().max().item()


def test_total_only_backward_equals_entrywise():
    """engine.train_step differentiates criterion.last_total; a caller summing entries must get the same gradients"""
    crit = _crit()
    outputs, targets, B = _fixed()
    o, la, ba, at = _stacked_gpu(outputs)
    crit(o, _cuda_targets(targets), None, slice(B))
    crit.last_total.backward()
    o2, la2, ba2, at2 = _stacked_gpu(outputs)
    ld, _ = crit(o2, _cuda_targets(targets), None, slice(B))
    sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict).backward()
    for a, b in ((la, la2), (ba, ba2), (at, at2)):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-8)
