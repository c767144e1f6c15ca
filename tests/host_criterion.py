"""TEST INFRASTRUCTURE, not product code: SetCriterion's losses with torch ops on HOST tensors.

The product computes every loss and gradient in one HIP launch (csrc/criterion.hip) and has no CPU implementation:
``SetCriterion.compute`` raises on non-GPU tensors.  The CPU test suite (no GPU in the build container) still has to exercise the HOST
part of the criterion - ``SetCriterion.prepare``: cost matrices, batched C++ Hungarian, fine_tune re-matching, the dense target tables
- against the reference's fixtures (G5, G9, G16), and needs the losses evaluated on those tables to compare with the fixtures'
values.  This module is that evaluator, installed by the tests through the ``criterion.host_compute`` hook:

    crit.host_compute = host_criterion.compute_host

It mirrors reference sedt/sedt.py:134-352 on the dense tables of ``prepare`` (loss names, weights, normalisation).  The GPU tests compare
the fused kernel with the fixtures and the oracle directly, never with this file."""
import torch
import torch.nn.functional as F

from sound_event_detection_transformer_amd.sedt.sedt import ALPHA_FL, GAMMA_FL


def compute_host(self, outputs, dense, fl=False):
    """(criterion, outputs, dense, fl) -> loss dict; sets criterion.last_total like the fused path"""
    layers = [outputs] + list(outputs.get('aux_outputs', []))
    L, ns, nb = dense['L'], dense['ns'], dense['num_boxes']
    if nb is None:
        nb = dense['wbox'][0].sum()
    C1 = self.num_classes + 1
    logits_all = torch.stack([o['pred_logits'] for o in layers]).float()          # [L,B,Q,C+1]
    boxes = torch.stack([o['pred_boxes'][:ns] for o in layers]).float()           # [L,ns,Q,2]
    logits = logits_all[:, :ns]
    out = {}
    vec = {}
    tc = dense['tc'].long()
    if 'labels' in self.losses:
        if fl:
            onehot = F.one_hot(tc, C1).float()
            p = logits.sigmoid()
            ce = F.binary_cross_entropy_with_logits(logits, onehot, pos_weight=self.empty_weight.to(logits.device), reduction='none')
            ce = ce * (1 - (p * onehot + (1 - p) * (1 - onehot))) ** GAMMA_FL
            if ALPHA_FL >= 0:
                ce = ce * (ALPHA_FL * onehot + (1 - ALPHA_FL) * (1 - onehot))
            ce = ce.sum(-1).view(L, -1)
        else:
            ce = F.cross_entropy(logits.reshape(-1, C1), tc.reshape(-1), self.empty_weight.to(logits.device),
                                 reduction='none').view(L, -1)
        vec['loss_ce'] = (ce * dense['coef'].view(L, -1)).sum(1) / nb
        with torch.no_grad():
            m = (dense['wbox'][0] > 0)
            hit = ((logits[0].argmax(-1) == tc[0]) & m).float().sum()
            out['class_error'] = 100 - 100 * hit / m.float().sum().clamp(min=1)
    if 'boxes' in self.losses:
        s1, e1 = boxes[..., 0] - boxes[..., 1] / 2, boxes[..., 0] + boxes[..., 1] / 2
        t = dense['tbox']
        s2, e2 = t[..., 0] - t[..., 1] / 2, t[..., 0] + t[..., 1] / 2
        l1 = (s1 - s2).abs() + (e1 - e2).abs()
        inter = (torch.min(e1, e2) - torch.max(s1, s2)).clamp(min=0)
        union = (e1 - s1) + (e2 - s2) - inter
        hull = (torch.max(e1, e2) - torch.min(s1, s2)).clamp(min=0)
        giou = inter / union - (hull - union) / hull
        w = dense['wbox']
        vec['loss_bbox'] = (l1 * w).view(L, -1).sum(1) / nb
        vec['loss_giou'] = ((1 - giou) * w).view(L, -1).sum(1) / nb
    if 'cardinality' in self.losses:
        with torch.no_grad():
            card = (logits_all.argmax(-1) != C1 - 1).sum(2).float()              # [L,B]
            vec['cardinality_error'] = (card - dense['tgt_len'][None]).abs().mean(1)
    if 'feature' in self.losses:
        feats = torch.stack([o['pred_feature'][:ns] for o in layers]).float()     # [L,ns,Q,F]
        gt = outputs['gt_feature'].float()
        gt = gt.view(ns, gt.shape[0] // ns, -1)
        tgt = gt[torch.arange(ns, device=gt.device)[None, :, None], dense['tidx'].long()]   # [L,ns,Q,F]
        mse = (F.normalize(feats, dim=-1) - F.normalize(tgt, dim=-1)).square().sum(-1)
        vec['loss_feature'] = (mse * (dense['wbox'] > 0).float()).view(L, -1).sum(1) / nb
    for k, v in vec.items():
        for li in range(L):
            out[k if li == 0 else f'{k}_{li - 1}'] = v[li]
    if 'weak' in self.losses and 'at' in outputs:
        pw, gw = outputs['at'][:dense['n_lab']].float(), dense['gt_weak']
        if fl:
            ce = F.binary_cross_entropy(pw, gw, reduction='none') * (1 - (pw * gw + (1 - pw) * (1 - gw))) ** GAMMA_FL
            if ALPHA_FL >= 0:
                ce = ce * (ALPHA_FL * gw + (1 - ALPHA_FL) * (1 - gw))
            out['loss_weak'] = ce.sum(1).mean()
        else:
            out['loss_weak'] = F.binary_cross_entropy(pw, gw)
    at_p = self._pooled(outputs, outputs.get('at'))
    if at_p is not None:
        if dense.get('wp_all', False) and at_p.shape[0] != dense['n_lab']:
            raise ValueError('loss_weak_p with weak_mask=None needs every clip labelled (reference sedt.py:184: BCELoss rejects the shapes)')
        r0 = 0 if dense.get('wp_all', False) else ns
        out['loss_weak_p'] = F.binary_cross_entropy(at_p[r0:dense['n_lab']].float(), dense['gt_weak'][r0:])
    wd = self.weight_dict
    total = None
    for k, v in vec.items():
        wts = [wd.get(k if li == 0 else f'{k}_{li - 1}', 0.0) for li in range(L)]
        if any(wts):
            term = (v * torch.tensor(wts, device=v.device, dtype=v.dtype)).sum()
            total = term if total is None else total + term
    for k in ('loss_weak', 'loss_weak_p'):
        if k in out and wd.get(k, 0.0):
            total = out[k] * wd[k] + (total if total is not None else 0.0)
    self.last_total = total
    return out
