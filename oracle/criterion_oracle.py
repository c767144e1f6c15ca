"""CPU restatement of the host-side loss path (TEST INFRASTRUCTURE).

Restates reference sedt/matcher.py (HungarianMatcher, incl. the fine_tune re-matching :99-121 and the focal
matching cost :73-78), sedt/sedt.py:134-352 (SetCriterion, incl. the focal-loss branches :176,211-218), sedt/sedt.py:412-433
(sigmoid_focal_loss / weak_focal_loss), sedt/sedt.py:355-396 (PostProcess) and utilities/box_ops.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Pinned against the reference through
tests/golden (G5: default flags, G9: fine_tune / normalize / focal loss, G10: PostProcess).
"""
from collections import Counter

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn


# ---- utilities/box_ops.py:9-56: a 1-D interval (centre, length) as the fake box [c-l/2, 0, c+l/2, 1]
def box_cl_to_xyxy(x):
    c, l = x.unbind(-1)
    return torch.stack([c - l / 2, torch.zeros_like(c), c + l / 2, torch.ones_like(c)], dim=-1)


def box_cl_to_se(x):
    c, l = x.unbind(-1)
    return torch.stack([c - l / 2, c + l / 2], dim=-1)


def _area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def generalized_box_iou(b1, b2):
    a1, a2 = _area(b1), _area(b2)
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


ALPHA_FL, GAMMA_FL = 0.5, 1.0          # config.py:71-72


def sigmoid_focal_loss(inputs, targets, weight=None, alpha=ALPHA_FL, gamma=GAMMA_FL):
    """sedt.py:412-422: BCE-with-logits (pos_weight = the class weights) x (1 - p_t)^gamma x alpha_t, summed over classes"""
    p = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, pos_weight=weight, reduction="none")
    pt = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - pt) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.sum(2)


def weak_focal_loss(prob, targets, alpha=ALPHA_FL, gamma=GAMMA_FL):
    """sedt.py:425-433: the same on clip-level probabilities; sum over classes, mean over clips"""
    ce = F.binary_cross_entropy(prob, targets, reduction="none")
    pt = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * (1 - pt) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.sum(1).mean()


class HungarianMatcher(nn.Module):
    """matcher.py:17-133.  ``rand`` is the uniform source of the fine_tune branch (matcher.py:116; tests inject a
    recorded sequence); ``last_rand`` keeps what was drawn per clip."""

    def __init__(self, cost_class=1., cost_bbox=5., cost_giou=2., epsilon=1., alpha=1.):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.epsilon, self.alpha = epsilon, alpha
        self.rand = torch.rand
        self.last_rand = []

    @torch.no_grad()
    def forward(self, outputs, targets, fine_tune=False, normalize=False, fl=False):
        bs, nq = outputs["pred_logits"].shape[:2]
        flat = outputs["pred_logits"].flatten(0, 1)
        out_prob = flat.sigmoid() if fl else flat.softmax(-1)
        out_bbox = outputs["pred_boxes"].flatten(0, 1)
        tgt_ids = torch.cat([v["labels"][:len(v["boxes"])] for v in targets])
        tgt_bbox = torch.cat([v["boxes"] for v in targets])
        if fl:                                     # matcher.py:73-78
            neg = (1 - ALPHA_FL) * out_prob ** GAMMA_FL * (-(1 - out_prob + 1e-8).log())
            pos = ALPHA_FL * (1 - out_prob) ** GAMMA_FL * (-(out_prob + 1e-8).log())
            cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        else:
            cost_class = -out_prob[:, tgt_ids]
        cost_bbox = torch.cdist(box_cl_to_xyxy(out_bbox), box_cl_to_xyxy(tgt_bbox), p=1)
        cost_giou = -generalized_box_iou(box_cl_to_xyxy(out_bbox), box_cl_to_xyxy(tgt_bbox))
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou
        C = C.view(bs, nq, -1).cpu()
        sizes = [len(v["boxes"]) for v in targets]
        idx = []
        for i, c in enumerate(C.split(sizes, -1)):
            r, col = linear_sum_assignment(c[i])
            idx.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(col, dtype=torch.int64)))
        if fine_tune:
            # matcher.py:99-121: localisation-only cost; a query stays matched only if its closest target is nearer than
            # epsilon; further close queries are added (to their closest target) with probability alpha * n_gt / n_queries
            loc = (self.cost_bbox * cost_bbox + self.cost_giou * cost_giou).view(bs, nq, -1).cpu()
            self.last_rand = []
            rematched = []
            for b, (pair, cl) in enumerate(zip(idx, loc.split(sizes, -1))):
                near_cost, near_tgt = cl[b].min(-1)
                src, tgt = pair
                n_gt = len(tgt)
                close = near_cost < self.epsilon
                kept = close[src]
                src, tgt = src[kept], tgt[kept]
                close[src] = False
                extra = torch.where(close)[0]
                u = self.rand(len(extra))
                self.last_rand.append(u.clone())
                close[extra[u > self.alpha * n_gt / nq]] = False
                rematched.append((torch.cat([src, torch.arange(nq)[close]]), torch.cat([tgt, near_tgt[close]])))
            idx = rematched
        coef = []
        for i, (_, tgt) in enumerate(idx):
            if normalize:
                num = Counter(tgt.tolist())
                coef.append(torch.tensor([1 / num[j] for j in tgt.tolist()], dtype=torch.float32))
            elif "ratio" in targets[i]:
                coef.append(targets[i]["ratio"].cpu())
            else:
                coef.append(torch.ones(len(tgt), dtype=torch.float32))
        return idx, coef


class SetCriterion(nn.Module):
    """sedt.py:134-352."""

    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, losses
        w = torch.ones(num_classes + 1)
        w[-1] = eos_coef
        self.register_buffer('empty_weight', w)

    @staticmethod
    def _src_idx(indices):
        b = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        return b, torch.cat([src for (src, _) in indices])

    def loss_weak(self, outputs, targets, indices, num_boxes, strong_mask, weak_mask, coef, fl=False):
        if 'at' not in outputs:
            return {}
        lm = slice(weak_mask.stop) if weak_mask is not None else slice(strong_mask.stop)
        pred = outputs['at'][lm]
        gt = torch.zeros(pred.shape)
        for i in range(pred.shape[0]):
            for j, l in enumerate(targets[i]["labels"]):
                gt[i, l] += targets[i]['ratio'][j] if 'ratio' in targets[i] else 1
        gt = gt.clamp(0, 1)
        losses = {'loss_weak': weak_focal_loss(pred, gt) if fl else F.binary_cross_entropy(pred, gt)}
        if 'at_p' in outputs:        # sedt.py:182-185 (--pooling): plain BCE of the pooled probabilities on the weak clips;
            # weak_mask None indexes with None = a new leading axis over ALL rows (at_p then needs as many rows as gt)
            losses['loss_weak_p'] = F.binary_cross_entropy(outputs['at_p'][weak_mask], gt[weak_mask])
        return losses

    def loss_labels(self, outputs, targets, indices, num_boxes, strong_mask, weak_mask, coef, log=True, fl=False):
        src_logits = outputs['pred_logits'][strong_mask]
        idx = self._src_idx(indices)
        tco = torch.cat([t["labels"][J] for t, (_, J) in zip(targets[strong_mask], indices)])
        cf = torch.cat(coef)
        tc = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64)
        cb = torch.ones(src_logits.shape[:2], dtype=torch.float32)
        tc[idx] = tco
        cb[idx] = cf
        if fl:       # sedt.py:211-218: one-hot over C+2 slots, last dropped -> C+1 columns, column C = "no event"
            onehot = torch.zeros(src_logits.shape[0], src_logits.shape[1], src_logits.shape[2] + 1)
            onehot.scatter_(2, tc.unsqueeze(-1), 1)
            ce = sigmoid_focal_loss(src_logits, onehot[:, :, :-1], self.empty_weight)
        else:
            ce = F.cross_entropy(src_logits.transpose(1, 2), tc, self.empty_weight, reduction='none')
        losses = {'loss_ce': (ce * cb).sum() / num_boxes}
        if log:
            if tco.numel() == 0:
                losses['class_error'] = torch.tensor(100.)
            else:
                acc = (src_logits[idx].argmax(-1) == tco).float().mean() * 100
                losses['class_error'] = 100 - acc
        return losses

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, num_boxes, strong_mask, weak_mask, coef, fl=False):
        pl = outputs['pred_logits']
        tl = torch.as_tensor([len(v["labels"]) for v in targets])
        card = (pl.argmax(-1) != pl.shape[-1] - 1).sum(1)
        return {'cardinality_error': F.l1_loss(card.float(), tl.float())}

    def loss_boxes(self, outputs, targets, indices, num_boxes, strong_mask, weak_mask, coef, fl=False):
        idx = self._src_idx(indices)
        src = outputs['pred_boxes'][idx]
        tgt = torch.cat([t['boxes'][i] for t, (_, i) in zip(targets, indices)], dim=0)
        l1 = F.l1_loss(box_cl_to_xyxy(src), box_cl_to_xyxy(tgt), reduction='none')
        giou = 1 - torch.diag(generalized_box_iou(box_cl_to_xyxy(src), box_cl_to_xyxy(tgt)))
        cf = torch.cat(coef)
        return {'loss_bbox': (l1.sum(dim=1) * cf).sum() / num_boxes, 'loss_giou': (giou * cf).sum() / num_boxes}

    def loss_feature(self, outputs, targets, indices, num_boxes, strong_mask, weak_mask, coef, fl=False):
        tf = outputs['gt_feature']
        idx = self._src_idx(indices)
        bs = len(indices)
        tf = tf.view(bs, tf.shape[0] // bs, -1)
        sf = outputs['pred_feature'][idx]
        tf = torch.cat([t[i] for t, (_, i) in zip(tf, indices)], dim=0)
        sf, tf = F.normalize(sf, dim=1), F.normalize(tf, dim=1)
        return {'loss_feature': F.mse_loss(sf, tf, reduction='none').sum() / num_boxes}

    def get_loss(self, loss, *a, **kw):
        return {'labels': self.loss_labels, 'cardinality': self.loss_cardinality, 'boxes': self.loss_boxes,
                'weak': self.loss_weak, 'feature': self.loss_feature}[loss](*a, **kw)

    def forward(self, outputs, targets, weak_mask=None, strong_mask=None, fine_tune=False, normalize=False, fl=False):
        owa = {k: v[strong_mask] for k, v in outputs.items() if k != 'aux_outputs'}
        indices, coef = self.matcher(owa, targets[strong_mask], fine_tune=fine_tune, normalize=normalize, fl=fl)
        num_boxes = torch.as_tensor([torch.cat(coef).sum()], dtype=torch.float)
        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, indices, num_boxes, strong_mask, weak_mask, coef, fl=fl))
        for i, aux in enumerate(outputs.get('aux_outputs', [])):
            aux_s = {k: v[strong_mask] for k, v in aux.items()}
            sub, cf = self.matcher(aux_s, targets[strong_mask], fl=fl)       # sedt.py:340: plain matching for aux layers
            for loss in self.losses:
                if loss == 'weak':
                    continue
                kw = {'log': False} if loss == 'labels' else {}
                d = self.get_loss(loss, aux, targets, sub, num_boxes, strong_mask, weak_mask, cf, fl=fl, **kw)
                losses.update({k + f'_{i}': v for k, v in d.items()})
        return losses, indices


class PostProcess(nn.Module):
    """sedt.py:355-396: softmax scores, optional fusion with clip-level audio tags (at_m 1/2/3), boxes as (onset, offset)
    scaled by the clip duration - or left as (centre, length) when is_semi."""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, audio_tags=None, at_m=2, is_semi=False, threshold=0.5):
        logits, box = outputs['pred_logits'], outputs['pred_boxes']
        B, Q, _ = logits.shape
        prob = F.softmax(logits, -1)
        if audio_tags is not None:
            top_q = prob[..., :-1].argmax(1)                      # (B, C): for every class, the query that scores highest
            cls = torch.arange(prob.shape[-1] - 1)
            tags = audio_tags.to(prob.dtype)
            if at_m in (2, 3):
                for b in range(B):
                    low = prob[b, top_q[b], cls] < threshold
                    if at_m == 3:
                        low = low & audio_tags[b].bool()
                    prob[b, top_q[b][low], cls[low]] = threshold
            if at_m in (1, 2):
                prob[..., :-1] = prob[..., :-1] * tags[:, None, :]
        scores, labels = prob[..., :-1].max(-1)
        boxes = box if is_semi else box_cl_to_se(box) * target_sizes.view(-1, 1, 1)
        return [{'scores': s, 'labels': l, 'boxes': b} for s, l, b in zip(scores, labels, boxes)]


def build_oracle_criterion(num_classes=10, dec_layers=3, dec_at=True, aux_loss=True, self_sup=False,
                           feature_recon=True, eos_coef=0.1, epsilon=1., alpha=1., pooling=None, weak_loss_p_coef=1.):
    """sedt/__init__.py:39-61."""
    wd = {'loss_ce': 1., 'loss_bbox': 5., 'loss_giou': 2.}
    losses = ['labels', 'boxes', 'cardinality']
    if not self_sup and dec_at:
        wd['loss_weak'] = 1.
        losses.append('weak')
    if not self_sup and pooling:         # __init__.py:44-45
        wd['loss_weak_p'] = weak_loss_p_coef
    if self_sup and feature_recon:
        losses.append('feature')
        wd['loss_feature'] = 1
    if aux_loss:
        aux = {}
        for i in range(dec_layers - 1):
            aux.update({k + f'_{i}': v for k, v in wd.items()})
        wd.update(aux)
    return SetCriterion(1 if self_sup else num_classes, HungarianMatcher(1., 5., 2., epsilon, alpha), wd, eos_coef, losses)


def synthetic_targets(batch, seed, num_classes=10, dataset='urbansed'):
    """URBAN-SED-shaped targets: n~clip(Poisson(4.5),1,9) events, label~U{0..C-1},
    length~U(0.02,0.5), centre~U(l/2,1-l/2); boxes are (centre, length) in [0,1]."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        n = int(torch.poisson(torch.tensor(4.5), generator=g).clamp(1, 9).item())
        l = torch.rand(n, generator=g) * 0.48 + 0.02
        c = l / 2 + torch.rand(n, generator=g) * (1 - l)
        out.append({'labels': torch.randint(0, num_classes, (n,), generator=g),
                    'boxes': torch.stack([c, l], dim=-1), 'orig_size': torch.tensor(10.0)})
    return out
