"""Hungarian matching (host side, stays off the HIP path by design) - counterpart of reference sedt/matcher.py.

Cost = cost_bbox * L1(fake boxes) + cost_class * (-p[class]) + cost_giou * (-GIoU), solved per clip with
scipy.optimize.linear_sum_assignment exactly as matcher.py:85-95.  Unlike the reference, the cost tensors of all
decoder layers are built in one batch on the device and cross to the host in ONE copy (``match_layers``), instead of
one device->host sync per layer."""
from collections import Counter

import torch
from scipy.optimize import linear_sum_assignment
from torch import nn

from ..utilities.box_ops import interval_giou_pairwise


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, epsilon=0, alpha=100):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.epsilon, self.alpha = epsilon, alpha
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"

    @torch.no_grad()
    def cost_matrices(self, logits, boxes, tgt_ids, tgt_bbox, fl=False, with_loc=False, alpha_fl=0.5, gamma_fl=1.0):
        """logits [L,B,Q,C+1], boxes [L,B,Q,2], targets concatenated over the batch -> (cost [L,B,Q,Nt], loc or None).
        fl: the focal matching cost on sigmoid probabilities (matcher.py:73-78); loc = the localisation-only part
        cost_bbox * L1 - cost_giou * GIoU that the fine-tune re-matching thresholds (matcher.py:99-104)."""
        if fl:
            p = logits.float().sigmoid()
            neg = (1 - alpha_fl) * p ** gamma_fl * (-(1 - p + 1e-8).log())
            pos = alpha_fl * (1 - p) ** gamma_fl * (-(p + 1e-8).log())
            cost_class = pos[..., tgt_ids] - neg[..., tgt_ids]
        else:
            cost_class = -logits.float().softmax(-1)[..., tgt_ids]
        c, l = boxes[..., 0].float(), boxes[..., 1].float()
        tc, tl = tgt_bbox[:, 0], tgt_bbox[:, 1]
        s1, e1, s2, e2 = c - l / 2, c + l / 2, tc - tl / 2, tc + tl / 2
        cost_bbox = (s1[..., None] - s2).abs() + (e1[..., None] - e2).abs()      # y extents (0,1) cancel
        giou = interval_giou_pairwise(c.reshape(-1), l.reshape(-1), tc, tl).view(*c.shape, -1)
        cost = self.cost_bbox * cost_bbox + self.cost_class * cost_class - self.cost_giou * giou
        return cost, (self.cost_bbox * cost_bbox - self.cost_giou * giou) if with_loc else None

    @torch.no_grad()
    def match_layers(self, logits, boxes, targets):
        """returns, per layer, the list over clips of (src_idx, tgt_idx) int64 CPU tensors"""
        L, B, Q = logits.shape[:3]
        dev = logits.device
        sizes = [len(v["boxes"]) for v in targets]
        if sum(sizes) == 0:
            e = torch.empty(0, dtype=torch.int64)
            return [[(e, e) for _ in range(B)] for _ in range(L)]
        tgt_ids = torch.cat([v["labels"][:len(v["boxes"])] for v in targets]).to(dev)
        tgt_bbox = torch.cat([v["boxes"] for v in targets]).to(dev).float()
        C = self.cost_matrices(logits, boxes, tgt_ids, tgt_bbox)[0].cpu()       # the one device->host sync
        out = []
        for l in range(L):
            per, off = [], 0
            for b, n in enumerate(sizes):
                i, j = linear_sum_assignment(C[l, b, :, off:off + n])
                per.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
                off += n
            out.append(per)
        return out

    @staticmethod
    def coefficients(idx, targets, normalize=False):
        coef = []
        for i, (_, tgt) in enumerate(idx):
            if normalize:
                num = Counter(tgt.tolist())
                coef.append(torch.tensor([1 / num[j] for j in tgt.tolist()], dtype=torch.float32))
            elif "ratio" in targets[i]:
                coef.append(targets[i]["ratio"].detach().float().cpu())
            else:
                coef.append(torch.ones(len(tgt), dtype=torch.float32))
        return coef

    @torch.no_grad()
    def forward(self, outputs, targets, fine_tune=False, normalize=False, fl=False, ft_rand=None):
        """reference matcher.py:42-133, single layer: (indices, coefficients) with every switch of the reference.
        fl: focal matching cost on sigmoid probabilities (:73-78).  fine_tune (:97-121): a Hungarian pair survives only when its
        query's nearest target (localisation cost only) is closer than ``epsilon``; every other query that close to a target joins
        it with probability alpha * num_gt / num_queries (one uniform per candidate, in query order).  normalize / 'ratio' (:123-132):
        the per-pair loss coefficients.  ft_rand: optional list (one row per clip) of the uniforms to consume instead of torch.rand
        (tests inject the draws the reference made).  SetCriterion.prepare / prepare_device run the same rules for all decoder
        layers at once; this entry is what a caller of ``build_matcher(args)(outputs, targets, ...)`` gets."""
        from .sedt import ALPHA_FL, GAMMA_FL
        logits, boxes = outputs["pred_logits"], outputs["pred_boxes"]
        B, Q = logits.shape[:2]
        sizes = [len(v["boxes"]) for v in targets]
        e = torch.empty(0, dtype=torch.int64)
        if sum(sizes) == 0:
            idx = [(e, e) for _ in range(B)]
            return idx, self.coefficients(idx, targets, normalize)
        dev = logits.device
        tgt_ids = torch.cat([v["labels"][:len(v["boxes"])] for v in targets]).to(dev)
        tgt_bbox = torch.cat([v["boxes"].reshape(-1, 2) for v in targets]).to(dev).float()
        cost, loc = self.cost_matrices(logits[None], boxes[None], tgt_ids, tgt_bbox, fl=fl, with_loc=fine_tune,
                                       alpha_fl=ALPHA_FL, gamma_fl=GAMMA_FL)
        cost = cost[0].cpu()
        loc = loc[0].cpu() if fine_tune else None
        idx, off = [], 0
        for b, n in enumerate(sizes):
            i, j = linear_sum_assignment(cost[b, :, off:off + n])
            i, j = torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)
            if fine_tune:
                if n == 0:
                    raise ValueError('fine_tune needs at least one event in every clip (matcher.py:103 takes a min over them)')
                near_c, near_t = loc[b, :, off:off + n].min(-1)
                close = near_c < self.epsilon
                num_gt = len(j)
                keep = close[i]
                i, j = i[keep], j[keep]
                cand = close.clone()
                cand[i] = False
                cq = torch.where(cand)[0]
                u = torch.as_tensor(ft_rand[b], dtype=torch.float32)[:len(cq)] if ft_rand is not None else torch.rand(len(cq))
                add = cq[~(u > (self.alpha * num_gt / Q))]
                i, j = torch.cat([i, add]), torch.cat([j, near_t[add]])
            idx.append((i, j))
            off += n
        return idx, self.coefficients(idx, targets, normalize)


def build_matcher(args):
    return HungarianMatcher(cost_class=args.set_cost_class, cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou,
                            epsilon=args.epsilon, alpha=args.alpha)
