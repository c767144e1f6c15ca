"""1-D interval helpers - counterpart of reference utilities/box_ops.py (an event (centre, length) is treated as the
fake box [c-l/2, 0, c+l/2, 1])."""
import torch


def box_cxcywh_to_xyxy(x):
    c, l = x.unbind(-1)
    return torch.stack([c - l / 2, torch.zeros_like(c), c + l / 2, torch.ones_like(c)], dim=-1)


def box_cxcywh_to_se(x):
    c, l = x.unbind(-1)
    return torch.stack([c - l / 2, c + l / 2], dim=-1)


def interval_giou_pairwise(c1, l1, c2, l2):
    """generalized IoU of intervals (c1,l1)[N] x (c2,l2)[M] -> [N,M]; equals box_ops.generalized_box_iou on the fake boxes
    (the y extent is [0,1] for both, so areas are lengths)"""
    s1, e1 = (c1 - l1 / 2)[:, None], (c1 + l1 / 2)[:, None]
    s2, e2 = (c2 - l2 / 2)[None, :], (c2 + l2 / 2)[None, :]
    inter = (torch.min(e1, e2) - torch.max(s1, s2)).clamp(min=0)
    union = (e1 - s1) + (e2 - s2) - inter
    hull = (torch.max(e1, e2) - torch.min(s1, s2)).clamp(min=0)
    return inter / union - (hull - union) / hull


def interval_giou_diag(c1, l1, c2, l2):
    s1, e1, s2, e2 = c1 - l1 / 2, c1 + l1 / 2, c2 - l2 / 2, c2 + l2 / 2
    inter = (torch.min(e1, e2) - torch.max(s1, s2)).clamp(min=0)
    union = (e1 - s1) + (e2 - s2) - inter
    hull = (torch.max(e1, e2) - torch.min(s1, s2)).clamp(min=0)
    return inter / union - (hull - union) / hull
