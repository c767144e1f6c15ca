"""Batch-level mixup - counterpart of reference utilities/mixup.py (mixup_data :13-127, mixup_label_unlabel :129-196): same
signatures, same np.random draws in the same order, same label bookkeeping (only clips with the same kind of label are mixed;
mixed clips carry per-event ``ratio`` coefficients).  The feature half - lam * x1 + (1 - lam) * x2 for the mixed clips, a
plain copy for the rest - is ONE launch for the whole output batch (sedt_mixup) instead of a chain of slices and torch.cat."""
import numpy as np
import torch

from .. import lib as L
from .utils import NestedTensor

_JOB = np.dtype([('src1', np.int32), ('src2', np.int32), ('mode', np.int32), ('lam', np.float32)])


def _se(boxes):
    c, l = boxes.unbind(-1)
    return torch.stack([c - l / 2, c + l / 2], dim=-1)


def _same_class_overlap(labels, boxes):
    """mixup.py:84-93: two events of one class that overlap in time -> the mixed clip is abandoned"""
    lab = labels.tolist()
    for e in set(lab):
        se = _se(boxes[(labels == e)[:len(boxes)]])
        se = se[se.argsort(dim=0)[:, 0]]
        if not (se[:, 1][:-1] < se[:, 0][1:]).all().item():
            return True
    return False


def _launch(x1, x2, jobs):
    n = len(jobs)
    out = torch.empty((n,) + tuple(x1.shape[1:]), device=x1.device, dtype=torch.float32)
    if n == 0:
        return out
    tab = torch.from_numpy(np.asarray(jobs, _JOB).view(np.uint8).reshape(-1).copy()).to(x1.device, non_blocking=True)
    L.check(L.load().sedt_mixup(L.p(x1), L.p(x2), L.p(tab), n, x1[0].numel(), L.p(out), L.stream_ptr()), 'mixup')
    return out


def _data(x):
    t = x.tensors if isinstance(x, NestedTensor) else x
    if not t.is_cuda:
        raise RuntimeError('mixup runs on the device (the HIP path has no CPU fallback)')
    return t.float().contiguous()


def mixup_data(x, y, mask_strong, mask_weak, mix_up_ratio=0.5, max_events=20, alpha=3):
    """reference mixup.py:13-127.  x: NestedTensor or (B,1,T,F) tensor; y: list of target dicts.  Returns
    (x', y', strong slice, weak slice); a NestedTensor input is updated in place like the reference does."""
    lam = float(np.random.beta(alpha, alpha)) if alpha > 0. else 1.0
    xt = _data(x)
    bs = xt.shape[0]
    mix_num = int(bs * mix_up_ratio)
    index = np.asarray(list(range(bs)))
    np.random.shuffle(index)
    dev = xt.device
    s_lab, s_job, w_lab, w_job = [], [], [], []
    for i in range(mix_num):
        l1, l2, j = y[i], y[int(index[i])], int(index[i])
        n1, n2 = len(l1["boxes"]), len(l2["boxes"])
        if n1 == 0 or n2 == 0:
            if n1 > 0:
                s_lab.append(l1); s_job.append((i, 0, 1, 0.0))
            elif n2 > 0:
                s_lab.append(l2); s_job.append((0, j, 2, 0.0))
            else:
                w_lab.append({"labels": torch.cat((l1["labels"], l2["labels"]), dim=0), "boxes": torch.tensor([], device=dev),
                              "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"]), device=dev),
                              "orig_size": l1["orig_size"]})
                w_job.append((i, j, 0, lam))
        elif n1 + n2 > max_events:
            s_lab.append(l1); s_job.append((i, 0, 1, 0.0))
        else:
            cand = {"labels": torch.cat((l1["labels"], l2["labels"]), dim=0), "boxes": torch.cat((l1["boxes"], l2["boxes"]), dim=0),
                    "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"]), device=dev),
                    "orig_size": l1["orig_size"]}
            if _same_class_overlap(cand["labels"], cand["boxes"]):
                s_lab.append(l1); s_job.append((i, 0, 1, 0.0))
            else:
                s_lab.append(cand); s_job.append((i, j, 0, lam))
    jobs, labels = list(s_job), list(s_lab)
    ns_total = mask_strong.stop
    for i in range(mix_num, ns_total):                     # strongly labelled clips that were not mixed
        jobs.append((i, 0, 1, 0.0)); labels.append(y[i])
    n_strong = len(labels)
    n_weak = 0
    if mask_weak is not None:
        jobs += w_job; labels += w_lab
        lw = max(0, mix_num - mask_strong.stop)
        for i in range(mask_weak.start + lw, mask_weak.stop):
            jobs.append((i, 0, 1, 0.0)); labels.append(y[i])
        n_weak = len(labels) - n_strong
        lu = max(0, mix_num - mask_weak.stop)
        for i in range(mask_weak.stop + lu, bs):
            jobs.append((i, 0, 1, 0.0)); labels.append(y[i])
    out = _launch(xt, xt, jobs)
    if isinstance(x, NestedTensor):
        x.tensors = out
        if x.mask is not None and x.mask.shape[0] != out.shape[0]:
            x.mask = x.mask[:out.shape[0]]
        out = x
    return out, labels, slice(n_strong), slice(n_strong, n_strong + n_weak)


def mixup_label_unlabel(x1, x2, y1, y2, mix_up_ratio=0.5, max_events=20, alpha=3):
    """reference mixup.py:129-196: the first int(bs * ratio) unlabelled clips (x2, pseudo labels y2) are mixed with labelled
    clips (x1, y1); returns (x2', y2')"""
    assert mix_up_ratio <= 0.5
    lam = float(np.random.beta(alpha, alpha)) if alpha > 0. else 1.0
    a, b = _data(x1), _data(x2)
    mix_num = int(a.shape[0] * mix_up_ratio)
    dev = a.device
    jobs, labels = [], []
    for i in range(mix_num):
        l1, l2 = y1[i], y2[i]
        if len(l1["boxes"]) + len(l2["boxes"]) > max_events:
            if len(l2["boxes"]):
                labels.append(l2); jobs.append((0, i, 2, 0.0))
            else:
                labels.append(l1); jobs.append((i, 0, 1, 0.0))
            continue
        cand = {"labels": torch.cat((l1["labels"], l2["labels"]), dim=0), "boxes": torch.cat((l1["boxes"], l2["boxes"]), dim=0),
                "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"]), device=dev),
                "orig_size": l1["orig_size"]}
        if _same_class_overlap(cand["labels"], cand["boxes"]):
            labels.append(l1); jobs.append((i, 0, 1, 0.0))
        else:
            labels.append(cand); jobs.append((i, i, 0, lam))
    for i in range(mix_num, b.shape[0]):
        labels.append(y2[i]); jobs.append((0, i, 2, 0.0))
    out = _launch(a, b, jobs)
    if isinstance(x2, NestedTensor):
        x2.tensors = out
        out = x2
    return out, labels
