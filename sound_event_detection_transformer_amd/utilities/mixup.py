"""Batch-level mixup - counterpart of reference utilities/mixup.py (mixup_data :13-127, mixup_label_unlabel :129-196): same
signatures, same np.random draws in the same order, same label bookkeeping (only clips with the same kind of label are mixed;
mixed clips carry per-event ``ratio`` coefficients).  The feature half - lam * x1 + (1 - lam) * x2 for the mixed clips, a
plain copy for the rest - is ONE launch for the whole output batch (sedt_mixup) instead of a chain of slices and torch.cat.

Split in two so that a captured step can use it: ``plan_mixup_data`` is the label half (host only: which clip goes where, the
merged targets, the new strong | weak split) and returns the job records of the feature half; ``mixup_data`` = plan + launch.
The label half of ``mixup_label_unlabel`` also exists on the device (ops.mixup_targets / sedt_mixup_targets) because inside the
mean-teacher step its second operand - the pseudo labels - never leaves the device."""
import numpy as np
import torch

from .. import lib as L
from .utils import NestedTensor

_JOB = np.dtype([('src1', np.int32), ('src2', np.int32), ('mode', np.int32), ('lam', np.float32)])


def _np_boxes(t):
    b = t["boxes"]
    return b.detach().cpu().numpy().reshape(-1, 2).astype(np.float32, copy=False)


def _same_class_overlap(labels, boxes):
    """mixup.py:84-93: two events of one class that overlap in time -> the mixed clip is abandoned.  labels: int list of the
    concatenated label list, boxes: f32 [n, 2] (centre, length); box j carries label j ((cur_labels == e)[:len(cur_boxes)]).
    The reference sorts a class's events by onset and requires end[k] < onset[k+1]: the same as every pair being disjoint,
    evaluated here in the same f32 arithmetic."""
    n = len(boxes)
    if n < 2:
        return False
    lab = np.asarray(labels[:n])
    s = boxes[:, 0] - boxes[:, 1] / np.float32(2)
    e = boxes[:, 0] + boxes[:, 1] / np.float32(2)
    same = lab[:, None] == lab[None, :]
    clash = same & ~(e[:, None] < s[None, :]) & ~(e[None, :] < s[:, None])
    np.fill_diagonal(clash, False)
    return bool(clash.any())


def _merge(l1, l2, lam, dev):
    return {"labels": torch.cat((l1["labels"], l2["labels"]), dim=0),
            "boxes": torch.cat((l1["boxes"], l2["boxes"]), dim=0) if len(l1["boxes"]) or len(l2["boxes"]) else torch.tensor([], device=dev),
            "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"]), device=dev),
            "orig_size": l1["orig_size"]}


def draw_mixup_data(bs, alpha=3):
    """the two np.random draws of mixup_data (mixup.py:22-29), in its order: the Beta weight, then the shuffled clip index"""
    lam = float(np.random.beta(alpha, alpha)) if alpha > 0. else 1.0
    index = np.asarray(list(range(bs)))
    np.random.shuffle(index)
    return lam, index


def plan_mixup_data(y, mask_strong, mask_weak, lam, index, mix_up_ratio=0.5, max_events=20):
    """label half of mixup_data (mixup.py:30-127) for the draws (lam, index): returns (jobs, targets', n_strong, n_weak) with
    jobs = one (src1, src2, mode, lam) record per output clip for ops.mixup(x, x, ...)."""
    bs = len(y)
    mix_num = int(bs * mix_up_ratio)
    dev = y[0]["labels"].device if bs else None
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
                w_lab.append(_merge(l1, l2, lam, dev))
                w_job.append((i, j, 0, lam))
        elif n1 + n2 > max_events:
            s_lab.append(l1); s_job.append((i, 0, 1, 0.0))
        else:
            lab = l1["labels"].tolist() + l2["labels"].tolist()
            if _same_class_overlap(lab, np.concatenate([_np_boxes(l1), _np_boxes(l2)])):
                s_lab.append(l1); s_job.append((i, 0, 1, 0.0))
            else:
                s_lab.append(_merge(l1, l2, lam, dev)); s_job.append((i, j, 0, lam))
    jobs, labels = list(s_job), list(s_lab)
    for i in range(mix_num, mask_strong.stop):             # strongly labelled clips that were not mixed
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
    return jobs, labels, n_strong, n_weak


def job_table(jobs, out=None):
    """job records -> the byte table sedt_mixup reads (a CPU uint8 tensor; ``out``: a pinned buffer to fill instead)"""
    raw = torch.from_numpy(np.asarray(jobs, _JOB).view(np.uint8).reshape(-1).copy())
    if out is None:
        return raw
    out[:raw.numel()].copy_(raw)
    return out


def _launch(x1, x2, jobs):
    from .. import ops
    if len(jobs) == 0:
        return torch.empty((0,) + tuple(x1.shape[1:]), device=x1.device, dtype=torch.float32)
    return ops.mixup(x1, x2, job_table(jobs).to(x1.device, non_blocking=True))


def _data(x):
    t = x.tensors if isinstance(x, NestedTensor) else x
    if not t.is_cuda:
        raise RuntimeError('mixup runs on the device (the HIP path has no CPU fallback)')
    return t.float().contiguous()


def mixup_data(x, y, mask_strong, mask_weak, mix_up_ratio=0.5, max_events=20, alpha=3):
    """reference mixup.py:13-127.  x: NestedTensor or (B,1,T,F) tensor; y: list of target dicts.  Returns
    (x', y', strong slice, weak slice); a NestedTensor input is updated in place like the reference does."""
    xt = _data(x)
    lam, index = draw_mixup_data(xt.shape[0], alpha)
    jobs, labels, n_strong, n_weak = plan_mixup_data(y, mask_strong, mask_weak, lam, index, mix_up_ratio, max_events)
    out = _launch(xt, xt, jobs)
    if isinstance(x, NestedTensor):
        x.tensors = out
        if x.mask is not None and x.mask.shape[0] != out.shape[0]:
            x.mask = x.mask[:out.shape[0]]
        out = x
    return out, labels, slice(n_strong), slice(n_strong, n_strong + n_weak)


def draw_mixup_label_unlabel(alpha=3):
    return float(np.random.beta(alpha, alpha)) if alpha > 0. else 1.0


def lam_pair(lam):
    """(lam, 1 - lam) as the reference's f32 tensors hold them (1 - lam is formed in double precision first)"""
    return np.asarray([lam, 1 - lam], np.float32)


def plan_mixup_label_unlabel(y1, y2, lam, bs1, mix_up_ratio=0.5, max_events=20):
    """label half of mixup_label_unlabel (mixup.py:146-190) for the draw lam: (jobs, targets')"""
    mix_num = int(bs1 * mix_up_ratio)
    dev = y2[0]["labels"].device if len(y2) else None
    jobs, labels = [], []
    for i in range(mix_num):
        l1, l2 = y1[i], y2[i]
        if len(l1["boxes"]) + len(l2["boxes"]) > max_events:
            if len(l2["boxes"]):
                labels.append(l2); jobs.append((0, i, 2, 0.0))
            else:
                labels.append(l1); jobs.append((i, 0, 1, 0.0))
            continue
        lab = l1["labels"].tolist() + l2["labels"].tolist()
        if _same_class_overlap(lab, np.concatenate([_np_boxes(l1), _np_boxes(l2)])):
            labels.append(l1); jobs.append((i, 0, 1, 0.0))
        else:
            labels.append(_merge(l1, l2, lam, dev)); jobs.append((i, i, 0, lam))
    for i in range(mix_num, len(y2)):
        labels.append(y2[i]); jobs.append((0, i, 2, 0.0))
    return jobs, labels


def mixup_label_unlabel(x1, x2, y1, y2, mix_up_ratio=0.5, max_events=20, alpha=3):
    """reference mixup.py:129-196: the first int(bs * ratio) unlabelled clips (x2, pseudo labels y2) are mixed with labelled
    clips (x1, y1); returns (x2', y2')"""
    assert mix_up_ratio <= 0.5
    lam = draw_mixup_label_unlabel(alpha)
    a, b = _data(x1), _data(x2)
    jobs, labels = plan_mixup_label_unlabel(y1, y2, lam, a.shape[0], mix_up_ratio, max_events)
    out = _launch(a, b, jobs)
    if isinstance(x2, NestedTensor):
        x2.tensors = out
        out = x2
    return out, labels
