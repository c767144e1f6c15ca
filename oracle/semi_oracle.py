"""CPU restatement of the mean-teacher step and its helpers (TEST INFRASTRUCTURE).

Restates reference engine.py:97-196 (the per-batch body of ``semi_train``), engine.py:300-348 (``get_pseudo_labels``),
utilities/mixup.py:13-196 (``mixup_data`` / ``mixup_label_unlabel``) and utilities/utils.py:46-81 (``EMA``).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Pinned against the reference itself
through tests/golden (G11 pseudo labels, G12 semi step, G13 mixup): tests/golden/make_golden.py imports the reference's
engine.py / mixup.py and records what they return for seeded inputs.

Random draws (Beta mixing weight, batch permutation) are arguments here, so that a test can replay what the reference drew.
"""
from collections import Counter

import torch

from .criterion_oracle import PostProcess, box_cl_to_se


# ------------------------------------------------------------------------------------------------ pseudo labels
@torch.no_grad()
def get_pseudo_labels(tea_outputs, postprocessor, orig_sizes, targets, counter, classwise_threshold, del_overlap=True):
    """engine.py:300-348.  Teacher outputs -> per-clip pseudo events written into ``targets[i]['labels'/'boxes']``:
    clip-level tags gate the class scores (PostProcess at_m=1), an event survives when its score reaches the threshold of
    its class and it is longer than 0.2 / clip-duration; then, per class, overlapping events are removed greedily in
    descending score order.  Boxes stay (centre, length) in [0, 1]."""
    tags = (tea_outputs["at"] >= classwise_threshold).long() if "at" in tea_outputs else None
    results = postprocessor(tea_outputs, orig_sizes, audio_tags=tags, at_m=1, is_semi=True, threshold=None)
    min_len = 0.2 / orig_sizes[0].item()
    for i, r in enumerate(results):
        ok = (r['scores'] >= classwise_threshold[r['labels']]) & (r['boxes'][:, 1] > min_len)
        labels, boxes, scores = r['labels'][ok], r['boxes'][ok], r['scores'][ok]
        if not del_overlap:
            targets[i]['labels'], targets[i]['boxes'] = labels, boxes
            continue
        order = scores.sort(descending=True)[1]
        on, off = boxes[:, 0] - boxes[:, 1] / 2, boxes[:, 0] + boxes[:, 1] / 2
        keep = []
        while order.numel() > 0:
            if order.numel() == 1:
                keep.append(order.item())
                break
            k = order[0].item()
            keep.append(k)
            rest = order[1:]
            shared = (off[rest].clamp(max=off[k]) - on[rest].clamp(min=on[k])).clamp(min=0)
            alive = ((shared == 0) + (labels[rest] != labels[k].item())).nonzero().squeeze()
            if alive.numel() == 0:
                break
            order = order[alive + 1]
        targets[i]['labels'], targets[i]['boxes'] = labels[keep], boxes[keep]
        counter.update(labels[keep].cpu().numpy().tolist())
    return targets


# ------------------------------------------------------------------------------------------------ mixup
def _same_class_overlap(labels, boxes):
    """mixup.py:84-93: True when two events of one class overlap in time (the mixed clip is then abandoned)"""
    for e in set(labels.tolist()):
        se = box_cl_to_se(boxes[(labels == e)[:len(boxes)]])
        se = se[se.argsort(dim=0)[:, 0]]
        if not (se[:, 1][:-1] < se[:, 0][1:]).all().item():
            return True
    return False


def mixup_data(x, y, mask_strong, mask_weak, lam, index, mix_up_ratio=0.5, max_events=20):
    """mixup.py:13-127 with the Beta draw ``lam`` and the shuffled ``index`` passed in.  x (B,1,T,F); returns
    (x', y', strong slice, weak slice).  Only clips with the same kind of label are mixed."""
    bs = x.shape[0]
    mix_num = int(bs * mix_up_ratio)
    d1, d2 = x[:mix_num], x[index][:mix_num]
    l1s, l2s = y[:mix_num], [y[i] for i in index[:mix_num]]
    mixed = lam * d1 + (1 - lam) * d2
    s_lab, s_dat, w_lab, w_dat, u_lab, u_dat = [], [], [], [], [], []
    for i, (l1, l2) in enumerate(zip(l1s, l2s)):
        n1, n2 = len(l1["boxes"]), len(l2["boxes"])
        if n1 == 0 or n2 == 0:
            if n1 > 0:
                s_lab.append(l1s[i]); s_dat.append(d1[i:i + 1])
            elif n2 > 0:
                s_lab.append(l2s[i]); s_dat.append(d2[i:i + 1])
            else:
                w_lab.append({"labels": torch.cat((l1["labels"], l2["labels"])), "boxes": torch.tensor([]),
                              "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"])),
                              "orig_size": l1["orig_size"]})
                w_dat.append(mixed[i:i + 1])
        elif n1 + n2 > max_events:
            s_lab.append(l1); s_dat.append(d1[i:i + 1])          # (n1 > 0 always holds here)
        else:
            cand = {"labels": torch.cat((l1["labels"], l2["labels"])), "boxes": torch.cat((l1["boxes"], l2["boxes"])),
                    "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"])),
                    "orig_size": l1["orig_size"]}
            if _same_class_overlap(cand["labels"], cand["boxes"]):
                s_lab.append(l1); s_dat.append(d1[i:i + 1])
            else:
                s_lab.append(cand); s_dat.append(mixed[i:i + 1])
    data, labels = [], []
    if len(x[mask_strong][mix_num:]):
        s_dat.append(x[mask_strong][mix_num:]); s_lab.extend(y[mask_strong][mix_num:])
    data += s_dat; labels += s_lab
    if mask_weak is not None:
        lw = max(0, mix_num - mask_strong.stop)
        if len(x[mask_weak][lw:]):
            w_dat.append(x[mask_weak][lw:]); w_lab.extend(y[mask_weak][lw:])
        data += w_dat; labels += w_lab
        lu = max(0, mix_num - mask_weak.stop)
        if len(x[mask_weak.stop:][lu:]):
            u_dat.append(x[mask_weak.stop:][lu:]); u_lab.extend(y[mask_weak.stop:][lu:])
        data += u_dat; labels += u_lab
    return torch.cat(data, dim=0), labels, slice(len(s_lab)), slice(len(s_lab), len(s_lab) + len(w_lab))


def mixup_label_unlabel(x1, x2, y1, y2, lam, mix_up_ratio=0.5, max_events=20):
    """mixup.py:129-196: the first ``mix_num`` unlabelled clips (student view) are mixed with labelled clips; their pseudo
    labels and the real labels are concatenated with the mixing weights as ``ratio``."""
    mix_num = int(x1.shape[0] * mix_up_ratio)
    d1, d2 = x1[:mix_num], x2[:mix_num]
    mixed = lam * d1 + (1 - lam) * d2
    labs, dat = [], []
    for i, (l1, l2) in enumerate(zip(y1[:mix_num], y2[:mix_num])):
        if len(l1["boxes"]) + len(l2["boxes"]) > max_events:
            if len(l2["boxes"]):
                labs.append(l2); dat.append(d2[i:i + 1])
            else:
                labs.append(l1); dat.append(d1[i:i + 1])
            continue
        cand = {"labels": torch.cat((l1["labels"], l2["labels"])), "boxes": torch.cat((l1["boxes"], l2["boxes"])),
                "ratio": torch.tensor([lam] * len(l1["labels"]) + [1 - lam] * len(l2["labels"])), "orig_size": l1["orig_size"]}
        if _same_class_overlap(cand["labels"], cand["boxes"]):
            labs.append(l1); dat.append(d1[i:i + 1])
        else:
            labs.append(cand); dat.append(mixed[i:i + 1])
    dat.append(x2[mix_num:])
    labs.extend(y2[mix_num:])
    return torch.cat(dat, dim=0), labs


# ------------------------------------------------------------------------------------------------ EMA teacher
class EMA(object):
    """utilities/utils.py:46-81"""

    def __init__(self, model, decay):
        self.model, self.decay, self.shadow, self.backup = model, decay, {}, {}

    def register(self):
        for n, p in self.model.named_parameters():
            if p.requires_grad:
                self.shadow[n] = p.data.clone()

    def update(self):
        for n, p in self.model.named_parameters():
            if p.requires_grad:
                self.shadow[n] = ((1.0 - self.decay) * p.data + self.decay * self.shadow[n]).clone()

    def apply_shadow(self):
        for n, p in self.model.named_parameters():
            if p.requires_grad:
                self.backup[n] = p.data
                p.data = self.shadow[n]

    def restore(self):
        for n, p in self.model.named_parameters():
            if p.requires_grad:
                p.data = self.backup[n]
        self.backup = {}


# ------------------------------------------------------------------------------------------------ one semi_train iteration
def semi_step(model, ema, criterion, optimizer, x_teacher, x_student, targets, mask_strong, mask_weak, mask_label,
              mask_unlabel, classwise_threshold, fine_tune=False, normalize=False, fl=False, max_norm=0.1, do_step=True,
              counter=None, mix_up_ratio=0, mix_draws=None, trace=None):
    """engine.py:117-181 for one batch.  x_teacher / x_student: (B,1,T,F) views of the same clips (the student's unlabelled
    part carries the extra augmentation).  mix_up_ratio > 0 (engine.py:128-133, 150-153): the labelled part goes through
    mixup_data, the student's unlabelled view and the pseudo labels through mixup_label_unlabel (mixed with the ALREADY MIXED
    labelled batch, as the reference passes it on); mix_draws = (lam of mixup_data, its shuffled index, lam of
    mixup_label_unlabel) - what np.random gave the reference.  trace (dict, optional) receives what the two mixups returned.
    Returns (sup dict, unsup dict, total, pseudo targets as fed to the student's criterion)."""
    counter = Counter() if counter is None else counter
    post = PostProcess()
    wd = criterion.weight_dict
    x_lab, t_lab = x_teacher[mask_label], targets[mask_label]
    if mix_up_ratio > 0:
        x_lab, t_lab, mask_strong, mask_weak = mixup_data(x_lab, t_lab, mask_strong, mask_weak, mix_draws[0], mix_draws[1],
                                                          mix_up_ratio=mix_up_ratio)
        if trace is not None:
            trace['md'] = (x_lab, t_lab, mask_strong, mask_weak)
    sup, _ = criterion(model(x_lab), t_lab, mask_weak, mask_strong, fine_tune, normalize, fl)
    sup_total = sum(sup[k] * wd[k] for k in sup if k in wd)
    unl = [dict(t) for t in targets[mask_unlabel]]
    ema.apply_shadow()
    with torch.no_grad():
        tea = model(x_teacher[mask_unlabel])
        sizes = torch.stack([t["orig_size"] for t in unl], dim=0)
        pseudo = get_pseudo_labels(tea, post, sizes, unl, counter, classwise_threshold)
    ema.restore()
    xs = x_student[mask_unlabel]
    if mix_up_ratio > 0:
        if trace is not None:
            trace['pseudo'] = [dict(t) for t in pseudo]
        xs, pseudo = mixup_label_unlabel(x_lab, xs, t_lab, pseudo, mix_draws[2])
        if trace is not None:
            trace['lu'] = (xs, pseudo)
    unsup, _ = criterion(model(xs), pseudo, None, slice(xs.shape[0]), fine_tune, normalize, fl)
    unsup_total = sum(unsup[k] * wd[k] for k in unsup if k in wd)
    total = sup_total + unsup_total
    total.backward()
    if do_step:
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
        optimizer.zero_grad()
        ema.update()
    return sup, unsup, total, pseudo


def train_step_mix(model, criterion, optimizer, x, targets, mask_strong, mask_weak, mix_up_ratio, mix_draws, max_norm=0.1,
                   do_step=True, trace=None):
    """engine.py:47-80 for one batch with mix-up: mixup_data (draws passed in) -> forward -> criterion with the masks mixup_data
    returned -> backward -> clip / step"""
    x, targets, ms, mw = mixup_data(x, targets, mask_strong, mask_weak, mix_draws[0], mix_draws[1], mix_up_ratio=mix_up_ratio)
    if trace is not None:
        trace['md'] = (x, targets, ms, mw)
    loss_dict, _ = criterion(model(x), targets, mw, ms)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    total.backward()
    if do_step:
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
        optimizer.zero_grad()
    return loss_dict, total
