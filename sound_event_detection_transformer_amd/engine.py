"""The per-batch body of the reference's ``engine.train`` loop (engine.py:56-80) as one function."""
import math

import torch

from .optim import FusedAdamW


def train_step(model, criterion, optimizer, batch_input, targets, mask_weak=None, mask_strong=None, max_norm=0.1,
               normalize=False, check_finite=True, patches=None):
    """forward -> SetCriterion -> weighted sum over weight_dict -> backward -> clip_grad_norm_(max_norm) -> step ->
    zero_grad.  Raises on a non-finite loss (the reference calls sys.exit(1), engine.py:70-73)."""
    outputs = model(batch_input, patches) if patches is not None else model(batch_input)
    loss_dict, _ = criterion(outputs, targets, mask_weak, mask_strong, False, normalize)
    wd = criterion.weight_dict
    losses = getattr(criterion, 'last_total', None)      # the same weighted sum, pre-reduced as one dot product
    if losses is None:
        losses = sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)
    if check_finite:
        v = losses.item()
        if not math.isfinite(v):
            raise FloatingPointError(f'Loss is {v}, stopping training: {loss_dict}')
    losses.backward()
    if isinstance(optimizer, FusedAdamW):
        optimizer.step(max_norm=max_norm)                # clip + AdamW fused (three launches for all tensors)
    else:
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    return losses.detach(), loss_dict


def build_optimizer(model, lr=1e-4, lr_backbone=1e-4, weight_decay=1e-4, fused=True):
    """AdamW with the reference's two parameter groups (train_sedt.py:234-240, 269-270)"""
    groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
              {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": lr_backbone}]
    if fused:
        return FusedAdamW(groups, lr=lr, weight_decay=weight_decay)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)
