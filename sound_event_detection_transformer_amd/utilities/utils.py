"""Boundary types of the SEDT model path (counterpart of reference utilities/utils.py:470-492, 526-560)."""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    """batched clips + padding mask (True = padded), reference utils.py:526-560"""

    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        mask = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), mask)

    def cuda(self, non_blocking=True):
        mask = self.mask.cuda(non_blocking=non_blocking) if self.mask is not None else None
        return NestedTensor(self.tensors.cuda(non_blocking=non_blocking), mask)

    def decompose(self):
        return self.tensors, self.mask

    def __getitem__(self, i):
        if isinstance(i, slice):
            return NestedTensor(self.tensors[i], self.mask[i])

    def __repr__(self):
        return str(self.tensors)


_MASKS = {}


def _no_padding_mask(b, h, w, device):
    """all-False (B,T,F) padding mask, created once per shape (read-only by convention: nothing in the path writes masks)"""
    key = (b, h, w, str(device))
    if key not in _MASKS:
        if len(_MASKS) > 16:
            _MASKS.clear()
        _MASKS[key] = torch.zeros((b, h, w), dtype=torch.bool, device=device)
    return _MASKS[key]


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """pad (C,T,F) clips to the batch maximum; mask is True on padding (reference utils.py:470-492)"""
    if isinstance(tensor_list, Tensor) and tensor_list.ndim == 4:
        # an already batched (B,C,T,F) tensor: no padding anywhere - use it as is, with a cached all-False mask
        b, _, h, w = tensor_list.shape
        return NestedTensor(tensor_list, _no_padding_mask(b, h, w, tensor_list.device))
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    c = max(t.shape[0] for t in tensor_list)
    h = max(t.shape[1] for t in tensor_list)
    w = max(t.shape[2] for t in tensor_list)
    b = len(tensor_list)
    dtype, device = tensor_list[0].dtype, tensor_list[0].device
    same = all(tuple(t.shape) == (c, h, w) for t in tensor_list)
    if same:                       # the training pipeline pads every clip to fixed frames: no per-clip copies
        tensor = torch.stack(list(tensor_list))
        return NestedTensor(tensor, _no_padding_mask(b, h, w, device))
    tensor = torch.zeros((b, c, h, w), dtype=dtype, device=device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], :img.shape[2]] = False
    return NestedTensor(tensor, mask)
