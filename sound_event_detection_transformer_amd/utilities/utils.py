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
_DERIVED = {}          # (id of a static mask, what) -> tensor derived from it (resized mask, position encoding)
_PINNED = []           # cached tensors a HIP-graph capture has seen: a graph keeps raw pointers, so these are never released


def pin_if_capturing(t):
    """a cached tensor handed to a capture in progress must outlive every replay of that graph: the caches below are evicted by
    .clear(), after which the allocator could reuse the memory under a live graph.  Small tensors (masks, position
    encodings, zero tokens); kept for the life of the process."""
    if t is not None and t.is_cuda and torch.cuda.is_current_stream_capturing() and not any(t is q for q in _PINNED):
        _PINNED.append(t)
    return t


def is_static_mask(mask):
    """True for the cached all-False masks of already batched inputs (and tensors derived from them): constant by construction"""
    return any(m is mask for m in _MASKS.values()) or any(v is mask for v in _DERIVED.values())


def derived_from_static_mask(mask, what, make):
    """make() computed once per static mask and kept (the resized padding mask and the position encoding of a batch without
    padding do not depend on the data: two launches per forward otherwise).  Nothing is cached while a HIP graph is being
    captured - a tensor created inside a capture holds its values only after a replay."""
    if not is_static_mask(mask):
        return make()
    key = (id(mask), what)
    hit = _DERIVED.get(key)
    if hit is not None:
        return pin_if_capturing(hit)
    out = make()
    if not (mask.is_cuda and torch.cuda.is_current_stream_capturing()):
        if len(_DERIVED) > 64:
            _DERIVED.clear()
        _DERIVED[key] = out
    return out


def _no_padding_mask(b, h, w, device):
    """all-False (B,T,F) padding mask, created once per shape (read-only by convention: nothing in the path writes masks)"""
    key = (b, h, w, str(device))
    if key not in _MASKS:
        if len(_MASKS) > 16:
            _MASKS.clear()
            _DERIVED.clear()
        _MASKS[key] = torch.zeros((b, h, w), dtype=torch.bool, device=device)
    return pin_if_capturing(_MASKS[key])


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


class EMA(object):
    """Mean-teacher weights - the reference's utilities/utils.py:46-81 API (register / load / update / apply_shadow /
    restore; ``shadow`` and ``backup`` dicts keyed by parameter name), with ``update`` as ONE launch over all tensors
    (sedt_multi_ema) instead of ~300 per-tensor ops.  apply_shadow / restore swap ``param.data`` pointers exactly like the
    reference; the HIP modules read ``data_ptr()`` on every call, so the swapped weights are what the next forward uses."""
    _CHUNK = 65536

    def __init__(self, model, decay):
        self.model, self.decay = model, decay
        self.shadow, self.backup = {}, {}
        self._tab = None

    def load(self, ema_model):
        for name, param in ema_model.named_parameters():
            self.shadow[name] = param.data.clone()
        self._tab = None

    def register(self):
        for name, param in self.model.named_parameters():
            if param.requires_grad:
                self.shadow[name] = param.data.clone()
        self._tab = None

    def _params(self):
        return [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    @torch.no_grad()
    def update(self):
        ps = self._params()
        for n, _ in ps:
            assert n in self.shadow
        fused = ps and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and self.shadow[n].is_contiguous()
                           for n, p in ps)
        if not fused:                                # host-side models (the CPU oracle in tests): the reference loop
            for n, p in ps:
                self.shadow[n] = ((1.0 - self.decay) * p.data + self.decay * self.shadow[n]).clone()
            return
        import numpy as np
        from .. import lib as L
        from ..optim import _DT
        key = tuple((p.data_ptr(), self.shadow[n].data_ptr(), p.numel()) for n, p in ps)
        if self._tab is None or self._tab[0] != key:
            rows = []
            for pp, sp, k in key:
                for c0 in range(0, k, self._CHUNK):
                    rows.append((pp + 4 * c0, sp + 4 * c0, min(self._CHUNK, k - c0)))
            t = np.zeros(len(rows), _DT)
            t['p'], t['m'], t['n'] = [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]
            dev = ps[0][1].device
            self._tab = (key, torch.from_numpy(t.view(np.uint8).copy()).to(dev), len(rows))
        _, tab, n = self._tab
        L.check(L.load().sedt_multi_ema(L.p(tab), n, float(self.decay), L.p(getattr(self, 'guard', None)), L.stream_ptr()), 'multi_ema')

    def apply_shadow(self):
        for name, param in self._params():
            assert name in self.shadow
            self.backup[name] = param.data
            param.data = self.shadow[name]

    def restore(self):
        for name, param in self._params():
            assert name in self.backup
            param.data = self.backup[name]
        self.backup = {}
