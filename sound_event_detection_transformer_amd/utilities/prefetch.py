"""Batch prefetcher - counterpart of reference data_utils/DataLoad.py:304-336 (``data_prefetcher``): while the model works on
batch k, batch k+1 is staged through PINNED host memory, copied on a side stream and (optionally) transformed there by the
device feature transform (utilities.transforms.DeviceBoxTransform), so neither the copy nor the augmentation sits on the
step's critical path.  ``next()`` makes the current stream wait for the side stream, exactly like the reference."""
import numpy as np
import torch

from .utils import NestedTensor


class DevicePrefetcher(object):
    def __init__(self, loader, device='cuda', transform=None, return_indexes=False, slots=2, targets_to_device=True):
        """loader yields (input, targets) [or ((input, targets), index)]: input = NestedTensor / tensor of features, or - with
        ``transform`` - a list of raw (T, F) mel-amplitude arrays that the transform turns into the (B,1,frames,F) batch.
        targets_to_device=False leaves the targets on the host: the graphed steppers lay them out in ONE pinned blob and send one
        copy per step (sedt.TargetTables), which is cheaper than a copy per tensor here"""
        self.targets_to_device = targets_to_device
        self.loader = iter(loader)
        self.dev = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.dev)
        self.transform, self.return_index = transform, return_indexes
        self._pin = [None] * slots                      # pinned staging buffers, reused round-robin
        self._slot = 0
        self._done = [None] * slots
        self.preload()

    def _pinned(self, shape):
        k = self._slot
        self._slot = (k + 1) % len(self._pin)
        if self._done[k] is not None:
            self._done[k].synchronize()                 # the copy that last read this buffer has finished long ago
        buf = self._pin[k]
        n = int(np.prod(shape))
        if buf is None or buf.numel() < n:
            buf = self._pin[k] = torch.empty(n, dtype=torch.float32).pin_memory()
        return k, buf[:n].view(*shape)

    def _to_device(self, x):
        if isinstance(x, NestedTensor):
            mask = x.mask.to(self.dev, non_blocking=True) if x.mask is not None else None
            return NestedTensor(self._to_device(x.tensors), mask)
        if torch.is_tensor(x):
            if x.is_cuda:
                return x
            if x.dtype == torch.float32 and x.numel() > 4096:
                k, stage = self._pinned(tuple(x.shape))
                stage.copy_(x)
                y = stage.to(self.dev, non_blocking=True)
                self._done[k] = torch.cuda.Event()
                self._done[k].record(self.stream)
                return y
            return x.to(self.dev, non_blocking=True)
        if isinstance(x, dict):
            return {k: self._to_device(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(self._to_device(v) for v in x)
        return x

    def preload(self):
        try:
            item = next(self.loader)
        except StopIteration:
            self.next_input = self.next_target = self.next_index = None
            return
        if self.return_index:
            (inp, tgt), self.next_index = item
        else:
            (inp, tgt), self.next_index = item, None
        with torch.cuda.stream(self.stream):
            if self.transform is not None and not torch.is_tensor(inp) and not isinstance(inp, NestedTensor):
                nraw = max(int(c.shape[0]) for c in inp)
                k, stage = self._pinned((len(inp), nraw, self.transform.F))      # (rows beyond a clip's own length are never read)
                self.next_input = self.transform(inp, staging=stage)
                self._done[k] = torch.cuda.Event()
                self._done[k].record(self.stream)
            else:
                self.next_input = self._to_device(inp)
            self.next_target = self._to_device(tgt) if self.targets_to_device else tgt

    def next(self):
        torch.cuda.current_stream(self.dev).wait_stream(self.stream)
        inp, tgt, idx = self.next_input, self.next_target, self.next_index
        if inp is not None:
            for t in ([inp.tensors] if isinstance(inp, NestedTensor) else [inp]):
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream(self.dev))
        self.preload()
        return ((inp, tgt), idx) if self.return_index else (inp, tgt)

    def __iter__(self):
        return self

    def __next__(self):
        inp, tgt = self.next()[0] if self.return_index else self.next()
        if inp is None:
            raise StopIteration
        return inp, tgt
