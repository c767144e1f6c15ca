"""Fused clip_grad_norm_ + AdamW over all parameter tensors in three launches (reference engine.py:77-80 issues
~1000 small ones).  Semantics equal torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.AdamW."""
import numpy as np
import torch

from . import lib as L

_CHUNK = 65536
_DT = np.dtype([('p', np.uint64), ('g', np.uint64), ('m', np.uint64), ('v', np.uint64), ('n', np.int32), ('lr', np.float32),
                ('wd', np.float32), ('pad', np.int32)])


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._static = None
        self._tail_ids = set()

    def set_tail_params(self, params):
        """put these parameters LAST in the flat state / gradient layout (call before the first step).  The data-parallel
        step reduces the head part of the flat gradient buffer while the backward of the tail parameters' layers still
        runs (engine.GraphedTrainStep): the tail is the part of the model whose gradients are produced last."""
        if self._static is not None:
            raise RuntimeError('set_tail_params must be called before the optimizer state is built (first step)')
        self._tail_ids = {id(p) for p in params}

    def _build(self):
        ps, gi = [], []
        for gidx, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.requires_grad:
                    if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                        raise RuntimeError('FusedAdamW needs contiguous f32 GPU parameters')
                    ps.append(p)
                    gi.append(gidx)
        order = sorted(range(len(ps)), key=lambda i: id(ps[i]) in self._tail_ids)      # stable: tail parameters last
        ps, gi = [ps[i] for i in order], [gi[i] for i in order]
        total = sum((p.numel() + 3) // 4 * 4 for p in ps)     # every tensor's state starts 16-byte aligned (float4 kernels)
        dev = ps[0].device
        self._m = torch.zeros(total, device=dev)
        self._v = torch.zeros(total, device=dev)
        owner, off_in_p, n, state_off = [], [], [], []
        so = 0
        self._tail_chunk0, self._tail_elem0 = None, None
        for i, p in enumerate(ps):
            k = p.numel()
            if id(p) in self._tail_ids and self._tail_chunk0 is None:
                self._tail_chunk0, self._tail_elem0 = len(owner), so
            for c0 in range(0, k, _CHUNK):
                owner.append(i); off_in_p.append(c0); n.append(min(_CHUNK, k - c0)); state_off.append(so + c0)
            so += (k + 3) // 4 * 4
        if self._tail_chunk0 is None:
            self._tail_chunk0, self._tail_elem0 = len(owner), so
        self._ps, self._gi = ps, np.asarray(gi)
        self._owner = np.asarray(owner)
        self._off = np.asarray(off_in_p, np.uint64) * 4
        self._tab = np.zeros(len(owner), _DT)
        self._tab['n'] = n
        so_b = np.asarray(state_off, np.uint64) * 4
        self._tab['m'] = np.uint64(self._m.data_ptr()) + so_b
        self._tab['v'] = np.uint64(self._v.data_ptr()) + so_b
        self._dev_tab = torch.empty(self._tab.nbytes, dtype=torch.uint8, device=dev)
        self._partial = torch.empty(len(owner), device=dev)
        self._sumsq = torch.zeros(1, device=dev)
        self._step_t = torch.zeros(1, dtype=torch.int32, device=dev)   # device-side step count (graph-replay safe)
        self._host_tab = torch.empty(self._tab.nbytes, dtype=torch.uint8).pin_memory()
        self._state_off_b = so_b
        self._flat_g = None
        self._static = True

    # ---- data-parallel support: all gradients in ONE flat buffer -> one RCCL all-reduce -> update from the flat buffer
    def enable_flat_grads(self):
        if self._static is None:
            self._build()
        if self._flat_g is None:
            self._flat_g = torch.zeros_like(self._m)
            self._gtab = self._tab.copy()
            self._dev_gtab = torch.empty_like(self._dev_tab)
            self._host_gtab = torch.empty(self._tab.nbytes, dtype=torch.uint8).pin_memory()
        return self._flat_g

    def head_tail_params(self):
        """(parameters before the tail, tail parameters) in flat-layout order"""
        if self._static is None:
            self._build()
        head = [p for p in self._ps if id(p) not in self._tail_ids]
        return head, [p for p in self._ps if id(p) in self._tail_ids]

    @torch.no_grad()
    def gather_grads(self, part=None):
        """copy p.grad of every parameter (part=None), of the head ('head') or of the tail parameters ('tail') into the
        flat gradient buffer (one launch); returns the matching view of the flat buffer"""
        flat = self.enable_flat_grads()
        c0, e0, n = self._tail_chunk0, self._tail_elem0, len(self._gtab)
        lo, hi, view = {None: (0, n, flat), 'head': (0, c0, flat[:e0]), 'tail': (c0, n, flat[e0:])}[part]
        sel = [p for p in self._ps if part is None or (id(p) in self._tail_ids) == (part == 'tail')]
        if any(p.grad is None for p in sel):
            raise RuntimeError('gather_grads: a parameter of the requested part has no gradient')
        gbase = np.fromiter((p.grad.data_ptr() if p.grad is not None else 0 for p in self._ps), np.uint64, len(self._ps))
        t = self._gtab
        t['p'] = np.uint64(flat.data_ptr()) + self._state_off_b
        t['g'][lo:hi] = (gbase[self._owner] + self._off)[lo:hi]
        if hi > lo:
            isz = _DT.itemsize
            self._host_gtab.numpy()[lo * isz:hi * isz] = t[lo:hi].view(np.uint8)
            self._dev_gtab[lo * isz:hi * isz].copy_(self._host_gtab[lo * isz:hi * isz], non_blocking=True)
            L.check(L.load().sedt_multi_gather(L.p(self._dev_gtab[lo * isz:]), hi - lo, L.stream_ptr()), 'multi_gather')
        return view

    @torch.no_grad()
    def step(self, closure=None, max_norm=0.0, from_flat=False):
        """from_flat: read the gradients from the flat buffer filled by gather_grads() (and all-reduced by the caller)"""
        if closure is not None:
            raise NotImplementedError
        if self._static is None:
            self._build()
        ps = self._ps
        if not from_flat and any(p.grad is None for p in ps):
            raise RuntimeError('FusedAdamW: every trainable parameter must have a gradient')
        pbase = np.fromiter((p.data_ptr() for p in ps), np.uint64, len(ps))          # live pointers, every call
        if from_flat:
            gbase = None
        else:
            gbase = np.fromiter((p.grad.data_ptr() for p in ps), np.uint64, len(ps))
        lrs = np.asarray([g['lr'] for g in self.param_groups], np.float32)
        wds = np.asarray([g['weight_decay'] for g in self.param_groups], np.float32)
        t = self._tab
        t['p'] = pbase[self._owner] + self._off
        t['g'] = (np.uint64(self._flat_g.data_ptr()) + self._state_off_b) if from_flat else (gbase[self._owner] + self._off)
        t['lr'] = lrs[self._gi][self._owner]
        t['wd'] = wds[self._gi][self._owner]
        self._host_tab.numpy()[:] = t.view(np.uint8)
        self._dev_tab.copy_(self._host_tab, non_blocking=True)
        self._step += 1
        self._step_t.add_(1)
        lib = L.load()
        g0 = self.param_groups[0]
        n = len(t)
        if max_norm > 0:
            L.check(lib.sedt_multi_sumsq(L.p(self._dev_tab), n, L.p(self._partial), L.p(self._sumsq), L.stream_ptr()), 'multi_sumsq')
        L.check(lib.sedt_multi_adamw(L.p(self._dev_tab), n, L.p(self._sumsq), float(max_norm), g0['betas'][0], g0['betas'][1],
                                     g0['eps'], L.p(self._step_t), L.stream_ptr()), 'multi_adamw')

    def refresh_hyperparams(self):
        """re-read lr / weight_decay of the param groups into the pinned chunk table (a captured graph re-uploads it)"""
        lrs = np.asarray([g['lr'] for g in self.param_groups], np.float32)
        wds = np.asarray([g['weight_decay'] for g in self.param_groups], np.float32)
        self._tab['lr'] = lrs[self._gi][self._owner]
        self._tab['wd'] = wds[self._gi][self._owner]
        self._host_tab.numpy()[:] = self._tab.view(np.uint8)

    def grad_norm(self):
        """global gradient norm of the last clipped step (device tensor)"""
        return self._sumsq.sqrt()
