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
        self._seg_of = {}                  # id(param) -> segment index of the flat layout (default: one segment)
        self._nseg = 1
        self._flat_dtype = torch.float32
        self._tabset = 'eager'
        self._sets = {}
        # optional device int32 word (engine's graphed steppers: the criterion's non-finite flag): while it is non-zero the
        # update kernels leave parameters, moments and the step count alone (include/sedt_hip.h "Non-finite guard")
        self.guard = None
        # optional device word (runtime.seed_ptr): the dropout-seed word advanced by the step itself (engine's graphed steppers:
        # saves the host-side increment launch per replay)
        self.seed_word = None

    def set_segments(self, segments):
        """lay the flat state / gradient buffers out as consecutive SEGMENTS: segments[k] = the parameters whose gradients the
        backward produces k-th (parameters not listed go to segment 0).  The data-parallel step (engine.GraphedTrainStep) cuts the
        backward at the segment boundaries and all-reduces segment k while the backward of segments k+1.. still runs.  Call before
        the first step."""
        if self._static is not None:
            raise RuntimeError('set_segments must be called before the optimizer state is built (first step)')
        self._seg_of = {id(p): k for k, seg in enumerate(segments) for p in seg}
        self._nseg = max(len(segments), 1)

    def set_tail_params(self, params):
        """two segments: everything else | these parameters (the ones whose gradients come last)"""
        params = list(params)
        ids = {id(p) for p in params}
        self.set_segments([[p for g in self.param_groups for p in g['params'] if id(p) not in ids], params])

    def _build(self):
        ps, gi = [], []
        for gidx, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.requires_grad:
                    if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                        raise RuntimeError('FusedAdamW needs contiguous f32 GPU parameters')
                    ps.append(p)
                    gi.append(gidx)
        order = sorted(range(len(ps)), key=lambda i: self._seg_of.get(id(ps[i]), 0))   # stable: segment by segment
        ps, gi = [ps[i] for i in order], [gi[i] for i in order]
        dev = ps[0].device
        owner, off_in_p, n, state_off = [], [], [], []
        so = 0
        self._seg_chunk0, self._seg_elem0 = [0] * (self._nseg + 1), [0] * (self._nseg + 1)      # segment k = [k], [k+1])
        seg_seen = -1
        for i, p in enumerate(ps):
            k = p.numel()
            sg = self._seg_of.get(id(p), 0)
            while seg_seen < sg:
                seg_seen += 1
                self._seg_chunk0[seg_seen], self._seg_elem0[seg_seen] = len(owner), so
            for c0 in range(0, k, _CHUNK):
                owner.append(i); off_in_p.append(c0); n.append(min(_CHUNK, k - c0)); state_off.append(so + c0)
            so += (k + 7) // 8 * 8                             # 32-byte aligned f32 state (16-byte aligned as a bf16 flat buffer)
        for sg in range(seg_seen + 1, self._nseg + 1):
            self._seg_chunk0[sg], self._seg_elem0[sg] = len(owner), so
        total = so
        self._m = torch.zeros(total, device=dev)
        self._v = torch.zeros(total, device=dev)
        self._ps, self._gi = ps, np.asarray(gi)
        self._pad = [(p.numel() + 7) // 8 * 8 for p in ps]
        self._owner = np.asarray(owner)
        self._off = np.asarray(off_in_p, np.uint64) * 4
        self._tab = np.zeros(len(owner), _DT)
        self._tab['n'] = n
        so_b = np.asarray(state_off, np.uint64) * 4
        self._tab['m'] = np.uint64(self._m.data_ptr()) + so_b
        self._tab['v'] = np.uint64(self._v.data_ptr()) + so_b
        self._partial = torch.empty(len(owner), device=dev)
        self._sumsq = torch.zeros(1, device=dev)
        self._step_t = torch.zeros(1, dtype=torch.int32, device=dev)   # device-side step count (graph-replay safe)
        self._state_off_b = so_b
        self._state_off_e = np.asarray(state_off, np.uint64)
        self._flat_g = None
        self._dev = dev
        self._static = True
        self._sets = {}
        self._use_set(self._tabset)
        pending, self._pending_state = getattr(self, '_pending_state', None), None
        if pending is not None:
            self.load_state_dict(pending)

    # ---- chunk tables: one pinned + one device copy PER USER.  The eager path owns the set 'eager'; every captured HIP
    # graph owns its own (``table_set(name)``): a captured upload re-reads its pinned buffer at every replay, so nothing else
    # may ever write other pointers into it (an eager step between replays used to do exactly that).
    def _use_set(self, name):
        if self._static is None:
            self._tabset = name
            return
        if name not in self._sets:
            nb = self._tab.nbytes
            self._sets[name] = dict(host=torch.empty(nb, dtype=torch.uint8).pin_memory(),
                                    dev=torch.empty(nb, dtype=torch.uint8, device=self._dev), ghost=None, gdev=None, hyper=None)
        self._tabset = name
        st = self._sets[name]
        self._host_tab, self._dev_tab = st['host'], st['dev']
        if self._flat_g is not None:
            if st['ghost'] is None:
                st['ghost'] = torch.empty(self._tab.nbytes, dtype=torch.uint8).pin_memory()
                st['gdev'] = torch.empty(self._tab.nbytes, dtype=torch.uint8, device=self._dev)
            self._host_gtab, self._dev_gtab = st['ghost'], st['gdev']

    def table_set(self, name):
        """context manager: inside it, step()/gather_grads() use the chunk tables called ``name`` (created on first use)"""
        opt = self

        class _Ctx(object):
            def __enter__(self_c):
                self_c.prev = opt._tabset
                if opt._static is None:
                    opt._build()
                opt._use_set(name)
                return opt

            def __exit__(self_c, *a):
                opt._use_set(self_c.prev)
                return False
        return _Ctx()

    # ---- data-parallel support: all gradients in ONE flat buffer -> one RCCL all-reduce -> update from the flat buffer
    def enable_flat_grads(self, dtype=None):
        """the flat gradient buffer (allocated on first use).  dtype torch.bfloat16: bf16 buckets - half the all-reduce bytes;
        gradients are rounded once when packed, the clip norm and AdamW read them back as f32 values"""
        if self._static is None:
            self._build()
        if self._flat_g is None:
            self._flat_dtype = dtype or torch.float32
            if self._flat_dtype not in (torch.float32, torch.bfloat16):
                raise ValueError('flat gradients are f32 or bf16')
            self._flat_g = torch.zeros(self._m.numel(), device=self._dev, dtype=self._flat_dtype)
            self._gtab = self._tab.copy()
            self._use_set(self._tabset)
        elif dtype is not None and dtype != self._flat_dtype:
            raise RuntimeError(f'the flat gradient buffer already exists as {self._flat_dtype}')
        return self._flat_g

    def flat_views(self):
        """{parameter data_ptr: view of its slot in the f32 flat gradient buffer, shaped like the parameter}: a kernel that
        writes a weight gradient there (ops.GRAD_SINK) has delivered it - gather_grads skips tensors whose gradient already
        lives in its slot.  None for a bf16 flat buffer (the gradients need the rounding pass)."""
        flat = self.enable_flat_grads()
        if self._flat_dtype != torch.float32:
            return None
        views, e0 = {}, 0
        for p, pad in zip(self._ps, self._pad):
            views[p.data_ptr()] = flat[e0:e0 + p.numel()].view(p.shape)
            e0 += pad
        return views

    @property
    def n_segments(self):
        return self._nseg

    def segment_params(self, k):
        if self._static is None:
            self._build()
        return [p for p in self._ps if self._seg_of.get(id(p), 0) == k]

    def head_tail_params(self):
        """(parameters of segment 0, parameters of the last segment) in flat-layout order (two-segment layouts)"""
        return self.segment_params(0), (self.segment_params(self._nseg - 1) if self._nseg > 1 else [])

    def flat_segment(self, k):
        """view of segment k of the flat gradient buffer"""
        return self.enable_flat_grads()[self._seg_elem0[k]:self._seg_elem0[k + 1]]

    @torch.no_grad()
    def gather_grads(self, part=None, accumulate=False):
        """copy (accumulate=True: add) p.grad of every parameter (part=None) or of the parameters of one segment (part = its
        index; 'head' = 0, 'tail' = the last one) into the flat gradient buffer (one launch); returns the matching view"""
        flat = self.enable_flat_grads()
        if part == 'head':
            part = 0
        elif part == 'tail':
            part = self._nseg - 1
        if part is None:
            lo, hi, view = 0, len(self._gtab), flat
        else:
            lo, hi, view = self._seg_chunk0[part], self._seg_chunk0[part + 1], self.flat_segment(part)
        sel = self._ps if part is None else self.segment_params(part)
        if any(p.grad is None for p in sel):
            raise RuntimeError('gather_grads: a parameter of the requested part has no gradient')
        gbase = np.fromiter((p.grad.data_ptr() if p.grad is not None else 0 for p in self._ps), np.uint64, len(self._ps))
        t = self._gtab
        t['p'] = np.uint64(flat.data_ptr()) + self._state_off_e * np.uint64(flat.element_size())
        t['g'][lo:hi] = (gbase[self._owner] + self._off)[lo:hi]
        if hi > lo:
            isz = _DT.itemsize
            rows = t[lo:hi]
            if not accumulate:
                rows = rows[rows['g'] != rows['p']]          # gradients written straight into their slot (flat_views)
            k = len(rows)
            if k:
                self._upload(self._dev_gtab[lo * isz:(lo + k) * isz], self._host_gtab[lo * isz:(lo + k) * isz],
                             np.ascontiguousarray(rows).view(np.uint8), ('g', lo, lo + k))
                mode = (1 if accumulate else 0) | (2 if self._flat_dtype == torch.bfloat16 else 0)
                L.check(L.load().sedt_multi_gather(L.p(self._dev_gtab[lo * isz:]), k, mode, L.stream_ptr()), 'multi_gather')
        return view

    @torch.no_grad()
    def step(self, closure=None, max_norm=0.0, from_flat=False):
        """from_flat: read the gradients from the flat buffer filled by gather_grads() (and all-reduced by the caller)"""
        if closure is not None:
            raise NotImplementedError
        from . import ops
        ops.x3_cache_clear()                      # (the backward is over: the bf16x3 operand images of this step can go)
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
        if from_flat:
            t['g'] = np.uint64(self._flat_g.data_ptr()) + self._state_off_e * np.uint64(self._flat_g.element_size())
            t['pad'] = 1 if self._flat_dtype == torch.bfloat16 else 0           # SedtChunk.gflags bit 0: bf16 gradients
        else:
            t['g'] = gbase[self._owner] + self._off
            t['pad'] = 0
        t['lr'] = lrs[self._gi][self._owner]
        t['wd'] = wds[self._gi][self._owner]
        self._sets[self._tabset]['hyper'] = (lrs.tobytes(), wds.tobytes())
        self._upload(self._dev_tab, self._host_tab, t.view(np.uint8), ('s',))
        self._step += 1
        lib = L.load()
        g0 = self.param_groups[0]
        n = len(t)
        if max_norm > 0:       # the norm's final reduction also advances the device-side step count (unless the guard is up)
            L.check(lib.sedt_multi_sumsq(L.p(self._dev_tab), n, L.p(self._partial), L.p(self._sumsq), L.p(self._step_t),
                                         L.p(self.guard), L.p(self.seed_word), L.stream_ptr()), 'multi_sumsq')
        else:
            self._step_t.add_(1)
        L.check(lib.sedt_multi_adamw(L.p(self._dev_tab), n, L.p(self._sumsq), float(max_norm), g0['betas'][0], g0['betas'][1],
                                     g0['eps'], L.p(self._step_t), L.p(self.guard), L.stream_ptr()), 'multi_adamw')

    def _upload(self, dev, host, content, what):
        """chunk table ``content`` (bytes) -> pinned ``host`` -> ``dev``.  The copy is asynchronous, so the pinned buffer must
        not be rewritten while an earlier copy from it is still queued behind the device's work (eager steps issued back to back
        used to do exactly that: a step whose gradients had moved could run on the NEXT step's pointers).  Hence: nothing is
        written or copied when the table is what the device already has (the steady state: the caching allocator hands the
        gradients the same blocks every step); otherwise the event of the last copy from this buffer is waited for first.
        While a HIP graph is being captured the copy is NOT recorded (it would be replayed with every step: one more node in
        front of the optimizer kernels for a table that only changes when a learning rate does): it is queued and
        ``flush_uploads`` - which the capturing stepper calls right after the capture - performs it once."""
        st = self._sets[self._tabset]
        marks = st.setdefault('marks', {})                  # per table part: the event of its last copy (None: copy pending in _deferred)
        hv = host.numpy()
        capturing = dev.is_cuda and torch.cuda.is_current_stream_capturing()
        if what in marks and np.array_equal(hv, content):
            return
        if not capturing:                                  # (a capture starts after a device synchronize: nothing is pending)
            for key, ev in marks.items():                  # every earlier copy that read bytes about to be rewritten
                if ev is not None and key[0] == what[0] and (len(key) == 1 or (key[1] < what[2] and what[1] < key[2])):
                    ev.synchronize()
        hv[:] = content
        if capturing:
            self.__dict__.setdefault('_deferred', []).append((dev, host, st, what))
            marks[what] = None
        else:
            dev.copy_(host, non_blocking=True)
            if dev.is_cuda:
                marks[what] = torch.cuda.Event()
                marks[what].record()
            else:
                marks[what] = None

    def flush_uploads(self):
        """perform the table uploads queued during a capture (see _upload); returns how many"""
        q, self._deferred = self.__dict__.get('_deferred', []), []
        for dev, host, st, what in q:
            dev.copy_(host, non_blocking=True)
        if q:
            torch.cuda.synchronize(self._dev)
        return len(q)

    def refresh_hyperparams(self, name=None):
        """re-read lr / weight_decay of the param groups into the pinned chunk table of set ``name`` (a captured graph
        re-uploads that table on every replay: engine.GraphedTrainStep calls this before each replay, so a StepLR /
        cosine schedule / manual ``param_group['lr']`` change reaches the graphed step).  The pinned buffer is rewritten only
        when a value changed, after waiting for the device (the previous replay's upload may still be reading it)."""
        name = self._tabset if name is None else name
        st = self._sets.get(name)
        if st is None:
            return False
        lrs = np.asarray([g['lr'] for g in self.param_groups], np.float32)
        wds = np.asarray([g['weight_decay'] for g in self.param_groups], np.float32)
        key = (lrs.tobytes(), wds.tobytes())
        if st['hyper'] == key:
            return False
        torch.cuda.synchronize(self._dev)
        tab = np.frombuffer(st['host'].numpy(), dtype=_DT)
        tab['lr'] = lrs[self._gi][self._owner]
        tab['wd'] = wds[self._gi][self._owner]
        st['hyper'] = key
        st['dev'].copy_(st['host'], non_blocking=True)       # (a captured step no longer uploads its table itself: _upload)
        return True

    # ---- checkpointing in torch.optim.AdamW's layout (reference train_sedt.py:272-283, 318-320 saves / restores it)
    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} exactly as torch.optim.AdamW writes it
        (parameter indices in param-group order), so checkpoints move freely between this optimizer and the reference's."""
        groups, index, i = [], {}, 0
        for g in self.param_groups:
            ids = []
            for p in g['params']:
                index[id(p)] = i
                ids.append(i)
                i += 1
            groups.append({**{k: v for k, v in g.items() if k != 'params'}, 'params': ids})
        state = {}
        if self._static is not None:
            step = float(self._step_t.item())
            if step > 0 or self._step > 0:
                off = 0
                for p in self._ps:
                    k = p.numel()
                    state[index[id(p)]] = {'step': torch.tensor(step), 'exp_avg': self._m[off:off + k].view_as(p).clone(),
                                           'exp_avg_sq': self._v[off:off + k].view_as(p).clone()}
                    off += (k + 7) // 8 * 8
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        if len(sd['param_groups']) != len(self.param_groups):
            raise ValueError('loaded state dict has a different number of parameter groups')
        for g, sg in zip(self.param_groups, sd['param_groups']):
            if len(g['params']) != len(sg['params']):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            g.update({k: v for k, v in sg.items() if k != 'params'})
        if self._static is None:
            self._pending_state = sd                  # applied as soon as the flat state exists (needs GPU parameters)
            if all(p.is_cuda for g in self.param_groups for p in g['params'] if p.requires_grad):
                self._build()
            return
        index, i = {}, 0
        for g in self.param_groups:
            for p in g['params']:
                index[id(p)] = i
                i += 1
        st = sd['state']
        steps = set()
        off = 0
        with torch.no_grad():
            for p in self._ps:
                k = p.numel()
                e = st.get(index[id(p)], st.get(str(index[id(p)])))
                if e is None:
                    self._m[off:off + k].zero_()
                    self._v[off:off + k].zero_()
                else:
                    self._m[off:off + k].copy_(e['exp_avg'].reshape(-1))
                    self._v[off:off + k].copy_(e['exp_avg_sq'].reshape(-1))
                    steps.add(int(float(e['step'])))
                off += (k + 7) // 8 * 8
        if len(steps) > 1:
            raise ValueError(f'FusedAdamW keeps ONE step count for all parameters; the checkpoint has {sorted(steps)}')
        n = steps.pop() if steps else 0
        self._step = n
        self._step_t.fill_(n)
        for stt in self._sets.values():
            stt['hyper'] = None

    def snapshot(self):
        """(moments, step counts) - restore() puts them back; used around the warm-up steps of a graph capture"""
        if self._static is None:
            self._build()
        return (self._m.clone(), self._v.clone(), self._step_t.clone(), self._step)

    def restore(self, snap):
        self._m.copy_(snap[0])
        self._v.copy_(snap[1])
        self._step_t.copy_(snap[2])
        self._step = snap[3]

    def grad_norm(self):
        """global gradient norm of the last clipped step (device tensor)"""
        return self._sumsq.sqrt()
