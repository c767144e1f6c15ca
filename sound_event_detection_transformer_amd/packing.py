"""All per-forward weight preparation of a model in two launches.

Every forward needs, for each weight tensor, its compute-dtype copy in the forward layout ([Cout][taps][Cin]) and - when
training - the dgrad layout ([Cin][taps][Cout], FrozenBN scale folded in), plus the folded scale/bias of every
FrozenBatchNorm.  Issued per tensor that is ~180 tiny launches per step; a ``PackPlan`` keeps persistent destination
buffers and device job tables and does it in ONE bn-fold launch + ONE pack launch.  Parameter pointers are re-read on
every run (EMA swaps / load_state_dict keep working); the tables are re-uploaded only when a pointer changed."""
import numpy as np
import torch

from . import lib as L
from .lib import F32, BF16, TORCH_DTYPE

_BN = np.dtype([('w', np.uint64), ('b', np.uint64), ('rm', np.uint64), ('rv', np.uint64), ('scale', np.uint64),
                ('bias', np.uint64), ('n', np.int32), ('pad', np.int32)])
_PK = np.dtype([('w', np.uint64), ('bnscale', np.uint64), ('wf', np.uint64), ('wb', np.uint64), ('e0', np.int64),
                ('ne', np.int32), ('Cout', np.int32), ('Cin', np.int32), ('taps', np.int32)])
_FJ = np.dtype([('w', np.uint64), ('wf', np.uint64), ('wb', np.uint64), ('N', np.int32), ('K', np.int32), ('blk0', np.int32),
                ('src_bf16', np.int32)])
_CHUNK = 32768

_active = []          # stack of plans whose prepared tensors are valid right now (inside a model forward)
_plans = None         # weak set of every PackPlan alive (weight_neighbours: the backward runs after the forward's plan scope has closed)


def _plan_set():
    global _plans
    if _plans is None:
        import weakref
        _plans = weakref.WeakSet()
    return _plans


def weight_neighbours(ptr, n):
    """bf16x3 operand images (ops._split3): the GEMM about to run needs the image of the weight operand at `ptr`; the n operands the
    model consumes NEXT are returned with it so that one split launch prepares them all, shortly before their use (an image made a whole
    forward ahead is read back cold from HBM: measured +2.5 % on the step; made a few GEMMs ahead it is still in L2 / Infinity Cache).
    Forward operands follow each other in model order, dgrad operands in reverse.  Returns a list of (tensor, rows, cols) - the operand
    containing `ptr` first (the whole tensor when ptr addresses a row slice of it) - or [] when no plan knows the pointer."""
    for plan in list(reversed(_active)) + [p for p in _plan_set() if p not in _active]:
        got = plan._neighbours(ptr, n)
        if got:
            return got
    return []


def lookup_frag(weight):
    """(fragment-major W, fragment-major W^T) prepared for this 2-D parameter by the active plan (csrc/slab.h), or None"""
    for plan in reversed(_active):
        hit = plan.frag_table.get(weight.data_ptr())
        if hit is not None:
            return hit
    return None


def lookup_conv_frag(weight):
    """(fragment-major forward operand, fragment-major dgrad operand) of a convolution weight the active plan lists under
    ``conv_frags`` (csrc/bneck.hip), or None"""
    for plan in reversed(_active):
        hit = plan.conv_frag_table.get(weight.data_ptr())
        if hit is not None:
            return hit
    return None


def lookup_operand_frag(ptr):
    """the fragment-major image of the packed convolution operand that starts at `ptr` (a plan's ``conv_frags``), or None: what
    SedtIgemm.bfrag points at.  The backward runs after the forward's plan scope has closed, so every live plan is asked."""
    for plan in list(reversed(_active)) + [p for p in _plan_set() if p not in _active]:
        hit = plan.operand_frag.get(ptr)
        if hit is not None:
            return hit
    return None


def lookup(weight):
    """(wf, wb, scale, bias) prepared for this parameter (or a view of it) by the active plan, or None"""
    for plan in reversed(_active):
        hit = plan.table.get(weight.data_ptr())
        if hit is not None:
            return hit
    return None


class PackPlan(object):
    def __init__(self, dt, device, convs, linears, bn_only=(), frags=(), conv_frags=()):
        """convs: list of (weight Parameter, (bn_w, bn_b, bn_rm, bn_rv) or None); linears: list of weight Parameters (2-D).
        A linear in f32 mode uses the parameter itself as the forward operand.
        frags (bf16 mode): 2-D weight Parameters [N][K] (N, K multiples of 32) that are ALSO packed fragment-major, W and W^T, for
        the x-stationary slab kernels (one more launch: sedt_pack_frag).
        conv_frags (bf16 mode): convolution weights out of `convs` whose two packed operands are ALSO laid out fragment-major (same
        launch, reading the packed bf16 operands): the fused Bottleneck kernels (csrc/bneck.hip)"""
        self.dt, self.device = dt, device
        td = TORCH_DTYPE[dt]
        es = 4 if dt == F32 else 2
        bns = [(w, bn) for w, bn in list(convs) + list(bn_only) if bn is not None]
        nbn = sum(bn[0].numel() for _, bn in bns)
        self.scale_buf = torch.empty(max(nbn, 1), device=device, dtype=torch.float32)
        self.bias_buf = torch.empty(max(nbn, 1), device=device, dtype=torch.float32)
        total = 0
        items = []
        for w, bn in convs:
            items.append((w, bn, True, True))
            total += 2 * w.numel()
        for w in linears:
            need_f = dt != F32
            items.append((w, None, need_f, True))
            total += (2 if need_f else 1) * w.numel()
        for w, bn in bn_only:
            items.append((w, bn, False, False))
        self.wbuf = torch.empty(total + 64, device=device, dtype=td)
        self.table = {}
        self._nblocks = 0
        self._entries = []
        self._params, self._bn_tensors = [], []
        bn_rows, pk_rows = [], []
        off, boff = 0, 0
        base = self.wbuf.data_ptr()
        for w, bn, need_f, need_b in items:
            Co, Ci = w.shape[0], w.shape[1]
            taps = w.numel() // (Co * Ci)
            n = w.numel()
            wf = wb = None
            wf_ptr = wb_ptr = 0
            if need_f:
                off = (off + 7) // 8 * 8
                wf = self.wbuf[off:off + n].view(Co, taps * Ci)
                wf_ptr = base + off * es
                off += n
            if need_b:
                off = (off + 7) // 8 * 8
                wb = self.wbuf[off:off + n].view(Ci, taps * Co)
                wb_ptr = base + off * es
                off += n
            sc = bi = None
            sc_ptr = 0
            if bn is not None:
                c = bn[0].numel()
                sc, bi = self.scale_buf[boff:boff + c], self.bias_buf[boff:boff + c]
                sc_ptr = sc.data_ptr()
                bn_rows.append((len(self._bn_tensors), sc_ptr, bi.data_ptr(), c))
                self._bn_tensors.append(bn)
                boff += c
            self._entries.append((w, wf if need_f else None, wb, sc, bi))
            if not need_f and not need_b:
                continue
            if taps > 9:
                raise ValueError('PackPlan handles kernels up to 3x3 (the 7x7 stem conv is folded separately)')
            pk_rows.append((len(self._params), sc_ptr, wf_ptr, wb_ptr, self._nblocks, 0, Co, Ci, taps))
            ts = 64 if taps == 1 else 32                      # tile edge of multi_pack_kernel (csrc/misc.hip)
            self._nblocks += ((Co + ts - 1) // ts) * ((Ci + ts - 1) // ts)
            self._params.append(w)
        self._bn = np.zeros(len(bn_rows), _BN)
        for r, (bi_, scp, bip, c) in enumerate(bn_rows):
            self._bn[r]['scale'], self._bn[r]['bias'], self._bn[r]['n'] = scp, bip, c
        self._bn_owner = np.asarray([r[0] for r in bn_rows], np.int64)
        self._pk = np.zeros(len(pk_rows), _PK)
        for r, (pi, scp, wfp, wbp, e0, ne, Co, Ci, taps) in enumerate(pk_rows):
            row = self._pk[r]
            row['bnscale'], row['wf'], row['wb'], row['e0'], row['ne'] = scp, wfp, wbp, e0, ne
            row['Cout'], row['Cin'], row['taps'] = Co, Ci, taps
        self._pk_owner = np.asarray([r[0] for r in pk_rows], np.int64)
        self._dev_bn = torch.empty(max(self._bn.nbytes, 8), dtype=torch.uint8, device=device)
        self._dev_pk = torch.empty(max(self._pk.nbytes, 8), dtype=torch.uint8, device=device)
        self._host_bn = torch.empty(max(self._bn.nbytes, 8), dtype=torch.uint8).pin_memory()
        self._host_pk = torch.empty(max(self._pk.nbytes, 8), dtype=torch.uint8).pin_memory()
        self._last = None
        self._init_frags(frags if dt == BF16 else (), device, conv_frags if dt == BF16 else ())
        _plan_set().add(self)
        self._ranges = None

    def _neighbours(self, ptr, n):
        if self.dt != F32:
            return []
        if self._ranges is None:       # (start, end, direction, index) of every f32 operand: packed forward / dgrad forms, linears' masters
            fwd, bwd = [], []
            for w, wf, wb, _, _ in self._entries:
                Co, Ci = w.shape[0], w.shape[1]
                taps = w.numel() // (Co * Ci)
                f = wf if wf is not None else (w if w.dim() == 2 else None)
                if f is not None:
                    fwd.append((f, Co * taps, Ci))
                if wb is not None:
                    bwd.append((wb, Ci * taps, Co))
            self._ord = {1: fwd, -1: bwd}
            self._ranges = sorted([(t.data_ptr(), t.data_ptr() + t.numel() * 4, d, i) for d, lst in self._ord.items() for i, (t, _, _) in enumerate(lst)])
            self._starts = [r[0] for r in self._ranges]
        import bisect
        k = bisect.bisect_right(self._starts, ptr) - 1
        if k < 0 or not (self._ranges[k][0] <= ptr < self._ranges[k][1]):
            return []
        _, _, d, i = self._ranges[k]
        lst = self._ord[d]
        idx = range(i, min(i + n + 1, len(lst))) if d > 0 else range(i, max(i - n - 1, -1), -1)
        return [lst[j] for j in idx]

    def _init_frags(self, frags, device, conv_frags):
        self.frag_table, self._fr_params, self._fr_entries = {}, [], []
        self.conv_frag_table, self._cf_entries = {}, []
        self.operand_frag = {}                                        # packed operand pointer (fixed, inside self.wbuf) -> its fragment-major image
        frags = [w for w in frags if w.dim() == 2 and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0]
        packed = {id(w): (wf, wb) for w, wf, wb, _, _ in self._entries}
        conv_frags = [w for w in conv_frags if id(w) in packed and packed[id(w)][0] is not None and packed[id(w)][1] is not None
                      and all(d % 32 == 0 for d in packed[id(w)][0].shape + packed[id(w)][1].shape)]
        self._fj = np.zeros(len(frags) + 2 * len(conv_frags), _FJ)
        self._fr_blocks = 0
        if not len(self._fj):
            return
        total = sum(2 * w.numel() for w in frags) + sum(2 * w.numel() for w in conv_frags)
        self.fbuf = torch.empty(total, device=device, dtype=torch.bfloat16)
        base, off = self.fbuf.data_ptr(), 0
        for r, w in enumerate(frags):
            N, K = w.shape
            n = w.numel()
            wf, wb = self.fbuf[off:off + n], self.fbuf[off + n:off + 2 * n]
            row = self._fj[r]
            row['wf'], row['wb'], row['N'], row['K'], row['blk0'] = base + 2 * off, base + 2 * (off + n), N, K, self._fr_blocks
            self._fr_blocks += (N // 32) * (K // 32)
            off += 2 * n
            self._fr_params.append(w)
            self._fr_entries.append((w, wf, wb))
        r = len(frags)
        for w in conv_frags:
            outs = []
            for src in packed[id(w)]:                              # the packed bf16 operands: fixed addresses inside self.wbuf
                N, K = src.shape
                n = src.numel()
                row = self._fj[r]
                row['wf'], row['N'], row['K'], row['blk0'], row['src_bf16'] = base + 2 * off, N, K, self._fr_blocks, 1
                outs.append(self.fbuf[off:off + n])
                self.operand_frag[src.data_ptr()] = outs[-1]
                self._fr_blocks += (N // 32) * (K // 32)
                off += n
                self._fr_params.append(src)
                r += 1
            self._cf_entries.append((w, outs[0], outs[1]))
        self._dev_fj = torch.empty(self._fj.nbytes, dtype=torch.uint8, device=device)
        self._host_fj = torch.empty(self._fj.nbytes, dtype=torch.uint8).pin_memory()

    def pointer_key(self):
        """the live parameter / buffer pointers this plan would read now (changes under EMA.apply_shadow, load_state_dict)"""
        wp = np.fromiter((w.data_ptr() for w in self._params), np.uint64, len(self._params))
        bp = np.asarray([[t.data_ptr() for t in bn] for bn in self._bn_tensors], np.uint64).reshape(-1, 4)
        return wp, bp, (wp.tobytes(), bp.tobytes())

    def run(self, key3=None):
        wp, bp, key = key3 if key3 is not None else self.pointer_key()
        if key != self._last:
            self._pk['w'] = wp[self._pk_owner]
            self._host_pk.numpy()[:self._pk.nbytes] = self._pk.view(np.uint8)
            self._dev_pk.copy_(self._host_pk, non_blocking=True)
            if len(self._bn):
                o = self._bn_owner
                self._bn['w'], self._bn['b'], self._bn['rm'], self._bn['rv'] = bp[o, 0], bp[o, 1], bp[o, 2], bp[o, 3]
                self._host_bn.numpy()[:self._bn.nbytes] = self._bn.view(np.uint8)
                self._dev_bn.copy_(self._host_bn, non_blocking=True)
            self._last = key
            self._ranges = None
            self.table = {w.data_ptr(): (wf if wf is not None else w, wb, sc, bi) for w, wf, wb, sc, bi in self._entries}
            if len(self._fj):
                # (the fragment-packed weights are a subset of the linears: a pointer change there changed `key` as well)
                self._fj['w'] = np.fromiter((w.data_ptr() for w in self._fr_params), np.uint64, len(self._fr_params))
                self._host_fj.numpy()[:] = self._fj.view(np.uint8)
                self._dev_fj.copy_(self._host_fj, non_blocking=True)
                self.frag_table = {w.data_ptr(): (wf, wb) for w, wf, wb in self._fr_entries}
                self.conv_frag_table = {w.data_ptr(): (ff, bf) for w, ff, bf in self._cf_entries}
        lib = L.load()
        if len(self._bn):
            L.check(lib.sedt_multi_bn_fold(L.p(self._dev_bn), len(self._bn), L.stream_ptr()), 'multi_bn_fold')
        L.check(lib.sedt_multi_pack(L.p(self._dev_pk), len(self._pk), self._nblocks, self.dt, L.stream_ptr()), 'multi_pack')
        if len(self._fj):
            L.check(lib.sedt_pack_frag(L.p(self._dev_fj), len(self._fj), self._fr_blocks, L.stream_ptr()), 'pack_frag')

    def __enter__(self):
        from . import ops
        ops.x3_cache_clear()                      # (a forward starts: no operand image of an earlier step is needed any more)
        self.run()
        _active.append(self)
        return self

    def __exit__(self, *a):
        _active.remove(self)
        return False


class PlanSet(object):
    """One PackPlan PER SET OF PARAMETER POINTERS.  A model whose ``param.data`` is swapped between the student weights and
    the EMA shadow (EMA.apply_shadow / restore, reference utilities/utils.py:69-81) alternates between two pointer sets; each
    gets its own plan - own packed-weight buffers, own pinned + device job tables - so that HIP graphs captured over either
    set stay valid (a captured host->device table upload re-reads its pinned buffer on every replay: the buffer must never
    be rewritten with the other set's pointers).  ``factory()`` builds a fresh plan of the model's structure."""
    MAX = 6

    def __init__(self, factory):
        self.factory = factory
        self.plans = [factory()]
        self._cur = None

    def current(self):
        key3 = self.plans[0].pointer_key()
        for p in self.plans:
            if p._last is None or p._last == key3[2]:
                self._key3 = key3
                return p
        if len(self.plans) >= self.MAX:
            raise RuntimeError(f'more than {self.MAX} distinct parameter-pointer sets seen by one model: parameters are being '
                               'reallocated every step (update them in place, or swap between fixed tensors)')
        self.plans.append(self.factory())
        self._key3 = key3
        return self.plans[-1]

    def __enter__(self):
        from . import ops
        ops.x3_cache_clear()
        self._cur = self.current()
        self._cur.run(self._key3)
        _active.append(self._cur)
        return self._cur

    def __exit__(self, *a):
        _active.remove(self._cur)
        self._cur = None
        return False
