"""Tensor-level wrappers over the C ABI (include/sedt_hip.h).

Every function takes torch tensors that live on the GPU, passes raw device pointers + sizes to
libsedt_hip.so on the current torch stream and returns torch tensors allocated with the caching
allocator.  No arithmetic happens in torch here; torch is device memory + streams only.
"""
import ctypes as C

import os
import numpy as np
import torch

from . import lib as L
from .lib import F32, BF16, ACT_NONE, ACT_RELU, ACT_SIGMOID, TORCH_DTYPE, p as _p  # noqa: F401


class ConvGeom(object):
    """geometry of one convolution over NHWC activations"""
    __slots__ = ('Hi', 'Wi', 'Ci', 'Ho', 'Wo', 'Co', 'KH', 'KW', 'sh', 'sw', 'ph', 'pw', 'dh', 'dw')

    def __init__(self, Hi, Wi, Ci, Co, k=1, stride=1, pad=0, dil=1):
        self.Hi, self.Wi, self.Ci, self.Co = Hi, Wi, Ci, Co
        self.KH = self.KW = k
        self.sh = self.sw = stride
        self.ph = self.pw = pad
        self.dh = self.dw = dil
        self.Ho = (Hi + 2 * pad - dil * (k - 1) - 1) // stride + 1
        self.Wo = (Wi + 2 * pad - dil * (k - 1) - 1) // stride + 1

    @property
    def taps(self):
        return self.KH * self.KW

    @property
    def plain(self):
        return self.KH == 1 and self.KW == 1 and self.sh == 1 and self.sw == 1 and self.ph == 0 and self.pw == 0


def _dev_env(name, default):
    """a developer A/B switch: the environment is consulted only when SEDT_DEV=1 is set as well (tools/README.md); a training job's
    behaviour never depends on a leaked variable"""
    return os.environ.get(name, default) if os.environ.get('SEDT_DEV') == '1' else default


IGEMM_BREG = _dev_env('SEDT_IGEMM_BREG', '0') != '0'   # developer A/B switch (csrc/igemm3.hip, BR = 1): weights through registers out of their fragment-major images
RELU_BITS = _dev_env('SEDT_RELU_BITS', '1') != '0'      # developer A/B switch: 0 = the backward masks with the bf16 activations
PROFILE = None   # bench.py sets this to a list: every GEMM launch then also records (argument block, dtype, shape, operands, hint)
PROFILE_FUSED = []     # with PROFILE on: (kernel-name prefix, algorithmic flop, algorithmic bytes) of every fused-Bottleneck launch of the recorded step
PROFILE_HINT = None   # how the un-profiled step launches the problem being recorded: 'conv3x3_c64_kernel' (direct kernel), ('group', kernel label, share of the recorded problem's flops the grouped form executes)


def _dev_check(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('sedt ops need GPU tensors: the HIP path has no CPU fallback')


def igemm_args(M, N, K, A, lda, B, ldb, Cout, ldc, *, trans=0, conv=None, transposed=0, out_f32=0, scale=None,
               bias=None, res=None, ldr=0, res_mod=0, mask=None, ldm=0, act=ACT_NONE, act_post_res=0, alpha=1.0, drop_p=0.0,
               seed=0, seed_ptr=None, splitk=1, slab=None, tile=(0, 0), colsum_out=None, mask_bits=False, bits_out=None):
    """the SedtIgemm argument block of one implicit GEMM; ``conv`` = (Hi, Wi, Ci, Ho, Wo, KH, KW, sh, sw, ph, pw, dh, dw)"""
    _dev_check(A, B, Cout)
    a = L.SedtIgemm()
    a.M, a.N, a.K = M, N, K
    a.A, a.B, a.lda, a.ldb = A.data_ptr(), B.data_ptr(), lda, ldb
    a.trans, a.transposed = trans, transposed
    if conv is not None:
        a.conv = 1
        (a.Hi, a.Wi, a.Ci, a.Ho, a.Wo, a.KH, a.KW, a.sh, a.sw, a.ph, a.pw, a.dh, a.dw) = conv
    else:
        a.conv = 0
        a.KH = a.KW = a.sh = a.sw = a.dh = a.dw = 1
    a.C = Cout.data_ptr() if Cout is not None else None
    a.ldc, a.out_f32 = ldc, out_f32
    a.scale = scale.data_ptr() if scale is not None else None
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.ldr, a.res_mod = ldr, res_mod
    a.mask = mask.data_ptr() if mask is not None else None
    a.ldm = ldm
    a.act, a.act_post_res, a.alpha = act, act_post_res, alpha
    a.drop_p, a.seed = drop_p, seed & 0xffffffff
    a.seed_ptr = seed_ptr.data_ptr() if seed_ptr is not None else None
    a.splitk = splitk
    a.slab = slab.data_ptr() if slab is not None else None
    a.tile_m, a.tile_n = tile
    a.colsum_out = colsum_out.data_ptr() if colsum_out is not None else None
    if mask_bits:                                   # 1-bit ReLU mask: uint8 [M, N/8] image, ldm in bytes
        assert mask is not None and mask.dtype == torch.uint8
        a.mask_bits = 1
    if bits_out is not None:                        # sign bits of the stored output, uint8 [M, N/8]
        assert bits_out.dtype == torch.uint8 and bits_out.shape == (M, N // 8) and N % 8 == 0 and bits_out.stride(1) == 1 and not trans
        a.bits_out, a.ldbits = bits_out.data_ptr(), bits_out.stride(0)
    if IGEMM_BREG and not trans and B.dtype == torch.bfloat16:
        from . import packing
        fr = packing.lookup_operand_frag(B.data_ptr())
        if fr is not None and fr.numel() == B.numel() and ldb * B.shape[0] == B.numel():
            a.bfrag = fr.data_ptr()
    return a


class WgradPool(object):
    """Weight-gradient GEMMs waiting for a ride (co-scheduling, csrc/igemm3.hip: igemm3_co_kernel).

    In the backward pass only the dgrad chain is ordered; a layer's weight gradients are needed by the optimizer alone.
    While ``coschedule()`` is active, ReduceBatch.flush() parks the layer's wgrad problems here instead of launching
    them, and every following forward/dgrad GEMM launch takes a few along in the workgroup slots it would leave idle
    (at batch 64 most GEMMs of the chain fill 2 of ~5 slots per CU).  ``drain()`` - called before anything reads the
    gradients - launches what is left and then all split-K reductions."""
    SLOTS = 1280                    # 256 CUs x 5 resident workgroups of the 32 KB-LDS GEMM kernels
    MAX_RIDERS = 10

    def __init__(self):
        self.gemms, self.reduces, self.keep, self.dtype = [], [], [], None

    def push(self, group, dtype, jobs, keep, operands):
        self.gemms.extend(group)
        self.reduces.extend(jobs)
        self.keep.append((keep, operands))
        self.dtype = dtype

    def take(self, main_tiles):
        if NO_RIDE:
            return []
        room = max(self.SLOTS - main_tiles, 512)
        out = []
        while self.gemms and len(out) < self.MAX_RIDERS:
            a = self.gemms[0][0]
            wgs = ((a.M + 63) // 64) * ((a.N + 63) // 64) * max(a.splitk, 1)
            if out and wgs > room:
                break
            out.append(self.gemms.pop(0))
            room -= wgs
        return out

    def give_back(self, items):
        self.gemms[0:0] = items

    def drain(self):
        lib = L.load()
        while self.gemms:
            chunk, self.gemms = self.gemms[:10], self.gemms[10:]
            arr = (L.SedtIgemm * len(chunk))(*[c[0] for c in chunk])
            L.check(lib.sedt_wgrad_group(arr, len(chunk), L.gemm_dtype(self.dtype), L.stream_ptr()), 'wgrad_group')
        while self.reduces:
            chunk, self.reduces = self.reduces[:L.MAX_REDUCE_JOBS], self.reduces[L.MAX_REDUCE_JOBS:]
            arr = (L.SedtReduceJob * len(chunk))(*chunk)
            L.check(lib.sedt_multi_wgrad_reduce(arr, len(chunk), None, L.stream_ptr()), 'multi_wgrad_reduce')
        self.keep = []


POOL = WgradPool()
_co = {'on': False}


class coschedule(object):
    """context manager: weight gradients computed inside are complete only after the scope exits (it drains the pool)"""

    def __init__(self, enable=True):
        self.enable = enable

    def __enter__(self):
        self.prev = _co['on']
        _co['on'] = bool(self.enable) and PROFILE is None
        return self

    def __exit__(self, *exc):
        _co['on'] = self.prev
        if not self.prev:
            POOL.drain()
        return False


# ---- the fast bf16x3 mode (csrc/split3.hip): an f32-mode GEMM whose shape fits the LDS-DMA kernels runs as ONE bf16 GEMM over a three
# times longer contraction axis, on operands split once into bf16 [hi | lo | hi] / [hi | hi | lo] images, with an f32 epilogue.
# X3_FAST = False keeps every contraction of the mode on the register-staged generic kernel (csrc/igemm.hip, the round-4 form).
X3_FAST = _dev_env('SEDT_X3_FAST', '1') != '0'


# An activation / gradient is a GEMM operand twice per step (x: its layer's forward and weight gradient; dY: input and weight gradient):
# its [hi | lo] image is kept from the first use to the second.  Keyed by (pointer, rows, cols, row stride); an entry holds the
# SOURCE tensor as well, so the caching allocator cannot hand that memory to another tensor while the entry lives, and the source's
# autograd version counter at the time the image was made: a torch in-place op on a cached operand bumps the counter and the stale
# image is dropped at the next lookup.  Writes through this library's own raw pointers do not bump it, so a GEMM that writes a tensor
# WITHOUT leaving its image (no split_out) pops that tensor's entry (igemm / _igemm_x3_fast).  Emptied when a model forward starts
# (packing.PlanSet / PackPlan) and by the optimizer step.
X3_CACHE = {}
# weight operand images [rows][3 cols] = [hi | hi | lo] of WHOLE packed operands, by base pointer: (rows, cols, ld, image, source tensor).
# A GEMM asks for its weight's image; on a miss ONE split launch makes it together with the next few operands the model consumes
# (packing.weight_neighbours), row slices of a cached operand (the q|k and v halves of an in_proj weight) are views of its image.
W3_CACHE = {}
W3_STARTS = []                      # sorted base pointers of W3_CACHE (range lookup for row slices)
X3_WGROUP = int(_dev_env('SEDT_X3_WGROUP', '3'))       # further operands prepared with the one that missed (0: one split launch per weight).
# Same-box A/B on the C2 bf16x3 step (profiles/r06_ab_x3_wgroup.txt): 0 -> 13.74-13.78 ms, 3 -> 13.69-13.72, 5 -> 13.85-13.87, 7 -> 13.83-13.93:
# an image made more than a few GEMMs ahead of its use is read back cold, which costs more than the launches it saves
X3_CACHE_ON = _dev_env('SEDT_X3_CACHE', '1') != '0'
X3_SPLIT_OUT = _dev_env('SEDT_X3_SPLIT_OUT', '1') != '0'      # GEMM epilogues write the operand image of outputs that feed GEMMs again


def x3_cache_clear():
    X3_CACHE.clear()
    W3_CACHE.clear()
    del W3_STARTS[:]


def _w3_lookup(ptr, rows, cols, ld):
    """image of the weight operand (ptr, rows, cols, ld) if it is a cached operand or a row slice of one"""
    import bisect
    k = bisect.bisect_right(W3_STARTS, ptr) - 1
    if k < 0:
        return None
    base = W3_STARTS[k]
    r_, c_, ld_, img, src = W3_CACHE[base]
    off = ptr - base
    if c_ != cols or ld_ != ld or off % (4 * ld) or off // (4 * ld) + rows > r_:
        return None
    r0 = off // (4 * ld)
    return img[r0:r0 + rows]


def _w3_plan(t, off, rows, cols, ld):
    """split jobs (tensor, 0, rows, cols, ld, 1) that make the missing weight image AND those of the operands consumed next"""
    from . import packing
    ptr = t.data_ptr() + 4 * off
    jobs = []
    if X3_WGROUP > 0 and X3_CACHE_ON:
        for wt, r_, c_ in packing.weight_neighbours(ptr, X3_WGROUP):
            if wt.data_ptr() in W3_CACHE or c_ % 64 or (wt.data_ptr() % 16):
                continue
            jobs.append((wt, 0, r_, c_, c_, 1))
    return jobs


SPLIT_MAXJ = 8                      # jobs per sedt_split3 launch (csrc/split3.hip)


def _split3(jobs):
    """jobs: (tensor, element offset of the view's first element, rows, cols, row stride, pattern); one launch for up to SPLIT_MAXJ.  Returns
    the bf16 images: [rows, 2 * cols] = [hi | lo] for pattern 0 (activations / gradients), [rows, 3 * cols] = [hi | hi | lo] for pattern 1 (weights)"""
    import bisect
    outs, todo = [None] * len(jobs), []           # todo: (job tuple, destination, index in outs or None)
    for i, (t, off, rows, cols, ld, pattern) in enumerate(jobs):
        key = (t.data_ptr() + 4 * off, rows, cols, ld)
        if pattern == 1 and X3_CACHE_ON:
            hit = _w3_lookup(key[0], rows, cols, ld)
            if hit is not None:
                outs[i] = hit
                continue
            extra = _w3_plan(t, off, rows, cols, ld)
            for wt, _, r_, c_, ld_, _ in extra[:SPLIT_MAXJ - len(jobs)]:
                d = torch.empty((r_, 3 * c_), device=wt.device, dtype=torch.bfloat16)
                todo.append(((wt, 0, r_, c_, ld_, 1), d, None))
                W3_CACHE[wt.data_ptr()] = (r_, c_, ld_, d, wt)
                bisect.insort(W3_STARTS, wt.data_ptr())
            hit = _w3_lookup(key[0], rows, cols, ld)          # (the operand itself is the first neighbour when a plan knows it)
            if hit is not None:
                outs[i] = hit
                continue
        hit = X3_CACHE.get(key) if (pattern == 0 and X3_CACHE_ON) else None
        if hit is not None and hit[2] == t._version:
            outs[i] = hit[1]
            continue
        d = torch.empty((rows, (3 if pattern else 2) * cols), device=t.device, dtype=torch.bfloat16)
        outs[i] = d
        todo.append(((t, off, rows, cols, ld, pattern), d, i))
        if pattern == 0 and X3_CACHE_ON:
            if len(X3_CACHE) >= 1024:           # (op-level callers outside a model forward never reach a clearing point: bound what is held)
                X3_CACHE.clear()
            X3_CACHE[key] = (t, d, t._version)
    for base in range(0, len(todo), SPLIT_MAXJ):
        chunk = todo[base:base + SPLIT_MAXJ]
        arr = (L.SedtSplitJob * len(chunk))()
        for n, ((t, off, rows, cols, ld, pattern), d, _) in enumerate(chunk):
            arr[n].src, arr[n].ld, arr[n].dst = t.data_ptr() + 4 * off, ld, d.data_ptr()
            arr[n].rows, arr[n].cols, arr[n].pattern = rows, cols, pattern
        L.check(L.load().sedt_split3(arr, len(chunk), L.stream_ptr()), 'split3')
    return outs


def _x3_rows(M, conv):
    """rows of the gathered operand: the pixels of the B images behind the M output rows (conv = igemm_args' geometry tuple)"""
    if conv is None:
        return M
    Hi, Wi, _, Ho, Wo = conv[:5]
    return (M // (Ho * Wo)) * Hi * Wi


def _x3_fast_ok(M, N, K, A, lda, B, ldb, Cout, ldc, kw):
    """envelope of the LDS-DMA forward / dgrad kernels on the tripled contraction (igemm_lds_try, csrc/igemm3.hip) with f32 epilogue operands"""
    conv = kw.get('conv')
    if not X3_FAST or kw.get('trans', 0) or kw.get('splitk', 1) > 1 or kw.get('act', ACT_NONE) == ACT_SIGMOID or kw.get('slab') is not None:
        return False
    if Cout is None or Cout.dtype != torch.float32 or A.dtype != torch.float32 or B.dtype != torch.float32:
        return False
    Ci = conv[2] if conv is not None else K
    if K % 64 or Ci % 64 or N % 8 or ldc % 8 or lda % 4 or ldb % 4 or A.data_ptr() % 16 or B.data_ptr() % 16 or Cout.data_ptr() % 16:
        return False
    if conv is not None:
        taps = conv[5] * conv[6]
        if ldb != K or taps * Ci != K or taps > 32 or M % (conv[3] * conv[4]):
            return False
        if kw.get('transposed', 0) and (conv[7] != conv[8] or (2 * Ci) % conv[7]):
            return False
    res, mask = kw.get('res'), kw.get('mask')
    if res is not None and (res.dtype != torch.float32 or res.data_ptr() % 16 or kw.get('ldr', 0) % 4):
        return False
    if mask is not None and not kw.get('mask_bits') and (mask.dtype != torch.float32 or mask.data_ptr() % 16 or kw.get('ldm', 0) % 4):
        return False
    if mask is not None and kw.get('mask_bits') and (mask.data_ptr() % 4 or kw.get('ldm', 0) % 4):     # 1-bit image: ldm in BYTES (igemm3.hip envelope)
        return False
    for k_ in ('scale', 'bias'):                      # per-column epilogue operands are read as 16-byte vectors (igemm_lds_try)
        t_ = kw.get(k_)
        if t_ is not None and t_.data_ptr() % 16:
            return False
    rows = _x3_rows(M, conv)
    return rows * 3 * Ci * 2 < (1 << 31) and N * 3 * K * 2 < (1 << 31)


def _igemm_x3_fast(M, N, K, A, lda, B, ldb, Cout, ldc, kw):
    conv = kw.get('conv')
    Ci = conv[2] if conv is not None else K
    rows = _x3_rows(M, conv)
    wjob = (B, 0, N * (K // Ci), Ci, Ci, 1) if conv is not None else (B, 0, N, K, ldb, 1)
    A3, B3 = _split3([(A, 0, rows, Ci, lda, 0), wjob])
    kw = dict(kw)
    if conv is not None:
        kw['conv'] = (conv[0], conv[1], 3 * Ci) + tuple(conv[3:])
    kw['out_f32'] = 1
    a = igemm_args(M, N, 3 * K, A3, 2 * Ci, B3, 3 * K, Cout, ldc, **kw)
    a.f32ep, a.awrap = 1, Ci                 # (the A image is [hi | lo]: the walk over 3 Ci wraps back onto hi)
    # outputs that feed GEMMs again - a ReLU'd activation (the Bottleneck chain, the FFN's hidden layer, the box MLP) or a masked
    # gradient (the dgrad chain: the next input gradient AND a weight gradient read it) - leave their operand image from the epilogue:
    # the consumers find it in the cache instead of running a split pass (read 4 B + write 6 B per element and a launch saved)
    img = None
    if X3_CACHE_ON and X3_SPLIT_OUT and (kw.get('act', ACT_NONE) == ACT_RELU or kw.get('mask') is not None) and N % 8 == 0:
        img = torch.empty((M, 2 * N), device=Cout.device, dtype=torch.bfloat16)
        a.split_out = img.data_ptr()
        X3_CACHE[(Cout.data_ptr(), M, N, ldc)] = (Cout, img, Cout._version)
    else:
        X3_CACHE.pop((Cout.data_ptr(), M, N, ldc), None)          # (rewritten without a new image: a cached one would be stale)
    if PROFILE is not None:
        PROFILE.append((a, BF16, (M, N, 3 * K, 0, 0 if conv is None else 1), (A, B, Cout, kw, A3, B3), PROFILE_HINT))
    if L.LAUNCH_LOG is not None:
        buf = C.create_string_buffer(160)
        if L.load().sedt_igemm_describe(C.byref(a), BF16, 0, buf, 160) == 0:
            L.LAUNCH_LOG['igemm_x3:' + buf.value.decode()] += 1
    L.check(L.load().sedt_igemm(C.byref(a), BF16, L.stream_ptr()), 'sedt_igemm_x3')


def igemm(dtype, M, N, K, A, lda, B, ldb, Cout, ldc, **kw):
    """raw implicit GEMM call (arguments as igemm_args)"""
    if dtype == F32 and L.GEMM_X3 and _x3_fast_ok(M, N, K, A, lda, B, ldb, Cout, ldc, kw):
        return _igemm_x3_fast(M, N, K, A, lda, B, ldb, Cout, ldc, kw)
    if dtype == F32 and L.GEMM_X3 and Cout is not None and X3_CACHE:
        X3_CACHE.pop((Cout.data_ptr(), M, N, ldc), None)          # the generic kernel rewrites Cout and leaves no operand image
    a = igemm_args(M, N, K, A, lda, B, ldb, Cout, ldc, **kw)
    conv, trans = kw.get('conv'), kw.get('trans', 0)
    if (dtype == BF16 and kw.get('out_f32') and not trans and kw.get('res') is None and (kw.get('mask') is None or kw.get('mask_bits'))
            and kw.get('act', ACT_NONE) != ACT_SIGMOID):
        # an f32 OUTPUT of bf16 operands (SP-SEDT's feature_align head: 12,000 x 2048 at C4): the LDS-DMA kernels write f32 through
        # their f32 epilogue (SedtIgemm.f32ep); asked for only when the dispatcher says the problem is inside their envelope
        a.f32ep = 1
        buf = C.create_string_buffer(96)
        if L.load().sedt_igemm_describe(C.byref(a), L.gemm_dtype(dtype), 0, buf, 96) != 0 or not buf.value.decode().startswith('igemm3'):
            a.f32ep = 0
    if _co['on'] and POOL.gemms and not trans and dtype == BF16:
        riders = POOL.take(((M + 63) // 64) * ((N + 63) // 64))
        arr = (L.SedtIgemm * len(riders))(*[r[0] for r in riders])
        taken = C.c_int(0)
        L.check(L.load().sedt_igemm_co(C.byref(a), arr, len(riders), L.gemm_dtype(dtype), L.stream_ptr(), C.byref(taken)), 'sedt_igemm_co')
        if not taken.value:
            POOL.give_back(riders)
        return
    if PROFILE is not None:     # the operand tensors are kept alive so that the launch can be replayed for timing
        PROFILE.append((a, L.gemm_dtype(dtype), (M, N, K, trans, 0 if conv is None else 1), (A, B, Cout, kw), PROFILE_HINT))
    if L.LAUNCH_LOG is not None:     # (lib.launch_log(): also which kernel instance the dispatcher picks for this problem)
        buf = C.create_string_buffer(160)
        if L.load().sedt_igemm_describe(C.byref(a), L.gemm_dtype(dtype), 0, buf, 160) == 0:
            L.LAUNCH_LOG['igemm:' + buf.value.decode().split('(')[0]] += 1
    L.check(L.load().sedt_igemm(C.byref(a), L.gemm_dtype(dtype), L.stream_ptr()), 'sedt_igemm')


def _geom_tuple(g, transposed=False):
    if transposed:   # rows run over the conv INPUT grid, gathered tensor is dY on the conv OUTPUT grid
        return (g.Ho, g.Wo, g.Co, g.Hi, g.Wi, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, g.dh, g.dw)
    return (g.Hi, g.Wi, g.Ci, g.Ho, g.Wo, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, g.dh, g.dw)


def linear(dtype, x, w, out=None, *, bias=None, out_f32=False, **ep):
    """y[M,N] = epilogue(x[M,K] @ w[N,K]^T); x/w may be row-strided views (last dim contiguous)"""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32 if out_f32 else TORCH_DTYPE[dtype])
    igemm(dtype, M, N, K, x, x.stride(0), w, w.stride(0), out, out.stride(0), bias=bias, out_f32=int(out_f32), **ep)
    return out


def skinny_ok(N, K):
    """envelope of the direct small-N linear kernels (csrc/skinny.hip)"""
    return N <= 16 and K % 64 == 0 and K <= 1024


def skinny_linear_fwd(dtype, x, w, bias=None, act=ACT_NONE, out_f32=False):
    """y = act(x @ w^T + b) for N <= 16 output features straight from the f32 master weight (no packing)"""
    _dev_check(x, w)
    M, K = x.shape
    N = w.shape[0]
    assert w.dtype == torch.float32 and w.is_contiguous() and x.stride(1) == 1
    y = torch.empty((M, N), device=x.device, dtype=torch.float32 if out_f32 else TORCH_DTYPE[dtype])
    L.check(L.load().sedt_skinny_linear_fwd(_p(x), x.stride(0), _p(w), _p(bias), _p(y), y.stride(0), M, N, K, act, int(out_f32), dtype,
                                            L.stream_ptr()), 'skinny_linear_fwd')
    return y


def skinny_linear_bwd(dtype, g, ysaved, w, x, act=ACT_NONE, mask=None, need_gx=True, need_gw=True, need_gb=True, gx_acc=None, batch=None):
    """(gx, dW, db) of skinny_linear_fwd; g f32 [M,N]; ysaved = the f32 output when act != NONE.  x may be a strided row view
    (stride(1) == 1).  gx_acc: a [M,K] (row-strided) view the input gradient is ADDED to instead of a fresh gx.
    batch (a ReduceBatch): the final sum of the 16 partial slabs rides in batch.flush() (dW / db are valid only then)"""
    M, K = x.shape
    N = w.shape[0]
    assert g.dtype == torch.float32 and g.is_contiguous() and (ysaved is None or (ysaved.dtype == torch.float32 and ysaved.is_contiguous()))
    assert x.stride(1) == 1
    if gx_acc is not None:
        assert gx_acc.shape == (M, K) and gx_acc.stride(1) == 1 and gx_acc.dtype == TORCH_DTYPE[dtype] and mask is None
        gx = gx_acc
    else:
        gx = torch.empty((M, K), device=x.device, dtype=TORCH_DTYPE[dtype]) if need_gx else None
    dw = torch.empty((N, K), device=x.device, dtype=torch.float32) if need_gw else None
    db = torch.empty((N,), device=x.device, dtype=torch.float32) if (need_gb and need_gw) else None
    lib = L.load()
    scratch = torch.empty((lib.sedt_skinny_linear_bwd_scratch(K) // 4,), device=x.device, dtype=torch.float32) if need_gw else None
    deferred = need_gw and batch is not None
    L.check(lib.sedt_skinny_linear_bwd(_p(g), _p(ysaved), g.stride(0), _p(w), _p(x), x.stride(0), _p(mask),
                                       mask.stride(0) if mask is not None else 0, _p(gx), gx.stride(0) if gx is not None else K,
                                       None if deferred else _p(dw), None if deferred else _p(db),
                                       _p(scratch), M, N, K, act, int(gx_acc is not None), dtype, L.stream_ptr()), 'skinny_linear_bwd')
    if deferred:                                  # sum of the 16 slabs [17][K] as a column-sum job of the batch's reduce launch
        tot = torch.empty((17 * K,), device=x.device, dtype=torch.float32)
        batch.add_colsum(scratch, 16, 17 * K, tot)
        dw = tot[:N * K].view(N, K)
        db = tot[16 * K:16 * K + N] if need_gb else None
    if need_gb and not need_gw:
        db = g.sum(0) if act == ACT_NONE else (g * ysaved * (1 - ysaved)).sum(0) if act == ACT_SIGMOID else (g * (ysaved > 0)).sum(0)
    return gx, dw, db


def _group_label(jobs):
    """bench.py's recorded step runs a group's problems one by one; this is the kernel instance the UN-recorded step launches them on as one
    group (sedt_igemm_group_describe), so that roofline.families books their flops on the row that actually ran"""
    arr = (L.SedtIgemm * len(jobs))(*jobs)
    buf = C.create_string_buffer(160)
    if L.load().sedt_igemm_group_describe(arr, len(jobs), BF16, buf, 160) != 0:
        return None
    return buf.value.decode() or None


def linear_group(dtype, items):
    """several independent ``linear`` calls (each item: (x, w, kwargs of linear)) as one launch when the kernel allows it
    (sedt_igemm_group) - the q / k / v projections of an attention block, or their three dgrads.  Returns the outputs."""
    outs, args = [], []
    for x, w, kw in items:
        kw = dict(kw)
        M, K = x.shape
        N = w.shape[0]
        out_f32 = kw.pop('out_f32', False)
        out = kw.pop('out', None)
        if out is None:
            out = torch.empty((M, N), device=x.device, dtype=torch.float32 if out_f32 else TORCH_DTYPE[dtype])
        outs.append(out)
        args.append(((M, N, K, x, x.stride(0), w, w.stride(0), out, out.stride(0)), dict(out_f32=int(out_f32), **kw)))
    if len(args) == 1 or PROFILE is not None or (_co['on'] and POOL.gemms) or (dtype == F32 and L.GEMM_X3 and X3_FAST):
        global PROFILE_HINT
        if len(args) > 1 and PROFILE is not None and dtype == BF16:
            PROFILE_HINT = ('group', _group_label([igemm_args(*a, **kw) for a, kw in args]), 1.0)
        for a, kw in args:
            igemm(dtype, *a, **kw)
        PROFILE_HINT = None
        return outs
    arr = (L.SedtIgemm * len(args))(*[igemm_args(*a, **kw) for a, kw in args])
    L.check(L.load().sedt_igemm_group(arr, len(args), L.gemm_dtype(dtype), L.stream_ptr()), 'igemm_group')
    return outs


def _conv3_c64_ok(dtype, t, g, ep, out, profiling_ok=False):
    """envelope of the direct 3x3 kernel (csrc/conv3x3_c64.hip): the layer1 conv2 geometry, plain epilogues"""
    return (CONV3_DIRECT and (PROFILE is None or profiling_ok) and dtype == BF16 and g.Ci == 64 and g.Co == 64 and g.KH == 3 and g.KW == 3
            and g.sh == 1 and g.sw == 1 and g.ph == 1 and g.pw == 1 and g.dh == 1 and g.dw == 1 and g.Wi == 16
            and t.stride(0) == 64 and out.stride(0) == 64 and set(ep) <= {'scale', 'bias', 'act', 'mask', 'ldm', 'tile'}
            and ep.get('act', 0) in (0, ACT_RELU) and tuple(ep.get('tile', (0, 0))) == (0, 0)
            and (ep.get('mask') is None or (ep.get('ldm', 64) == 64 and ep.get('act', 0) == 0)))


def _conv3_c64(t, B, g, w, flip, out, ep):
    L.check(L.load().sedt_conv3x3_c64(_p(t), _p(w), flip, _p(ep.get('scale')), _p(ep.get('bias')), 1 if ep.get('act', 0) == ACT_RELU else 0,
                                      _p(ep.get('mask')), _p(out), B, g.Hi, g.Wi, 64, L.stream_ptr()), 'conv3x3_c64')
    return out


def conv_fwd(dtype, x, B, g, wf, out=None, **ep):
    """NHWC conv forward: x [B*Hi*Wi, Ci] (row stride = x.stride(0)), wf packed [Co][taps][Ci]"""
    M = B * g.Ho * g.Wo
    if out is None:
        out = torch.empty((M, g.Co), device=x.device, dtype=TORCH_DTYPE[dtype])
    if _conv3_c64_ok(dtype, x, g, ep, out):
        return _conv3_c64(x, B, g, wf, 0, out, ep)
    if _dil_halves_ok(dtype, x, g, wf, out, ep, False):
        return _conv_dil_halves(x, B, g, wf, out, ep, False, dtype == F32)
    conv = None if g.plain else _geom_tuple(g)
    global PROFILE_HINT
    if PROFILE is not None and _conv3_c64_ok(dtype, x, g, ep, out, True):
        PROFILE_HINT = 'conv3x3_c64_kernel'
    elif PROFILE is not None and dtype == BF16 and _dil_halves_ok(dtype, x, g, wf, out, ep, False, True):
        # the un-recorded step runs this conv as two column halves in one grouped launch, 6 of the 9 taps each (2/3 of the flops)
        PROFILE_HINT = ('group', _conv_dil_halves(x, B, g, wf, out, ep, False, False, describe=True), 2.0 / 3.0)
    igemm(dtype, M, g.Co, g.taps * g.Ci, x, x.stride(0), wf, g.taps * g.Ci, out, out.stride(0), conv=conv, **ep)
    PROFILE_HINT = None
    return out


# The input gradient of a stride-2 3x3 convolution (layer2 / layer3 block 0) by OUTPUT PARITY.  The transposed gather of the plain form walks
# all nine taps for every input pixel, but a pixel (hi, wi) only ever receives taps with kh = hi + 1 and kw = wi + 1 (mod 2): three of four
# (tap, pixel) pairs multiply rows of zeros.  Split by the parity of (hi, wi) the problem is four small REGULAR convolutions over dY - 1, 2,
# 2 and 4 taps - whose outputs interleave in dx: one grouped launch (sedt_igemm_group) of four problems that read their tap blocks of the
# packed dgrad weight in place (SedtIgemm.btap) and write every second row / column of dx (SedtIgemm.omap), epilogue operands (residual,
# ReLU mask bits) indexed by the dx pixel.  A quarter of the MFMAs, the same sums in the same order.
S2_PARITY = _dev_env('SEDT_S2_PARITY', '1') != '0'


def _dgrad_s2_ok(dtype, dy, g, wb, out, ep, profiling_ok=False):
    """bf16, or the fast bf16x3 form of the f32 mode (split operand images, f32 epilogue)"""
    x3 = dtype == F32 and L.GEMM_X3 and X3_FAST
    if not (dtype == BF16 or x3):
        return False
    if x3 and not (dy.dtype == torch.float32 and out.dtype == torch.float32 and dy.data_ptr() % 16 == 0 and dy.stride(0) % 4 == 0
                   and (ep.get('res') is None or ep['res'].dtype == torch.float32)
                   and (ep.get('mask') is None or ep.get('mask_bits') or ep['mask'].dtype == torch.float32)):
        return False
    return (S2_PARITY and (PROFILE is None or profiling_ok) and not _co['on'] and g.KH == 3 and g.KW == 3 and g.sh == 2 and g.sw == 2
            and g.ph == 1 and g.pw == 1 and g.dh == 1 and g.dw == 1 and g.Co % 64 == 0 and g.Ci % 8 == 0 and wb.stride(0) == 9 * g.Co
            and dy.stride(0) % 8 == 0 and out.stride(0) % 8 == 0 and set(ep) <= {'mask', 'ldm', 'mask_bits', 'res', 'ldr', 'alpha'})


def _conv_dgrad_s2(dy, B, g, wb, out, ep, x3=False, describe=False):
    """describe: build the four problems and return the label of the kernel their grouped launch runs on (nothing is launched)"""
    Co = g.Co
    if x3 and not describe:                                               # operand images: dY [pixels][2 Co] = [hi | lo], the weight [Ci][9][3 Co] = [hi | hi | lo] per tap
        dy, wb = _split3([(dy, 0, B * g.Ho * g.Wo, g.Co, dy.stride(0), 0), (wb, 0, g.Ci * 9, g.Co, g.Co, 1)])
        wb = wb.view(g.Ci, 27 * g.Co)
        Co = 3 * g.Co
    jobs = []
    for ph_ in (0, 1):                                   # parity of the input row hi = 2a + ph_
        nh, khs = (g.Hi + 1 - ph_) // 2, ([1] if ph_ == 0 else [0, 2])
        for pw_ in (0, 1):
            nw, kws = (g.Wi + 1 - pw_) // 2, ([1] if pw_ == 0 else [0, 2])
            if nh == 0 or nw == 0:
                continue
            # a "transposed" stride-1 gather over the dY grid: tap (i, j) of the walk reads dY[a + ph_ - i][b + pw_ - j], i.e. the original
            # taps in ascending (kh, kw) order - rows / columns beyond the grid are masked by the kernel
            conv = (g.Ho, g.Wo, Co, nh, nw, len(khs), len(kws), 1, 1, ph_, pw_, 1, 1)
            taps = [(kh, kw) for kh in khs for kw in kws]
            a = igemm_args(B * nh * nw, g.Ci, len(taps) * Co, dy, dy.stride(0), wb, wb.stride(0), out, out.stride(0), conv=conv,
                           transposed=1, tile=(64, 64), out_f32=int(x3), **ep)
            a.omap, a.o_Hi, a.o_Wi, a.o_sh, a.o_sw, a.o_h0, a.o_w0 = 1, g.Hi, g.Wi, 2, 2, ph_, pw_
            a.btap_on, a.f32ep, a.awrap = 1, int(x3), (g.Co if x3 else 0)
            for t_, (kh, kw) in enumerate(taps):
                a.btap[t_] = (kh * 3 + kw) * Co
            jobs.append(a)
    if describe:
        return _group_label(jobs)
    arr = (L.SedtIgemm * len(jobs))(*jobs)
    L.check(L.load().sedt_igemm_group(arr, len(jobs), BF16, L.stream_ptr()), 'igemm_group_s2')
    return out


# layer4's dilated 3x3 convolutions (dilation 2, padding 2) run on a map of FOUR columns: the left tap column (kw = 0) lands inside the image
# only for output columns 2, 3 and the right one (kw = 2) only for columns 0, 1 - a third of the tap walk multiplies zero rows.  Split by
# column half, each half is a regular 3 x 2-tap convolution over all four input columns with its own six tap blocks of the packed
# weight (SedtIgemm.btap) and its own two output columns (SedtIgemm.omap): two problems of 128 tiles of 128x128 in ONE grouped ping-pong
# launch (igemm3_w8_group_kernel), 2/3 of the K loop each.  Forward and input gradient (the mirror image) alike.
DIL_HALVES = _dev_env('SEDT_DIL_HALVES', '1') != '0'


def _dil_halves_ok(dtype, t, g, w, out, ep, transposed, profiling_ok=False):
    x3 = dtype == F32 and L.GEMM_X3 and X3_FAST
    if not (DIL_HALVES and (dtype == BF16 or x3) and (PROFILE is None or profiling_ok) and not _co['on']):
        return False
    d = g.dh
    cin, cout = (g.Co, g.Ci) if transposed else (g.Ci, g.Co)          # channels of the gathered tensor / of the output
    if not (g.KH == 3 and g.KW == 3 and g.sh == 1 and g.sw == 1 and g.dw == d and d >= 1 and g.ph == d and g.pw == d and g.Wi == 2 * d
            and g.Wo == g.Wi and g.Ho == g.Hi and cin % 64 == 0 and cout % 128 == 0 and w.stride(0) == 9 * cin and t.stride(0) % 8 == 0
            and out.stride(0) % 8 == 0):
        return False
    if x3 and not (t.dtype == torch.float32 and out.dtype == torch.float32 and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0
                   and all(ep.get(k_) is None or ep[k_].dtype == torch.float32 for k_ in ('res',))
                   and (ep.get('mask') is None or ep.get('mask_bits') or ep['mask'].dtype == torch.float32)):
        return False
    # (the halves pay while the two problems fill the chip with 128x128 tiles)
    B_rows = t.shape[0]
    return set(ep) <= {'scale', 'bias', 'act', 'mask', 'ldm', 'mask_bits', 'res', 'ldr', 'alpha', 'bits_out', 'act_post_res'} \
        and (B_rows // 2 // 128) * (cout // 128) >= 96


def _conv_dil_halves(t, B, g, w, out, ep, transposed, x3, describe=False):
    """t = x (forward) or dY (input gradient); w = the packed forward / dgrad operand [cout][9][cin]"""
    d = g.dh
    cin, cout = (g.Co, g.Ci) if transposed else (g.Ci, g.Co)
    C = cin
    if x3 and not describe:
        t, w = _split3([(t, 0, B * g.Hi * g.Wi, cin, t.stride(0), 0), (w, 0, cout * 9, cin, cin, 1)])
        w = w.view(cout, 27 * cin)
        C = 3 * cin
    jobs = []
    for half in (0, 1):
        # forward: output columns [0, d) see taps kw in {1, 2}, columns [d, 2d) taps {0, 1}; the input gradient mirrors it
        kw0 = (1 - half) if not transposed else half
        conv = (g.Hi, g.Wi, C, g.Ho, d, 3, 2, 1, 1, d, (d if transposed else 0), d, d)
        a = igemm_args(B * g.Ho * d, cout, 6 * C, t, t.stride(0), w, w.stride(0), out, out.stride(0), conv=conv, transposed=int(transposed),
                       tile=(128, 128), out_f32=int(x3), **ep)
        a.omap, a.o_Hi, a.o_Wi, a.o_sh, a.o_sw, a.o_h0, a.o_w0 = 1, g.Ho, g.Wo, 1, 1, 0, half * d
        a.btap_on, a.f32ep, a.awrap = 1, int(x3), (cin if x3 else 0)
        for kh in range(3):
            for j in range(2):
                a.btap[kh * 2 + j] = (kh * 3 + kw0 + j) * C
        jobs.append(a)
    if describe:
        return _group_label(jobs)
    arr = (L.SedtIgemm * 2)(*jobs)
    L.check(L.load().sedt_igemm_group(arr, 2, BF16, L.stream_ptr()), 'igemm_group_dil')
    return out


def conv_dgrad(dtype, dy, B, g, wb, out=None, **ep):
    """dx [B*Hi*Wi, Ci] = conv_transpose(dy [B*Ho*Wo, Co]); wb packed [Ci][taps][Co] (BN scale folded in)"""
    M = B * g.Hi * g.Wi
    if out is None:
        out = torch.empty((M, g.Ci), device=dy.device, dtype=TORCH_DTYPE[dtype])
    if _dgrad_s2_ok(dtype, dy, g, wb, out, ep):
        return _conv_dgrad_s2(dy, B, g, wb, out, ep, x3=dtype == F32)
    if _dil_halves_ok(dtype, dy, g, wb, out, ep, True):
        return _conv_dil_halves(dy, B, g, wb, out, ep, True, dtype == F32)
    if _conv3_c64_ok(dtype, dy, g, ep, out) and 'scale' not in ep and 'bias' not in ep:
        return _conv3_c64(dy, B, g, wb, 1, out, ep)       # the input gradient of a stride-1 3x3 conv is the same conv, taps flipped
    conv = None if g.plain else _geom_tuple(g, transposed=True)
    global PROFILE_HINT
    if PROFILE is not None and _conv3_c64_ok(dtype, dy, g, ep, out, True) and 'scale' not in ep and 'bias' not in ep:
        PROFILE_HINT = 'conv3x3_c64_kernel'
    elif PROFILE is not None and dtype == BF16 and _dgrad_s2_ok(dtype, dy, g, wb, out, ep, True):
        # stride-2 input gradient by output parity: four problems of 1 + 2 + 2 + 4 taps over a quarter of the pixels each = 1/4 of the walk
        PROFILE_HINT = ('group', _conv_dgrad_s2(dy, B, g, wb, out, ep, describe=True), 0.25)
    elif PROFILE is not None and dtype == BF16 and _dil_halves_ok(dtype, dy, g, wb, out, ep, True, True):
        PROFILE_HINT = ('group', _conv_dil_halves(dy, B, g, wb, out, ep, True, False, describe=True), 2.0 / 3.0)
    igemm(dtype, M, g.Ci, g.taps * g.Co, dy, dy.stride(0), wb, g.taps * g.Co, out, out.stride(0), conv=conv,
          transposed=0 if g.plain else 1, **ep)
    PROFILE_HINT = None
    return out


def _fused_bias_ok(dtype, dy, x, g):
    """envelope of the bf16 LDS-DMA wgrad kernel (csrc/wgrad3.hip / wgrad4.hip, envelope in wgrad_lds_envelope): only there the bias gradient rides along"""
    return (dtype == BF16 and L.load() is not None and g.Co % 8 == 0 and (g.taps * g.Ci) % 8 == 0 and dy.stride(0) % 8 == 0
            and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and g.Ci % 8 == 0)


class ReduceBatch(object):
    """collects the split-K reductions of several wgrads and issues them as ONE launch (flush); keeps the slabs alive.
    Inside a runtime.async_wgrad() scope the wgrad GEMMs themselves are deferred too: flush() then issues all of them
    plus the reduction as ONE block on the side stream (one fork per layer instead of one per GEMM)."""

    def __init__(self, shared=False):
        """shared: this batch outlives the backward of the layer that fills it (WgradShare): it then holds NO reference to the gradient
        tensors - autograd's AccumulateGrad steals a gradient whose reference count is 1 and CLONES one that is still referenced, and a
        clone taken before the deferred launch has written the tensor would be garbage"""
        self.jobs, self.keep, self.deferred, self.operands = [], [], [], []
        self.group, self.group_dtype = [], None
        self.shared = shared
        self.expect = []            # shared: (parameter, data_ptr of the gradient tensor handed to autograd) - checked at the final flush

    def add_gemm(self, dtype, args, shape, code=None):
        """a weight-gradient GEMM to be issued with the others of this batch in one grouped launch.  code: the dtype code the entry point
        gets when it is not gemm_dtype(dtype) - the fast bf16x3 problems are bf16 problems on split operands"""
        self.group.append((args, shape, L.gemm_dtype(dtype) if code is None else code))
        self.group_dtype = dtype

    def _launch_group(self):
        if not self.group:
            return
        lib = L.load()
        if PROFILE is not None:                       # bench.py replays every GEMM on its own for the per-launch timing
            for a, shape, code in self.group:
                L.check(lib.sedt_igemm(C.byref(a), code, L.stream_ptr()), 'sedt_igemm')
                PROFILE.append((a, code, shape, (list(self.keep), list(self.operands)), 'wgrad_group'))
        else:
            i = 0
            while i < len(self.group):                # runs of equal dtype code: one grouped launch each
                j = i
                while j < len(self.group) and self.group[j][2] == self.group[i][2]:
                    j += 1
                arr = (L.SedtIgemm * (j - i))(*[g[0] for g in self.group[i:j]])
                L.check(lib.sedt_wgrad_group(arr, j - i, self.group[i][2], L.stream_ptr()), 'wgrad_group')
                i = j
        self.group = []

    def defer(self, body, operands):
        self.deferred.append(body)
        self.operands.extend(t for t in operands if t is not None)

    def add(self, slab, sk, R, taps, Ci, rowscale, out, cs, bias_out, cs_splitk=0):
        j = L.SedtReduceJob()
        j.slab, j.out, j.splitk, j.R, j.taps, j.Ci = slab.data_ptr(), out.data_ptr(), sk, R, taps, Ci
        j.cs_splitk = cs_splitk
        j.rowscale = rowscale.data_ptr() if rowscale is not None else None
        j.colsum_slab = cs.data_ptr() if cs is not None else None
        j.bias_out = bias_out.data_ptr() if (cs is not None and bias_out is not None) else None
        self.jobs.append(j)
        self.keep.append((slab, cs, rowscale, None, None) if self.shared else (slab, cs, rowscale, out, bias_out))
        if len(self.jobs) == L.MAX_REDUCE_JOBS and not _co['on'] and not self.shared:
            self._launch_group()
            self._launch()

    def add_colsum(self, partial, nrows, ncols, out):
        """out[c] = sum_r partial[r][c] in the same launch as the split-K reductions (LayerNorm gamma/beta partials)"""
        j = L.SedtReduceJob()
        j.colsum_slab, j.bias_out, j.splitk, j.R, j.taps, j.Ci = partial.data_ptr(), out.data_ptr(), nrows, ncols, 1, 0
        self.jobs.append(j)
        self.keep.append((partial, None, None, None if self.shared else out, None))

    def _launch(self):
        for i in range(0, len(self.jobs), L.MAX_REDUCE_JOBS):
            chunk = self.jobs[i:i + L.MAX_REDUCE_JOBS]
            arr = (L.SedtReduceJob * len(chunk))(*chunk)
            pf = None
            if self.prefetch is not None and RED_PREFETCH and i + L.MAX_REDUCE_JOBS >= len(self.jobs):      # (the last launch of the batch)
                pf = L.prefetch_arg(self.prefetch)
            L.check(L.load().sedt_multi_wgrad_reduce(arr, len(chunk), pf, L.stream_ptr()), 'multi_wgrad_reduce')
        self.jobs, self.keep = [], []

    prefetch = None

    def collect(self):
        """shared batches, at the end of a layer's backward: run the layer's deferred bodies NOW - they allocate the split-K slabs and
        build the argument blocks, nothing is launched (bf16x3: the operand splits are) - so that no closure keeps the gradient tensors
        referenced when the layer returns them to autograd.  The launches happen in the stack's final flush."""
        for body in self.deferred:
            body()
        self.deferred = []

    def flush(self, prefetch=None):
        """prefetch: up to three tensors (weights) the launch AFTER this batch's reduce launch will stream: that launch touches them"""
        from . import runtime
        self.prefetch = prefetch
        if _co['on'] and not runtime.async_wgrad_on():
            for body in self.deferred:              # allocate slabs / build the argument blocks; the GEMMs are not launched:
                body()                              # they ride along with later launches of the dgrad chain
            self.deferred = []
            if self.group_dtype in (None, BF16):
                # (not the gradient tensors themselves: an extra reference would make autograd clone them before they are written)
                POOL.push(self.group, BF16, self.jobs, [k[:3] for k in self.keep], self.operands)
                self.group, self.jobs, self.keep, self.operands = [], [], [], []
                return
        if self.deferred:
            with runtime.side(*self.operands):      # (the current stream itself unless runtime.async_wgrad is on)
                for body in self.deferred:
                    body()
                self._launch_group()
                self._launch()
            self.deferred, self.operands = [], []
        else:
            self._launch_group()
            self._launch()


DEFER_LAYER_WGRADS = _dev_env('SEDT_DEFER_LAYER_WGRADS', '1') != '0'
_defer = {'on': False}


class defer_layer_wgrads(object):
    """scope (the captured steppers' bodies): the weight gradients of ALL layers of a transformer stack are issued as ONE grouped GEMM
    launch + ONE reduce launch when the backward leaves the stack, instead of two launches per layer (three decoder layers at M = 704
    rows: 6 latency-bound launches -> 2; six per-op encoder layers of the B = 32 configurations: 12 -> 2).  Only inside a stepper: the
    gradients a layer returns are written later in stream order, which a consumer that reads them at AccumulateGrad time (torch DDP's
    hooks on the eager path) must not see."""

    def __init__(self, enable=True):
        self.enable = bool(enable) and DEFER_LAYER_WGRADS

    def __enter__(self):
        self.prev = _defer['on']
        _defer['on'] = self.enable
        return self

    def __exit__(self, *exc):
        _defer['on'] = self.prev
        return False


class WgradShare(object):
    """the ReduceBatch shared by the layers of one stack in one forward / backward pass (functional.EncoderLayerFn / DecoderLayerFn,
    cfg['wg_share']); None-safe helpers keep the layers' code paths identical with and without it"""

    def __init__(self):
        self.rb = None
        self.mark = 0               # rb.expect[:mark]: gradients of layers whose AccumulateGrad nodes have run by now

    @staticmethod
    def make():
        return WgradShare() if (_defer['on'] and not _co['on']) else None

    def batch(self):
        if self.rb is None:
            self.rb = ReduceBatch(shared=True)
        return self.rb

    def layer_done(self, last, prefetch=None):
        """end of a layer's backward; last: this is the last layer of the stack to run (layer 0)"""
        rb = self.rb
        if rb is None:
            return
        rb.collect()
        if last:
            for param, ptr in rb.expect[:self.mark]:          # (this layer's own gradients reach AccumulateGrad after it returns)
                g = param.grad                                # None: torch.autograd.grad captured the tensor itself (the DP segments)
                if g is not None and g.data_ptr() != ptr:
                    raise RuntimeError('a deferred weight gradient was cloned or accumulated by autograd before it was written: '
                                       'defer_layer_wgrads needs parameters whose .grad is None when the backward starts')
            rb.expect = []
            rb.flush(prefetch=prefetch)
            self.rb, self.mark = None, 0
        else:
            self.mark = len(rb.expect)


GRAD_SINK = None      # {parameter data_ptr: f32 view}: where the weight gradient of that parameter is to be written (grad_sink())


class grad_sink(object):
    """inside the scope, weight-gradient kernels called with ``param=`` write into the view registered for that parameter - the data-
    parallel steppers register the parameters' slots in the flat gradient buffer (FusedAdamW.flat_views), which spares the
    130 MB gather pass for the tensors that are produced this way"""

    def __init__(self, views):
        self.views = dict(views) if views else None      # own copy: a slot is handed out ONCE per scope (see _sink)

    def __enter__(self):
        global GRAD_SINK
        self.prev, GRAD_SINK = GRAD_SINK, self.views
        return self

    def __exit__(self, *a):
        global GRAD_SINK
        GRAD_SINK = self.prev
        return False


def _sink(param, shape):
    if GRAD_SINK is None or param is None:
        return None
    # pop: a parameter that receives a second contribution in the same backward (a module applied twice) gets an ordinary
    # buffer for it - autograd then sums the two into a new tensor and the gather pass copies that one
    v = GRAD_SINK.pop(param.data_ptr(), None)
    return v.view(shape) if v is not None and v.numel() == int(np.prod(shape)) else None


def _x3_wgrad_ok(dy, x, g, B):
    """envelope of the LDS-DMA weight-gradient kernels (wgrad_lds_envelope + wgrad3_conv_ok, csrc/wgrad3.hip) on the split operands"""
    if not X3_FAST or dy.dtype != torch.float32 or x.dtype != torch.float32:
        return False
    Co, Ci = g.Co, g.Ci
    if Co % 8 or Ci % 8 or dy.stride(0) % 4 or x.stride(0) % 4 or dy.data_ptr() % 16 or x.data_ptr() % 16:
        return False
    if not g.plain and ((64 % g.Wo) != 0 or g.Ho * g.Wo < 64 or g.Ho < 64 // g.Wo):
        return False
    return B * g.Ho * g.Wo * 3 * Co * 2 < (1 << 31) and B * g.Hi * g.Wi * 3 * Ci * 2 < (1 << 31)


def wgrad(dtype, dy, x, B, g, rowscale=None, out=None, bias_out=None, batch=None, param=None):
    """dW (Co, Ci, KH, KW) f32 = sum over pixels dy[pix][co] * gather(x)[pix][tap][ci] (* rowscale[co]).
    bias_out (f32 [Co]): also receives sum over pixels of dy (fused into the wgrad kernel when it can, else a colsum).
    batch: a ReduceBatch - the split-K reduction is deferred until batch.flush() (the result is valid only then).
    param: the parameter this is the gradient of - with a grad_sink() active the result lands in the registered view"""
    from . import runtime
    lib = L.load()
    Mo, No, Kp = g.Co, g.taps * g.Ci, B * g.Ho * g.Wo
    sk = lib.sedt_igemm_splitk(Mo, No, Kp, dtype)
    own_out = False                     # allocated here (not the caller's buffer, not a slot of the flat gradient buffer)
    if out is None:
        out = _sink(param, (g.Co, g.Ci, g.KH, g.KW))
        if out is None:
            out = torch.empty((g.Co, g.Ci, g.KH, g.KW), device=dy.device, dtype=torch.float32)
            own_out = True

    def body():
        conv = None if g.plain else _geom_tuple(g)
        if dtype == F32 and L.GEMM_X3 and _x3_wgrad_ok(dy, x, g, B):
            # fast bf16x3 (csrc/split3.hip): both operands split once, hi hi + lo hi as ONE problem over [dY_hi | dY_lo], hi lo as a second,
            # 3 sk slabs of [Co][taps Ci] for the reduce launch
            sk3 = max(1, lib.sedt_igemm_splitk(2 * Mo, No, Kp, BF16))
            rows_x = B * g.Hi * g.Wi
            dy3, x3 = _split3([(dy, 0, Kp, Mo, dy.stride(0), 0), (x, 0, rows_x, g.Ci, x.stride(0), 0)])
            slab = torch.empty((3 * sk3, Mo, No), device=dy.device, dtype=torch.float32)
            cs = torch.empty((2 * sk3, Mo), device=dy.device, dtype=torch.float32) if bias_out is not None else None
            a1 = igemm_args(2 * Mo, No, Kp, dy3, 2 * Mo, x3, 2 * g.Ci, slab, No, trans=1, conv=conv, out_f32=1, splitk=sk3, slab=slab,
                            colsum_out=cs)
            x3lo, slab2 = x3[:, g.Ci:2 * g.Ci], slab[2 * sk3:]
            a2 = igemm_args(Mo, No, Kp, dy3, 2 * Mo, x3lo, 2 * g.Ci, slab2, No, trans=1, conv=conv, out_f32=1, splitk=sk3, slab=slab2)
            tb = batch if batch is not None else ReduceBatch()
            tb.add_gemm(dtype, a1, (2 * Mo, No, Kp, 1, 0 if conv is None else 1), code=BF16)
            tb.add_gemm(dtype, a2, (Mo, No, Kp, 1, 0 if conv is None else 1), code=BF16)
            tb.add(slab, 3 * sk3, Mo, g.taps, g.Ci, rowscale, out, cs, bias_out, cs_splitk=2 * sk3)
            tb.keep.append((dy3, x3, None, None, None))
            if batch is None:
                tb._launch_group()
                tb._launch()
            return
        slab = torch.empty((sk, Mo, No), device=dy.device, dtype=torch.float32)
        fused = bias_out is not None and _fused_bias_ok(dtype, dy, x, g)
        cs = torch.empty((sk, Mo), device=dy.device, dtype=torch.float32) if fused else None
        if batch is not None:
            batch.add_gemm(dtype, igemm_args(Mo, No, Kp, dy, dy.stride(0), x, x.stride(0), slab, No, trans=1, conv=conv,
                                             out_f32=1, splitk=sk, slab=slab, colsum_out=cs),
                           (Mo, No, Kp, 1, 0 if conv is None else 1))
            batch.add(slab, sk, Mo, g.taps, g.Ci, rowscale, out, cs, bias_out if fused else None)
        else:
            igemm(dtype, Mo, No, Kp, dy, dy.stride(0), x, x.stride(0), slab, No, trans=1, conv=conv, out_f32=1, splitk=sk,
                  slab=slab, colsum_out=cs)
            L.check(lib.sedt_wgrad_reduce_bias(_p(slab), sk, Mo, g.taps, g.Ci, _p(rowscale), _p(out), _p(cs),
                                               _p(bias_out) if fused else None, L.stream_ptr()), 'wgrad_reduce')
        if bias_out is not None and not fused:
            colsum(dtype, dy, out=bias_out)

    if batch is not None:
        batch.defer(body, (dy, x, rowscale))              # issued by batch.flush(): one grouped launch for the whole layer
        if batch.shared and param is not None and own_out and tuple(param.shape) == tuple(out.shape[:param.dim()]):
            batch.expect.append((param, out.data_ptr()))      # (checked at the stack's final flush: autograd must have STOLEN this tensor)
    else:
        with runtime.side(dy, x, rowscale):               # off the dgrad critical path when runtime.async_wgrad is on
            body()
    return out


def linear_wgrad(dtype, dy, x, out=None, bias_out=None, batch=None, param=None):
    """dW [N,K] f32 = dy[M,N]^T @ x[M,K]; bias_out f32 [N] optionally receives the column sums of dy"""
    N, K = dy.shape[1], x.shape[1]
    g = ConvGeom(1, 1, K, N)
    return wgrad(dtype, dy, x, dy.shape[0], g, out=out.view(N, K, 1, 1) if out is not None else None,
                 bias_out=bias_out, batch=batch, param=param).view(N, K)


def colsum(dtype, x, out=None):
    """column sums of x[rows, cols] (compute dtype or f32) -> f32[cols]"""
    lib = L.load()
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols,), device=x.device, dtype=torch.float32)
    nb = lib.sedt_colsum_scratch(rows, cols)
    scratch = torch.empty((max(nb // 4, 1),), device=x.device, dtype=torch.float32)
    L.check(lib.sedt_colsum(_p(x), x.stride(0), rows, cols, int(x.dtype == torch.float32), dtype, _p(out), _p(scratch),
                            nb, L.stream_ptr()), 'colsum')
    return out


def dropout_grad(dtype, g, p, seed, seed_ptr=None, out=None):
    rows, cols = g.shape
    if out is None:
        out = torch.empty((rows, cols), device=g.device, dtype=g.dtype)
    L.check(L.load().sedt_dropout_grad(_p(g), g.stride(0), _p(out), out.stride(0), rows, cols, p, seed & 0xffffffff,
                                       _p(seed_ptr), dtype, L.stream_ptr()), 'dropout_grad')
    return out


def add(dtype, a, b, b_mod=0, out=None):
    rows, cols = a.shape
    if out is None:
        out = torch.empty_like(a)
    L.check(L.load().sedt_add(_p(a), _p(b), _p(out), rows, cols, b_mod, dtype, L.stream_ptr()), 'add')
    return out


def add_n(dtype, tensors, out=None):
    """sum of equally shaped contiguous tensors of the compute dtype: one launch for up to 8 (sedt_add_n has 8 pointer slots), more are
    folded in groups of 8 with the running sum as the first source of the next group (--dec_layers >= 5 hands over 2 shares per
    layer of the query-position gradient, functional.DecoderLayerFn)"""
    _dev_check(*tensors)
    tensors = list(tensors)
    n = len(tensors)
    assert n >= 1 and all(t.shape == tensors[0].shape and t.dtype == TORCH_DTYPE[dtype] and t.is_contiguous() for t in tensors)
    while n > 8:
        tensors = [add_n(dtype, tensors[:8])] + tensors[8:]
        n = len(tensors)
    if out is None:
        out = torch.empty_like(tensors[0])
    arr = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    L.check(L.load().sedt_add_n(arr, n, _p(out), tensors[0].numel(), dtype, L.stream_ptr()), 'add_n')
    return out


def cast(x, out_dtype):
    """f32 <-> compute dtype"""
    code = {torch.float32: F32, torch.bfloat16: BF16}
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=TORCH_DTYPE[out_dtype])
    L.check(L.load().sedt_cast(_p(x), code[x.dtype], _p(out), out_dtype, x.numel(), L.stream_ptr()), 'cast')
    return out


def relu_mask(dtype, g, y):
    g, y = g.contiguous(), y.contiguous()
    out = torch.empty_like(g)
    L.check(L.load().sedt_relu_mask(_p(g), _p(y), _p(out), g.numel(), dtype, L.stream_ptr()), 'relu_mask')
    return out


def copy2d(jobs):
    """jobs: (src tensor view start, dst tensor view start, outer, inner_bytes, src_stride_bytes, dst_stride_bytes) given as
    (src_ptr, dst_ptr, outer, inner, sstride, dstride) integers; up to 8 per launch (sedt_copy2d)"""
    for base in range(0, len(jobs), 8):
        chunk = jobs[base:base + 8]
        arr = (L.SedtCopyJob * len(chunk))()
        for n, (sp, dp, outer, inner, ss, ds) in enumerate(chunk):
            arr[n].src, arr[n].dst, arr[n].outer, arr[n].inner, arr[n].src_stride, arr[n].dst_stride = sp, dp, outer, inner, ss, ds
        L.check(L.load().sedt_copy2d(arr, len(chunk), L.stream_ptr()), 'copy2d')


def spsedt_dec_in(dtype, patch, query, B, Q, P, qpp, train, ratio=0.0, keep=None, seed=0, seed_ptr=None):
    """SP-SEDT decoder input [B*Q, D] from the patch queries [B*P, D] and the query embedding rows [Q, D] (f32); returns (dec_in, keep
    [Q, B] f32 or None) - reference spsedt.py:48-69 in one launch"""
    D = patch.shape[1]
    assert patch.is_contiguous() and patch.shape[0] == B * P and query.dtype == torch.float32 and query.is_contiguous() and query.shape == (Q, D)
    out = torch.empty((B * Q, D), device=patch.device, dtype=patch.dtype)
    keep_out = torch.empty((Q, B), device=patch.device, dtype=torch.float32) if train else None
    if keep is not None:
        keep = keep.to(device=patch.device, dtype=torch.float32).reshape(Q, B).contiguous()
    L.check(L.load().sedt_spsedt_dec_in(_p(patch), _p(query), _p(keep), _p(keep_out), _p(out), B, Q, P, qpp, D, int(train),
                                        float(ratio), seed & 0xffffffff, _p(seed_ptr), dtype, L.stream_ptr()), 'spsedt_dec_in')
    return out, keep_out


def spsedt_dec_in_bwd(dtype, g, keep, B, Q, P, qpp, train, need_patch=True):
    D = g.shape[1]
    g = g.contiguous()
    d_patch = torch.empty((B * P, D), device=g.device, dtype=g.dtype) if need_patch else None
    d_query = torch.empty((Q, D), device=g.device, dtype=torch.float32)
    L.check(L.load().sedt_spsedt_dec_in_bwd(_p(g), _p(keep), _p(d_patch), _p(d_query), B, Q, P, qpp, D, int(train), dtype, L.stream_ptr()),
            'spsedt_dec_in_bwd')
    return d_patch, d_query


def gelu_fwd(dtype, h, p=0.0, seed=0, seed_ptr=None):
    """a = dropout(gelu(h)) in one launch (reference transformer.py:423-431 'gelu' + the FFN dropout); h: the linear1 output"""
    h = h.contiguous()
    a = torch.empty_like(h)
    L.check(L.load().sedt_gelu_fwd(_p(h), _p(a), h.numel(), p, seed & 0xffffffff, _p(seed_ptr), dtype, L.stream_ptr()), 'gelu_fwd')
    return a


def gelu_bwd(dtype, g, h, p=0.0, seed=0, seed_ptr=None):
    """gradient wrt the pre-activation h of a = dropout(gelu(h)), given the gradient wrt a"""
    g, h = g.contiguous(), h.contiguous()
    out = torch.empty_like(g)
    L.check(L.load().sedt_gelu_bwd(_p(g), _p(h), _p(out), g.numel(), p, seed & 0xffffffff, _p(seed_ptr), dtype, L.stream_ptr()), 'gelu_bwd')
    return out


def sigmoid_grad(g, s):
    g, s = g.contiguous(), s.contiguous()
    out = torch.empty_like(g)
    L.check(L.load().sedt_sigmoid_grad(_p(g), _p(s), _p(out), g.numel(), L.stream_ptr()), 'sigmoid_grad')
    return out


def layernorm_fwd(dtype, x, gamma, beta, add_t=None, out=None):
    """out = (y, mean, rstd) preallocated (contiguous row ranges of larger buffers are fine)"""
    rows, D = x.shape
    y2 = torch.empty_like(x) if add_t is not None else None
    if out is not None:
        y, mean, rstd = out
    else:
        y = torch.empty_like(x)
        mean = torch.empty((rows,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
    L.check(L.load().sedt_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(add_t), _p(y), _p(y2), _p(mean), _p(rstd), rows, D,
                                        dtype, L.stream_ptr()), 'layernorm_fwd')
    return y, y2, mean, rstd


def layernorm_bwd(dtype, dy, x, gamma, mean, rstd, dy2=None, dres=None, want_param_grads=True, batch=None, drop=None, dres2=None):
    """batch (a ReduceBatch): the gamma/beta reduction - which only feeds the optimizer - is deferred to batch.flush() and
    rides in the layer's split-K reduction launch.
    drop = (p, seed, seed_ptr): also returns dx passed through that dropout mask (the gradient entering the sub-layer whose
    dropped output fed this LayerNorm) as a 4th result - one launch instead of layernorm_bwd + dropout_grad"""
    lib = L.load()
    rows, D = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty((D,), device=x.device, dtype=torch.float32) if want_param_grads else None
    db = torch.empty((D,), device=x.device, dtype=torch.float32) if want_param_grads else None
    nb = lib.sedt_layernorm_bwd_scratch(rows, D)
    scratch = torch.empty((nb // 4,), device=x.device, dtype=torch.float32)
    dxd, p_, seed_, sp_ = None, 0.0, 0, None
    if drop is not None and drop[0] > 0:
        p_, seed_, sp_ = drop
        if _dev_env('SEDT_LN_DROP_FUSE', '1') == '0':        # A/B switch: separate dropout_grad launch
            dx, dg, db = layernorm_bwd(dtype, dy, x, gamma, mean, rstd, dy2, dres, want_param_grads, batch, dres2=dres2)
            return dx, dg, db, dropout_grad(dtype, dx, p_, seed_, sp_)
        dxd = torch.empty_like(x)
    deferred = want_param_grads and batch is not None
    L.check(lib.sedt_layernorm_bwd_drop(_p(dy), _p(dy2), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dres2), _p(dx),
                                        None if deferred else _p(dg), None if deferred else _p(db), _p(scratch), nb, rows, D,
                                        _p(dxd), p_, seed_ & 0xffffffff, _p(sp_), dtype, L.stream_ptr()), 'layernorm_bwd')
    if deferred:
        # the gamma/beta reduction of the per-workgroup partials rides in batch.flush() as one more job
        dgb = torch.empty((2 * D,), device=x.device, dtype=torch.float32)
        batch.add_colsum(scratch, nb // (2 * D * 4), 2 * D, dgb)
        dg, db = dgb[:D], dgb[D:]
    if drop is not None:
        return dx, dg, db, (dxd if dxd is not None else dx)
    return dx, dg, db


def attention_fwd(dtype, q, k, v, B, H, Lq, Lk, kpm=None, amask=None, drop_p=0.0, seed=0, seed_ptr=None, out=None):
    """q [B*Lq, >=H*32] / k, v [B*Lk, ...] row-strided views; returns (o [B*Lq, H*32], lse [B,H,Lq])"""
    if out is None:
        out = torch.empty((B * Lq, H * 32), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Lq), device=q.device, dtype=torch.float32)
    L.check(L.load().sedt_attention_fwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                        _p(lse), _p(kpm), _p(amask), B, H, Lq, Lk, drop_p, seed & 0xffffffff, _p(seed_ptr),
                                        dtype, L.stream_ptr()), 'attention_fwd')
    return out, lse


def attention_bwd(dtype, q, k, v, o, do, lse, B, H, Lq, Lk, dq, dk, dv, kpm=None, amask=None, drop_p=0.0, seed=0,
                  seed_ptr=None):
    L.check(L.load().sedt_attention_bwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0),
                                        _p(do), do.stride(0), _p(lse), _p(kpm), _p(amask), _p(dq), dq.stride(0), _p(dk),
                                        dk.stride(0), _p(dv), dv.stride(0), B, H, Lq, Lk, drop_p, seed & 0xffffffff,
                                        _p(seed_ptr), dtype, L.stream_ptr()), 'attention_bwd')
    return dq, dk, dv


NO_RIDE = _dev_env('SEDT_CO_NORIDE', '0') == '1'      # experiment: park all wgrads until the end of the backward, no riders
CONV3_DIRECT = _dev_env('SEDT_CONV3_DIRECT', '1') != '0'   # direct 3x3 kernel for the layer1 conv2 geometry (conv3x3_c64.hip)
STEM_DIRECT = _dev_env('SEDT_STEM_DIRECT', '1') != '0'     # one-launch stem forward / backward (stem.hip)
# The pre-norm encoder layer on the x-stationary slab kernels (csrc/enc_slab.hip: two launches per layer instead of seven; a workgroup
# owns 32 tokens, only weights stream).  Default in the bf16 mode; SLAB_ENC = False (tests, A/B) keeps the per-op chain.
SLAB_ENC = _dev_env('SEDT_SLAB_ENC', '1') != '0'
SLAB_ENC_BWD = _dev_env('SEDT_SLAB_ENC_BWD', '1') != '0'      # the input-gradient chain of those layers on the slab kernels as well


SLAB_MIN_WGS = 192      # slabs (= workgroups) below which the per-op chain fills the chip better (B = 32: 128 slabs; same-box A/B on C3)
SLAB_MAX_WGS = 320      # ... and above which it wins again: every slab streams the layer's 3.4 MB of weights, the per-op GEMMs amortise them
                        # over 64-128 rows per tile (C4, B = 200: 800 slabs, 17.0 ms with the slab encoder against 16.65 ms without)


def encoder_slab_ok(dtype, D, H, S, FF, amask, B=None):
    return bool(SLAB_ENC and dtype == BF16 and amask is None and (B is None or SLAB_MIN_WGS <= B * ((S + 31) // 32) <= SLAB_MAX_WGS)
                and L.load().sedt_encoder_slab_ok(D, H, S, FF, dtype))


ENC_PREFETCH = _dev_env('SEDT_ENC_PREFETCH', '1') != '0'
RED_PREFETCH = _dev_env('SEDT_RED_PREFETCH', '1') != '0'


class BackwardChain(object):
    """what the backward that runs AFTER a node's own will stream first, handed from node to node while one forward is traced (backward
    order is the reverse of forward order): a node's forward reads ``top`` - the operands of the node before it, whose backward follows
    its own - keeps it on its autograd ctx, and leaves its own operands in ``top`` for the node after it.  The reduce launch that closes
    the node's backward then touches those operands (SedtPrefetch) so that they are L2-resident when the next backward streams them.
    One chain object per model forward (sedt.backbone.ResNet50Body.forward, sedt.transformer.Transformer.forward): no module- or
    process-level state, nothing survives the forward that created it"""
    __slots__ = ('top',)

    def __init__(self):
        self.top = None


def encoder_qkv_fwd(x, pos, gamma, beta, w_in_frag, b_in, B, S, train=True, prefetch=None):
    """xn = LayerNorm1(x); q | k = (xn + pos) Wqk^T + b; v = xn Wv^T + b.  x, pos [B*S, 256] bf16 contiguous; w_in_frag = the
    fragment-major in_proj_weight (packing.lookup_frag).  Returns (qk [B*S, 512], v [B*S, 256], (xn, xnp, mean, rstd) or None)"""
    _dev_check(x, pos, w_in_frag)
    M, D = x.shape
    assert x.is_contiguous() and pos.is_contiguous() and pos.shape == x.shape and M == B * S and D == 256 and x.dtype == torch.bfloat16
    qk = torch.empty((M, 2 * D), device=x.device, dtype=x.dtype)
    v = torch.empty((M, D), device=x.device, dtype=x.dtype)
    by = None
    if train:
        by = (torch.empty_like(x), torch.empty_like(x), torch.empty((M,), device=x.device, dtype=torch.float32),
              torch.empty((M,), device=x.device, dtype=torch.float32))
    s = by if by is not None else (None,) * 4
    # the next launch's weights: touched by this one so that they are L2-resident when it streams them
    pf = L.prefetch_arg(prefetch) if (prefetch and ENC_PREFETCH) else None
    L.check(L.load().sedt_encoder_qkv_fwd(_p(x), _p(pos), _p(gamma), _p(beta), _p(w_in_frag), _p(b_in), _p(qk), _p(v), _p(s[0]), _p(s[1]),
                                          _p(s[2]), _p(s[3]), B, S, pf, L.stream_ptr()), 'encoder_qkv_fwd')
    return qk, v, by


def encoder_attn_ffn_fwd(x, qk, v, kpm, w_o_frag, b_o, gamma2, beta2, w1_frag, b1, w2_frag, b2, B, S, FF, drop_p=0.0, seeds=(0, 0, 0, 0),
                         seed_ptr=None, train=True):
    """attention over the clip's keys + out-proj + residual + LayerNorm2 + the FFN pair + residual for every 32-token slab in ONE
    launch.  Returns (x2, (ctx, lse, x1, mean2, rstd2, x1n, h) or None); seeds = (attention, out-proj, hidden, FFN output)"""
    _dev_check(x, qk, v)
    M, D = x.shape
    assert x.is_contiguous() and qk.is_contiguous() and v.is_contiguous() and M == B * S
    x2 = torch.empty_like(x)
    by = None
    if train:
        f32 = dict(device=x.device, dtype=torch.float32)
        by = (torch.empty_like(x), torch.empty((B, 8, S), **f32), torch.empty_like(x), torch.empty((M,), **f32), torch.empty((M,), **f32),
              torch.empty_like(x), torch.empty((M, FF), device=x.device, dtype=x.dtype))
    s = by if by is not None else (None,) * 7
    L.check(L.load().sedt_encoder_attn_ffn_fwd(_p(x), _p(qk), _p(v), _p(kpm), _p(w_o_frag), _p(b_o), _p(gamma2), _p(beta2), _p(w1_frag),
                                               _p(b1), _p(w2_frag), _p(b2), _p(x2), _p(s[0]), _p(s[1]), _p(s[2]), _p(s[3]), _p(s[4]),
                                               _p(s[5]), _p(s[6]), B, S, FF, drop_p, seeds[0] & 0xffffffff, seeds[1] & 0xffffffff,
                                               seeds[2] & 0xffffffff, seeds[3] & 0xffffffff, _p(seed_ptr), L.stream_ptr()),
            'encoder_attn_ffn_fwd')
    return x2, by


# The prediction heads in ONE launch each way (csrc/heads_slab.hip).  Default in the bf16 mode.
FUSED_BNECK = int(_dev_env('SEDT_BNECK', '5'))      # 0 off, 1 layer1's identity blocks, 2 + layer2's, 3 + layer1's block 0, 4 + layer2's block 0 (forward), 5 + layer3's identity blocks


def bneck_ok(dtype, blk, W, B=0, H=0):
    """the fused-Bottleneck envelope: an identity block of layer1 (256/64 on 16 columns) or layer2 (512/128 on 8) (csrc/bneck.hip), or of
    layer3 (1024/256 on 4 columns, csrc/bneck3.hip) while B * ceil(H / 8) strips cover the chip about once; bf16"""
    if dtype != BF16:
        return False
    lvl = int(FUSED_BNECK)
    if blk.cin == 1024:
        return bool(lvl >= 5 and L.load().sedt_bneck3_ok(blk.cin, blk.planes, W, blk.stride, blk.dil, int(blk.ds), B, H, dtype))
    return bool(lvl >= (1 if blk.cin == 256 else 2) and L.load().sedt_bneck_ok(blk.cin, blk.planes, W, blk.stride, blk.dil, int(blk.ds), dtype))


BNECK_PREFETCH = _dev_env('SEDT_BNECK_PREFETCH', '1') != '0'


def _bneck3_prefetch(nxt):
    return L.prefetch_arg(nxt) if (nxt is not None and BNECK_PREFETCH) else None


def bneck_fwd(x, B, H, W, wf, sb, train=True, want_bits=True, want_ab=False, nxt=None):
    """x [B*H*W, C] bf16 contiguous; wf = the three fragment-major forward operands; sb = ((s1, b1), (s2, b2), (s3, b3)).
    Returns (y, a, b, ybits, abits, bbits).  Training: the sign bits of the two intermediates (all bneck_bwd needs of them; [M, P/8] bytes)
    and, on request, the sign bits of y; want_ab: the intermediates themselves as well (the weight-gradient GEMMs of a trainable block)"""
    _dev_check(x)
    M, C = x.shape
    P = C // 4
    assert x.is_contiguous() and M == B * H * W
    y = torch.empty_like(x)
    a = b = bits = abits = bbits = None
    if train:
        abits = torch.empty((M, P // 8), device=x.device, dtype=torch.uint8)
        bbits = torch.empty_like(abits)
        if want_ab:
            a = torch.empty((M, P), device=x.device, dtype=torch.bfloat16)
            b = torch.empty_like(a)
        if want_bits:
            bits = torch.empty((M, C // 8), device=x.device, dtype=torch.uint8)
    (s1, b1), (s2, b2), (s3, b3) = sb
    if PROFILE is not None:
        PROFILE_FUSED.append(('bneck3_kernel<false>' if C == 1024 else 'bneck_kernel<BG<%d, %d, %d>, false' % (C, P, W),
                              2.0 * M * (2 * C * P + 9 * P * P),
                              M * (2.0 * C * 2 + (C // 8 if bits is not None else 0) + (2 * (P // 8) if abits is not None else 0)
                                   + (2 * P * 2 if a is not None else 0)) + 2.0 * (2 * C * P + 9 * P * P)))
    if C == 1024:
        # (nxt = the next block's three operands: touched by this launch, L2-resident for the next)
        L.check(L.load().sedt_bneck3_fwd(_p(x), _p(y), _p(wf[0]), _p(wf[1]), _p(wf[2]), _p(s1), _p(b1), _p(s2), _p(b2), _p(s3), _p(b3), _p(a),
                                         _p(b), _p(abits), _p(bbits), _p(bits), B, H, _bneck3_prefetch(nxt), L.stream_ptr()), 'bneck3_fwd')
    else:
        L.check(L.load().sedt_bneck_fwd(_p(x), _p(y), _p(wf[0]), _p(wf[1]), _p(wf[2]), _p(s1), _p(b1), _p(s2), _p(b2), _p(s3), _p(b3), _p(a),
                                        _p(b), _p(abits), _p(bbits), _p(bits), C, P, W, B, H, L.stream_ptr()), 'bneck_fwd')
    return y, a, b, bits, abits, bbits


def bneck_bwd(gy, B, H, W, wt, abits, bbits, xbits, want_g=False, chain_only=False, nxt=None):
    """input gradient of the fused Bottleneck: gy [B*H*W, C] bf16 (already masked by [y > 0]); wt = the three fragment-major dgrad
    operands (conv1, conv2, conv3 order); abits / bbits from bneck_fwd; xbits = sign bits of the block input or None.
    Returns (gx, gb, ga): gb / ga [M, P] = the gradients of the two intermediates (want_g: a trainable block's weight gradients read them).
    chain_only: stop at ga (gx None; wt[0] and xbits unused) - the caller's block has another first convolution (layer1's block 0)"""
    assert gy.is_contiguous() and abits.is_contiguous() and bbits.is_contiguous() and (xbits is None or xbits.is_contiguous())
    M, C = gy.shape
    gx = None if chain_only else torch.empty_like(gy)
    gb = ga = None
    if want_g or chain_only:
        ga = torch.empty((M, C // 4), device=gy.device, dtype=torch.bfloat16)
        gb = torch.empty_like(ga) if want_g else None
    if PROFILE is not None:
        P_ = C // 4
        name = ('bneck3_kernel<true>' if C == 1024 else
                'bneck_kernel<BG<%d, %d, %d>, true, %s>' % (C, P_, W, 'true' if chain_only else 'false'))
        PROFILE_FUSED.append((name, 2.0 * M * ((1 if chain_only else 2) * C * P_ + 9 * P_ * P_),
                              M * (C * 2.0 * (1 if chain_only else 2) + C // 8 + 2 * (P_ // 8) + ((2 if want_g else 1 if chain_only else 0) * P_ * 2))
                              + 2.0 * (2 * C * P_ + 9 * P_ * P_)))
    if C == 1024:
        assert not chain_only
        L.check(L.load().sedt_bneck3_bwd(_p(gy), _p(gx), _p(wt[2]), _p(wt[1]), _p(wt[0]), _p(abits), _p(bbits), _p(xbits), _p(gb), _p(ga), B, H,
                                         _bneck3_prefetch(nxt), L.stream_ptr()), 'bneck3_bwd')
        return gx, gb, ga
    L.check(L.load().sedt_bneck_bwd(_p(gy), _p(gx), _p(wt[2]), _p(wt[1]), None if chain_only else _p(wt[0]), _p(abits), _p(bbits),
                                    None if chain_only else _p(xbits), _p(gb), _p(ga), C, C // 4, W, B, H, L.stream_ptr()), 'bneck_bwd')
    return gx, gb, ga


def bneck0_ok(dtype, blk, W):
    """the fused forward of layer1's first Bottleneck (csrc/bneck.hip: 64 -> 64 -> 64 -> 256 with the projection skip, 16 columns), bf16"""
    return bool(int(FUSED_BNECK) >= 3 and dtype == BF16 and L.load().sedt_bneck0_ok(blk.cin, blk.planes, W, blk.stride, blk.dil, int(blk.ds), dtype))


def bneck0_fwd(x, B, H, wf, sb, train=True, want_bits=True, want_ab=False):
    """x [B*H*16, 64] bf16 contiguous; wf = the fragment-major forward operands of conv1, conv2, conv3 and the projection; sb = their four
    (scale, bias) pairs.  Returns (y, a, b, ybits); a, b (what the per-op backward reads) and ybits only when training"""
    _dev_check(x)
    M = x.shape[0]
    assert x.is_contiguous() and M == B * H * 16 and x.shape[1] == 64
    y = torch.empty((M, 256), device=x.device, dtype=torch.bfloat16)
    a = b = bits = abits = bbits = None
    if train:
        abits = torch.empty((M, 8), device=x.device, dtype=torch.uint8)
        bbits = torch.empty_like(abits)
        if want_ab:
            a = torch.empty((M, 64), device=x.device, dtype=torch.bfloat16)
            b = torch.empty_like(a)
        if want_bits:
            bits = torch.empty((M, 32), device=x.device, dtype=torch.uint8)
    (s1, b1), (s2, b2), (s3, b3), (sd, bd) = sb
    if PROFILE is not None:
        PROFILE_FUSED.append(('bneck0_fwd_kernel', 2.0 * M * (64 * 64 + 9 * 64 * 64 + 2 * 64 * 256),
                              M * (64 * 2.0 + 256 * 2 + (32 if bits is not None else 0) + (16 if abits is not None else 0) + (256 if a is not None else 0))))
    L.check(L.load().sedt_bneck0_fwd(_p(x), _p(y), _p(wf[0]), _p(wf[1]), _p(wf[2]), _p(wf[3]), _p(s1), _p(b1), _p(s2), _p(b2), _p(s3), _p(b3),
                                     _p(sd), _p(bd), _p(a), _p(b), _p(abits), _p(bbits), _p(bits), B, H, L.stream_ptr()), 'bneck0_fwd')
    return y, a, b, bits, abits, bbits


def bneck2_ok(dtype, blk, W):
    """the fused forward of layer2's first Bottleneck (csrc/bneck.hip: 256 -> 128, 3x3 stride 2, -> 512 with the stride-2 projection), bf16"""
    return bool(int(FUSED_BNECK) >= 4 and dtype == BF16 and L.load().sedt_bneck2_ok(blk.cin, blk.planes, W, blk.stride, blk.dil, int(blk.ds), dtype))


def bneck2_fwd(x, B, H, wf, sb, train=True, want_bits=True):
    """x [B*H*16, 256] bf16 contiguous; wf = the fragment-major forward operands of conv1, conv2, conv3 and the projection; sb = their four
    (scale, bias) pairs.  Returns (y [B*H2*8, 512], a [B*H*16, 128], b [B*H2*8, 128], ybits); a, b, ybits only when training"""
    _dev_check(x)
    M = x.shape[0]
    assert x.is_contiguous() and M == B * H * 16 and x.shape[1] == 256
    M2 = B * ((H - 1) // 2 + 1) * 8
    y = torch.empty((M2, 512), device=x.device, dtype=torch.bfloat16)
    a = b = bits = None
    if train:
        a = torch.empty((M, 128), device=x.device, dtype=torch.bfloat16)
        b = torch.empty((M2, 128), device=x.device, dtype=torch.bfloat16)
        if want_bits:
            bits = torch.empty((M2, 64), device=x.device, dtype=torch.uint8)
    (s1, b1), (s2, b2), (s3, b3), (sd, bd) = sb
    if PROFILE is not None:
        PROFILE_FUSED.append(('bneck2_fwd_kernel', 2.0 * (M * 256 * 128 + M2 * (9 * 128 * 128 + 128 * 512 + 256 * 512)),
                              M * 256 * 2.0 + M2 * (512 * 2.0 + (64 if bits is not None else 0)) + ((M + M2) * 128 * 2.0 if a is not None else 0)))
    L.check(L.load().sedt_bneck2_fwd(_p(x), _p(y), _p(wf[0]), _p(wf[1]), _p(wf[2]), _p(wf[3]), _p(s1), _p(b1), _p(s2), _p(b2), _p(s3), _p(b3),
                                     _p(sd), _p(bd), _p(a), _p(b), _p(bits), B, H, L.stream_ptr()), 'bneck2_fwd')
    return y, a, b, bits


SLAB_HEADS = _dev_env('SEDT_SLAB_HEADS', '1') != '0'


HEADS_SLAB_MAX_ROWS = 12288     # (C4: 25,200 head rows = 788 slabs ran 0.5 % slower than the per-op heads, same-box A/B; C2 / C3 / C5: 2,000 rows, -1.2 %)


def heads_slab_ok(dtype, D, C1, CA, rows=0):
    return bool(SLAB_HEADS and dtype == BF16 and rows <= HEADS_SLAB_MAX_ROWS and L.load().sedt_heads_slab_ok(D, C1, CA, dtype))


def heads_fwd(x, wc, bc, w1f, b1, w2f, b2, w3, b3, wa, ba, Lh, B, Qp, train=True):
    """x [Lh*B*Qp, 256] bf16 contiguous; wc / w3 / wa f32 masters; w1f / w2f fragment-major.  Returns (cls [rows, C1], box [rows, 2], at
    [B, CA] or None, (h1, h2) or None)"""
    _dev_check(x, wc, w3)
    rows, D = x.shape
    assert x.is_contiguous() and rows == Lh * B * Qp and wc.is_contiguous() and w3.is_contiguous() and (wa is None or wa.is_contiguous())
    C1, CA = wc.shape[0], 0 if wa is None else wa.shape[0]
    f32 = dict(device=x.device, dtype=torch.float32)
    cls, box = torch.empty((rows, C1), **f32), torch.empty((rows, 2), **f32)
    at = torch.empty((B, CA), **f32) if CA else None
    hh = (torch.empty_like(x), torch.empty_like(x)) if train else None
    L.check(L.load().sedt_heads_fwd(_p(x), _p(wc), _p(bc), _p(w1f), _p(b1), _p(w2f), _p(b2), _p(w3), _p(b3), _p(wa), _p(ba), _p(cls), _p(box),
                                    _p(at), _p(hh[0]) if hh else None, _p(hh[1]) if hh else None, Lh, B, Qp, C1, CA, L.stream_ptr()), 'heads_fwd')
    return cls, box, at, hh


def heads_bwd(x, h1, h2, box, at, g_cls, g_box, g_at, wc, w3, wa, w2t, w1t, Lh, B, Qp):
    """returns (dhs, g_h1, g_h2, part [slabs, (C1 + CA + 2) * 257])"""
    rows, D = x.shape
    C1, CA = wc.shape[0], 0 if wa is None else wa.shape[0]
    dhs, g_h1, g_h2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    nsl = (rows + 31) // 32
    part = torch.empty((nsl, (C1 + CA + 2) * 257), device=x.device, dtype=torch.float32)
    for g in (g_cls, g_box, g_at):
        assert g is None or (g.dtype == torch.float32 and g.is_contiguous())
    L.check(L.load().sedt_heads_bwd(_p(x), _p(h1), _p(h2), _p(box), _p(at), _p(g_cls), _p(g_box), _p(g_at), _p(wc), _p(w3), _p(wa), _p(w2t),
                                    _p(w1t), _p(dhs), _p(g_h1), _p(g_h2), _p(part), Lh, B, Qp, C1, CA, L.stream_ptr()), 'heads_bwd')
    return dhs, g_h1, g_h2, part


def encoder_ffn_bwd(gx2, h, x1, mean2, rstd2, gamma2, w2t_frag, w1t_frag, wot_frag, B, S, drop_p=0.0, seeds=(0, 0), seed_ptr=None):
    """FFN backward + LayerNorm2 backward + out-proj input gradient per 32-token slab in ONE launch (csrc/enc_slab.hip).
    seeds = (FFN-output dropout, out-proj dropout).  Returns (g2, gh, gx1, g1, gctx, ln_part [slabs, 512])"""
    _dev_check(gx2, h, x1)
    M, D = gx2.shape
    FF = h.shape[1]
    assert gx2.is_contiguous() and h.is_contiguous() and x1.is_contiguous() and M == B * S
    nsl = B * ((S + 31) // 32)
    g2 = torch.empty_like(gx2) if drop_p > 0 else None
    g1 = torch.empty_like(gx2) if drop_p > 0 else None
    gh = torch.empty_like(h)
    gx1, gctx = torch.empty_like(gx2), torch.empty_like(gx2)
    part = torch.empty((nsl, 2 * D), device=gx2.device, dtype=torch.float32)
    L.check(L.load().sedt_encoder_ffn_bwd(_p(gx2), _p(h), _p(x1), _p(mean2), _p(rstd2), _p(gamma2), _p(w2t_frag), _p(w1t_frag),
                                          _p(wot_frag), _p(g2), _p(gh), _p(gx1), _p(g1), _p(gctx), _p(part), B, S, FF, drop_p,
                                          seeds[0] & 0xffffffff, seeds[1] & 0xffffffff, _p(seed_ptr), L.stream_ptr()), 'encoder_ffn_bwd')
    return (g2 if g2 is not None else gx2), gh, gx1, (g1 if g1 is not None else gx1), gctx, part


def encoder_qkv_bwd(dqk, dv, x, mean1, rstd1, gamma1, gx1, wint_frag, B, S):
    """gx = LayerNorm1'(dq|dk Wqk + dv Wv) + gx1 per slab in one launch.  Returns (gx, ln_part [slabs, 512])"""
    _dev_check(dqk, dv, x, gx1)
    M, D = x.shape
    assert dqk.is_contiguous() and dv.is_contiguous() and x.is_contiguous() and gx1.is_contiguous() and M == B * S
    gx = torch.empty_like(x)
    part = torch.empty((B * ((S + 31) // 32), 2 * D), device=x.device, dtype=torch.float32)
    L.check(L.load().sedt_encoder_qkv_bwd(_p(dqk), _p(dv), _p(x), _p(mean1), _p(rstd1), _p(gamma1), _p(gx1), _p(wint_frag), _p(gx),
                                          _p(part), B, S, L.stream_ptr()), 'encoder_qkv_bwd')
    return gx, part


def posenc(dtype, mask_u8, D):
    B, H, W = mask_u8.shape
    pos = torch.empty((B, H * W, D), device=mask_u8.device, dtype=TORCH_DTYPE[dtype])
    L.check(L.load().sedt_posenc(_p(mask_u8), _p(pos), B, H, W, D, dtype, L.stream_ptr()), 'posenc')
    return pos


def mask_resize(mask_u8, Hout, Wout):
    B, Hin, Win = mask_u8.shape
    out = torch.empty((B, Hout, Wout), device=mask_u8.device, dtype=torch.uint8)
    L.check(L.load().sedt_mask_resize(_p(mask_u8), _p(out), B, Hin, Win, Hout, Wout, L.stream_ptr()), 'mask_resize')
    return out


def bn_fold(w, b, rm, rv, scale=None, bias=None):
    n = w.numel()
    if scale is None:
        scale = torch.empty((n,), device=w.device, dtype=torch.float32)
        bias = torch.empty_like(scale)
    L.check(L.load().sedt_bn_fold(_p(w), _p(b), _p(rm), _p(rv), _p(scale), _p(bias), n, L.stream_ptr()), 'bn_fold')
    return scale, bias


def pack_conv(dtype, w, bnscale=None, want_fwd=True, want_bwd=True, wf=None, wb=None):
    """w (Co, Ci, KH, KW) or (N, K) f32 -> wf [Co][taps][Ci], wb [Ci][taps][Co]*bnscale in the compute dtype"""
    Co, Ci = w.shape[0], w.shape[1]
    taps = w.numel() // (Co * Ci)
    td = TORCH_DTYPE[dtype]
    if want_fwd and wf is None:
        wf = torch.empty((Co, taps * Ci), device=w.device, dtype=td)
    if want_bwd and wb is None:
        wb = torch.empty((Ci, taps * Co), device=w.device, dtype=td)
    L.check(L.load().sedt_pack_conv(_p(w), Co, Ci, taps, _p(bnscale), _p(wf), _p(wb), dtype, L.stream_ptr()), 'pack_conv')
    return wf, wb


def stem_prep(dtype, w0, b0, w1):
    wcat = torch.empty((64, 128), device=w1.device, dtype=TORCH_DTYPE[dtype])
    L.check(L.load().sedt_stem_prep(_p(w0), _p(b0), _p(w1), _p(wcat), dtype, L.stream_ptr()), 'stem_prep')
    return wcat


def stem_im2col(dtype, x, B, H, W):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    col = torch.empty((B * Ho * Wo, 128), device=x.device, dtype=TORCH_DTYPE[dtype])
    L.check(L.load().sedt_stem_im2col(_p(x), _p(col), B, H, W, dtype, L.stream_ptr()), 'stem_im2col')
    return col, Ho, Wo


def stem_pool_ok(dtype, W):
    """the one-launch stem (conv0 o conv1 o FrozenBN o ReLU o max-pool) exists for the bf16 mode and 64 mel bands"""
    return dtype == BF16 and W == 64 and STEM_DIRECT


def stem_pool_fwd(x, wcat, scale, bias, B, H, W, want_idx=True, want_s1=False):
    """x f32 [B][H][64] -> (pool bf16 [B*Hp*16, 64], idx uint8 or None, Hp, Wp[, s1 bf16 [B*Ho*32, 64]])"""
    Ho = (H - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, 16
    pool = torch.empty((B * Hp * Wp, 64), device=x.device, dtype=torch.bfloat16)
    idx = torch.empty((B * Hp * Wp, 64), device=x.device, dtype=torch.uint8) if want_idx else None
    s1 = torch.empty((B * Ho * 32, 64), device=x.device, dtype=torch.bfloat16) if want_s1 else None
    L.check(L.load().sedt_stem_pool_fwd(_p(x), _p(wcat), _p(scale), _p(bias), _p(pool), _p(idx), _p(s1), B, H, W, L.stream_ptr()),
            'stem_pool_fwd')
    return (pool, idx, Hp, Wp, s1) if want_s1 else (pool, idx, Hp, Wp)


def stem_pool_wgrad(x, g, idx, pool, rowscale, B, H, W):
    """G f32 [64][128] (wcat's layout) = bn_scale[co] * sum over pixels of maxpool_bwd(g)[pixel][co] * patch[pixel][k]"""
    lib = L.load()
    ns = lib.sedt_stem_pool_wgrad_slabs(B, H)
    slab = torch.empty((ns, 64, 128), device=x.device, dtype=torch.float32)
    G = torch.empty((64, 128), device=x.device, dtype=torch.float32)
    L.check(lib.sedt_stem_pool_wgrad(_p(x), _p(g), _p(idx), _p(pool), _p(slab), ns, B, H, W, L.stream_ptr()), 'stem_pool_wgrad')
    L.check(lib.sedt_wgrad_reduce_bias(_p(slab), ns, 64, 1, 128, _p(rowscale), _p(G), None, None, L.stream_ptr()), 'wgrad_reduce')
    return G


def stem_conv0_grad(G, w1):
    from . import runtime
    dw0 = torch.empty((3, 1, 1, 1), device=G.device, dtype=torch.float32)
    db0 = torch.empty((3,), device=G.device, dtype=torch.float32)
    with runtime.side(G):                       # G comes from a wgrad: stay on its stream
        L.check(L.load().sedt_stem_conv0_grad(_p(G), _p(w1), _p(dw0), _p(db0), L.stream_ptr()), 'stem_conv0_grad')
    return dw0, db0


def maxpool_fwd(dtype, x, B, H, W, Cch):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B * Ho * Wo, Cch), device=x.device, dtype=x.dtype)
    idx = torch.empty((B * Ho * Wo, Cch), device=x.device, dtype=torch.uint8)
    L.check(L.load().sedt_maxpool_fwd(_p(x), _p(y), _p(idx), B, H, W, Cch, dtype, L.stream_ptr()), 'maxpool_fwd')
    return y, idx, Ho, Wo


def maxpool_bwd(dtype, dy, idx, relu_src, B, H, W, Cch, y=None):
    """dx of the 3x3/s2 max-pool, optionally through the ReLU that produced the pooled tensor: relu_src = that tensor at
    full resolution, or (cheaper) y = the pooled output itself"""
    dx = torch.empty((B * H * W, Cch), device=dy.device, dtype=dy.dtype)
    L.check(L.load().sedt_maxpool_bwd_y(_p(dy), _p(idx), _p(relu_src), _p(y), _p(dx), B, H, W, Cch, dtype, L.stream_ptr()),
            'maxpool_bwd')
    return dx


def avgpool(dtype, x, B, P, Cch):
    out = torch.empty((B, Cch), device=x.device, dtype=torch.float32)
    L.check(L.load().sedt_avgpool(_p(x), _p(out), B, P, Cch, dtype, L.stream_ptr()), 'avgpool')
    return out


def sumsq(g, out, accumulate=False):
    lib = L.load()
    nb = lib.sedt_sumsq_scratch(g.numel())
    scratch = torch.empty((nb // 4,), device=g.device, dtype=torch.float32)
    L.check(lib.sedt_sumsq(_p(g), g.numel(), _p(out), _p(scratch), nb, int(accumulate), L.stream_ptr()), 'sumsq')
    return out


def adamw_clip(p, g, m, v, sumsq_t, max_norm, lr, beta1, beta2, eps, weight_decay, step):
    L.check(L.load().sedt_adamw_clip(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(sumsq_t), max_norm, lr, beta1, beta2, eps,
                                     weight_decay, step, L.stream_ptr()), 'adamw_clip')


def set_criterion(logits, boxes, at, dense, empty_weight, layer_of, w_ce, w_bbox, w_giou, w_weak, fl=False, alpha_fl=0.5,
                  gamma_fl=1.0, nonfinite=None, q0=0, at_p=None, w_weak_p=0.0, wp_all=False):
    """device half of SetCriterion in one launch (csrc/criterion.hip).  logits [L,B,Qs,C+1], boxes [L,B,Qs,2] (the Q queries
    q0 .. q0+Q-1 of every clip take part, Q from the dense targets), at [Bat,C] or None - all f32 contiguous; dense =
    SetCriterion.dense_views(...).  fl: the focal-loss variant (sedt.py:176, 211-218).  nonfinite: optional int32 device word set
    to 1 when the weighted total is NaN/inf.  at_p [Bp,C]: the pooled clip-level probabilities of a --pooling model (loss_weak_p over
    the weak clips, or over all labelled ones when wp_all = the reference's weak_mask None).  Returns (out[4L+6], total scalar,
    state); state feeds set_criterion_bwd."""
    _dev_check(logits, boxes)
    Lh, B, Qs, C1 = logits.shape
    Q = dense['tc'].shape[2]
    assert logits.dtype == torch.float32 and boxes.dtype == torch.float32 and logits.is_contiguous() and boxes.is_contiguous()
    assert boxes.shape == (Lh, B, Qs, 2) and Lh <= L.CRIT_MAXL and len(layer_of) == Lh and 0 <= q0 and q0 + Q <= Qs
    a = L.SedtCriterion()
    dl = torch.empty((Lh, B, Q, C1), device=logits.device, dtype=torch.float32)
    db, db2 = torch.empty((Lh, B, Q, 2), device=logits.device, dtype=torch.float32), torch.empty((Lh, B, Q, 2), device=logits.device, dtype=torch.float32)
    dat = None
    out = torch.empty(4 * Lh + 6, device=logits.device, dtype=torch.float32)
    total = torch.empty((), device=logits.device, dtype=torch.float32)
    a.logits, a.boxes, a.out, a.total = logits.data_ptr(), boxes.data_ptr(), out.data_ptr(), total.data_ptr()
    a.dlogits, a.dboxes, a.dboxes2 = dl.data_ptr(), db.data_ptr(), db2.data_ptr()
    for k in ('tc', 'coef', 'wbox', 'tbox', 'tgt_len'):
        setattr(a, k, dense[k].data_ptr())
    if dense.get('num_boxes') is not None:          # None: the kernel sums the final layer's box weights itself
        a.num_boxes = dense['num_boxes'].data_ptr()
    a.empty_weight = empty_weight.data_ptr()
    a.L, a.B, a.ns, a.Q, a.C, a.n_lab, a.Qs, a.q0 = Lh, B, dense['ns'], Q, C1 - 1, dense['n_lab'], Qs, q0
    assert dense['tgt_len'].numel() == B and dense['L'] == Lh and empty_weight.is_cuda and empty_weight.numel() == C1
    if at is not None:
        assert at.dtype == torch.float32 and at.is_contiguous() and at.dim() == 2 and at.shape[1] == C1 - 1
        dat = torch.empty_like(at)
        a.at, a.dat, a.gt_weak, a.Bat = at.data_ptr(), dat.data_ptr(), dense['gt_weak'].data_ptr(), at.shape[0]
        assert dense['n_lab'] <= at.shape[0]
    dat_p = None
    if at_p is not None:
        assert at is not None, 'loss_weak_p needs the audio-tag output (the reference builds its targets in loss_weak)'
        if wp_all and at_p.shape[0] != dense['n_lab']:
            # reference sedt.py:184 with weak_mask None indexes both tensors with None: [1, B, C] against [1, n_lab, C] - BCELoss raises
            # on the shape mismatch whenever unlabelled clips ride in the batch
            raise ValueError(f"loss_weak_p with weak_mask=None needs every clip of the batch labelled (at_p has {at_p.shape[0]} clips, "
                             f"{dense['n_lab']} are labelled): the reference's BCELoss rejects the shapes (sedt.py:184)")
        assert at_p.dtype == torch.float32 and at_p.is_contiguous() and at_p.dim() == 2 and at_p.shape[1] == C1 - 1
        assert dense['n_lab'] <= at_p.shape[0]
        dat_p = torch.empty_like(at_p)
        a.at_p, a.dat_p, a.Bp, a.w_weak_p, a.wp_all = at_p.data_ptr(), dat_p.data_ptr(), at_p.shape[0], w_weak_p, int(bool(wp_all))
    for i in range(Lh):
        a.layer_of[i], a.w_ce[i], a.w_bbox[i], a.w_giou[i] = layer_of[i], w_ce[i], w_bbox[i], w_giou[i]
    a.w_weak = w_weak
    a.fl, a.alpha_fl, a.gamma_fl = int(bool(fl)), alpha_fl, gamma_fl
    if nonfinite is not None:
        assert nonfinite.dtype == torch.int32 and nonfinite.is_cuda
        a.nonfinite = nonfinite.data_ptr()
    if dense.get('split') is not None:              # {ns, n_lab} of this batch as device words (mix-up moves the boundary)
        assert dense['split'].dtype == torch.int32 and dense['split'].is_cuda and dense['split'].numel() >= 2
        a.split = dense['split'].data_ptr()
    scratch = torch.empty(Lh * B * Q * 5, device=logits.device, dtype=torch.float32)
    L.check(L.load().sedt_set_criterion(a, _p(scratch), L.stream_ptr()), 'set_criterion')
    return out, total, (a, dl, db, db2, dat, dat_p, (Lh, B, Qs, C1))


def set_criterion_bwd(state, g, gtotal=None):
    """gradients of (logits, boxes, at, at_p) - in the layout of the head outputs, zero rows for queries outside the window - given the
    gradient g[4L+6] of the loss vector and / or the gradient of the separately returned total (either may be None)"""
    a, dl, db, db2, dat, dat_p, (Lh, B, Qs, C1) = state
    g = None if g is None else g.contiguous().float()
    gtotal = None if gtotal is None else gtotal.contiguous().float()
    gl = torch.empty((Lh, B, Qs, C1), device=dl.device, dtype=torch.float32)
    gb = torch.empty((Lh, B, Qs, 2), device=dl.device, dtype=torch.float32)
    gat = None if dat is None else torch.empty_like(dat)
    gat_p = None if dat_p is None else torch.empty_like(dat_p)
    L.check(L.load().sedt_set_criterion_bwd(a, _p(g), _p(gtotal), _p(gl), _p(gb), _p(gat), _p(gat_p), L.stream_ptr()), 'set_criterion_bwd')
    return gl, gb, gat, gat_p


def match_targets(logits, boxes, tables, dense, layer_of, w_class, w_bbox, w_giou, max_targets, assign=None, fl=False,
                  fine_tune=False, normalize=False, epsilon=1.0, alpha=1.0, alpha_fl=0.5, gamma_fl=1.0, ft_rand=None, ft_seed=0,
                  seed_ptr=None, q0=0):
    """device-side Hungarian matching + dense targets in one launch (csrc/criterion.hip).  tables: dict with lab_cat (int64),
    lab_off (int32 [B+1]), box_cat (f32 [N,2]), box_off (int32 [ns+1]), ratio_cat (f32 or None); dense: the views of
    SetCriterion.dense_views, written in place.  fl / fine_tune / normalize: the matcher variants of matcher.py:73-78,
    99-121, 124-132 (ft_rand [ns,Q] f32 injects the uniforms of the fine-tune branch; else a counter hash of ft_seed)."""
    _dev_check(logits, boxes)
    Lh, B, Qs, C1 = logits.shape
    Q = dense['tc'].shape[2]                        # queries q0 .. q0+Q-1 of the Qs head rows per clip are matched
    assert logits.dtype == torch.float32 and boxes.dtype == torch.float32 and logits.is_contiguous() and boxes.is_contiguous()
    assert 0 <= q0 and q0 + Q <= Qs
    a = L.SedtMatch()
    a.logits, a.boxes = logits.data_ptr(), boxes.data_ptr()
    assert tables['lab_cat'].dtype == torch.int64 and tables['lab_off'].dtype == torch.int32 and tables['box_off'].dtype == torch.int32
    assert tables['box_cat'].dtype == torch.float32 and tables['lab_off'].numel() >= B + 1 and tables['box_off'].numel() >= dense['ns'] + 1
    for k in ('lab_cat', 'lab_off', 'box_cat', 'box_off'):
        assert tables[k].is_cuda and tables[k].is_contiguous()
        setattr(a, k, tables[k].data_ptr())
    if tables.get('ratio_cat') is not None:
        assert tables['ratio_cat'].dtype == torch.float32 and tables['ratio_cat'].is_cuda
        a.ratio_cat = tables['ratio_cat'].data_ptr()
    for k in ('tc', 'coef', 'wbox', 'tbox', 'tidx', 'tgt_len'):
        setattr(a, k, dense[k].data_ptr())
    if dense['gt_weak'].numel():
        a.gt_weak = dense['gt_weak'].data_ptr()
    if assign is not None:
        assert assign.dtype == torch.int32 and assign.numel() == Lh * dense['ns'] * Q
        a.assign = assign.data_ptr()
    a.L, a.B, a.ns, a.Q, a.C, a.n_lab, a.max_targets, a.Qs, a.q0 = Lh, B, dense['ns'], Q, C1 - 1, dense['n_lab'], max_targets, Qs, q0
    assert dense['L'] == Lh and dense['tgt_len'].numel() == B
    for i in range(Lh):
        a.layer_of[i] = layer_of[i]
    a.w_class, a.w_bbox, a.w_giou = w_class, w_bbox, w_giou
    a.fl, a.fine_tune, a.normalize = int(bool(fl)), int(bool(fine_tune)), int(bool(normalize))
    a.alpha_fl, a.gamma_fl, a.epsilon, a.alpha = alpha_fl, gamma_fl, epsilon, alpha
    if ft_rand is not None:
        assert ft_rand.dtype == torch.float32 and ft_rand.is_cuda and ft_rand.is_contiguous() and ft_rand.numel() >= dense['ns'] * Q
        a.ft_rand = ft_rand.data_ptr()
    a.ft_seed = ft_seed & 0xffffffff
    if seed_ptr is not None:
        a.seed_ptr = seed_ptr.data_ptr()
    if dense.get('split') is not None:
        assert dense['split'].dtype == torch.int32 and dense['split'].is_cuda and dense['split'].numel() >= 2
        a.split = dense['split'].data_ptr()
    L.check(L.load().sedt_match_targets(a, L.stream_ptr()), 'match_targets')


def sum_f32(x, out=None):
    """out[0] = sum(x) in a fixed order (one small launch)"""
    x = x.contiguous()
    assert x.dtype == torch.float32 and x.is_cuda
    if out is None:
        out = torch.empty(1, device=x.device, dtype=torch.float32)
    L.check(L.load().sedt_sum_f32(_p(x), x.numel(), _p(out), L.stream_ptr()), 'sum_f32')
    return out


def _pool_args(logits, boxes, attn, mode, q0, Q):
    assert logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 3 and logits.is_cuda
    B, Qs, C1 = logits.shape
    a = L.SedtPoolAt()
    a.logits, a.B, a.Qs, a.q0, a.Q, a.C, a.mode = logits.data_ptr(), B, Qs, q0, Q, C1 - 1, L.POOL_MODES[mode]
    if mode == 'weighted_sum':
        assert boxes is not None and boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.shape == (B, Qs, 2)
        a.boxes = boxes.data_ptr()
    if mode == 'attn':
        assert attn is not None and attn.dtype == torch.float32 and attn.is_contiguous() and attn.shape == (B, Q, C1 - 1)
        a.attn = attn.data_ptr()
    return a


def pool_at(logits, boxes, attn, mode, q0, Q):
    """--pooling variants (sedt.py:96-119; csrc/pool_at.hip): at_p [B,C] from the final layer's class logits [B,Qs,C+1] (event
    queries q0 .. q0+Q-1), the boxes [B,Qs,2] (mode 'weighted_sum') or the attention logits [B,Q,C] (mode 'attn'); all f32."""
    a = _pool_args(logits, boxes, attn, mode, q0, Q)
    out = torch.empty((a.B, a.C), device=logits.device, dtype=torch.float32)
    L.check(L.load().sedt_pool_at(a, _p(out), L.stream_ptr()), 'pool_at')
    return out


def pool_at_bwd(logits, boxes, attn, mode, q0, Q, g):
    """(glogits [B,Qs,C+1], gboxes [B,Qs,2] | None, gattn [B,Q,C] | None) for the gradient g [B,C] that reached at_p"""
    a = _pool_args(logits, boxes, attn, mode, q0, Q)
    g = g.contiguous().float()
    assert g.shape == (a.B, a.C)
    gl = torch.empty_like(logits)
    gb = torch.empty_like(boxes) if mode == 'weighted_sum' else None
    ga = torch.empty_like(attn) if mode == 'attn' else None
    L.check(L.load().sedt_pool_at_bwd(a, _p(g), _p(gl), _p(gb), _p(ga), L.stream_ptr()), 'pool_at_bwd')
    return gl, gb, ga


def feature_loss(pred, gt, dense, layer_of, num_boxes, w=None, nonfinite=None, base=None):
    """SP-SEDT feature-reconstruction loss (sedt.py:263-283) of every decoder layer in one launch.  pred [L,B,Q,F] f32,
    gt [B*P,F] f32, w [L] f32 layer weights.  Returns (out[L+1]: loss per dense layer + their weighted sum,
    dpred [L,B,Q,F] = d loss[d] / d pred, unweighted) - and, with base (a device scalar: the criterion's running weighted total), a third
    result total = base + out[L]."""
    _dev_check(pred, gt)
    Lh, B, Q, F = pred.shape
    assert pred.dtype == torch.float32 and gt.dtype == torch.float32 and pred.is_contiguous() and gt.is_contiguous()
    ns = dense['ns']
    P = gt.shape[0] // max(ns, 1)
    assert gt.shape == (ns * P, F) and dense['L'] == Lh and num_boxes.numel() == 1
    dpred = torch.empty_like(pred)
    rowloss = torch.empty(Lh * ns * Q, device=pred.device, dtype=torch.float32)
    out = torch.empty(Lh + 1, device=pred.device, dtype=torch.float32)
    lay = (C.c_int32 * Lh)(*layer_of)
    total = torch.empty((), device=pred.device, dtype=torch.float32) if base is not None else None
    if base is not None:
        assert base.dtype == torch.float32 and base.numel() == 1 and base.is_cuda
    L.check(L.load().sedt_feature_loss(_p(pred), _p(gt), _p(dense['wbox']), _p(dense['tidx']), _p(num_boxes), lay, _p(w), Lh, B, ns, Q, P, F,
                                       _p(rowloss), _p(out), _p(dpred), _p(nonfinite), _p(base), _p(total), L.stream_ptr()), 'feature_loss')
    return (out, dpred) if base is None else (out, dpred, total)


def scale_layers(x, g, gtot, w, idx=None):
    """in place: x[l] *= g[d] + gtot[0] * w[d], d = idx[l] (identity when None)   (x [L, ...] f32; g, w [L]; gtot [1])"""
    Lh = x.shape[0]
    per = x.numel() // Lh
    ia = None if idx is None else (C.c_int32 * Lh)(*idx)
    L.check(L.load().sedt_scale_layers(_p(x), _p(g), _p(gtot), _p(w), ia, Lh, per, L.stream_ptr()), 'scale_layers')
    return x


def postprocess(logits, boxes, sizes=None, tags=None, at_m=2, is_semi=False, threshold=0.5):
    """PostProcess.forward on the device (sedt.py:355-396): (scores [B,Q], labels [B,Q] int64, boxes [B,Q,2])"""
    _dev_check(logits, boxes)
    B, Q, C1 = logits.shape
    logits, boxes = logits.detach().float().contiguous(), boxes.detach().float().contiguous()
    tg = None if tags is None else tags.to(device=logits.device, dtype=torch.float32).contiguous()
    sz = None if sizes is None else sizes.to(device=logits.device, dtype=torch.float32).reshape(-1).contiguous()
    scores = torch.empty((B, Q), device=logits.device, dtype=torch.float32)
    labels = torch.empty((B, Q), device=logits.device, dtype=torch.int64)
    out = torch.empty((B, Q, 2), device=logits.device, dtype=torch.float32)
    L.check(L.load().sedt_postprocess(_p(logits), _p(boxes), _p(tg), _p(sz), B, Q, C1 - 1, at_m, 0.0 if threshold is None else float(threshold),
                                      int(bool(is_semi)), _p(scores), _p(labels), _p(out), L.stream_ptr()), 'postprocess')
    return scores, labels, out


def pseudo_labels(logits, boxes, at, thr, min_len, tables, counter=None, del_overlap=True):
    """engine.get_pseudo_labels on the device (engine.py:300-348): fills the flat target tables (dict with lab_cat int64,
    box_cat f32 [N,2], lab_off / box_off int32 [B+1]) that match_targets reads; counter int32 [C] accumulates the kept events
    per class."""
    _dev_check(logits, boxes, thr)
    B, Q, C1 = logits.shape
    logits, boxes = logits.detach().float().contiguous(), boxes.detach().float().contiguous()
    at = None if at is None else at.detach().float().contiguous()
    assert thr.dtype == torch.float32 and thr.numel() == C1 - 1 and thr.is_contiguous()
    assert tables['lab_cat'].dtype == torch.int64 and tables['box_cat'].dtype == torch.float32
    assert tables['lab_off'].dtype == torch.int32 and tables['box_off'].dtype == torch.int32
    assert tables['lab_off'].numel() >= B + 1 and tables['box_off'].numel() >= B + 1
    cap = min(tables['lab_cat'].numel(), tables['box_cat'].numel() // 2)
    if counter is not None:
        assert counter.dtype == torch.int32 and counter.numel() >= C1 - 1 and counter.is_cuda
    L.check(L.load().sedt_pseudo_labels(_p(logits), _p(boxes), _p(at), _p(thr), float(min_len), B, Q, C1 - 1, int(bool(del_overlap)),
                                        _p(tables['lab_cat']), _p(tables['box_cat']), _p(tables['lab_off']), _p(tables['box_off']),
                                        _p(counter), cap, L.stream_ptr()), 'pseudo_labels')


def mixup(x1, x2, jobs, out=None):
    """feature half of utilities/mixup.py: out[i] = lam * x1[src1] + (1 - lam) * x2[src2] / x1[src1] / x2[src2] per job record
    (jobs: uint8 device tensor of n 16-byte records {int32 src1, src2, mode; f32 lam}); x1 / x2 / out f32 [*, clip...]"""
    _dev_check(x1, x2, jobs)
    assert x1.dtype == torch.float32 and x2.dtype == torch.float32 and x1.is_contiguous() and x2.is_contiguous()
    assert jobs.dtype == torch.uint8 and jobs.numel() % 16 == 0
    n = jobs.numel() // 16
    if out is None:
        out = torch.empty((n,) + tuple(x1.shape[1:]), device=x1.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == n and out[0].numel() == x1[0].numel() == x2[0].numel()
    L.check(L.load().sedt_mixup(_p(x1), _p(x2), _p(jobs), n, x1[0].numel(), _p(out), L.stream_ptr()), 'mixup')
    return out


def mixup_targets(tab1, tab2, lam, mix_num, tab_out, jobs, max_events=20):
    """label half of mixup_label_unlabel (utilities/mixup.py:129-196) on the device: tab1 = the labelled targets, tab2 = the pseudo
    targets (sedt.TargetTables), lam = f32 device [2] {lam, 1 - lam}; writes the merged targets into tab_out (every clip strong)
    and the B2 feature-mixing records into jobs (uint8 [16 * B2])."""
    d1, d2, do = tab1.as_dict(), tab2.as_dict(), tab_out.as_dict()
    B2 = tab2.B
    assert tab_out.B == B2 and tab_out.ns == B2 and jobs.dtype == torch.uint8 and jobs.numel() >= 16 * B2 and jobs.is_cuda
    assert lam.dtype == torch.float32 and lam.numel() >= 2 and lam.is_cuda and 0 <= mix_num <= min(tab1.B, B2)
    cap = min(do['lab_cat'].numel(), do['box_cat'].numel() // 2)
    L.check(L.load().sedt_mixup_targets(_p(d1['lab_cat']), _p(d1['lab_off']), _p(d1['box_cat']), _p(d1['box_off']), _p(d1['ratio_cat']),
                                        _p(tab1.split), tab1.B, tab1.ns, _p(d2['lab_cat']), _p(d2['lab_off']), _p(d2['box_cat']),
                                        _p(d2['box_off']), B2, _p(lam), mix_num, max_events, _p(do['lab_cat']), _p(do['lab_off']),
                                        _p(do['box_cat']), _p(do['box_off']), _p(do['ratio_cat']), cap, _p(jobs), L.stream_ptr()),
            'mixup_targets')
    return tab_out
