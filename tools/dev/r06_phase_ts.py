"""developer build: where a workgroup of the LDS-DMA GEMM spends its life (csrc/igemm3.hip, SEDT_TS): prologue (descriptors, gather
offsets, epilogue-operand prefetch, first tiles issued) | K loop | epilogue (LDS staging, residual / mask, stores issued), shader clocks,
for one problem alone in five cache states (profiles/r06_ab_bpf.txt; SEDT_IGEMM_BPF=0 SEDT_IGEMM_CPF=0 shows the kernels without their prefetches).
usage (on the GPU box): SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so SEDT_IGEMM_BREG=1 python tools/dev/r06_phase_ts.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sound_event_detection_transformer_amd import ops, lib as L   # noqa: E402

lib = L.load()
lib.sedt_dev_phase_ts.argtypes = [C.c_void_p, C.c_int]
g = torch.Generator().manual_seed(1)
flush = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')
small = torch.empty(48 << 20, dtype=torch.uint8, device='cuda')      # evicts the L2s (32 MB), stays inside the Infinity Cache next to the problem


def frag(w):
    N, K = w.shape
    return w.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def run(M, N, K, tile, use_frag, cold, with_res=True):
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()
    sc, bi = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
    fr = frag(w)
    x2, res2 = x.clone(), res.clone()
    wcopies = [w.clone() for _ in range(max(8, (352 << 20) // (w.numel() * 2)))] if cold == 'Bhbm' else []
    wi = [0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    if with_res:
        a = ops.igemm_args(M, N, K, x, K, w, K, y, N, scale=sc, bias=bi, res=res, ldr=N, act=L.ACT_RELU, act_post_res=1, tile=tile)
    elif with_res is False:
        a = ops.igemm_args(M, N, K, x, K, w, K, y, N, scale=sc, bias=bi, act=L.ACT_RELU, tile=tile)
    else:
        a = ops.igemm_args(M, N, K, x, K, w, K, y, N, tile=tile)
    if use_frag:
        a.bfrag = fr.data_ptr()
    buf = C.create_string_buffer(160)
    lib.sedt_igemm_describe(C.byref(a), L.BF16, 0, buf, 160)
    name = buf.value.decode()
    bm, bn = [int(v) for v in name.split('<')[1].split(',')[:2]]
    nwg = min(4096, ((M + bm - 1) // bm) * ((N + bn - 1) // bn))
    stats = []
    for rep in range(6):
        if cold == 'Bhbm':                      # the weight from HBM (a fresh copy out of 320+ MB of copies every launch), everything else warm
            L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
            wi[0] = (wi[0] + 1) % len(wcopies)
            a.B = wcopies[wi[0]].data_ptr()
        elif cold == 'wrA':                       # A freshly WRITTEN by another kernel (as inside the step), everything else warm
            L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
            x.copy_(x2)
        elif cold == 'wrAR':                    # A and the residual freshly written
            L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
            x.copy_(x2)
            res.copy_(res2)
        elif cold == 'mall':
            L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
            small.fill_(rep)
        elif cold:
            flush.fill_(rep)
        else:
            L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
        e1.record()
        torch.cuda.synchronize()
        ts = np.zeros((nwg, 5), np.uint64)
        assert lib.sedt_dev_phase_ts(ts.ctypes.data, nwg) == 0
        ts = ts.astype(np.int64)
        pro, loop, epi = ts[:, 1] - ts[:, 0], ts[:, 2] - ts[:, 1], ts[:, 3] - ts[:, 2]
        span_rt = (ts[:, 4].max() - ts[:, 4].min()) / 100.0           # us between the first and the last workgroup START (100 MHz clock)
        stats.append((e0.elapsed_time(e1) * 1e3, np.median(pro), np.median(loop), np.median(epi), np.percentile(loop, 90), span_rt))
    s = np.asarray(stats[1:]).mean(0)
    print(f'{M:6d} {N:5d} {K:5d} {name:34s} {"res  " if with_res is True else "nores" if with_res is False else "plain"} {cold if isinstance(cold, str) else "cold" if cold else "hot ":4s} launch {s[0]:7.1f} us | wg clocks: prologue {s[1]:7.0f}  K loop {s[2]:7.0f} (p90 {s[4]:7.0f})  '
          f'epilogue {s[3]:7.0f} | first..last wg start {s[5]:6.1f} us, {nwg} wgs', flush=True)


for (M, N, K, tile) in ((8192, 2048, 512, (0, 0)), (8192, 2048, 1024, (0, 0)), (8192, 512, 2048, (0, 0)), (8192, 512, 1024, (64, 128)), (32256, 256, 1024, (64, 128))):
    for use_frag in (False, True) if os.environ.get('SEDT_IGEMM_BREG') == '1' else (False,):
        for with_res in (True,):
            for cold in (True, 'mall', 'wrAR', 'Bhbm', False):      # flushed | L2 evicted | A + residual rewritten | weight from HBM | repeated
                run(M, N, K, tile, use_frag, cold, with_res)
