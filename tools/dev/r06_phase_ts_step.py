"""developer build: the phase stamps (csrc/igemm3.hip, SEDT_TS) of ONE problem shape as it runs INSIDE the captured C2 step - the last launch of
that (M, N, K) in a replay - next to the same argument block launched alone right after (inputs warm in the Infinity Cache).
usage (GPU box): SEDT_DEV=1 SEDT_LIB_AB=build/dev/libsedt_hip_dev.so python tools/dev/r06_phase_ts_step.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                                                    # noqa: E402
from sound_event_detection_transformer_amd import lib as L, runtime                            # noqa: E402

lib = L.load()
lib.sedt_dev_phase_ts.argtypes = [C.c_void_p, C.c_int]
sys.argv = ['bench.py', '--config', 'c2']
args = bench.parse()
runtime.set_compute_dtype('bf16')
dev = torch.device('cuda:0')
step, clips, flop, what, graphed, ex = bench.build_workload(args, dev, 0, 1)
for _ in range(30):
    step()
torch.cuda.synchronize()


def stamps(nwg):
    ts = np.zeros((nwg, 5), np.uint64)
    assert lib.sedt_dev_phase_ts(ts.ctypes.data, nwg) == 0
    ts = ts.astype(np.int64)
    life = ts[:, 3] - ts[:, 0]
    ok = life > 0
    ts = ts[ok]
    pro, loop, epi = ts[:, 1] - ts[:, 0], ts[:, 2] - ts[:, 1], ts[:, 3] - ts[:, 2]
    span = (ts[:, 4].max() - ts[:, 4].min()) / 100.0
    return f'prologue {np.median(pro):6.0f}  K loop {np.median(loop):7.0f} (p90 {np.percentile(loop, 90):7.0f})  epilogue {np.median(epi):6.0f} | wg starts span {span:6.1f} us, {len(ts)} wgs'


for (M, N, K, bm, bn) in ((8192, 2048, 512, 64, 128), (8192, 2048, 1024, 64, 128), (8192, 512, 2048, 128, 128), (8192, 512, 1024, 64, 128), (32256, 256, 1024, 64, 128),
                          (32256, 1024, 256, 64, 64), (8192, 256, 2048, 64, 128)):
    nwg = min(4096, ((M + bm - 1) // bm) * ((N + bn - 1) // bn))
    assert lib.sedt_dev_ts_filter(M, N, K) == 0
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(f'{M:6d} {N:5d} {K:5d} in step : {stamps(nwg)}', flush=True)
lib.sedt_dev_ts_filter(0, 0, 0)

# ---- code-cold, data-warm: a whole C2 step runs between two launches of an isolated problem (its code leaves the instruction caches and,
#      with 13 GB of traffic, the Infinity Cache), then copy kernels rewrite the problem's operands (data warm again) and the problem runs
if os.environ.get('SEDT_TS_CODECOLD') == '1':
    from sound_event_detection_transformer_amd import ops
    g = torch.Generator().manual_seed(1)
    for (M, N, K, tile) in ((8192, 2048, 512, (0, 0)), (8192, 512, 2048, (0, 0)), (8192, 512, 1024, (64, 128))):
        x = torch.randn(M, K, generator=g).bfloat16().cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
        res = torch.randn(M, N, generator=g).bfloat16().cuda()
        sc, bi = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()
        x2, w2, res2, sc2, bi2 = x.clone(), w.clone(), res.clone(), sc.clone(), bi.clone()
        y = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
        a = ops.igemm_args(M, N, K, x, K, w, K, y, N, scale=sc, bias=bi, res=res, ldr=N, act=L.ACT_RELU, act_post_res=1, tile=tile)
        buf = C.create_string_buffer(160)
        lib.sedt_igemm_describe(C.byref(a), L.BF16, 0, buf, 160)
        bm, bn = [int(v) for v in buf.value.decode().split('<')[1].split(',')[:2]]
        nwg = min(4096, ((M + bm - 1) // bm) * ((N + bn - 1) // bn))
        for mode in ('hot', 'codecold'):
            for rep in range(4):
                lib.sedt_dev_ts_filter(1, 1, 1)                     # nothing inside the step records
                if mode == 'codecold':
                    step()
                x.copy_(x2); w.copy_(w2); res.copy_(res2); sc.copy_(sc2); bi.copy_(bi2)
                y.zero_()
                lib.sedt_dev_ts_filter(0, 0, 0)
                L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
                torch.cuda.synchronize()
            print(f'{M:6d} {N:5d} {K:5d} alone, {mode:8s}: {stamps(nwg)}', flush=True)
