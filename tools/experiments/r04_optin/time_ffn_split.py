"""the 2-D tiled FFN forward alone (graph-captured); developer build: SEDT_SLAB_DBG bit 0 no linear1 GEMM, 1 no linear2 GEMM, 2 no partial
stores / reduction, 3 no h stores, 4 no linear1 epilogue"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, packing, lib as L      # noqa: E402
dev = torch.device('cuda')
M, E, FF = int(os.environ.get('M', 8192)), 256, 2048
g = torch.Generator().manual_seed(1)
rnd = lambda *sh, s=1.0, dt=torch.bfloat16: (s * torch.randn(*sh, generator=g)).to(dev).to(dt)


def timeit(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for _ in range(reps):
            fn()
    g_.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g_.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


w1, w2 = torch.nn.Parameter(rnd(FF, E, s=0.06, dt=torch.float32)), torch.nn.Parameter(rnd(E, FF, s=0.02, dt=torch.float32))
plan = packing.PackPlan(L.BF16, dev, [], [w1, w2], (), [w1, w2])
plan.run()
torch.cuda.synchronize()
f1, f2 = plan.frag_table[w1.data_ptr()], plan.frag_table[w2.data_ptr()]
x1n, x1 = rnd(M, E), rnd(M, E)
b1, b2 = rnd(FF, dt=torch.float32), rnd(E, dt=torch.float32)
for tr in (True, False):
    t = timeit(lambda: ops.ffn_split_fwd(x1n, x1, f1[0], b1, f2[0], b2, FF, 0.1, (5, 6), None, train=tr))
    print('dbg %s M %d ffn_split_fwd train=%d %7.2f us' % (os.environ.get('SEDT_SLAB_DBG', '0'), M, tr, t))
