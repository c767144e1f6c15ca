"""how much does a GEMM launch lose when its weights are not L2-resident?  A launch timed hot (same launch repeated) and cold (a 256 MB
buffer written between two launches: L2 and most of the MALL hold other data)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, runtime, lib as L      # noqa: E402
runtime.set_compute_dtype('bf16')
dev = torch.device('cuda')
g = torch.Generator().manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev).bfloat16()
big = torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.float32)


def timeit(fn, reps=10):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
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


flush = lambda: big.fill_(1.0)
t_flush = timeit(flush)
for M, N, K in ((8192, 256, 1024), (8192, 1024, 256), (8192, 256, 2304), (8192, 512, 2048), (8192, 2048, 512), (8192, 2048, 256), (704, 256, 256)):
    x, w = rnd(M, K), rnd(N, K)
    hot = timeit(lambda: ops.linear(L.BF16, x, w))
    cold = timeit(lambda: (flush(), ops.linear(L.BF16, x, w))) - t_flush
    # activations hot, weights cold: touch x after the flush
    xw = timeit(lambda: (flush(), x.add_(0), ops.linear(L.BF16, x, w))) - timeit(lambda: (flush(), x.add_(0)))
    print('M %5d N %5d K %5d  hot %6.2f us   all cold %6.2f us   weights cold, x hot %6.2f us' % (M, N, K, hot, cold, xw))
