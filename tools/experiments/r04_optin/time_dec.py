"""decoder slab kernel alone (graph-captured), with the developer build's phase ablation: SEDT_SLAB_DBG bit 0 no attention cores, bit 1 no
FFN, bit 2 no K / V image writes"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, packing, runtime, lib as L      # noqa: E402
from sound_event_detection_transformer_amd.sedt.transformer import TransformerDecoderLayer   # noqa: E402
runtime.set_compute_dtype('bf16')
dev = torch.device('cuda')
B, S, E, Q = int(os.environ.get('B', 64)), 128, 256, int(os.environ.get('Q', 11))
g = torch.Generator().manual_seed(1)
rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev).bfloat16()


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


layer = TransformerDecoderLayer(256, 8, 2048, 0.1, 'relu', True).cuda().train()
a_, c_ = layer.self_attn, layer.multihead_attn
lin = [a_.in_proj_weight, a_.out_proj.weight, c_.in_proj_weight, c_.out_proj.weight, layer.linear1.weight, layer.linear2.weight]
plan = packing.PackPlan(L.BF16, dev, [], lin, (), lin)
plan.run()
torch.cuda.synchronize()
fr = [plan.frag_table[w.data_ptr()][0] for w in lin]
tgt, qpos, kc, vc = rnd(B * Q, E), rnd(B * Q, E), rnd(B * S, E), rnd(B * S, E)
vecs = (a_.in_proj_bias, a_.out_proj.bias, c_.in_proj_bias, c_.out_proj.bias, layer.linear1.bias, layer.linear2.bias, layer.norm1.weight,
        layer.norm1.bias, layer.norm2.weight, layer.norm2.bias, layer.norm3.weight, layer.norm3.bias)
vecs = tuple(v.detach() for v in vecs)
for tr in (False, True):
    t = timeit(lambda: ops.decoder_layer_fwd(tgt, qpos, kc, vc, None, None, fr, vecs, B, Q, S, 2048, 0.1, (1, 2, 3, 4, 5, 6), None, train=tr))
    print('dbg %s B %d Q %d decoder_layer_fwd train=%d %7.2f us' % (os.environ.get('SEDT_SLAB_DBG', '0'), B, Q, tr, t))
