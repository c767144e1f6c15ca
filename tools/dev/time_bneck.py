"""fused layer1 Bottleneck (csrc/bneck.hip) against the per-op launches, forward (training / no-grad) and input-gradient chain, graph-captured.
Developer build: SEDT_BNECK_R=4|8 picks the strip height."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sound_event_detection_transformer_amd import ops, packing, runtime, lib as L      # noqa: E402
from sound_event_detection_transformer_amd.ops import ConvGeom, ACT_RELU               # noqa: E402
from sound_event_detection_transformer_amd.sedt.backbone import ResNet50Body           # noqa: E402
runtime.set_compute_dtype('bf16')
dev = torch.device('cuda')
B, H = int(os.environ.get('B', 64)), int(os.environ.get('H', 125))
g = torch.Generator().manual_seed(1)


def timeit(fn, reps=10):
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


body = ResNet50Body(True).cuda()
blk = body.layer1[1]
ws = (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight)
convs = [(blk.conv1.weight, blk.bn1.tensors()), (blk.conv2.weight, blk.bn2.tensors()), (blk.conv3.weight, blk.bn3.tensors())]
plan = packing.PackPlan(L.BF16, dev, convs, [], (), (), list(ws))
plan.run()
torch.cuda.synchronize()
M = B * H * 16
x = torch.randn(M, 256, generator=g).to(dev).bfloat16().relu()
gy = torch.randn(M, 256, generator=g).to(dev).bfloat16()
cf = [plan.conv_frag_table[w.data_ptr()] for w in ws]
pk = [plan.table[w.data_ptr()] for w in ws]
sb = [p[2:] for p in pk]
y, a, b, bits, abits, bbits = ops.bneck_fwd(x, B, H, [c[0] for c in cf], sb, want_ab=True)
g1, g2, g3 = ConvGeom(H, 16, 256, 64, 1), ConvGeom(H, 16, 64, 64, 3, 1, 1, 1), ConvGeom(H, 16, 64, 256, 1)
ybits = torch.empty((M, 32), device=dev, dtype=torch.uint8)


def per_op_fwd():
    a_ = ops.conv_fwd(L.BF16, x, B, g1, pk[0][0], scale=sb[0][0], bias=sb[0][1], act=ACT_RELU)
    b_ = ops.conv_fwd(L.BF16, a_, B, g2, pk[1][0], scale=sb[1][0], bias=sb[1][1], act=ACT_RELU)
    return ops.conv_fwd(L.BF16, b_, B, g3, pk[2][0], scale=sb[2][0], bias=sb[2][1], res=x, ldr=256, act=ACT_RELU, act_post_res=1, bits_out=ybits)


def per_op_bwd():
    gb = ops.conv_dgrad(L.BF16, gy, B, g3, pk[2][1], mask=b, ldm=64)
    ga = ops.conv_dgrad(L.BF16, gb, B, g2, pk[1][1], mask=a, ldm=64)
    return ops.conv_dgrad(L.BF16, ga, B, g1, pk[0][1], res=gy, ldr=256, mask=bits, ldm=32, mask_bits=True)


tag = 'R %s B %d H %d' % (os.environ.get('SEDT_BNECK_R', '8'), B, H)
print(tag, 'per-op fwd        %7.2f us' % timeit(per_op_fwd))
print(tag, 'fused  fwd train  %7.2f us' % timeit(lambda: ops.bneck_fwd(x, B, H, [c[0] for c in cf], sb)))
print(tag, 'fused  fwd tr+ab  %7.2f us' % timeit(lambda: ops.bneck_fwd(x, B, H, [c[0] for c in cf], sb, want_ab=True)))
print(tag, 'fused  fwd nograd %7.2f us' % timeit(lambda: ops.bneck_fwd(x, B, H, [c[0] for c in cf], sb, train=False)))
print(tag, 'per-op bwd        %7.2f us' % timeit(per_op_bwd))
print(tag, 'fused  bwd        %7.2f us' % timeit(lambda: ops.bneck_bwd(gy, B, H, [c[1] for c in cf], abits, bbits, bits)))
