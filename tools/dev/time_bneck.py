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
LAYER = int(os.environ.get('LAYER', 1))
B, H = int(os.environ.get('B', 64)), int(os.environ.get('H', {1: 125, 2: 63, 3: 32}[LAYER]))
W, C, P = {1: (16, 256, 64), 2: (8, 512, 128), 3: (4, 1024, 256)}[LAYER]
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
blk = {1: body.layer1, 2: body.layer2, 3: body.layer3}[LAYER][1]
ws = (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight)
convs = [(blk.conv1.weight, blk.bn1.tensors()), (blk.conv2.weight, blk.bn2.tensors()), (blk.conv3.weight, blk.bn3.tensors())]
plan = packing.PackPlan(L.BF16, dev, convs, [], (), (), list(ws))
plan.run()
torch.cuda.synchronize()
M = B * H * W
x = torch.randn(M, C, generator=g).to(dev).bfloat16().relu()
gy = torch.randn(M, C, generator=g).to(dev).bfloat16()
cf = [plan.conv_frag_table[w.data_ptr()] for w in ws]
pk = [plan.table[w.data_ptr()] for w in ws]
sb = [p[2:] for p in pk]
y, a, b, bits, abits, bbits = ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb, want_ab=True)
g1, g2, g3 = ConvGeom(H, W, C, P, 1), ConvGeom(H, W, P, P, 3, 1, 1, 1), ConvGeom(H, W, P, C, 1)
ybits = torch.empty((M, C // 8), device=dev, dtype=torch.uint8)


def per_op_fwd():
    a_ = ops.conv_fwd(L.BF16, x, B, g1, pk[0][0], scale=sb[0][0], bias=sb[0][1], act=ACT_RELU)
    b_ = ops.conv_fwd(L.BF16, a_, B, g2, pk[1][0], scale=sb[1][0], bias=sb[1][1], act=ACT_RELU)
    return ops.conv_fwd(L.BF16, b_, B, g3, pk[2][0], scale=sb[2][0], bias=sb[2][1], res=x, ldr=C, act=ACT_RELU, act_post_res=1, bits_out=ybits)


def per_op_bwd():
    gb = ops.conv_dgrad(L.BF16, gy, B, g3, pk[2][1], mask=b, ldm=P)
    ga = ops.conv_dgrad(L.BF16, gb, B, g2, pk[1][1], mask=a, ldm=P)
    return ops.conv_dgrad(L.BF16, ga, B, g1, pk[0][1], res=gy, ldr=C, mask=bits, ldm=C // 8, mask_bits=True)


tag = 'layer%d B %d H %d' % (LAYER, B, H)
print(tag, 'per-op fwd        %7.2f us' % timeit(per_op_fwd))
print(tag, 'fused  fwd train  %7.2f us' % timeit(lambda: ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb)))
print(tag, 'fused  fwd tr+ab  %7.2f us' % timeit(lambda: ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb, want_ab=True)))
print(tag, 'fused  fwd nograd %7.2f us' % timeit(lambda: ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb, train=False)))
print(tag, 'per-op bwd        %7.2f us' % timeit(per_op_bwd))
print(tag, 'fused  bwd        %7.2f us' % timeit(lambda: ops.bneck_bwd(gy, B, H, W, [c[1] for c in cf], abits, bbits, bits)))
print(tag, 'fused  bwd + g    %7.2f us' % timeit(lambda: ops.bneck_bwd(gy, B, H, W, [c[1] for c in cf], abits, bbits, bits, want_g=True)))
