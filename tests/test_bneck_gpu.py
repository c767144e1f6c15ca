"""GPU: the fused identity Bottlenecks of layer1 / layer2 (csrc/bneck.hip) against the three per-op launches each replaces, each way.

Both paths round the two 64-channel intermediates (and their gradients) to bf16 and accumulate in f32; the fused kernel sums the
contraction in a different order, so an intermediate may land on the neighbouring bf16 value.  Tolerance: 2e-2 of the tensor's largest
entry (observed <= 8e-3); the sign bits must describe the y the kernel wrote exactly.  Strip edge cases: H not a multiple of the 8-row
strip, H smaller than one strip, a single row."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def cos(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def _stage(seed, which=1):
    """layer1 (frozen) or layer2 (trainable) of the backbone with seeded weights and non-trivial FrozenBN statistics, and the pack plan of
    its convolutions"""
    from sound_event_detection_transformer_amd import packing
    from sound_event_detection_transformer_amd.lib import BF16
    from sound_event_detection_transformer_amd.sedt.backbone import ResNet50Body
    torch.manual_seed(seed)
    body = ResNet50Body(True).cuda()
    layer = {1: body.layer1, 2: body.layer2, 3: body.layer3}[which]
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for b in layer:
            for bn in (b.bn1, b.bn2, b.bn3) + ((b.downsample[1],) if b.downsample is not None else ()):
                bn.weight.copy_(1 + 0.2 * torch.randn(bn.weight.shape, generator=g))
                bn.bias.copy_(0.1 * torch.randn(bn.bias.shape, generator=g))
                bn.running_mean.copy_(0.1 * torch.randn(bn.bias.shape, generator=g))
                bn.running_var.copy_(1 + 0.3 * torch.rand(bn.bias.shape, generator=g))
    for p in layer.parameters():
        p.requires_grad_(which != 1)                    # reference backbone.py:60-62: layer1 is frozen, layer2 trains
    convs, cfr = [], []
    for i, b in enumerate(layer):
        convs += [(b.conv1.weight, b.bn1.tensors()), (b.conv2.weight, b.bn2.tensors()), (b.conv3.weight, b.bn3.tensors())]
        if b.downsample is not None:
            convs.append((b.downsample[0].weight, b.downsample[1].tensors()))
            cfr += [b.conv1.weight, b.conv2.weight, b.conv3.weight, b.downsample[0].weight]      # the projection blocks' own fused forwards
        else:
            cfr += [b.conv1.weight, b.conv2.weight, b.conv3.weight]
    plan = packing.PackPlan(BF16, torch.device('cuda'), convs, [], (), (), cfr)
    return layer, plan


def _run(layer, plan, x, B, H, fused, gy=None, mask_input=False, W=16):
    from sound_event_detection_transformer_amd import functional as Fn, ops
    from sound_event_detection_transformer_amd.lib import BF16
    keep = ops.FUSED_BNECK
    ops.FUSED_BNECK = 5 if fused else 0
    try:
        xin = x.clone().requires_grad_(gy is not None)
        holder = {}
        for p in layer.parameters():
            p.grad = None
        meta = dict(dt=BF16, B=B, H=H, W=W, blocks=[b.cfg for b in layer], mask_input=mask_input, grad_premasked=True, x_bits=None,
                    holder=holder)
        ts = [t for b in layer for t in b.tensors()]
        with plan:
            if gy is None:
                with torch.no_grad():
                    return Fn.StageFn.apply(xin, meta, *ts), None, None
            y = Fn.StageFn.apply(xin, meta, *ts)
            bits = holder.get('bits')
            g = gy * (y.detach() > 0)                     # the consumer's promise: masked by the stage's final ReLU
            y.backward(g)
        return y.detach(), xin.grad, bits
    finally:
        ops.FUSED_BNECK = keep


@pytest.mark.parametrize('B,H', [(2, 125), (3, 13), (1, 8), (2, 1), (1, 32), (40, 125), (50, 125)])
def test_fused_bottleneck_matches_the_per_op_chain(B, H):
    from sound_event_detection_transformer_amd import ops
    layer, plan = _stage(5)
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B * H * 16, 64, generator=g).cuda().bfloat16().relu()          # (layer1's input is the pooled stem output: >= 0)
    gy = torch.randn(B * H * 16, 256, generator=g).cuda().bfloat16()
    assert ops.bneck_ok(ops.BF16, layer[1].cfg, 16) and not ops.bneck_ok(ops.BF16, layer[0].cfg, 16)
    y1, gx1, bits1 = _run(layer, plan, x, B, H, True, gy)
    y0, gx0, bits0 = _run(layer, plan, x, B, H, False, gy)
    assert rel(y1, y0) < 2e-2
    # the two runs round a and b independently: a ReLU mask bit flips where a pre-activation is within rounding of zero, and a flipped
    # unit moves its whole term.  Over 80 k+ pixels the max-norm then finds such an element (observed 4.6e-2 at B = 40, 8e-3 at B = 2);
    # the direction is the check that scales - and the kernel-level test below, where both sides use the SAME masks, holds 1e-2
    assert rel(gx1, gx0) < (2e-2 if B * H < 1000 else 8e-2) and cos(gx1, gx0) > 0.9995
    # the sign bits are those of the y this path wrote: bit c % 8 of byte c / 8
    want = (y1.float() > 0).view(-1, 32, 8).to(torch.uint8)
    packed = (want << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8)
    assert torch.equal(bits1, packed)
    # no-grad forward (teacher / evaluation): same output, no by-products
    y2, _, _ = _run(layer, plan, x, B, H, True)
    assert torch.equal(y2, y1)
    # a trainable block keeps its intermediates and takes the per-op backward (weight gradients): same forward output
    for p in layer[2].parameters():
        p.requires_grad_(True)
    y3, gx3, _ = _run(layer, plan, x, B, H, True, gy)
    assert torch.equal(y3, y1) and rel(gx3, gx0) < (2e-2 if B * H < 1000 else 8e-2) and cos(gx3, gx0) > 0.9995
    assert layer[2].conv2.weight.grad is not None


@pytest.mark.parametrize('B,H', [(2, 125), (3, 13), (1, 17), (70, 125)])
def test_fused_layer2_bottlenecks_match_the_per_op_chain(B, H):
    """layer2 trains: the fused forward keeps a and b, the fused input-gradient chain hands the two intermediate gradients to the
    weight-gradient GEMMs.  Block 0 (stride 2, downsample) stays per-op; the map behind it is ceil(H / 2) x 8"""
    from sound_event_detection_transformer_amd import ops
    layer, plan = _stage(7, which=2)
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B * H * 16, 256, generator=g).cuda().bfloat16().relu()
    H2 = (H - 1) // 2 + 1
    gy = torch.randn(B * H2 * 8, 512, generator=g).cuda().bfloat16()
    assert ops.bneck_ok(ops.BF16, layer[1].cfg, 8) and not ops.bneck_ok(ops.BF16, layer[0].cfg, 16)
    y1, gx1, _ = _run(layer, plan, x, B, H, True, gy, mask_input=True)
    w1 = {n_: p.grad.clone() for n_, p in layer.named_parameters() if p.grad is not None}
    y0, gx0, _ = _run(layer, plan, x, B, H, False, gy, mask_input=True)
    w0 = {n_: p.grad.clone() for n_, p in layer.named_parameters() if p.grad is not None}
    assert rel(y1, y0) < 2e-2
    assert set(w1) == set(w0) and len(w1) == 13          # 4 blocks x 3 convolutions + the downsample projection
    # the two runs round their intermediates independently, so a ReLU mask bit flips where a pre-activation is within rounding of zero
    # (~1e-3 of the elements): a whole term of an input-gradient / weight-gradient sum that has only ~1e3 terms here.  6e-2 covers that
    # (observed 3.3e-2 / 2.6e-2); the kernel-level tests below, where both sides use the SAME masks, hold 1e-2
    assert rel(gx1, gx0) < (6e-2 if B * H < 1000 else 1e-1) and cos(gx1, gx0) > 0.999
    for n_ in w0:
        assert rel(w1[n_], w0[n_]) < 6e-2 and cos(w1[n_], w0[n_]) > 0.999, n_


# (which, B, H).  The small cases run ONE strip per workgroup (spw = ceil(B * ceil(H / 8) / 256) = 1); the production strip walk - a
# workgroup takes spw consecutive strips, prefetching strip s + 1's tile while strip s computes, rotating the weight chunks across
# strips and crossing clip boundaries - needs more than 256 strips: C2 runs layer1 at spw = 4 (B = 64, 16 strips per clip), C4 at
# spw = 13 (B = 200), which does not divide the strips of a clip.  (1, 40, 125): 640 strips, spw 3 (16 per clip); (1, 70, 125): 1120,
# spw 5; (2, 70, 63): 560 strips of layer2's 63 x 8 map, spw 3 (8 per clip); (2, 100, 63): 800, spw 4; (1, 200, 13): spw 2 with 2 strips
# per clip, the second one partial (13 = 8 + 5 rows)
@pytest.mark.parametrize('which,B,H', [(1, 2, 19), (2, 2, 19), (3, 2, 19), (1, 40, 125), (1, 70, 125), (2, 70, 63), (2, 100, 63), (1, 200, 13)])
def test_fused_bottleneck_kernels_against_torch(which, B, H):
    """the two entry points on their own against an f32 torch restatement of the block (conv / FrozenBN affine / ReLU), operands and the
    two intermediates rounded to bf16 as the kernels do; the masks of the input-gradient chain are the kernel's own sign bits"""
    import torch.nn.functional as F
    from sound_event_detection_transformer_amd import ops, packing
    layer, plan = _stage(11, which)
    blk = layer[2]
    W, C, P = {1: (16, 256, 64), 2: (8, 512, 128), 3: (4, 1024, 256)}[which]
    g = torch.Generator().manual_seed(1)
    x = (0.5 * torch.randn(B * H * W, C, generator=g)).cuda().bfloat16().relu()
    gy = torch.randn(B * H * W, C, generator=g).cuda().bfloat16()
    with plan:
        cf = [packing.lookup_conv_frag(w) for w in (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight)]
        sb = [packing.lookup(w)[2:] for w in (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight)]
        y, a, b, bits, abits, bbits = ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb, want_ab=True)
        for t_, tb in ((a, abits), (b, bbits)):             # the sign bits describe the intermediates the kernel wrote
            w_ = (t_.float() > 0).view(-1, P // 8, 8).to(torch.uint8)
            assert torch.equal(tb, (w_ << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8))
        xb = (x.float() > 0).view(-1, C // 8, 8).to(torch.uint8)
        xbits = (xb << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8)
        gym = gy * (y > 0)
        gx, gb, ga = ops.bneck_bwd(gym, B, H, W, [c[1] for c in cf], abits, bbits, xbits, want_g=True)
        gx2, _, _ = ops.bneck_bwd(gym, B, H, W, [c[1] for c in cf], abits, bbits, xbits)
        assert torch.equal(gx, gx2)
        # the no-grad form (no by-products) and the sign-bits-only training form write the same y
        y_ng, a_ng, _, bits_ng, ab_ng, _ = ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb, train=False)
        y_sb, a_sb, _, _, ab_sb, bb_sb = ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], sb)
        assert torch.equal(y_ng, y) and a_ng is None and bits_ng is None and ab_ng is None
        assert torch.equal(y_sb, y) and a_sb is None and torch.equal(ab_sb, abits) and torch.equal(bb_sb, bbits)
        torch.cuda.synchronize()

    def q(t):
        return t.bfloat16().float()

    def nchw(t, Cn):
        return t.float().view(B, H, W, Cn).permute(0, 3, 1, 2)

    def tok(t):
        return t.permute(0, 2, 3, 1).reshape(B * H * W, -1)

    (s1, b1), (s2, b2), (s3, b3) = [(s_.view(1, -1, 1, 1), bb.view(1, -1, 1, 1)) for s_, bb in sb]
    w1, w2, w3 = q(blk.conv1.weight), q(blk.conv2.weight), q(blk.conv3.weight)
    X = nchw(x, C)
    A = q(F.relu(F.conv2d(X, w1) * s1 + b1))
    Bt = q(F.relu(F.conv2d(A, w2, padding=1) * s2 + b2))
    Y = F.relu(F.conv2d(Bt, w3) * s3 + b3 + X)
    assert rel(a, tok(A)) < 1e-2 and rel(b, tok(Bt)) < 1e-2 and rel(y, tok(Y)) < 1e-2
    # input gradients with the BN scale folded into bf16 weights, as the dgrad operands are packed
    GY = nchw(gym, C)
    w3s, w2s, w1s = q(w3 * s3.view(-1, 1, 1, 1)), q(w2 * s2.view(-1, 1, 1, 1)), q(w1 * s1.view(-1, 1, 1, 1))
    GB = q(F.conv_transpose2d(GY, w3s) * (nchw(b, P) > 0))
    GA = q(F.conv_transpose2d(GB, w2s, padding=1) * (nchw(a, P) > 0))
    GX = (F.conv_transpose2d(GA, w1s) + GY) * (X > 0)
    assert rel(gb, tok(GB)) < 1e-2 and rel(ga, tok(GA)) < 1e-2 and rel(gx, tok(GX)) < 1e-2


@pytest.mark.parametrize('B,H', [(3, 21), (40, 125), (70, 125)])
def test_fused_first_block_forward_against_torch(B, H):
    """layer1's block 0 (projection skip) in one forward launch against the f32 torch restatement; the skip path is rounded to bf16 before
    the sum, as the per-op chain stores it.  (40, 125) / (70, 125): 3 / 5 consecutive strips per workgroup (see the note above)"""
    import torch.nn.functional as F
    from sound_event_detection_transformer_amd import ops, packing
    layer, plan = _stage(13, 1)
    blk = layer[0]
    g = torch.Generator().manual_seed(2)
    x = (0.7 * torch.randn(B * H * 16, 64, generator=g)).cuda().bfloat16().relu()
    ws = (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight, blk.downsample[0].weight)
    assert ops.bneck0_ok(ops.BF16, blk.cfg, 16) and not ops.bneck0_ok(ops.BF16, layer[1].cfg, 16)
    with plan:
        cf = [packing.lookup_conv_frag(w) for w in ws]
        sb = [packing.lookup(w)[2:] for w in ws]
        y, a, b, bits, abits, bbits = ops.bneck0_fwd(x, B, H, [c[0] for c in cf], sb, want_ab=True)
        for t_, tb in ((a, abits), (b, bbits)):
            w_ = (t_.float() > 0).view(-1, 8, 8).to(torch.uint8)
            assert torch.equal(tb, (w_ << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8))
        y2, a2, b2, bits2, _, _ = ops.bneck0_fwd(x, B, H, [c[0] for c in cf], sb, train=False)
        torch.cuda.synchronize()
    assert torch.equal(y, y2) and a2 is None and bits2 is None
    want = (y.float() > 0).view(-1, 32, 8).to(torch.uint8)
    assert torch.equal(bits, (want << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8))

    def q(t):
        return t.bfloat16().float()

    def nchw(t, Cn):
        return t.float().view(B, H, 16, Cn).permute(0, 3, 1, 2)

    def tok(t):
        return t.permute(0, 2, 3, 1).reshape(B * H * 16, -1)

    (s1, b1), (s2, b2_), (s3, b3), (sd, bd) = [(s_.view(1, -1, 1, 1), bb.view(1, -1, 1, 1)) for s_, bb in sb]
    w1, w2, w3, wd = (q(w) for w in ws)
    X = nchw(x, 64)
    A = q(F.relu(F.conv2d(X, w1) * s1 + b1))
    Bt = q(F.relu(F.conv2d(A, w2, padding=1) * s2 + b2_))
    I = q(F.conv2d(X, wd) * sd + bd)
    Y = F.relu(F.conv2d(Bt, w3) * s3 + b3 + I)
    assert rel(a, tok(A)) < 1e-2 and rel(b, tok(Bt)) < 1e-2 and rel(y, tok(Y)) < 1e-2


@pytest.mark.parametrize('B,H', [(2, 125), (3, 21), (1, 8), (2, 1), (1, 32), (40, 125), (70, 125)])
def test_fused_layer2_first_block_forward_against_torch(B, H):
    """layer2's block 0 (3x3 stride 2, stride-2 projection skip) in one forward launch against the f32 torch restatement: odd and even input
    heights (the last output row then reads a zero row below the image), fewer output rows than a strip; (40, 125) / (70, 125): 640 / 1120
    strips of 4 output rows = 3 / 5 consecutive strips per workgroup, 16 strips per clip"""
    import torch.nn.functional as F
    from sound_event_detection_transformer_amd import ops, packing
    layer, plan = _stage(17, 2)
    blk = layer[0]
    g = torch.Generator().manual_seed(B * 10 + H)
    x = (0.5 * torch.randn(B * H * 16, 256, generator=g)).cuda().bfloat16().relu()
    ws = (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight, blk.downsample[0].weight)
    assert ops.bneck2_ok(ops.BF16, blk.cfg, 16) and not ops.bneck2_ok(ops.BF16, layer[1].cfg, 8)
    with plan:
        cf = [packing.lookup_conv_frag(w) for w in ws]
        sb = [packing.lookup(w)[2:] for w in ws]
        y, a, b, bits = ops.bneck2_fwd(x, B, H, [c[0] for c in cf], sb)
        y2, a2, b2, bits2 = ops.bneck2_fwd(x, B, H, [c[0] for c in cf], sb, train=False)
        torch.cuda.synchronize()
    assert torch.equal(y, y2) and a2 is None and bits2 is None
    want = (y.float() > 0).view(-1, 64, 8).to(torch.uint8)
    assert torch.equal(bits, (want << torch.arange(8, device='cuda', dtype=torch.uint8)).sum(-1).to(torch.uint8))
    H2 = (H - 1) // 2 + 1

    def q(t):
        return t.bfloat16().float()

    def tok(t):
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])

    (s1, b1), (s2, b2_), (s3, b3), (sd, bd) = [(s_.view(1, -1, 1, 1), bb.view(1, -1, 1, 1)) for s_, bb in sb]
    w1, w2, w3, wd = (q(w) for w in ws)
    X = x.float().view(B, H, 16, 256).permute(0, 3, 1, 2)
    A = q(F.relu(F.conv2d(X, w1) * s1 + b1))
    Bt = q(F.relu(F.conv2d(A, w2, stride=2, padding=1) * s2 + b2_))
    I = q(F.conv2d(X, wd, stride=2) * sd + bd)
    Y = F.relu(F.conv2d(Bt, w3) * s3 + b3 + I)
    assert Y.shape[2] == H2 and Y.shape[3] == 8
    assert rel(a, tok(A)) < 1e-2 and rel(b, tok(Bt)) < 1e-2 and rel(y, tok(Y)) < 1e-2


def test_fused_layer3_bottlenecks_match_the_per_op_chain():
    """layer3's five identity blocks (csrc/bneck3.hip: one 32-pixel strip per workgroup, all weights streamed per strip) inside the stage,
    at a batch whose strips cover the chip once (the envelope: 192 <= B * ceil(H / 8) <= 512); block 0 (stride 2) stays per-op"""
    from sound_event_detection_transformer_amd import ops
    layer, plan = _stage(19, which=3)
    B, H = 48, 63                                            # layer3's input is layer2's output: 63 x 8; behind block 0: 32 x 4
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * H * 8, 512, generator=g).cuda().bfloat16().relu()
    gy = torch.randn(B * 32 * 4, 1024, generator=g).cuda().bfloat16()
    assert ops.bneck_ok(ops.BF16, layer[1].cfg, 4, B, 32) and not ops.bneck_ok(ops.BF16, layer[1].cfg, 4, 8, 32)
    y1, gx1, _ = _run(layer, plan, x, B, H, True, gy, mask_input=True, W=8)
    w1 = {n_: p.grad.clone() for n_, p in layer.named_parameters() if p.grad is not None}
    y0, gx0, _ = _run(layer, plan, x, B, H, False, gy, mask_input=True, W=8)
    w0 = {n_: p.grad.clone() for n_, p in layer.named_parameters() if p.grad is not None}
    assert rel(y1, y0) < 2e-2 and rel(gx1, gx0) < 6e-2
    assert set(w1) == set(w0) and len(w1) == 19              # 6 blocks x 3 convolutions + the downsample projection
    for n_ in w0:
        assert rel(w1[n_], w0[n_]) < 6e-2, n_
