"""GPU: the x-stationary slab kernels (csrc/slab.h, enc_slab.hip) against the per-op chain they replace.

Both paths make the same bf16 rounding decisions at every tensor the per-op chain materialises and draw identical dropout masks
(counter hashes of (seed, element index)); what differs is the f32 summation order inside the GEMMs.  Tolerance: 2e-2 of the tensor's
largest entry for activations and gradients in bf16 (observed <= 8e-3), exact for the weight packing."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def test_pack_frag_layout():
    """fragment-major packing (include/sedt_hip.h: SedtFragJob): block (n/32, k/16) of 1 KB, lane 32*((k%16)/8) + n%32 owns 8 k"""
    from sound_event_detection_transformer_amd import packing
    from sound_event_detection_transformer_amd.lib import BF16
    g = torch.Generator().manual_seed(3)
    ws = [torch.nn.Parameter(torch.randn(n, k, generator=g).cuda()) for n, k in ((768, 256), (256, 2048), (64, 32))]
    plan = packing.PackPlan(BF16, ws[0].device, [], ws, (), ws)
    with plan:
        for w in ws:
            wf, wb = packing.lookup_frag(w)
            N, K = w.shape
            ref = w.detach().bfloat16().cpu()
            # W: [N/32][K/16][2 halves][32 rows][8]
            f = wf.cpu().view(N // 32, K // 16, 2, 32, 8)
            want = ref.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4)
            assert torch.equal(f, want)
            # W^T: features = k, contraction = n
            t = wb.cpu().view(K // 32, N // 16, 2, 32, 8)
            want_t = ref.t().contiguous().view(K // 32, 32, N // 16, 2, 8).permute(0, 2, 3, 1, 4)
            assert torch.equal(t, want_t)


@pytest.fixture(autouse=True)
def _small_batches_take_the_slab_path():
    from sound_event_detection_transformer_amd import ops
    keep = ops.SLAB_MIN_WGS
    ops.SLAB_MIN_WGS = 1            # (the fill rule B * ceil(S / 32) >= 192 would send these small test batches down the per-op chain)
    yield
    ops.SLAB_MIN_WGS = keep


def _layer_and_plan(seed):
    from sound_event_detection_transformer_amd import packing
    from sound_event_detection_transformer_amd.lib import BF16
    from sound_event_detection_transformer_amd.sedt.transformer import TransformerEncoderLayer
    torch.manual_seed(seed)
    layer = TransformerEncoderLayer(256, 8, 2048, 0.1, 'relu', True).cuda().train()
    with torch.no_grad():
        for n_, p in layer.named_parameters():
            if 'norm' in n_:
                p.add_(0.1 * torch.randn_like(p))
            elif p.dim() == 1:
                p.normal_(0, 0.05)
    a = layer.self_attn
    lin = [a.in_proj_weight, a.out_proj.weight, layer.linear1.weight, layer.linear2.weight]
    return layer, packing.PackPlan(BF16, torch.device('cuda'), [], lin, (), lin)


@pytest.mark.parametrize('B,S,pad', [(3, 128, 0), (2, 124, 0), (2, 128, 37), (5, 40, 9)])
@pytest.mark.parametrize('train', [True, False])
def test_encoder_slab_layer_matches_per_op_chain(B, S, pad, train):
    from sound_event_detection_transformer_amd import ops, runtime
    runtime.set_compute_dtype('bf16')
    try:
        layer, plan = _layer_and_plan(11)
        if not train:
            layer.eval()
        g = torch.Generator().manual_seed(5)
        x0 = torch.randn(B * S, 256, generator=g).cuda().bfloat16()
        pos = (0.5 * torch.randn(B * S, 256, generator=g)).cuda().bfloat16()
        kpm = torch.zeros(B, S, dtype=torch.uint8)
        if pad:
            kpm[0, S - pad:] = 1                      # one clip with padded keys
        kpm = kpm.cuda()
        gy = torch.randn(B * S, 256, generator=g).cuda().bfloat16()
        res = {}
        for mode in ('slab', 'chain'):
            ops.SLAB_ENC = mode == 'slab'
            runtime.manual_seed(99)
            x = x0.clone().requires_grad_(train)
            for p in layer.parameters():
                p.grad = None
            with plan:
                assert ops.encoder_slab_ok(1, 256, 8, S, 2048, None) == (mode == 'slab')
                y = layer.forward_tokens(x, pos, kpm, B, S)
                if train:
                    y.backward(gy)
            res[mode] = (y.detach().clone(), None if not train else x.grad.clone(),
                         {n_: p.grad.clone() for n_, p in layer.named_parameters()} if train else {})
        live = (kpm.view(B, S, 1) == 0).expand(B, S, 256).reshape(B * S, 256)      # (padded QUERY rows still produce defined values: compare all)
        assert rel(res['slab'][0], res['chain'][0]) < 2e-2
        assert live.any()
        if train:
            assert rel(res['slab'][1], res['chain'][1]) < 2e-2
            for n_ in res['chain'][2]:
                assert rel(res['slab'][2][n_], res['chain'][2][n_]) < 2e-2, n_
    finally:
        ops.SLAB_ENC = True
        runtime.set_compute_dtype('f32')


def test_encoder_slab_layer_dropout_masks_equal_the_chains():
    """with the FFN weights zeroed except the biases the layer output is x + dropout(attn) + dropout(b2 + 0): the positions the two
    paths zero must coincide exactly (same hashes), not just statistically"""
    from sound_event_detection_transformer_amd import ops, runtime
    runtime.set_compute_dtype('bf16')
    try:
        layer, plan = _layer_and_plan(12)
        with torch.no_grad():
            layer.linear2.weight.zero_()
            layer.linear2.bias.fill_(3.0)
            layer.self_attn.out_proj.weight.zero_()
            layer.self_attn.out_proj.bias.fill_(5.0)
        B, S = 2, 128
        x = torch.zeros(B * S, 256).cuda().bfloat16()
        pos = torch.zeros_like(x)
        outs = []
        for mode in (True, False):
            ops.SLAB_ENC = mode
            runtime.manual_seed(7)
            with plan, torch.no_grad():
                outs.append(layer.forward_tokens(x, pos, None, B, S).float())
        # values are in {0, 5/0.9, 3/0.9, 8/0.9} (+ rounding): identical patterns
        assert torch.equal(outs[0], outs[1])
        kept = (outs[0] > 1).float().mean().item()
        assert 0.95 < kept < 1.0
    finally:
        ops.SLAB_ENC = True
        runtime.set_compute_dtype('f32')


def test_encoder_slab_backward_matches_per_op_backward():
    """the slab input-gradient chain (sedt_encoder_ffn_bwd | attention | sedt_encoder_qkv_bwd) against the per-op backward kernels fed
    the SAME saved tensors: the slab forward in both runs, ops.SLAB_ENC_BWD toggled"""
    from sound_event_detection_transformer_amd import ops, runtime
    runtime.set_compute_dtype('bf16')
    try:
        layer, plan = _layer_and_plan(13)
        B, S = 3, 124
        g = torch.Generator().manual_seed(6)
        x0 = torch.randn(B * S, 256, generator=g).cuda().bfloat16()
        pos = (0.5 * torch.randn(B * S, 256, generator=g)).cuda().bfloat16()
        kpm = torch.zeros(B, S, dtype=torch.uint8)
        kpm[1, 100:] = 1
        kpm = kpm.cuda()
        gy = torch.randn(B * S, 256, generator=g).cuda().bfloat16()
        res = {}
        for mode in (True, False):
            ops.SLAB_ENC_BWD = mode
            runtime.manual_seed(98)
            x = x0.clone().requires_grad_(True)
            for p in layer.parameters():
                p.grad = None
            with plan:
                layer.forward_tokens(x, pos, kpm, B, S).backward(gy)
            res[mode] = (x.grad.clone(), {n_: p.grad.clone() for n_, p in layer.named_parameters()})
        assert rel(res[True][0], res[False][0]) < 2e-2
        for n_ in res[False][1]:
            assert rel(res[True][1][n_], res[False][1][n_]) < 2e-2, n_
    finally:
        ops.SLAB_ENC_BWD = True
        runtime.set_compute_dtype('f32')


def _dec_layer_and_plan(seed):
    from sound_event_detection_transformer_amd import packing
    from sound_event_detection_transformer_amd.lib import BF16
    from sound_event_detection_transformer_amd.sedt.transformer import TransformerDecoderLayer
    torch.manual_seed(seed)
    layer = TransformerDecoderLayer(256, 8, 2048, 0.1, 'relu', True).cuda().train()
    with torch.no_grad():
        for n_, p in layer.named_parameters():
            if 'norm' in n_:
                p.add_(0.1 * torch.randn_like(p))
            elif p.dim() == 1:
                p.normal_(0, 0.05)
    a, c = layer.self_attn, layer.multihead_attn
    lin = [a.in_proj_weight, a.out_proj.weight, c.in_proj_weight, c.out_proj.weight, layer.linear1.weight, layer.linear2.weight]
    return layer, packing.PackPlan(BF16, torch.device('cuda'), [], lin, (), lin)


@pytest.mark.parametrize('L_,B,Qp,dec_at', [(3, 4, 11, True), (3, 2, 21, True), (3, 5, 20, False), (1, 3, 11, True)])
@pytest.mark.parametrize('train', [True, False])
def test_heads_slab_kernels_match_per_op_heads(L_, B, Qp, dec_at, train):
    """functional.HeadsFn on csrc/heads_slab.hip (one launch each way) against its per-op form (3 skinny kernels + 2 GEMMs forward, 10
    launches backward): outputs, the input gradient and all head parameter gradients"""
    from sound_event_detection_transformer_amd import ops, runtime, packing, functional as Fn
    from sound_event_detection_transformer_amd.lib import BF16
    runtime.set_compute_dtype('bf16')
    try:
        g = torch.Generator().manual_seed(31)
        P = lambda *sh, s=0.06: torch.nn.Parameter((s * torch.randn(*sh, generator=g)).cuda())
        wc, bc = P(11, 256), P(11)
        w1, b1, w2, b2, w3, b3 = P(256, 256), P(256), P(256, 256), P(256), P(2, 256), P(2)
        wa, ba = (P(10, 256), P(10)) if dec_at else (None, None)
        params = [p for p in (wc, bc, w1, b1, w2, b2, w3, b3, wa, ba) if p is not None]
        plan = packing.PackPlan(BF16, torch.device('cuda'), [], [w1, w2], (), [w1, w2])
        hs0 = torch.randn(L_, B, Qp, 256, generator=g).cuda().bfloat16()
        gc, gb = torch.randn(L_, B, Qp, 11, generator=g).cuda(), torch.randn(L_, B, Qp, 2, generator=g).cuda()
        ga = torch.randn(B, 10, generator=g).cuda()
        res = {}
        for mode in (True, False):
            ops.SLAB_HEADS = mode
            hs = hs0.clone().requires_grad_(train)
            for p in params:
                p.grad = None
            with plan:
                out = Fn.HeadsFn.apply(hs, wc, bc, w1, b1, w2, b2, w3, b3, wa, ba, BF16)
                if train:
                    loss = (out[0] * gc).sum() + (out[1] * gb).sum() + ((out[2] * ga).sum() if dec_at else 0)
                    loss.backward()
            res[mode] = ([o.detach().clone() for o in out], hs.grad.clone() if train else None, [p.grad.clone() for p in params] if train else [])
        for a_, b_ in zip(res[True][0], res[False][0]):
            assert rel(a_, b_) < 5e-3
        if train:
            assert rel(res[True][1], res[False][1]) < 2e-2
            for a_, b_ in zip(res[True][2], res[False][2]):
                assert a_.is_contiguous() and rel(a_, b_) < 2e-2
    finally:
        ops.SLAB_HEADS = True
        runtime.set_compute_dtype('f32')
