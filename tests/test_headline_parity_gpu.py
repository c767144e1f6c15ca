"""GPU: the kernels the HEADLINE benchmark runs (C2: B = 64, bf16) against the CPU oracle, with the dispatch asserted.

The fused kernels of the bf16 throughput mode choose themselves by batch-dependent fill rules (ops.encoder_slab_ok: 192 <= B * ceil(S/32)
<= 320 slabs; sedt_bneck3_ok: 192 <= B * ceil(H/8) <= 512 strips; ops.HEADS_SLAB_MAX_ROWS), so the B = 2 / B = 4 fixture tests never
reach `enc_slab.hip` or `bneck3.hip`: their parity claims would be about another kernel set than the one the throughput claim is made
on.  Every test here runs at a batch inside the windows, compares with the ORACLE (oracle/sedt_oracle.py, pinned to the reference by
fixtures G1-G15; its autograd runs on the box's host cores), and asserts through ``lib.launch_log()`` which entry points - and which
GEMM kernel instances - the run dispatched.

Tolerances (bf16 operands and activations, f32 accumulation; `rel` = max |difference| / max |reference| per tensor):
  * model outputs at B = 64: the bounds of test_bf16_mode_error_vs_fixtures... (pred_logits 4.5e-2, pred_boxes 3e-2, at 2.2e-2;
    measured here: printed by the test);
  * gradient directions at B = 64 under the smooth surrogate loss: cosine >= 0.997 per tensor, conv0's six scalars >= 0.95 (the
    bounds of test_gradient_parity_gpu.py at B = 4);
  * one slab encoder layer against the oracle's TransformerEncoderLayer on bf16-rounded inputs and weights: forward 1e-2, input
    gradient 4e-2 / cosine 0.9999, weight gradients 3e-2 / cosine 0.9999 - except those behind the FFN's ReLU kink (linear1, norm2:
    1.5e-1 / cosine 0.998; why: at the assertion)."""
import numpy as np
import pytest
import torch

from oracle import sedt_oracle as O

pytestmark = pytest.mark.gpu

BF16_OUT_BOUNDS = {'pred_logits': 4.5e-2, 'pred_boxes': 3e-2, 'at': 2.2e-2}


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def _rc(got, ref):
    """(max |difference| / max |reference|, cosine)"""
    a, b = got.detach().double().flatten().cpu(), ref.detach().double().flatten().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300)), float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


@pytest.fixture(scope='module')
def pkg():
    from sound_event_detection_transformer_amd import lib, ops, runtime, sedt
    assert torch.cuda.is_available()
    return lib, ops, runtime, sedt


def _pair(sedt, seed, train):
    oracle = O.build_oracle_model(10, 10, 3, 3, True, True, True, dropout=0.0)
    sd = O.seeded_state_dict(oracle.state_dict(), seed)
    oracle.load_state_dict(sd)
    model, _, _ = sedt.build_model(sedt.default_args(dropout=0.0))
    model.load_state_dict(sd)
    model.cuda()
    return (oracle.train(), model.train()) if train else (oracle.eval(), model.eval())


# entry points one C2-shaped bf16 forward must go through (E = 3 encoder layers; layer1: block 0 fused forward + 2 identity blocks,
# layer2: block 0 fused forward + 3 identity blocks, layer3: 5 identity blocks on bneck3; one heads launch; one-launch stem)
FWD_DISPATCH = {'encoder_qkv_fwd': 3, 'encoder_attn_ffn_fwd': 3, 'bneck0_fwd': 1, 'bneck2_fwd': 1, 'bneck_fwd': 5, 'bneck3_fwd': 5,
                'heads_fwd': 1, 'stem_pool_fwd': 1}
# ... and its backward: slab encoder chain around the attention backward (3 encoder + 2 x 3 decoder attention backwards), the fused
# input-gradient chains (layer1: 2 identity + block 0's chain-only form, layer2: 3; layer3: 5), one heads launch, one stem launch
BWD_DISPATCH = {'encoder_ffn_bwd': 3, 'encoder_qkv_bwd': 3, 'attention_bwd': 9, 'bneck_bwd': 6, 'bneck3_bwd': 5, 'heads_bwd': 1,
                'stem_pool_wgrad': 1}


def _assert_dispatch(log, want, what):
    got = {k: log.get(k, 0) for k in want}
    assert got == want, (what, got, {k: v for k, v in log.items() if not k.startswith('igemm:')})


def test_b64_bf16_forward_runs_the_headline_kernels_and_matches_the_oracle(pkg, capsys):
    """C2's batch through the bf16 path, no-grad (eval / teacher form of the fused kernels) and with autograd recording (training form,
    by-products written), against the oracle on sampled clips"""
    lib, ops, runtime, sedt = pkg
    B, pick = 64, [0, 21, 42, 63]
    x = torch.randn(B, 1, 500, 64, generator=torch.Generator().manual_seed(15))
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    out = {}
    for form in ('nograd', 'train'):
        oracle, model = _pair(sedt, 2020, form == 'train')
        with torch.no_grad():
            ref = oracle(x[pick])
        runtime.set_compute_dtype('bf16')
        try:
            with lib.launch_log() as log:
                if form == 'nograd':
                    with torch.no_grad():
                        o = model(x.cuda())
                else:
                    o = model(x.cuda())
            torch.cuda.synchronize()
        finally:
            runtime.set_compute_dtype('f32')
        _assert_dispatch(log, FWD_DISPATCH, form)
        # layer4 (40 % of the forward flops) on the 128x128 ping-pong tile, layer3's projection block / the FFN-sized problems on the
        # 16-wave form, nothing of the forward on the generic register-staged GEMM except the tiny unaligned heads (none here: one launch)
        kinds = {k[6:]: v for k, v in log.items() if k.startswith('igemm:')}
        assert any(k.startswith('igemm3_w8_kernel<128, 128') for k in kinds), kinds
        assert not any(k.startswith('igemm_kernel<') for k in kinds), kinds
        errs = {k: rel(o[k][pick], ref[k]) for k in BF16_OUT_BOUNDS}
        for i, a in enumerate(o['aux_outputs']):
            errs[f'aux{i}_logits'] = rel(a['pred_logits'][pick], ref['aux_outputs'][i]['pred_logits'])
            errs[f'aux{i}_boxes'] = rel(a['pred_boxes'][pick], ref['aux_outputs'][i]['pred_boxes'])
        out[form] = errs
        for k, v in errs.items():
            bound = BF16_OUT_BOUNDS['pred_logits' if 'logits' in k else 'pred_boxes' if 'boxes' in k else k]
            assert v < bound, (form, k, v, bound)
    with capsys.disabled():
        for form, errs in out.items():
            print(f'\n[B = 64 bf16 forward on the headline kernels vs the oracle, {form}] ' + ', '.join(f'{k}={v:.2e}' for k, v in errs.items()))


def _smooth_loss(o):
    t = o['pred_logits'].float().square().mean() + 3.0 * o['pred_boxes'].float().square().mean() + o['at'].float().square().mean()
    for i, a in enumerate(o['aux_outputs']):
        t = t + (0.5 + 0.25 * i) * (a['pred_logits'].float().square().mean() + 3.0 * a['pred_boxes'].float().square().mean())
    return t


def test_b64_bf16_gradient_directions_on_the_headline_kernels_against_the_oracle(pkg, capsys):
    """forward + backward of C2's batch through the fused kernels (slab encoder both ways, fused Bottleneck chains of layer1/2/3, one-launch
    heads, 256x128 weight-gradient tiles) against the oracle's f32 autograd on the same 64 clips: every trainable tensor's direction"""
    lib, ops, runtime, sedt = pkg
    B = 64
    x = torch.randn(B, 1, 500, 64, generator=torch.Generator().manual_seed(41))
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    oracle, model = _pair(sedt, 42, True)
    _smooth_loss(oracle(x)).backward()
    runtime.set_compute_dtype('bf16')
    try:
        with lib.launch_log() as log:
            _smooth_loss(model(x.cuda())).backward()
        torch.cuda.synchronize()
    finally:
        runtime.set_compute_dtype('f32')
    _assert_dispatch(log, FWD_DISPATCH, 'forward')
    _assert_dispatch(log, BWD_DISPATCH, 'backward')
    assert log.get('wgrad_group', 0) >= 10, dict(log)
    po = dict(oracle.named_parameters())
    cosines, rels = {}, {}
    for n, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, n
            continue
        ref = po[n].grad
        if ref.abs().max().item() == 0:
            continue
        a, b = p.grad.detach().double().flatten().cpu(), ref.double().flatten()
        cosines[n] = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        rels[n] = float((a - b).abs().max() / b.abs().max())
    v = np.array(list(cosines.values()))
    low = {n: c for n, c in cosines.items() if c < 0.997}
    with capsys.disabled():
        print(f'\n[B = 64 bf16 gradient directions on the headline kernels vs the oracle, smooth loss, {len(v)} tensors] min {v.min():.5f} '
              f'({min(cosines, key=cosines.get)}), 1st percentile {np.percentile(v, 1):.5f}, median {np.median(v):.6f}; '
              f'worst max-rel {max(rels.values()):.2e} ({max(rels, key=rels.get)}); below 0.997: {sorted(low.items(), key=lambda kv: kv[1])[:6]}')
    assert len(v) >= 150
    assert all('conv0' in n for n in low), low
    assert all(c > 0.95 for c in low.values()), low


def _oracle_layer(layer):
    """the oracle's pre-norm encoder layer with this HIP layer's weights rounded to bf16 (what the slab kernels stream), f32 arithmetic"""
    ol = O.TransformerEncoderLayer(256, 8, 2048, dropout=0.0, normalize_before=True)
    sd = {k: v.detach().cpu().clone() for k, v in layer.state_dict().items()}
    for k in sd:
        if k.endswith('weight') and sd[k].dim() == 2:
            sd[k] = sd[k].bfloat16().float()
    ol.load_state_dict(sd)
    return ol.train()


# (B, S, pad): 64 x 128 = C2 (256 slabs); 64 x 124 = the DCASE map at C2's batch (the last slab of a clip holds 28 tokens); a padded
# clip; 48 x 128 = 192 slabs and 80 x 128 = 320 slabs: the two edges of the fill window (still the slab path)
@pytest.mark.parametrize('B,S,pad', [(64, 128, 0), (64, 124, 0), (64, 128, 37), (48, 128, 0), (80, 128, 5)])
def test_slab_encoder_layer_against_the_oracle_layer(pkg, B, S, pad, capsys):
    """layer.forward_tokens in the slab mode (enc_qkv + enc_attn_ffn forward, enc_ffn_bwd + attention backward + enc_qkv_bwd) against the
    ORACLE's TransformerEncoderLayer (reference sedt/transformer.py:192-204) - not against this library's per-op chain"""
    lib, ops, runtime, sedt = pkg
    from sound_event_detection_transformer_amd import packing
    from sound_event_detection_transformer_amd.lib import BF16
    from sound_event_detection_transformer_amd.sedt.transformer import TransformerEncoderLayer
    torch.manual_seed(11)
    layer = TransformerEncoderLayer(256, 8, 2048, 0.0, 'relu', True).cuda().train()
    with torch.no_grad():
        for n_, p in layer.named_parameters():
            if 'norm' in n_:
                p.add_(0.1 * torch.randn_like(p))
            elif p.dim() == 1:
                p.normal_(0, 0.05)
    a = layer.self_attn
    lin = [a.in_proj_weight, a.out_proj.weight, layer.linear1.weight, layer.linear2.weight]
    plan = packing.PackPlan(BF16, torch.device('cuda'), [], lin, (), lin)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B * S, 256, generator=g).bfloat16()
    pos = (0.5 * torch.randn(B * S, 256, generator=g)).bfloat16()
    gy = torch.randn(B * S, 256, generator=g).bfloat16()
    kpm = torch.zeros(B, S, dtype=torch.bool)
    if pad:
        kpm[1, S - pad:] = True
    # ---- oracle (seq-first (S, B, d) like the reference), f32 on the bf16-rounded inputs
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    ol = _oracle_layer(layer)
    xo = x0.float().view(B, S, 256).transpose(0, 1).clone().requires_grad_(True)
    yo = ol(xo, src_key_padding_mask=kpm, pos=pos.float().view(B, S, 256).transpose(0, 1))
    yo.backward(gy.float().view(B, S, 256).transpose(0, 1))
    ref_y = yo.detach().transpose(0, 1).reshape(B * S, 256)
    ref_gx = xo.grad.transpose(0, 1).reshape(B * S, 256)
    # ---- HIP slab path, real dispatch rule (no patched window)
    runtime.set_compute_dtype('bf16')
    try:
        x = x0.cuda().requires_grad_(True)
        with plan, lib.launch_log() as log:
            assert ops.encoder_slab_ok(BF16, 256, 8, S, 2048, None, B)
            y = layer.forward_tokens(x, pos.cuda(), kpm.cuda().view(torch.uint8), B, S)
            y.backward(gy.cuda())
        torch.cuda.synchronize()
    finally:
        runtime.set_compute_dtype('f32')
    _assert_dispatch(log, {'encoder_qkv_fwd': 1, 'encoder_attn_ffn_fwd': 1, 'encoder_ffn_bwd': 1, 'encoder_qkv_bwd': 1, 'attention_bwd': 1,
                           'layernorm_fwd': 0, 'layernorm_bwd': 0}, 'slab layer')
    live = ~kpm.view(B * S)                      # rows of padded QUERY tokens are never read downstream (key padding): compare the live ones
    errs = {'y': _rc(y[live.cuda()], ref_y[live]), 'gx': _rc(x.grad[live.cuda()], ref_gx[live])}
    po = dict(ol.named_parameters())
    for n_, p in layer.named_parameters():
        errs[n_] = _rc(p.grad, po[n_].grad)
    with capsys.disabled():
        wk = max((k for k in errs if k not in ('y', 'gx')), key=lambda k: errs[k][0])
        print(f'\n[slab encoder layer vs the oracle layer, B={B} S={S} pad={pad}] y {errs["y"][0]:.2e}, gx {errs["gx"][0]:.2e} (cos '
              f'{errs["gx"][1]:.6f}), worst weight gradient {wk} {errs[wk][0]:.2e} (cos {errs[wk][1]:.6f})')
    # Measured (B = 64, S = 128; the per-op chain reads the same to three digits - tools/dev/slab_vs_oracle_diag.py): y 4.4e-3; gx 2.1e-2,
    # cosine 0.99996; attention / LayerNorm1 gradients 1.0-1.2e-2, cosine 0.99995; linear2.weight 2.5e-3.  The gradients that pass
    # through the FFN's ReLU kink - linear1.*, norm2.* - read 4-6e-2 with cosine 0.9992: the hidden pre-activation carries ~1e-3 of its
    # spread as bf16 error (x1n is stored in bf16), so ~1.5e-3 of the 16.8 M hidden units sit on the other side of zero than the oracle's,
    # and a flipped unit contributes its WHOLE gradient term: relative L2 error sqrt(1.5e-3) = 4 % whichever kernel computes it.  That is
    # the bf16 data flow, not an arithmetic error of a kernel (the f32 mode holds 5e-3 / cosine 1 - 5e-6 on the same path).
    assert errs['y'][0] < 1e-2, errs['y']
    assert errs['gx'][0] < 4e-2 and errs['gx'][1] > 0.9999, errs['gx']
    for n_, (r_, c_) in errs.items():
        if n_ in ('y', 'gx'):
            continue
        kink = n_.startswith('linear1') or n_.startswith('norm2')
        assert r_ < (1.5e-1 if kink else 3e-2) and c_ > (0.998 if kink else 0.9999), (n_, r_, c_)


@pytest.mark.parametrize('B,S,slab', [(47, 128, False), (48, 128, True), (80, 128, True), (81, 128, False), (64, 124, True), (32, 124, False),
                                      (200, 124, False)])
def test_encoder_dispatch_window_edges(pkg, B, S, slab):
    """the fill rule itself (ops.SLAB_MIN_WGS .. SLAB_MAX_WGS slabs): exactly 192 and 320 slabs are inside, 188 and 324 outside; C3 / C5
    (B = 32) and C4 (B = 200) take the per-op chain"""
    lib, ops, runtime, sedt = pkg
    from sound_event_detection_transformer_amd.lib import BF16, F32
    assert ops.encoder_slab_ok(BF16, 256, 8, S, 2048, None, B) == slab
    assert not ops.encoder_slab_ok(F32, 256, 8, S, 2048, None, B)
