"""GPU: deeper parity of the HIP path against the reference fixtures.

* every per-stage digest the fixture G2 holds (conv1+bn1, layer1-4, input_proj, each encoder / decoder layer) is checked on
  the HIP side too - a wrong-but-compensating stage cannot hide behind correct final outputs;
* the sine position encoding against fixture G6 itself (not only against the oracle);
* the BASELINE batch size (B = 64) forward in f32 against the CPU oracle on a strided sample of clips (clips are independent
  end to end, so clip i of the B = 64 batch must equal the oracle's output for clip i alone);
* the bf16 throughput mode: the MEASURED error of every output and every gradient norm against the f32 reference is
  printed, and bounded at about 3x what this build shows (outputs: max |delta| / max |ref| per tensor).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sedt_oracle as O                                   # noqa: E402
from oracle.criterion_oracle import build_oracle_criterion, synthetic_targets   # noqa: E402


def rel(got, ref):
    got = torch.as_tensor(got).float().cpu()
    ref = torch.as_tensor(ref).float().cpu()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def digest(t, n=64):
    t = t.detach().float().flatten().cpu()
    idx = torch.linspace(0, t.numel() - 1, n).long()
    return torch.cat([t.mean()[None], t.abs().mean()[None], t[idx]])


@pytest.fixture(scope='module')
def pkg():
    from sound_event_detection_transformer_amd import runtime, sedt, ops
    assert torch.cuda.is_available()
    return runtime, sedt, ops


def _seed_load(model, seed):
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), seed))
    return model


@pytest.mark.parametrize('name,E,Q,T', [('urban', 3, 10, 500), ('dcase', 6, 20, 496)])
def test_g2_stage_digests_on_the_hip_path(pkg, golden_dir, name, E, Q, T):
    runtime, sedt, ops = pkg
    from sound_event_detection_transformer_amd import lib as L
    runtime.set_compute_dtype('f32')
    g = np.load(os.path.join(golden_dir, 'g2_g3_sedt.npz'))
    model, _, _ = sedt.build_model(sedt.default_args(enc_layers=E, num_queries=Q, dropout=0.0))
    _seed_load(model, 2020).cuda().eval()
    B = 2
    x = torch.randn(B, 1, T, 64, generator=torch.Generator().manual_seed(7)).cuda()
    body = model.backbone[0].body
    got = {}
    # --- stem output before the max-pool (the reference hooks bn1, whose output the in-place ReLU that follows overwrites:
    #     the fixture holds relu(bn1(conv1(conv0(x))))); the HIP stem never materialises it inside StemFn, so the same
    #     kernels are called here
    with torch.no_grad():
        wcat = ops.stem_prep(L.F32, body.conv0.weight, body.conv0.bias, body.conv1.weight)
        col, Ho, Wo = ops.stem_im2col(L.F32, x.contiguous(), B, T, 64)
        sc, bi = ops.bn_fold(*body.bn1.tensors())
        s1 = ops.linear(L.F32, col, wcat, scale=sc, bias=bi, act=L.ACT_RELU)
    got['bn1'] = s1.view(B, Ho, Wo, 64).permute(0, 3, 1, 2)
    # --- stage outputs, input_proj, every transformer layer
    body.keep_stage_out = True
    model.input_proj.register_forward_hook(lambda m, i, o: got.__setitem__('input_proj', o))
    S = {}

    def wrap(layer, key, rows_of):
        orig = layer.forward_tokens

        def f(*a, **k):
            y = orig(*a, **k)
            got[key] = (y, rows_of)
            return y
        layer.forward_tokens = f
    for li, l in enumerate(model.transformer.encoder.layers):
        wrap(l, f'enc{li}', 'S')
    for li, l in enumerate(model.transformer.decoder.layers):
        wrap(l, f'dec{li}', 'Q')
    with torch.no_grad():
        model(x)
    body.keep_stage_out = False
    H, W = T, 64
    H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    for li, stride in enumerate((1, 2, 2, 1)):
        H, W = (H - 1) // stride + 1, (W - 1) // stride + 1
        tok = body.stage_out[li]
        got[f'layer{li + 1}'] = tok.view(B, H, W, tok.shape[1]).permute(0, 3, 1, 2)
    checked = 0
    for key in [k for k in g.files if k.startswith(f'{name}_stage_')]:
        stage = key[len(name) + 7:]
        v = got[stage]
        if isinstance(v, tuple):                       # token matrix [B*rows, d] -> the reference's (rows, B, d)
            y, _ = v
            v = y.view(B, y.shape[0] // B, y.shape[1]).transpose(0, 1)
        d = digest(v.contiguous())
        r = torch.from_numpy(g[key])
        assert rel(d[2:], r[2:]) < 1e-3, (stage, rel(d[2:], r[2:]))
        assert abs(d[1] - r[1]) < 1e-3 * abs(r[1]), (stage, 'abs-mean')
        checked += 1
    assert checked == 6 + E + 3


def test_g6_posenc_against_the_fixture(pkg, golden_dir):
    runtime, sedt, ops = pkg
    from sound_event_detection_transformer_amd import lib as L
    g = np.load(os.path.join(golden_dir, 'g6_posenc.npz'))
    for h in (32, 31, 8):
        mask = torch.zeros(1, h, 4, dtype=torch.uint8, device='cuda')
        pos = ops.posenc(L.F32, mask, 256).view(1, h, 4, 256)
        assert rel(pos[0, :, 0, :], g[f'pos_{h}']) < 1e-5
        assert torch.equal(pos[0, :, 0, :], pos[0, :, 3, :])
    mask = torch.zeros(1, 32, 4, dtype=torch.uint8, device='cuda')
    mask[0, 23:, :] = 1
    pos = ops.posenc(L.F32, mask, 256).view(1, 32, 4, 256)
    assert rel(pos[0, :, 0, :], g['pos_32_pad23']) < 1e-5
    # and through the module API (PositionEmbeddingSine of the product), bf16 output within bf16 rounding
    pe = sedt.build_model(sedt.default_args())[0].backbone[1]
    from sound_event_detection_transformer_amd.utilities.utils import NestedTensor
    runtime.set_compute_dtype('f32')
    p = pe(NestedTensor(torch.zeros(1, 2048, 32, 4, device='cuda'), torch.zeros(1, 32, 4, dtype=torch.bool, device='cuda')))
    assert rel(p[0, :, :, 0].t(), g['pos_32']) < 1e-5


def test_b64_forward_f32_matches_oracle_on_sampled_clips(pkg):
    """BASELINE batch size through the HIP path (f32 parity mode) vs the CPU oracle on clips 0, 21, 42, 63"""
    runtime, sedt, ops = pkg
    runtime.set_compute_dtype('f32')
    B = 64
    pick = [0, 21, 42, 63]
    x = torch.randn(B, 1, 500, 64, generator=torch.Generator().manual_seed(15))
    oracle = _seed_load(O.build_oracle_model(10, 10, 3, 3, True, True, True, dropout=0.0), 2020).eval()
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = oracle(x[pick])
    model, _, _ = sedt.build_model(sedt.default_args(dropout=0.0))
    _seed_load(model, 2020).cuda().eval()
    with torch.no_grad():
        o = model(x.cuda())
    for k in ('pred_logits', 'pred_boxes', 'at'):
        assert rel(o[k][pick], ref[k]) < 1e-3, (k, rel(o[k][pick], ref[k]))
    for i, a in enumerate(o['aux_outputs']):
        assert rel(a['pred_logits'][pick], ref['aux_outputs'][i]['pred_logits']) < 1e-3
        assert rel(a['pred_boxes'][pick], ref['aux_outputs'][i]['pred_boxes']) < 1e-3


# measured on this build (printed by the test): pred_logits 1.5e-2, pred_boxes 1.0e-2, at 4.8e-3, total loss 6e-5 .. 6e-4; gradient
# norms under the smooth surrogate loss: median 2e-3 .. 4e-3, worst 1.5e-2 .. 7e-2 (conv0, the end of the longest backward chain).
# Bounds = about 3x the larger observations.
BF16_BOUNDS = {'pred_logits': 4.5e-2, 'pred_boxes': 3e-2, 'at': 2.2e-2, 'loss': 3e-3, 'gradnorm': 0.2, 'gradnorm_median': 1.2e-2}


def _smooth_loss(o):
    """a kink-free scalar of every model output (all decoder layers): what the gradient comparison below differentiates"""
    t = o['pred_logits'].float().square().mean() + 3.0 * o['pred_boxes'].float().square().mean() + o['at'].float().square().mean()
    for i, a in enumerate(o['aux_outputs']):
        t = t + (0.5 + 0.25 * i) * (a['pred_logits'].float().square().mean() + 3.0 * a['pred_boxes'].float().square().mean())
    return t


def test_bf16_mode_error_vs_fixtures_and_gradient_norms_vs_this_librarys_f32_mode(pkg, golden_dir, capsys):
    """bf16 throughput mode against the f32 reference, measured and printed:
    * eval outputs against fixture G2 (the reference's own numbers);
    * the training loss against fixture G3 with the Hungarian ASSIGNMENT taken from the f32-mode forward of the same model (pinned
      to the reference's by G2/G3): at random init several queries predict nearly the same event, so matching costs are
      near-tied and a last-bit change upstream may swap two queries - a legitimate answer to a tie, not an arithmetic error;
    * per-parameter gradient norms against the f32 MODE of this library (itself within 2e-3 of the reference's, G3) under a
      SMOOTH surrogate loss.  SetCriterion's own gradient has kinks exactly where a random-init model sits (L1 sign at
      pred = target, the clamps of the 1-D GIoU with boxes hanging over the clip start): a 1e-2 forward difference flips a few
      of its +-1 entries and moves every gradient norm by 5-25 % although nothing is wrong - the first form of this test
      (criterion gradients against G3) passed or failed with the f32 summation ORDER of an unrelated kernel.  The criterion's
      backward kernel is pinned separately in f32 (G3, G5, G9) where such flips cannot happen."""
    runtime, sedt, ops = pkg
    g = np.load(os.path.join(golden_dir, 'g2_g3_sedt.npz'))
    B = 2
    x = torch.randn(B, 1, 500, 64, generator=torch.Generator().manual_seed(7))
    model, crit, _ = sedt.build_model(sedt.default_args(dropout=0.0))
    _seed_load(model, 2020).cuda()
    runtime.set_compute_dtype('bf16')
    model.eval()
    with torch.no_grad():
        o = model(x.cuda())
    errs = {k: rel(o[k], g[f'urban_eval_{k}']) for k in ('pred_logits', 'pred_boxes', 'at')}
    model.train()
    tg = [{k: v.cuda() for k, v in t.items()} for t in synthetic_targets(B, 99, 10)]
    runtime.set_compute_dtype('f32')
    with torch.no_grad():
        dense, _ = crit.prepare(model(x.cuda()), tg, None, slice(B))
    model.zero_grad(set_to_none=True)
    _smooth_loss(model(x.cuda())).backward()
    ref_gn = {n: p.grad.norm().item() for n, p in model.named_parameters() if p.grad is not None}
    runtime.set_compute_dtype('bf16')
    with torch.no_grad():
        crit.compute(model(x.cuda()), dense)
    total = crit.last_total
    errs['loss'] = abs(total.item() - float(g['urban_train_total'])) / abs(float(g['urban_train_total']))
    model.zero_grad(set_to_none=True)
    _smooth_loss(model(x.cuda())).backward()
    runtime.set_compute_dtype('f32')
    names = [n for n in ref_gn if ref_gn[n] > 0]
    params = dict(model.named_parameters())
    gn = np.array([params[n].grad.norm().item() for n in names])
    rg = np.array([ref_gn[n] for n in names])
    relg = np.abs(gn - rg) / (rg + 1e-12)
    errs['gradnorm'] = float(relg.max())
    errs['gradnorm_median'] = float(np.median(relg))
    worst = names[int(relg.argmax())]
    with capsys.disabled():
        print('\n[bf16 vs f32: outputs / loss against fixtures G2 / G3, gradient norms against the f32 mode under a smooth loss] '
              + ', '.join(f'{k}={v:.3e}' for k, v in errs.items()) + f' (worst grad: {worst})')
    for k, bound in BF16_BOUNDS.items():
        assert errs[k] < bound, (k, errs[k], bound)
