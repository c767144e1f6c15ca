"""GPU: the kernels and autograd nodes added in round 6, each against a plain torch statement of the same operation (the reference lines
they replace are cited at the entry points in include/sedt_hip.h):
  sedt_spsedt_dec_in / _bwd   reference sedt/spsedt.py:48-69      (decoder input of SP-SEDT)
  sedt_copy2d + SplitClipsFn  reference engine.py:134-165          (two criterion calls on clip ranges of one forward)
  sedt_avgpool (16-byte form) reference sedt/spsedt.py:44-46       (adaptive average pool of the patch features)
  FanoutFn                    reference sedt/spsedt.py:79-83       (three heads on one decoder output)
  sedt_feature_loss base      reference sedt/sedt.py:263-283, engine.py:64-66 (sum of the weighted losses)
  ops.defer_layer_wgrads      the layers' weight gradients of a stack as one launch pair: same gradients, fewer launches"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pkg():
    from sound_event_detection_transformer_amd import functional as Fn, lib, ops, runtime
    assert torch.cuda.is_available()
    return Fn, lib, ops, runtime


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('train', [True, False])
def test_spsedt_decoder_input_forward_and_backward(pkg, dtype, train):
    Fn, lib, ops, runtime = pkg
    dt, td = (lib.F32, torch.float32) if dtype == 'f32' else (lib.BF16, torch.bfloat16)
    B, P, qpp, D = 7, 10, 2, 256
    Q = P * qpp
    g = torch.Generator().manual_seed(3)
    patch = torch.randn(B * P, D, generator=g).to(td).cuda().requires_grad_(True)
    query = torch.randn(Q, D, generator=g).cuda().requires_grad_(True)
    keep = (torch.rand(Q, B, 1, generator=g) > 0.3).float().cuda() if train else None
    out = Fn.SpDecInFn.apply(patch, query, keep, B, Q, P, qpp, train, 0.3, dt)
    assert out.shape == (B * Q, D) and out.dtype == td
    gy = torch.randn(B * Q, D, generator=g).to(td).cuda()
    out.backward(gy)
    # torch statement (spsedt.py:48-69): patches repeated qpp times per clip, (Q, B, d) layout, 2 * query + patch * mask
    pf = patch.detach().float().requires_grad_(True)
    qf = query.detach().clone().requires_grad_(True)
    pq = pf.view(B, P, 1, D).repeat(1, 1, qpp, 1).flatten(1, 2).permute(1, 0, 2)                 # (Q, B, D)
    ref = (qf.unsqueeze(1) * 2 + pq * keep) if train else (pq + qf.unsqueeze(1))
    ref_tok = ref.permute(1, 0, 2).reshape(B * Q, D)
    ref_tok.backward(gy.float())
    tol = 1e-6 if dtype == 'f32' else 8e-3
    assert ((out.float() - ref_tok).abs().max() / ref_tok.abs().max()).item() < tol
    assert ((patch.grad.float() - pf.grad).abs().max() / pf.grad.abs().max()).item() < tol
    assert ((query.grad - qf.grad).abs().max() / qf.grad.abs().max()).item() < (1e-5 if dtype == 'f32' else 2e-3)


def test_spsedt_decoder_input_draws_its_own_mask(pkg):
    """no injected mask: Bernoulli(1 - ratio) from the counter hash, a different draw per seed bump; ratio <= 0 keeps every patch"""
    Fn, lib, ops, runtime = pkg
    B, P, qpp, D = 200, 10, 2, 256
    Q = P * qpp
    patch = torch.ones(B * P, D, device='cuda')
    query = torch.zeros(Q, D, device='cuda')
    sp = runtime.seed_ptr(patch.device)
    o1, k1 = ops.spsedt_dec_in(lib.F32, patch, query, B, Q, P, qpp, True, 0.1, None, 1234, sp)
    assert torch.equal(o1[:, 0].view(B, Q).t().contiguous(), k1)                  # out = 2 * 0 + keep * 1
    rate = k1.mean().item()
    assert abs(rate - 0.9) < 0.02, rate
    runtime.bump_seed(patch.device)
    _, k2 = ops.spsedt_dec_in(lib.F32, patch, query, B, Q, P, qpp, True, 0.1, None, 1234, sp)
    assert (k1 != k2).float().mean().item() > 0.05
    _, k3 = ops.spsedt_dec_in(lib.F32, patch, query, B, Q, P, qpp, True, -1.0, None, 1234, sp)
    assert bool((k3 == 1).all())


def test_split_clips_is_one_launch_each_way_and_exact(pkg):
    Fn, lib, ops, runtime = pkg
    g = torch.Generator().manual_seed(5)
    L, B, Qs, C1, n = 3, 64, 21, 11, 32
    la = torch.randn(L, B, Qs, C1, generator=g).cuda().requires_grad_(True)
    ba = torch.rand(L, B, Qs, 2, generator=g).cuda().requires_grad_(True)
    at = torch.rand(B, 10, generator=g).cuda().requires_grad_(True)
    with lib.launch_log() as log:
        a0, b0, t0, a1, b1, t1 = Fn.SplitClipsFn.apply(n, (1, 1, 0), la, ba, at)
    assert log['copy2d'] == 1
    assert torch.equal(a0, la[:, :n]) and torch.equal(a1, la[:, n:]) and torch.equal(b0, ba[:, :n]) and torch.equal(b1, ba[:, n:])
    assert torch.equal(t0, at[:n]) and torch.equal(t1, at[n:]) and a0.is_contiguous() and a1.is_contiguous()
    w = [torch.randn_like(t) for t in (a0, b0, t0, a1, b1, t1)]
    with lib.launch_log() as log:
        sum((t * w_).sum() for t, w_ in zip((a0, b0, t0, a1, b1, t1), w)).backward()
    assert log['copy2d'] == 1
    assert torch.equal(la.grad, torch.cat([w[0], w[3]], 1)) and torch.equal(ba.grad, torch.cat([w[1], w[4]], 1))
    assert torch.equal(at.grad, torch.cat([w[2], w[5]], 0))
    # one part without a gradient: the other part's range is written, the rest is zero
    la.grad = None
    a0, b0, t0, a1, b1, t1 = Fn.SplitClipsFn.apply(n, (1, 1, 0), la, ba, at)
    (a1 * w[3]).sum().backward()
    assert torch.equal(la.grad[:, n:], w[3]) and bool((la.grad[:, :n] == 0).all())


@pytest.mark.parametrize('dtype,C', [('bf16', 2048), ('f32', 2048), ('bf16', 2044)])
def test_avgpool_matches_torch_mean(pkg, dtype, C):
    """C a multiple of 8: the 16-byte kernel; otherwise the scalar one - same summation order, so both equal a sequential f32 sum"""
    Fn, lib, ops, runtime = pkg
    dt, td = (lib.F32, torch.float32) if dtype == 'f32' else (lib.BF16, torch.bfloat16)
    B, P = 50, 32
    x = torch.randn(B * P, C, generator=torch.Generator().manual_seed(2)).to(td).cuda()
    y = ops.avgpool(dt, x, B, P, C)
    ref = x.float().view(B, P, C).cpu().double().sum(1) / P
    assert y.shape == (B, C) and y.dtype == torch.float32
    assert ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-6


def test_fanout_sums_the_consumers_gradients_in_one_launch(pkg):
    Fn, lib, ops, runtime = pkg
    x = torch.randn(3, 20, 20, 256, generator=torch.Generator().manual_seed(1)).bfloat16().cuda().requires_grad_(True)
    a, b, c = Fn.FanoutFn.apply(x, 3, lib.BF16)
    ga, gb, gc = (torch.randn_like(x) for _ in range(3))
    with lib.launch_log() as log:
        ((a * ga).sum() + (b * gb).sum() + (c * gc).sum()).backward()
    assert log['add_n'] == 1
    ref = (ga.float() + gb.float() + gc.float())
    assert ((x.grad.float() - ref).abs().max() / ref.abs().max()).item() < 8e-3
    x.grad = None
    a, b, c = Fn.FanoutFn.apply(x, 3, lib.BF16)
    (b * gb).sum().backward()                                     # consumers without a gradient are skipped
    assert torch.equal(x.grad, gb)


def test_feature_loss_folds_the_running_total(pkg):
    Fn, lib, ops, runtime = pkg
    g = torch.Generator().manual_seed(4)
    L, B, Q, F, P = 3, 4, 20, 256, 10
    pred = torch.randn(L, B, Q, F, generator=g).cuda()
    gt = torch.randn(B * P, F, generator=g).cuda()
    dense = {'ns': B, 'L': L, 'wbox': (torch.rand(L, B, Q, generator=g) > 0.4).float().cuda(),
             'tidx': torch.randint(0, P, (L, B, Q), generator=g).float().cuda()}
    nb = dense['wbox'][0].sum().reshape(1)
    w = torch.tensor([1.0, 0.5, 0.25], device='cuda')
    base = torch.tensor(3.25, device='cuda')
    out, dpred = ops.feature_loss(pred, gt, dense, [2, 0, 1], nb, w)
    out2, dpred2, total = ops.feature_loss(pred, gt, dense, [2, 0, 1], nb, w, None, base)
    assert torch.equal(out, out2) and torch.equal(dpred, dpred2)
    assert abs(total.item() - (base.item() + out[L].item())) < 1e-6 * max(1.0, abs(total.item()))


def test_stack_level_weight_gradient_launches_change_nothing_but_the_launch_count(pkg):
    """ops.defer_layer_wgrads (inside the captured steppers): the per-op encoder and the decoder hand their layers' weight-gradient
    problems to one shared batch, launched by layer 0's backward - bit-identical gradients, fewer wgrad / reduce launches"""
    Fn, lib, ops, runtime = pkg
    from oracle import sedt_oracle as O
    from sound_event_detection_transformer_amd import sedt
    runtime.set_compute_dtype('bf16')
    try:
        g = torch.Generator().manual_seed(9)
        B = 4                                                       # 16 slabs: the per-op encoder path
        src = torch.randn(B, 256, 31, 4, generator=g).cuda()
        pos = (torch.randn(B, 256, 31, 4, generator=g) * 0.5).cuda()
        query = torch.randn(21, 256, generator=g).cuda()
        mask = torch.zeros(B, 31, 4, dtype=torch.bool).cuda()
        res = {}
        for defer in (False, True):
            m = sedt.Transformer(256, 8, 6, 3, 2048, 0.0, 'relu', True, True, False)
            m.load_state_dict(O.seeded_state_dict(m.state_dict(), 5))
            m.cuda().train()
            with ops.defer_layer_wgrads(defer), lib.launch_log() as log:
                hs, mem = m(src, mask, query, pos)
                (hs.float().square().mean() + mem.float().square().mean()).backward()
            torch.cuda.synchronize()
            res[defer] = (log['wgrad_group'], log['multi_wgrad_reduce'],
                          torch.cat([p.grad.flatten() for p in m.parameters()]))
        assert res[True][0] < res[False][0] and res[True][1] < res[False][1], (res[False][:2], res[True][:2])
        assert res[False][0] >= 9 and res[True][0] <= 7, (res[False][:2], res[True][:2])
        assert torch.equal(res[False][2], res[True][2])
        assert bool(torch.isfinite(res[True][2]).all()) and res[True][2].abs().max().item() > 0
    finally:
        runtime.set_compute_dtype('f32')
