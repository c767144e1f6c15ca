"""GPU: every HIP op through the C ABI against plain PyTorch fp32 on the CPU.

Tolerances: f32 mode rtol 2e-4 (exact-f32 MFMA, different summation order); bf16 mode compares
against the f32 reference evaluated on bf16-rounded inputs with 2e-2 of the output scale."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1
TD = {F32: torch.float32, BF16: torch.bfloat16}


@pytest.fixture(scope='module')
def ops():
    from sound_event_detection_transformer_amd import ops as o
    assert torch.cuda.is_available()
    return o


def dev(t, dt):
    return t.to(TD[dt]).cuda().contiguous()


def rnd(t, dt):
    """what the kernel sees after the dtype cast"""
    return t.to(TD[dt]).float()


def close(got, ref, dt, f32_tol=2e-4, bf16_tol=2e-2):
    got = got.float().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    tol = f32_tol if dt == F32 else bf16_tol
    assert err < tol, f'rel err {err:.3e} (tol {tol}) scale {scale:.3e}'


G = torch.Generator().manual_seed(1234)


def randn(*s):
    return torch.randn(*s, generator=G)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('M,N,K', [(256, 256, 256), (704, 11, 256), (300, 200, 96), (8192, 768, 256), (130, 70, 64),
                                   (64, 2, 256), (257, 129, 40)])
def test_linear_epilogues(ops, dt, M, N, K):
    x, w, b = randn(M, K), randn(N, K) / math.sqrt(K), randn(N)
    res, msk, sc = randn(M, N), randn(M, N), torch.rand(N, generator=G) + 0.5
    xr, wr, rr, mr = rnd(x, dt), rnd(w, dt), rnd(res, dt), rnd(msk, dt)
    # plain + bias
    y = ops.linear(dt, dev(x, dt), dev(w, dt), bias=b.cuda())
    close(y, xr @ wr.t() + b, dt)
    # scale, bias, residual then relu (bottleneck conv3 form), f32 output
    y = ops.linear(dt, dev(x, dt), dev(w, dt), bias=b.cuda(), scale=sc.cuda(), res=dev(res, dt), ldr=N,
                   act=ops.ACT_RELU, act_post_res=1, out_f32=True)
    assert y.dtype == torch.float32
    close(y, F.relu((xr @ wr.t()) * sc + b + rr), dt)
    # relu before residual, mask, alpha (dgrad form)
    y = ops.linear(dt, dev(x, dt), dev(w, dt), res=dev(res, dt), ldr=N, mask=dev(msk, dt), ldm=N, alpha=0.5)
    close(y, (xr @ wr.t() + rr) * (mr > 0) * 0.5, dt)
    # sigmoid, broadcast residual rows (res_mod)
    rm = 7
    y = ops.linear(dt, dev(x, dt), dev(w, dt), bias=b.cuda(), act=ops.ACT_SIGMOID, res=dev(res[:rm], dt), ldr=N, res_mod=rm)
    close(y, torch.sigmoid(xr @ wr.t() + b) + rr[torch.arange(M) % rm], dt)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('tile', [(128, 128), (128, 64), (64, 64)])
def test_linear_tiles_and_strided_views(ops, dt, tile):
    M, N, K = 384, 256, 256
    big = randn(M, 3 * K)
    w = randn(N, K) / 16
    xb = dev(big, dt)
    xv = xb[:, K:2 * K]                      # column slice: lda = 3K
    out = torch.zeros((M, 2 * N), device='cuda', dtype=TD[dt])
    ops.igemm(dt, M, N, K, xv, xv.stride(0), dev(w, dt), K, out[:, N:], out.stride(0), tile=tile)
    close(out[:, N:], rnd(big[:, K:2 * K], dt) @ rnd(w, dt).t(), dt)
    assert out[:, :N].abs().max().item() == 0


CONVS = [  # Hi, Wi, Ci, Co, k, stride, pad, dil
    (20, 8, 64, 64, 3, 1, 1, 1), (21, 8, 64, 128, 3, 2, 1, 1), (8, 4, 128, 64, 3, 1, 2, 2), (13, 6, 64, 128, 1, 2, 0, 1),
    (10, 4, 128, 256, 1, 1, 0, 1)]


def _nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def _nchw(tok, B, H, W):
    return tok.view(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('cfg', CONVS)
def test_conv_fwd_dgrad_wgrad(ops, dt, cfg):
    Hi, Wi, Ci, Co, k, s, pd, dl = cfg
    B = 3
    g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
    x = randn(B, Ci, Hi, Wi)
    w = randn(Co, Ci, k, k) / math.sqrt(Ci * k * k)
    sc, bi = torch.rand(Co, generator=G) + 0.5, randn(Co)
    xr, wr = rnd(x, dt).requires_grad_(True), rnd(w, dt).requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=pd, dilation=dl)
    assert y_ref.shape[2:] == (g.Ho, g.Wo)
    wf, wb = ops.pack_conv(dt, w.cuda(), bnscale=sc.cuda())
    xd = dev(_nhwc(x).reshape(-1, Ci), dt)
    # forward with folded FrozenBN + relu
    y = ops.conv_fwd(dt, xd, B, g, wf, scale=sc.cuda(), bias=bi.cuda(), act=ops.ACT_RELU)
    ref = F.relu(y_ref * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1))
    close(y, _nhwc(ref).reshape(-1, Co).detach(), dt)
    # backward through scale: upstream grad gy (already relu-masked by the caller)
    gy = randn(B, Co, g.Ho, g.Wo)
    gyr = rnd(gy, dt)
    (y_ref * sc.view(1, -1, 1, 1)).backward(gyr)
    gyd = dev(_nhwc(gy).reshape(-1, Co), dt)
    dx = ops.conv_dgrad(dt, gyd, B, g, wb)
    close(dx, _nhwc(xr.grad).reshape(-1, Ci), dt)
    dw = ops.wgrad(dt, gyd, xd, B, g, rowscale=sc.cuda())
    assert dw.shape == w.shape and dw.dtype == torch.float32
    close(dw, wr.grad, dt)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('M,N,K', [(704, 11, 256), (8192, 256, 2048), (1920, 2, 256), (300, 768, 256), (77, 10, 256)])
def test_linear_wgrad_and_colsum(ops, dt, M, N, K):
    x, gy = randn(M, K), randn(M, N)
    dw = ops.linear_wgrad(dt, dev(gy, dt), dev(x, dt))
    close(dw, rnd(gy, dt).t() @ rnd(x, dt), dt)
    cs = ops.colsum(dt, dev(gy, dt))
    close(cs, rnd(gy, dt).sum(0), dt, bf16_tol=1e-3)
    cs = ops.colsum(dt, gy.cuda())           # f32 input in either mode
    close(cs, gy.sum(0), dt, bf16_tol=1e-4)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_epilogue_dropout_matches_dropout_grad(ops, dt):
    M, N, K, pdrop, seed = 512, 256, 64, 0.1, 777
    x, w = randn(M, K), randn(N, K)
    seedbuf = torch.tensor([5], dtype=torch.int32).cuda()
    y = ops.linear(dt, dev(x, dt), dev(w, dt), drop_p=pdrop, seed=seed, seed_ptr=seedbuf)
    keep = ops.dropout_grad(dt, torch.ones(M, N, dtype=TD[dt]).cuda(), pdrop, seed, seedbuf).float().cpu()
    rate = (keep > 0).float().mean().item()
    assert abs(rate - 0.9) < 0.01
    np.testing.assert_allclose(keep[keep > 0].numpy(), 1 / 0.9, rtol=1e-2)
    close(y, (rnd(x, dt) @ rnd(w, dt).t()) * keep, dt)
    keep2 = ops.dropout_grad(dt, torch.ones(M, N, dtype=TD[dt]).cuda(), pdrop, seed + 1, seedbuf).float().cpu()
    assert ((keep > 0) != (keep2 > 0)).float().mean().item() > 0.1      # a different seed is a different mask
    seedbuf += 1                                                         # device-side seed word changes the mask too
    keep3 = ops.dropout_grad(dt, torch.ones(M, N, dtype=TD[dt]).cuda(), pdrop, seed, seedbuf).float().cpu()
    assert torch.equal(keep2 > 0, keep3 > 0)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('rows', [8192, 704, 5])
def test_layernorm(ops, dt, rows):
    D = 256
    x, pos = randn(rows, D) * 2 + 0.3, randn(rows, D)
    gam, bet = torch.rand(D, generator=G) + 0.5, randn(D) * 0.1
    xr = rnd(x, dt).requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (D,), gr, br, 1e-5)
    y, y2, mean, rstd = ops.layernorm_fwd(dt, dev(x, dt), gam.cuda(), bet.cuda(), dev(pos, dt))
    close(y, y_ref.detach(), dt)
    close(y2, (y_ref + rnd(pos, dt)).detach(), dt)
    dy, dy2, dres = randn(rows, D), randn(rows, D), randn(rows, D)
    y_ref.backward(rnd(dy, dt) + rnd(dy2, dt))
    dx, dg, db = ops.layernorm_bwd(dt, dev(dy, dt), dev(x, dt), gam.cuda(), mean, rstd, dy2=dev(dy2, dt), dres=dev(dres, dt))
    close(dx, xr.grad + rnd(dres, dt), dt)
    close(dg, gr.grad, dt, bf16_tol=5e-3)
    close(db, br.grad, dt, bf16_tol=5e-3)


def _attn_ref(q, k, v, B, H, Lq, Lk, kpm, amask, keep=None):
    """q [B*Lq, H*32] etc -> o [B*Lq, H*32]; explicit math of torch MHA's core"""
    qh = q.view(B, Lq, H, 32).transpose(1, 2) / math.sqrt(32)
    kh = k.view(B, Lk, H, 32).transpose(1, 2)
    vh = v.view(B, Lk, H, 32).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if amask is not None:
        s = s + amask
    if kpm is not None:
        s = s.masked_fill(kpm.view(B, 1, 1, Lk).bool(), float('-inf'))
    p = s.softmax(-1)
    if keep is not None:
        p = p * keep
    return (p @ vh).transpose(1, 2).reshape(B * Lq, H * 32), p


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('Lq,Lk,masks', [(128, 128, False), (124, 124, True), (11, 128, True), (11, 11, False),
                                         (20, 20, True), (70, 200, False)])
def test_attention_fwd_bwd(ops, dt, Lq, Lk, masks):
    B, H = 3, 8
    q, k, v, do = randn(B * Lq, 256), randn(B * Lk, 256), randn(B * Lk, 256), randn(B * Lq, 256)
    kpm = amask = None
    if masks:
        kpm = torch.zeros(B, Lk, dtype=torch.uint8)
        kpm[1, Lk - 5:] = 1
        amask = torch.zeros(Lq, Lk)
        amask[torch.rand(Lq, Lk, generator=G) < 0.2] = float('-inf')
        amask[:, 0] = 0                                           # never a fully masked row
    qr, kr, vr = (rnd(t, dt).requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, B, H, Lq, Lk, kpm, amask)
    o_ref.backward(rnd(do, dt))
    # q/k live in one buffer like the packed QK projection output (row stride 512)
    qk = dev(torch.cat([q, q], 1), dt)
    qd = qk[:, :256]
    kd, vd = dev(k, dt), dev(v, dt)
    kpm_d = kpm.cuda() if kpm is not None else None
    am_d = amask.cuda() if amask is not None else None
    o, lse = ops.attention_fwd(dt, qd, kd, vd, B, H, Lq, Lk, kpm_d, am_d)
    close(o, o_ref.detach(), dt)
    dq, dk, dv = (torch.empty_like(dev(t, dt)) for t in (q, k, v))
    ops.attention_bwd(dt, qd, kd, vd, o, dev(do, dt), lse, B, H, Lq, Lk, dq, dk, dv, kpm_d, am_d)
    close(dq, qr.grad, dt, bf16_tol=3e-2)
    close(dk, kr.grad, dt, bf16_tol=3e-2)
    close(dv, vr.grad, dt, bf16_tol=3e-2)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_attention_dropout_consistency(ops, dt):
    """V = identity exposes the dropped probabilities; backward must use the same regenerated mask."""
    B, H, Lq, Lk, pd, seed = 2, 8, 40, 32, 0.1, 4242
    q, k, do = randn(B * Lq, 256), randn(B * Lk, 256), randn(B * Lq, 256)
    v = torch.eye(32).repeat(B, 8)                                 # [B*32, 256]: every head sees I
    o, lse = ops.attention_fwd(dt, dev(q, dt), dev(k, dt), dev(v, dt), B, H, Lq, Lk, drop_p=pd, seed=seed)
    pdrop = o.float().cpu().view(B, Lq, H, 32).transpose(1, 2)     # = P*keep/(1-p)
    _, p = _attn_ref(rnd(q, dt), rnd(k, dt), rnd(v, dt), B, H, Lq, Lk, None, None)
    keep = (pdrop > 0).float()
    assert abs(keep.mean().item() - 0.9) < 0.02
    close(pdrop, p * keep / 0.9, dt)
    qr, kr, vr = (rnd(t, dt).requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, B, H, Lq, Lk, None, None, keep / 0.9)
    o_ref.backward(rnd(do, dt))
    dq, dk, dv = (torch.empty_like(dev(t, dt)) for t in (q, k, v))
    ops.attention_bwd(dt, dev(q, dt), dev(k, dt), dev(v, dt), o, dev(do, dt), lse, B, H, Lq, Lk, dq, dk, dv, drop_p=pd, seed=seed)
    close(dq, qr.grad, dt, bf16_tol=3e-2)
    close(dk, kr.grad, dt, bf16_tol=3e-2)
    close(dv, vr.grad, dt, bf16_tol=3e-2)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_maxpool(ops, dt):
    B, H, W, Cc = 2, 25, 16, 64
    x = randn(B, Cc, H, W)
    xr = rnd(x, dt).requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    y, idx, Ho, Wo = ops.maxpool_fwd(dt, dev(_nhwc(x).reshape(-1, Cc), dt), B, H, W, Cc)
    assert (Ho, Wo) == tuple(y_ref.shape[2:])
    close(y, _nhwc(y_ref).reshape(-1, Cc).detach(), dt, bf16_tol=1e-6)
    gy = randn(B, Cc, Ho, Wo)
    y_ref.backward(rnd(gy, dt))
    relu_src = randn(B, Cc, H, W)
    dx = ops.maxpool_bwd(dt, dev(_nhwc(gy).reshape(-1, Cc), dt), idx, dev(_nhwc(relu_src).reshape(-1, Cc), dt), B, H, W, Cc)
    ref = xr.grad * (rnd(relu_src, dt) > 0)
    if dt == F32:
        close(dx, _nhwc(ref).reshape(-1, Cc), dt)
    else:   # bf16 ties may route the gradient to another tap of equal value: allow a small mismatch fraction
        frac = ((dx.float().cpu() - _nhwc(ref).reshape(-1, Cc)).abs() > 1e-2).float().mean().item()
        assert frac < 0.02
    # the ReLU mask taken from the pooled output (stem path): pool(relu(x)) backward == maxpool_bwd(..., y=pooled) bit for bit
    xp = dev(_nhwc(F.relu(x)).reshape(-1, Cc), dt)
    y2, idx2, _, _ = ops.maxpool_fwd(dt, xp, B, H, W, Cc)
    gyd = dev(_nhwc(gy).reshape(-1, Cc), dt)
    a = ops.maxpool_bwd(dt, gyd, idx2, xp, B, H, W, Cc)
    b = ops.maxpool_bwd(dt, gyd, idx2, None, B, H, W, Cc, y=y2)
    assert torch.equal(a, b)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('H,W', [(500, 64), (496, 64), (128, 64), (37, 20)])
def test_stem(ops, dt, H, W):
    """conv0 (1->3 1x1 + bias) folded into conv1 7x7 s2 p3 with border-aware bias, + FrozenBN + ReLU; conv0 grads"""
    B = 2
    x = randn(B, 1, H, W)
    w0, b0 = (randn(3, 1, 1, 1) * 0.5).requires_grad_(True), (randn(3) * 0.5).requires_grad_(True)
    w1 = randn(64, 3, 7, 7) / math.sqrt(147)
    sc, bi = torch.rand(64, generator=G) + 0.5, randn(64) * 0.1
    pre = F.conv2d(F.conv2d(x, w0, b0), w1, stride=2, padding=3)
    ref = F.relu(pre * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1))
    wcat = ops.stem_prep(dt, w0.detach().cuda(), b0.detach().cuda(), w1.cuda())
    col, Ho, Wo = ops.stem_im2col(dt, x.cuda().contiguous(), B, H, W)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    y = ops.linear(dt, col, wcat, scale=sc.cuda(), bias=bi.cuda(), act=ops.ACT_RELU)
    close(y, _nhwc(ref).reshape(-1, 64).detach(), dt)
    gy = randn(B, 64, Ho, Wo)
    gpre = rnd(gy, dt)
    (pre * sc.view(1, -1, 1, 1)).backward(gpre)
    gyd = dev(_nhwc(gy).reshape(-1, 64), dt)
    Gm = ops.linear_wgrad(dt, gyd, col) * sc.cuda().view(-1, 1)        # [64][128], BN scale applied per row
    dw0, db0 = ops.stem_conv0_grad(Gm.contiguous(), w1.cuda())
    close(dw0, w0.grad, dt)
    close(db0, b0.grad, dt)


@pytest.mark.parametrize('B,H', [(2, 500), (3, 496), (5, 128), (2, 37), (1, 9), (20, 500)])   # the last: 2-3 tiles per persistent workgroup
def test_stem_one_launch_forward_and_backward(ops, B, H):
    """csrc/stem.hip: conv0 o conv1 o FrozenBN o ReLU o max-pool in one launch, and the conv0 gradients from the pooled gradient
    in one launch + reduce, against plain PyTorch f32 (bf16-rounded operands) and against the unfused HIP chain"""
    dt, W = BF16, 64
    x = randn(B, 1, H, W)
    w0, b0 = (randn(3, 1, 1, 1) * 0.5).requires_grad_(True), (randn(3) * 0.5).requires_grad_(True)
    w1 = randn(64, 3, 7, 7) / math.sqrt(147)
    sc, bi = torch.rand(64, generator=G) + 0.5, randn(64) * 0.1
    pre = F.conv2d(F.conv2d(x, w0, b0), w1, stride=2, padding=3)
    s1_ref = F.relu(pre * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1))
    xd = x.cuda().contiguous()
    wcat = ops.stem_prep(dt, w0.detach().cuda(), b0.detach().cuda(), w1.cuda())
    pool, idx, Hp, Wp, s1 = ops.stem_pool_fwd(xd, wcat, sc.cuda(), bi.cuda(), B, H, W, want_s1=True)
    Ho = (H - 1) // 2 + 1
    assert (Hp, Wp) == ((Ho - 1) // 2 + 1, 16)
    close(s1, _nhwc(s1_ref).reshape(-1, 64).detach(), dt)
    # the pooling half is exact given the un-pooled tile the kernel itself produced (values, argmax bytes, first-maximum rule)
    p2, i2, _, _ = ops.maxpool_fwd(dt, s1, B, Ho, 32, 64)
    assert torch.equal(pool, p2) and torch.equal(idx, i2)
    # without the argmax bytes (no-grad form): same pooled values
    p3, i3, _, _ = ops.stem_pool_fwd(xd, wcat, sc.cuda(), bi.cuda(), B, H, W, want_idx=False)
    assert i3 is None and torch.equal(p3, pool)
    # ---- backward: conv0 gradients from the pooled gradient
    gy = randn(B, 64, Hp, Wp)
    gyd = dev(_nhwc(gy).reshape(-1, 64), dt)
    Gd = ops.stem_pool_wgrad(xd, gyd, idx, pool, sc.cuda(), B, H, W)
    # unfused HIP chain on the same tensors: max-pool backward through the ReLU, then the GEMM weight gradient
    col, _, _ = ops.stem_im2col(dt, xd, B, H, W)
    gs = ops.maxpool_bwd(dt, gyd, idx, None, B, Ho, 32, 64, y=pool)
    Gu = ops.linear_wgrad(dt, gs, col) * sc.cuda().view(-1, 1)
    close(Gd, Gu.float().cpu(), F32, f32_tol=2e-3)                       # same bf16 products, different summation order
    assert torch.equal(Gd[:, 49:64], torch.zeros_like(Gd[:, 49:64])) and torch.equal(Gd[:, 113:], torch.zeros_like(Gd[:, 113:]))
    dw0, db0 = ops.stem_conv0_grad(Gd, w1.cuda())
    # independent reference for the gradient matrix: the kernel's own argmax bytes route the pooled gradient on the CPU (which
    # window element a bf16 near-tie selects is the forward's business and is pinned above; a torch max_pool2d reference re-decides
    # the ties and moves whole +- terms: 10 % of the matrix scale), then einsum against unfolded bf16-rounded patches
    gp = rnd(gy, dt).permute(0, 2, 3, 1) * (pool.float().cpu().view(B, Hp, Wp, 64) > 0)
    code = idx.cpu().view(B, Hp, Wp, 64).long()
    gs_ref = torch.zeros(B, Ho, 32, 64)
    hh, ww = torch.arange(Hp).view(1, Hp, 1, 1), torch.arange(Wp).view(1, 1, Wp, 1)
    ho, wo = (2 * hh - 1 + code // 3), (2 * ww - 1 + code % 3)
    bb_ = torch.arange(B).view(B, 1, 1, 1).expand_as(code)
    cc_ = torch.arange(64).view(1, 1, 1, 64).expand_as(code)
    gs_ref.index_put_((bb_.reshape(-1), ho.reshape(-1), wo.reshape(-1), cc_.reshape(-1)), gp.reshape(-1), accumulate=True)
    gs_ref = rnd(gs_ref, dt).permute(0, 3, 1, 2).reshape(B, 64, -1)
    patches = F.unfold(rnd(x, dt), 7, padding=3, stride=2)                   # [B, 49, Ho*Wo]
    inb = F.unfold(torch.ones_like(x), 7, padding=3, stride=2)
    Gx = torch.einsum('bcp,btp->ct', gs_ref, patches) * sc.view(-1, 1)
    Gi = torch.einsum('bcp,btp->ct', gs_ref, inb) * sc.view(-1, 1)
    close(Gd[:, :49], Gx, dt)
    close(Gd[:, 64:113], Gi, dt)
    w1v = w1.view(64, 3, 49)
    for got, Gref in ((dw0.view(3), Gx), (db0, Gi)):
        ref = torch.einsum('oct,ot->c', w1v, Gref)
        scale = torch.einsum('oct,ot->c', w1v.abs(), Gref.abs()).max().item()
        assert (got.cpu() - ref).abs().max().item() < 2e-2 * scale
    # bit-reproducible
    assert torch.equal(ops.stem_pool_wgrad(xd, gyd, idx, pool, sc.cuda(), B, H, W), Gd)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_posenc_mask_resize_against_oracle(ops, dt):
    from oracle import sedt_oracle as O
    B, T = 3, 500
    m = torch.zeros(B, T, 64, dtype=torch.bool)
    m[1, 360:, :] = True
    m[2, 100:, :] = True
    small = F.interpolate(m[None].float(), size=(32, 4)).to(torch.bool)[0]
    got = ops.mask_resize(m.to(torch.uint8).cuda(), 32, 4)
    assert torch.equal(got.cpu().bool(), small)
    pe = O.PositionEmbeddingSine(256, normalize=True)(O.NestedTensor(torch.zeros(B, 1, 32, 4), small))   # (B,256,32,4)
    ref = pe.flatten(2).permute(0, 2, 1)                                                                   # (B,S,256)
    pos = ops.posenc(dt, got, 256)
    close(pos, ref, dt, f32_tol=2e-5, bf16_tol=5e-3)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_small_elementwise(ops, dt):
    a, b = randn(130, 256), randn(10, 256)
    close(ops.add(dt, dev(a, dt), dev(b, dt), b_mod=10), rnd(a, dt) + rnd(b, dt)[torch.arange(130) % 10], dt, bf16_tol=1e-2)
    close(ops.cast(a.cuda(), dt), rnd(a, dt), dt, bf16_tol=1e-6)
    close(ops.cast(dev(a, dt), F32), rnd(a, dt), F32)
    s = torch.sigmoid(a)
    close(ops.sigmoid_grad(b.repeat(13, 1).cuda(), s.cuda()), b.repeat(13, 1) * s * (1 - s), F32)
    x = randn(6, 32, 2048)
    close(ops.avgpool(dt, dev(x.reshape(-1, 2048), dt), 6, 32, 2048), rnd(x, dt).mean(1), dt, bf16_tol=1e-3)
    w, bb, rm, rv = torch.rand(64, generator=G), randn(64), randn(64), torch.rand(64, generator=G) + 0.5
    sc, bi = ops.bn_fold(w.cuda(), bb.cuda(), rm.cuda(), rv.cuda())
    close(sc, w * (rv + 1e-5).rsqrt(), F32)
    close(bi, bb - rm * w * (rv + 1e-5).rsqrt(), F32)


def test_adamw_clip_matches_torch(ops):
    n = 100003
    p0, g0 = randn(n), randn(n) * 3
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-4, weight_decay=1e-4)
    p, m, v = p0.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    ss = torch.zeros(1).cuda()
    for step in (1, 2, 3):
        g = g0 * step
        pr.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 0.1)
        opt.step()
        ops.sumsq(g.cuda(), ss)
        assert abs(ss.sqrt().item() - tn.item()) < 1e-3 * tn.item()
        ops.adamw_clip(p, g.cuda(), m, v, ss, 0.1, 1e-4, 0.9, 0.999, 1e-8, 1e-4, step)
        np.testing.assert_allclose(p.cpu().numpy(), pr.detach().numpy(), rtol=1e-5, atol=1e-7)


def test_fused_multi_tensor_adamw_matches_torch(ops):
    from sound_event_detection_transformer_amd.optim import FusedAdamW
    shapes = [(300, 70), (65536 * 2 + 5,), (11,), (64, 3, 7, 7), (1,)]
    ref = [randn(*s).requires_grad_(True) for s in shapes]
    mine = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    o_ref = torch.optim.AdamW([{'params': ref[:3]}, {'params': ref[3:], 'lr': 3e-4}], lr=1e-4, weight_decay=1e-4)
    o_my = FusedAdamW([{'params': mine[:3]}, {'params': mine[3:], 'lr': 3e-4}], lr=1e-4, weight_decay=1e-4)
    for step in range(3):
        gs = [randn(*s) * (step + 1) for s in shapes]
        for r, m, g in zip(ref, mine, gs):
            r.grad = g.clone()
            m.grad = g.clone().cuda()
        tn = torch.nn.utils.clip_grad_norm_(ref, 0.1)
        o_ref.step()
        o_my.step(max_norm=0.1)
        assert abs(o_my.grad_norm().item() - tn.item()) < 1e-4 * tn.item()
        for r, m in zip(ref, mine):
            np.testing.assert_allclose(m.detach().cpu().numpy(), r.detach().numpy(), rtol=2e-5, atol=1e-7)


def test_fused_adamw_steps_issued_back_to_back_keep_their_own_gradient_pointers(ops):
    """two eager steps queued behind a busy device, the second with its gradients at OTHER addresses: each step must run on
    the chunk table it was issued with (the pinned table used to be rewritten while the first step's upload was still queued,
    so both steps read the second step's gradients); a third step with unchanged pointers re-uses the device table as is"""
    from sound_event_detection_transformer_amd.optim import FusedAdamW
    shapes = [(300, 70), (40000,), (11,)]

    def run(sync):
        ps = [(randn(*s) * 0 + 1.0).cuda().requires_grad_(True) for s in shapes]
        opt = FusedAdamW(ps, lr=1e-2, weight_decay=0.0)
        g1 = [torch.full(s, 0.5, device='cuda') for s in shapes]
        g2 = [torch.full(s, -3.0, device='cuda') for s in shapes]
        for p, g in zip(ps, g1):
            p.grad = g
        opt.step(max_norm=0.0)                       # (builds the tables; everything below re-uses them)
        torch.cuda.synchronize()
        busy = torch.randn(4096, 4096, device='cuda')
        for _ in range(60):                          # ~10 ms of queued work: the host runs ahead of the device
            busy = busy @ busy * 1e-3
        opt.step(max_norm=0.0)
        if sync:
            torch.cuda.synchronize()
        for p, g in zip(ps, g2):
            p.grad = g
        opt.step(max_norm=0.0)
        if sync:
            torch.cuda.synchronize()
        opt.step(max_norm=0.0)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in ps]

    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('M,N,K', [(8192, 256, 2048), (300, 768, 256), (704, 11, 256), (1000, 64, 64)])
def test_linear_wgrad_with_fused_bias_grad(ops, dt, M, N, K):
    """bias gradient = column sums of dY: rides along in the bf16 LDS-DMA wgrad kernel, separate colsum otherwise"""
    x, gy = randn(M, K), randn(M, N)
    db = torch.full((N,), float('nan')).cuda()
    dw = ops.linear_wgrad(dt, dev(gy, dt), dev(x, dt), bias_out=db)
    close(dw, rnd(gy, dt).t() @ rnd(x, dt), dt)
    close(db, rnd(gy, dt).sum(0), dt, bf16_tol=2e-3)


def test_igemm_co_matches_separate_launches(ops):
    """co-scheduled launch (a dgrad GEMM + weight-gradient problems riding in the same grid) == the same GEMMs launched
    one by one: bit-identical outputs (same kernels bodies, same per-tile arithmetic)"""
    import ctypes as C
    from sound_event_detection_transformer_amd import lib as L
    B = 8
    torch.manual_seed(3)

    def problem(Hi, Wi, Ci, Co, k, s, pd, dl):
        g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
        x = torch.randn(B * Hi * Wi, Ci, device='cuda').bfloat16()
        gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda').bfloat16()
        return g, x, gy

    def wgrad_args(g, x, gy):
        Mo, No, Kp = g.Co, g.taps * g.Ci, B * g.Ho * g.Wo
        sk = L.load().sedt_igemm_splitk(Mo, No, Kp, L.BF16)
        slab = torch.zeros((sk, Mo, No), device='cuda', dtype=torch.float32)
        conv = None if g.plain else ops._geom_tuple(g)
        a = ops.igemm_args(Mo, No, Kp, gy, gy.stride(0), x, x.stride(0), slab, No, trans=1, conv=conv, out_f32=1, splitk=sk,
                           slab=slab)
        return a, slab

    # main problem: a plain GEMM with K = 128 -> the 64x64 2-stage kernel, the configuration that can carry riders
    xm = torch.randn(4096, 128, device='cuda').bfloat16()
    wm = (torch.randn(64, 128, device='cuda') / 11.0).bfloat16()
    riders = [problem(32, 4, 256, 128, 3, 1, 1, 1), problem(32, 4, 512, 128, 1, 1, 0, 1), problem(16, 8, 64, 64, 3, 1, 1, 1)]
    outs = {}
    for mode in ('separate', 'co'):
        dx = torch.zeros((4096, 64), device='cuda', dtype=torch.bfloat16)
        main = ops.igemm_args(4096, 64, 128, xm, xm.stride(0), wm, wm.stride(0), dx, dx.stride(0), act=ops.ACT_RELU)
        ws = [wgrad_args(*r) for r in riders]
        arr = (L.SedtIgemm * len(ws))(*[a for a, _ in ws])
        lib = L.load()
        if mode == 'co':
            taken = C.c_int(0)
            L.check(lib.sedt_igemm_co(C.byref(main), arr, len(ws), L.BF16, L.stream_ptr(), C.byref(taken)), 'igemm_co')
            assert taken.value == 1
        else:
            L.check(lib.sedt_igemm(C.byref(main), L.BF16, L.stream_ptr()), 'igemm')
            for a, _ in ws:
                L.check(lib.sedt_igemm(C.byref(a), L.BF16, L.stream_ptr()), 'igemm')
        torch.cuda.synchronize()
        outs[mode] = [dx] + [s for _, s in ws]
    for a, b in zip(outs['separate'], outs['co']):
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a, b)
    assert outs['co'][0].abs().sum() > 0 and all(o.abs().sum() > 0 for o in outs['co'][1:])


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('M,N,K,act', [(2112, 11, 256, 'none'), (2112, 2, 256, 'sigmoid'), (64, 10, 256, 'none'),
                                       (333, 16, 128, 'relu'), (17, 1, 64, 'sigmoid')])
def test_skinny_linear_fwd_bwd(ops, dt, M, N, K, act):
    """direct small-N head kernels (csrc/skinny.hip) vs torch: y, dx (with and without the relu-input mask), dW, db"""
    ACT = {'none': ops.ACT_NONE, 'sigmoid': ops.ACT_SIGMOID, 'relu': ops.ACT_RELU}[act]
    fn = {'none': lambda t: t, 'sigmoid': torch.sigmoid, 'relu': F.relu}[act]
    assert ops.skinny_ok(N, K)
    x, w, b, g = randn(M, K), randn(N, K) / math.sqrt(K), randn(N), randn(M, N)
    xr = rnd(x, dt).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = fn(xr @ wr.t() + br)
    yr.backward(g)
    xd = dev(x, dt)
    y = ops.skinny_linear_fwd(dt, xd, w.cuda(), b.cuda(), ACT, out_f32=True)
    close(y, yr.detach(), dt, bf16_tol=1e-5 if dt == BF16 else 2e-2)      # f32 math on bf16-rounded inputs: exact to rounding
    gx, gw, gb = ops.skinny_linear_bwd(dt, g.cuda(), y if act != 'none' else None, w.cuda(), xd, ACT)
    close(gx, xr.grad, dt)
    close(gw, wr.grad, dt, bf16_tol=1e-4)
    close(gb, br.grad, dt, bf16_tol=1e-4)
    # relu-input mask on dx (x is the saved post-relu activation of the previous layer)
    gx2, _, _ = ops.skinny_linear_bwd(dt, g.cuda(), y if act != 'none' else None, w.cuda(), xd, ACT, mask=xd, need_gw=False, need_gb=False)
    close(gx2, xr.grad * (rnd(x, dt) > 0), dt)
    # bf16-typed output variant
    y2 = ops.skinny_linear_fwd(dt, xd, w.cuda(), b.cuda(), ACT, out_f32=False)
    assert y2.dtype == TD[dt]
    close(y2, yr.detach(), dt)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_pack_plan_layouts(dt):
    """one-launch weight packing (vectorized and scalar tiles) vs torch permutes: forward [Co][taps][Ci], dgrad [Ci][taps][Co] * bn scale"""
    from sound_event_detection_transformer_amd.packing import PackPlan
    shapes = [(64, 64, 1), (256, 64, 1), (72, 136, 1), (64, 64, 3), (40, 96, 3), (512, 512, 3), (11, 256, 1), (256, 20, 1), (24, 12, 3)]
    convs, linears = [], []
    for co, ci, k in shapes:
        w = torch.nn.Parameter(randn(co, ci, k, k).cuda())
        bn = tuple(t.cuda() for t in (torch.rand(co, generator=G) + 0.5, randn(co), randn(co), torch.rand(co, generator=G) + 0.5))
        convs.append((w, bn))
    for n, k in [(256, 256), (2048, 256), (10, 256), (256, 2048)]:
        linears.append(torch.nn.Parameter(randn(n, k).cuda()))
    plan = PackPlan(dt, torch.device('cuda'), convs, linears)
    with plan:
        for w, bn in convs:
            wf, wb, sc, bi = plan.table[w.data_ptr()]
            co, ci, k = w.shape[0], w.shape[1], w.shape[2]
            scale = (bn[0] * torch.rsqrt(bn[3] + 1e-5)).cpu()
            ref_f = w.detach().cpu().permute(0, 2, 3, 1).reshape(co, k * k * ci)
            ref_b = (w.detach().cpu() * scale[:, None, None, None]).permute(1, 2, 3, 0).reshape(ci, k * k * co)
            close(wf, rnd(ref_f, dt), dt, f32_tol=1e-6, bf16_tol=1e-6)
            close(wb, ref_b, dt, f32_tol=1e-6, bf16_tol=8e-3)
            close(sc, scale, F32, f32_tol=1e-5)
        for w in linears:
            wf, wb, _, _ = plan.table[w.data_ptr()]
            close(wf, rnd(w.detach().cpu(), dt), dt, f32_tol=1e-6, bf16_tol=1e-6)
            close(wb, w.detach().cpu().t(), dt, f32_tol=1e-6, bf16_tol=8e-3)


@pytest.mark.parametrize('dt', [F32, BF16])
def test_layernorm_bwd_with_fused_dropout_grad(ops, dt):
    """second output of layernorm_bwd(drop=...) == dropout_grad(dx) with the same (p, seed): bit-identical"""
    rows, D = 700, 256
    x, dy, gam = randn(rows, D), randn(rows, D), torch.rand(D, generator=G) + 0.5
    xd, dyd = dev(x, dt), dev(dy, dt)
    y, _, mean, rstd = ops.layernorm_fwd(dt, xd, gam.cuda(), torch.zeros(D).cuda())
    dx0, dg0, db0 = ops.layernorm_bwd(dt, dyd, xd, gam.cuda(), mean, rstd)
    dx, dg, db, dxd = ops.layernorm_bwd(dt, dyd, xd, gam.cuda(), mean, rstd, drop=(0.1, 1234, None))
    assert torch.equal(dx, dx0) and torch.equal(dg, dg0) and torch.equal(db, db0)
    ref = ops.dropout_grad(dt, dx, 0.1, 1234)
    assert torch.equal(dxd, ref)
    kept = (dxd != 0).float().mean().item()
    assert 0.85 < kept < 0.95
    # p = 0: the extra result is dx itself
    assert ops.layernorm_bwd(dt, dyd, xd, gam.cuda(), mean, rstd, drop=(0.0, 0, None))[3] is not None


@pytest.mark.parametrize('dt', [F32, BF16])
@pytest.mark.parametrize('cfg', [(32, 4, 128, 512, 3, 1, 2, 2, 5), (16, 4, 512, 640, 1, 1, 0, 1, 7), (17, 4, 640, 512, 1, 1, 0, 1, 3),
                                 (16, 4, 256, 256, 3, 1, 1, 1, 6), (16, 4, 256, 1024, 1, 1, 0, 1, 5)])
def test_wgrad_large_tiles(ops, dt, cfg):
    """weight gradients with Cout, taps*Cin >= 512: the 128x128 ping-pong kernel (csrc/wgrad4.hip) in bf16 mode -
    layer4-like dilated 3x3 and 1x1 problems, K a multiple of the 64-pixel tile or not, every split-K slice count the library picks"""
    Hi, Wi, Ci, Co, k, s, pd, dl, B = cfg
    g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
    x, gy = randn(B, Ci, Hi, Wi), randn(B, Co, g.Ho, g.Wo)
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    F.conv2d(rnd(x, dt), w, stride=s, padding=pd, dilation=dl).backward(rnd(gy, dt))
    xd, gyd = dev(_nhwc(x).reshape(-1, Ci), dt), dev(_nhwc(gy).reshape(-1, Co), dt)
    dw = ops.wgrad(dt, gyd, xd, B, g)
    close(dw, w.grad, dt, bf16_tol=2e-3)          # same bf16-rounded inputs, f32 accumulation: only summation order differs
    # with the bias gradient (column sums of dY) fused into the same launch
    db = torch.empty(Co, device='cuda', dtype=torch.float32)
    dwb = ops.wgrad(dt, gyd, xd, B, g, bias_out=db)
    close(dwb, w.grad, dt, bf16_tol=2e-3)          # (same kernel as dw only under SEDT_WGRAD4_BIAS=1)
    close(db, rnd(gy, dt).sum((0, 2, 3)), dt, bf16_tol=1e-4)
    # the grouped launch (one wide + one small problem) gives the same numbers
    rb = ops.ReduceBatch()
    dw2 = ops.wgrad(dt, gyd, xd, B, g, batch=rb)
    g1 = ops.ConvGeom(Hi, Wi, Ci, 64, 1, 1, 0, 1)
    gy1 = randn(B * Hi * Wi, 64)
    dw1 = ops.wgrad(dt, dev(gy1, dt), xd, B, g1, batch=rb)
    rb.flush()
    assert torch.equal(dw2, dw)
    close(dw1.view(64, Ci), rnd(gy1, dt).t() @ rnd(_nhwc(x).reshape(-1, Ci), dt), dt)


@pytest.mark.parametrize('cfg', [(125, 16, 256, 64, 1, 1, 0, 1), (125, 16, 64, 256, 1, 1, 0, 1), (125, 16, 64, 64, 3, 1, 1, 1),
                                 (32, 4, 512, 512, 3, 1, 2, 2), (32, 4, 2048, 512, 1, 1, 0, 1), (128, 1, 256, 2048, 1, 1, 0, 1)])
def test_gemm_repeat_runs_are_bit_identical(ops, cfg):
    """bf16 forward / dgrad / wgrad of the step's shapes at full size (B = 64): repeated launches on fixed inputs must be
    bit-identical (no atomics, fixed summation orders) - and any LDS race shows up here as sporadic differing elements (the
    K = 64 single-stage kernel did before every barrier retired the wave's LDS reads)"""
    Hi, Wi, Ci, Co, k, s, pd, dl = cfg
    B, dt = 64, BF16
    g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
    gen = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B * Hi * Wi, Ci, device='cuda', generator=gen).bfloat16()
    gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda', generator=gen).bfloat16()
    res = torch.randn(B * g.Ho * g.Wo, Co, device='cuda', generator=gen).bfloat16()
    w = torch.randn(Co, Ci, k, k, device='cuda', generator=gen) / math.sqrt(Ci * k * k)
    wf, wb = ops.pack_conv(dt, w)
    fns = {'fwd': lambda: ops.conv_fwd(dt, x, B, g, wf, act=ops.ACT_RELU, res=res, ldr=Co),
           'fwd_plain': lambda: ops.conv_fwd(dt, x, B, g, wf),
           'dgrad': lambda: ops.conv_dgrad(dt, gy, B, g, wb, mask=x, ldm=Ci),
           'wgrad': lambda: ops.wgrad(dt, gy, x, B, g)}
    for name, fn in fns.items():
        first = fn().clone()
        assert torch.isfinite(first.float()).all(), name
        for _ in range(8):
            junk = torch.full_like(first, float('nan'))          # recycle allocator blocks with NaNs: unwritten outputs show
            del junk
            assert torch.equal(fn(), first), name


@pytest.mark.parametrize('B,H', [(2, 125), (3, 124), (1, 16), (2, 7), (5, 33), (70, 125), (300, 32)])   # the last two: 3 tiles per persistent workgroup
def test_direct_conv3x3_c64_forward_and_dgrad(ops, B, H):
    """csrc/conv3x3_c64.hip (layer1 conv2 geometry: 64 -> 64 channels, 16-wide map) against F.conv2d and against the implicit-GEMM
    kernel it replaces, forward (FrozenBN scale / bias + ReLU) and input gradient (taps flipped, ReLU mask of the consumer)"""
    dt, W, C = BF16, 16, 64
    g = ops.ConvGeom(H, W, C, C, 3, 1, 1, 1)
    x = randn(B, C, H, W)
    w = randn(C, C, 3, 3) / math.sqrt(C * 9)
    sc, bi = torch.rand(C, generator=G) + 0.5, randn(C) * 0.1
    xd = dev(_nhwc(x).reshape(-1, C), dt)
    wf, wb = ops.pack_conv(dt, w.cuda(), sc.cuda())
    assert ops.CONV3_DIRECT
    y = ops.conv_fwd(dt, xd, B, g, wf, scale=sc.cuda(), bias=bi.cuda(), act=ops.ACT_RELU)
    ref = F.relu(F.conv2d(rnd(x, dt), rnd(w, dt), padding=1) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1))
    close(y, _nhwc(ref).reshape(-1, C), dt)
    y0 = ops.conv_fwd(dt, xd, B, g, wf)                                    # no epilogue operands at all
    close(y0, _nhwc(F.conv2d(rnd(x, dt), rnd(w, dt), padding=1)).reshape(-1, C), dt)
    gy = randn(B, C, H, W)
    gyd = dev(_nhwc(gy).reshape(-1, C), dt)
    msrc = dev(_nhwc(randn(B, C, H, W)).reshape(-1, C), dt)
    dx = ops.conv_dgrad(dt, gyd, B, g, wb, mask=msrc, ldm=C)
    wsc = rnd(w * sc.view(-1, 1, 1, 1), dt)                                 # the dgrad pack carries the BN scale
    refdx = F.conv_transpose2d(rnd(gy, dt), wsc, padding=1) * (_nchw(msrc.float().cpu(), B, H, W) > 0)
    close(dx, _nhwc(refdx).reshape(-1, C), dt)
    dx0 = ops.conv_dgrad(dt, gyd, B, g, wb)
    close(dx0, _nhwc(F.conv_transpose2d(rnd(gy, dt), wsc, padding=1)).reshape(-1, C), dt)
    # against the implicit GEMM on the same operands: same bf16 products, f32 accumulation in a different order
    ops.CONV3_DIRECT = False
    try:
        yi = ops.conv_fwd(dt, xd, B, g, wf, scale=sc.cuda(), bias=bi.cuda(), act=ops.ACT_RELU)
        dxi = ops.conv_dgrad(dt, gyd, B, g, wb, mask=msrc, ldm=C)
    finally:
        ops.CONV3_DIRECT = True
    assert (y.float() - yi.float()).abs().max().item() <= 2e-2 * yi.float().abs().max().item()
    assert (dx.float() - dxi.float()).abs().max().item() <= 2e-2 * dxi.float().abs().max().item()
    assert torch.equal(ops.conv_fwd(dt, xd, B, g, wf, scale=sc.cuda(), bias=bi.cuda(), act=ops.ACT_RELU), y)     # reproducible


@pytest.mark.parametrize('cfg', [(32, 4, 256, 256, 3, 1, 1, 1), (32, 4, 1024, 256, 1, 1, 0, 1), (32, 4, 2048, 256, 1, 1, 0, 1),
                                 (31, 2, 256, 256, 3, 1, 1, 1), (31, 2, 1024, 256, 1, 1, 0, 1)])
def test_sixteen_wave_gemm_matches_the_four_wave_tiles(ops, cfg):
    """M = 8192, N = 256, K >= 1024 (layer3's convolutions, the FFN's second linear: exactly one 64x128 tile per CU) run as a
    1024-thread workgroup - two 8-wave ping-pong teams, half of K each (csrc/igemm3.hip, KH = 2); tile = (64, 64) keeps the
    problem on the 4-wave kernel; the last two cases (M = 3968 rows, the B = 32 configurations) take the same form on the 64x64
    tile.  Same bf16 inputs, f32 sums in another order: equal to a bf16 rounding; forward with the
    Bottleneck epilogue, input gradient with a 1-bit mask; repeats are bit-identical"""
    Hi, Wi, Ci, Co, k, s, pd, dl = cfg
    B, dt = 64, BF16
    g = ops.ConvGeom(Hi, Wi, Ci, Co, k, s, pd, dl)
    gen = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(B * Hi * Wi, Ci, device='cuda', generator=gen).bfloat16()
    gy = torch.randn(B * g.Ho * g.Wo, Co, device='cuda', generator=gen).bfloat16()
    res = torch.randn(B * g.Ho * g.Wo, Co, device='cuda', generator=gen).bfloat16()
    sc, bi = torch.rand(Co, device='cuda', generator=gen) + 0.5, torch.randn(Co, device='cuda', generator=gen)
    w = torch.randn(Co, Ci, k, k, device='cuda', generator=gen) / math.sqrt(Ci * k * k)
    wf, wb = ops.pack_conv(dt, w)
    bits = torch.zeros((x.shape[0], Ci // 8), dtype=torch.uint8, device='cuda')
    bits.copy_(((x.float() > 0).view(-1, Ci // 8, 8).to(torch.int32) * (2 ** torch.arange(8, device='cuda', dtype=torch.int32))).sum(-1).to(torch.uint8))
    for fn in (lambda **t: ops.conv_fwd(dt, x, B, g, wf, scale=sc, bias=bi, res=res, ldr=Co, act=ops.ACT_RELU, act_post_res=1, **t),
               lambda **t: ops.conv_dgrad(dt, gy, B, g, wb, mask=bits, ldm=bits.stride(0), mask_bits=True, **t)):
        other = (64, 64) if x.shape[0] >= 8192 else (64, 128)      # a tile that keeps the problem on an 8- or 4-wave kernel
        a, b = fn(), fn(tile=other)
        assert torch.isfinite(a.float()).all()
        assert (a.float() - b.float()).abs().max().item() <= 2e-2 * b.float().abs().max().item()
        assert ((a == 0) != (b == 0)).float().mean().item() < 1e-3
        assert torch.equal(fn(), a)


@pytest.mark.parametrize('dt,M,N,K', [(BF16, 640, 256, 128), (BF16, 200, 72, 40), (F32, 130, 64, 96), (BF16, 8192, 1024, 256)])
def test_one_bit_relu_masks_in_the_gemm_epilogue(ops, dt, M, N, K):
    """SedtIgemm.bits_out / mask_bits (round 3): a GEMM that stores relu(x w^T + res) also leaves the 1-bit image of [out > 0];
    a later GEMM masked by that image equals the GEMM masked by the bf16 / f32 tensor itself - on the LDS-DMA kernels (aligned bf16
    shapes) and on the general kernel (f32 mode, unaligned shapes)"""
    td = torch.bfloat16 if dt == BF16 else torch.float32
    x, w = dev(randn(M, K), dt), dev(randn(N, K) / math.sqrt(K), dt)
    res = dev(randn(M, N), dt)
    bits = torch.zeros((M, N // 8), dtype=torch.uint8, device='cuda')
    y = ops.linear(dt, x, w, res=res, ldr=res.stride(0), act=ops.ACT_RELU, act_post_res=1, bits_out=bits)
    y2 = ops.linear(dt, x, w, res=res, ldr=res.stride(0), act=ops.ACT_RELU, act_post_res=1)
    assert torch.equal(y, y2)
    want = (y.float() > 0).view(M, N // 8, 8).to(torch.int32)
    packed = (want * (2 ** torch.arange(8, device='cuda', dtype=torch.int32))).sum(-1).to(torch.uint8)
    assert torch.equal(bits, packed)
    assert 0.2 < want.float().mean().item() < 0.8
    g, w2 = dev(randn(M, K), dt), dev(randn(N, K) / math.sqrt(K), dt)
    a = ops.linear(dt, g, w2, mask=y, ldm=y.stride(0))
    b = ops.linear(dt, g, w2, mask=bits, ldm=bits.stride(0), mask_bits=True)
    assert a.dtype == td and torch.equal(a, b)


@pytest.mark.parametrize('B,Hi,Wi,C', [(2, 63, 8, 256), (3, 125, 16, 128), (2, 64, 8, 256), (1, 7, 5, 64), (64, 63, 8, 256), (2, 1, 16, 128)])
@pytest.mark.parametrize('epilogue', ['plain', 'mask_bits+res'])
def test_stride2_dgrad_by_output_parity_equals_the_transposed_gather(ops, B, Hi, Wi, C, epilogue):
    """the input gradient of a stride-2 3x3 convolution as four regular sub-convolutions by output parity (ops._conv_dgrad_s2: one grouped
    launch, tap blocks of the packed weight read in place, outputs interleaved through SedtIgemm.omap) against (a) the plain transposed
    gather it replaces and (b) torch's conv_transpose2d in f32 on the bf16-rounded operands; odd and even heights / widths, a single row,
    the residual operand and the 1-bit ReLU mask indexed by the dx pixel"""
    import torch.nn.functional as F
    from sound_event_detection_transformer_amd import lib as L
    g = torch.Generator().manual_seed(B * 1000 + Hi * 10 + Wi)
    geo = ops.ConvGeom(Hi, Wi, C, C, 3, 2, 1, 1)
    w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    _, wb = ops.pack_conv(L.BF16, w)
    dy = torch.randn(B * geo.Ho * geo.Wo, C, generator=g).cuda().bfloat16()
    ep = {}
    if epilogue != 'plain':
        bits = torch.randint(0, 256, (B * Hi * Wi, C // 8), generator=g, dtype=torch.uint8).cuda()
        res = torch.randn(B * Hi * Wi, C, generator=g).cuda().bfloat16()
        ep = dict(mask=bits, ldm=bits.stride(0), mask_bits=True, res=res, ldr=res.stride(0))
    out = {}
    for parity in (True, False):
        keep, ops.S2_PARITY = ops.S2_PARITY, parity
        try:
            with L.launch_log() as log:
                out[parity] = ops.conv_dgrad(L.BF16, dy, B, geo, wb, **ep)
            torch.cuda.synchronize()
        finally:
            ops.S2_PARITY = keep
        assert log['igemm_group_s2'] == (1 if parity else 0) and log['sedt_igemm'] == (0 if parity else 1), dict(log)
    dyn = dy.float().view(B, geo.Ho, geo.Wo, C).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(dyn, w.bfloat16().float(), stride=2, padding=1,
                             output_padding=(Hi - ((geo.Ho - 1) * 2 + 1), Wi - ((geo.Wo - 1) * 2 + 1)))
    ref = ref.permute(0, 2, 3, 1).reshape(B * Hi * Wi, C)
    if ep:
        keepm = ((bits.view(-1, C // 8, 1) >> torch.arange(8, device='cuda', dtype=torch.uint8)) & 1).bool().view(-1, C)
        ref = (ref + res.float()) * keepm
    scale = ref.abs().max().item()
    assert ((out[True].float() - ref).abs().max().item()) < 1e-2 * scale
    # same products, same order of the non-zero taps; the two kernels run different tiles (64x64 against 128x128 with two K groups), so an
    # f32 sum may differ in its last bit and land on the neighbouring bf16 value: one bf16 step of the largest entry = 2^-8 = 3.9e-3 of it
    assert ((out[True].float() - out[False].float()).abs().max().item()) < 8e-3 * scale


@pytest.mark.parametrize('H', [32, 31])
def test_dilated_conv_by_column_halves_equals_the_plain_gather(ops, H):
    """layer4's dilated 3x3 (dilation 2, padding 2, four columns) as two column halves of six taps each in one grouped ping-pong launch
    (ops._conv_dil_halves) against the nine-tap gather it replaces and against torch in f32 on the bf16-rounded operands: forward with
    FrozenBN + ReLU, input gradient with the 1-bit ReLU mask; URBAN-SED (32 rows) and DCASE (31 rows) maps at C2's batch"""
    import torch.nn.functional as F
    from sound_event_detection_transformer_amd import lib as L
    B, W, C = 64, 4, 512
    g = torch.Generator().manual_seed(H)
    geo = ops.ConvGeom(H, W, C, C, 3, 1, 2, 2)
    w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).cuda()
    sc, bi = (1 + 0.1 * torch.randn(C, generator=g)).cuda(), (0.1 * torch.randn(C, generator=g)).cuda()
    wf, wb = ops.pack_conv(L.BF16, w, bnscale=sc)
    x = torch.randn(B * H * W, C, generator=g).cuda().bfloat16()
    dy = torch.randn(B * H * W, C, generator=g).cuda().bfloat16()
    bits = torch.randint(0, 256, (B * H * W, C // 8), generator=g, dtype=torch.uint8).cuda()
    out = {}
    for halves in (True, False):
        keep, ops.DIL_HALVES = ops.DIL_HALVES, halves
        try:
            with L.launch_log() as log:
                y = ops.conv_fwd(L.BF16, x, B, geo, wf, scale=sc, bias=bi, act=ops.ACT_RELU)
                dx = ops.conv_dgrad(L.BF16, dy, B, geo, wb, mask=bits, ldm=bits.stride(0), mask_bits=True)
            torch.cuda.synchronize()
        finally:
            ops.DIL_HALVES = keep
        assert log['igemm_group_dil'] == (2 if halves else 0) and log['sedt_igemm'] == (0 if halves else 2), dict(log)
        out[halves] = (y, dx)
    wq = w.bfloat16().float()
    xn = x.float().view(B, H, W, C).permute(0, 3, 1, 2)
    ref_y = F.relu(F.conv2d(xn, wq, padding=2, dilation=2) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(-1, C)
    wsq = (w * sc.view(-1, 1, 1, 1)).bfloat16().float()                    # (the dgrad operand carries the BN scale)
    keepm = ((bits.view(-1, C // 8, 1) >> torch.arange(8, device='cuda', dtype=torch.uint8)) & 1).bool().view(-1, C)
    ref_dx = F.conv_transpose2d(dy.float().view(B, H, W, C).permute(0, 3, 1, 2), wsq, padding=2, dilation=2).permute(0, 2, 3, 1).reshape(-1, C)
    ref_dx = ref_dx * keepm
    for got, ref in ((out[True][0], ref_y), (out[True][1], ref_dx)):
        assert (got.float() - ref).abs().max().item() < 1e-2 * ref.abs().max().item()
    for a_, b_ in zip(out[True], out[False]):
        assert (a_.float() - b_.float()).abs().max().item() < 8e-3 * b_.float().abs().max().item()
