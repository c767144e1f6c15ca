"""CPU: the oracle restatement reproduces the REFERENCE's outputs (tests/golden, made by
tests/golden/make_golden.py from /root/reference).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import sedt_oracle as O
from oracle.criterion_oracle import build_oracle_criterion, synthetic_targets

TOL = dict(rtol=1e-4, atol=2e-5)


def _load(model, seed):
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), seed))
    return model


def _digest(t, n=64):
    t = t.detach().float().flatten()
    idx = torch.linspace(0, t.numel() - 1, n).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item()], t[idx].numpy()]).astype(np.float32)


@pytest.mark.parametrize('name,E,pre', [('pre_e3', 3, True), ('post_e3', 3, False), ('pre_e6', 6, True)])
def test_g1_transformer(golden_dir, name, E, pre):
    g = np.load(os.path.join(golden_dir, 'g1_transformer.npz'))
    m = _load(O.Transformer(256, 8, E, 3, 2048, 0.1, pre, True, False).eval(), 11)
    gen = torch.Generator().manual_seed(21)
    src = torch.randn(2, 256, 32, 4, generator=gen)
    pos = torch.randn(2, 256, 32, 4, generator=gen) * 0.5
    query = torch.randn(11, 256, generator=gen)
    mask = torch.zeros(2, 32, 4, dtype=torch.bool)
    mask[1, 25:, :] = True
    with torch.no_grad():
        hs, mem = m(src, mask, query, pos)
    np.testing.assert_allclose(hs.numpy(), g[f'{name}_hs'], **TOL)
    np.testing.assert_allclose(mem.numpy(), g[f'{name}_mem'], **TOL)


def test_g1_selfsup(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g1_transformer.npz'))
    m = _load(O.Transformer(256, 8, 3, 3, 2048, 0.1, True, True, True).eval(), 12)
    gen = torch.Generator().manual_seed(22)
    src = torch.randn(2, 256, 31, 4, generator=gen)
    pos = torch.randn(2, 256, 31, 4, generator=gen) * 0.5
    qe = torch.randn(20, 2, 256, generator=gen)
    am = torch.ones(20, 20) * float('-inf')
    for i in range(10):
        am[2 * i:2 * i + 2, 2 * i:2 * i + 2] = 0
    with torch.no_grad():
        hs, mem = m(src, torch.zeros(2, 31, 4, dtype=torch.bool), qe, pos, decoder_mask=am)
    np.testing.assert_allclose(hs.numpy(), g['selfsup_hs'], **TOL)
    np.testing.assert_allclose(mem.numpy(), g['selfsup_mem'], **TOL)


def test_g6_posenc(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g6_posenc.npz'))
    pe = O.PositionEmbeddingSine(256, normalize=True)
    for h in (32, 31, 8):
        p = pe(O.NestedTensor(torch.zeros(1, 2048, h, 4), torch.zeros(1, h, 4, dtype=torch.bool)))
        np.testing.assert_allclose(p[0, :, :, 0].t().numpy(), g[f'pos_{h}'], rtol=1e-6, atol=1e-6)
    m = torch.zeros(1, 32, 4, dtype=torch.bool)
    m[0, 23:, :] = True
    p = pe(O.NestedTensor(torch.zeros(1, 2048, 32, 4), m))
    np.testing.assert_allclose(p[0, :, :, 0].t().numpy(), g['pos_32_pad23'], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('name,E,Q,T', [('urban', 3, 10, 500), ('dcase', 6, 20, 496)])
def test_g2_g3_sedt(golden_dir, name, E, Q, T):
    g = np.load(os.path.join(golden_dir, 'g2_g3_sedt.npz'))
    model = _load(O.build_oracle_model(10, Q, E, 3, True, True, True, dropout=0.0), 2020)
    assert sum(p.numel() for p in model.backbone.parameters()) == 23454918          # SURVEY 8c check
    x = torch.randn(2, 1, T, 64, generator=torch.Generator().manual_seed(7))
    model.eval()
    with torch.no_grad():
        o = model(x)
        feat, stages = model.backbone[0].body(x, return_stages=True)
    for k in ('pred_logits', 'pred_boxes', 'at'):
        np.testing.assert_allclose(o[k].numpy(), g[f'{name}_eval_{k}'], rtol=2e-4, atol=5e-5)
    for i, a in enumerate(o['aux_outputs']):
        np.testing.assert_allclose(a['pred_logits'].numpy(), g[f'{name}_eval_aux{i}_logits'], rtol=2e-4, atol=5e-5)
        np.testing.assert_allclose(a['pred_boxes'].numpy(), g[f'{name}_eval_aux{i}_boxes'], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(_digest(stages['conv1']), g[f'{name}_stage_bn1'], rtol=1e-4, atol=1e-5)
    for s in ('layer1', 'layer2', 'layer3', 'layer4'):
        np.testing.assert_allclose(_digest(stages[s]), g[f'{name}_stage_{s}'], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        o = model([x[0], x[1][:, :T - 140, :]])
    for k in ('pred_logits', 'pred_boxes', 'at'):
        np.testing.assert_allclose(o[k].numpy(), g[f'{name}_ragged_{k}'], rtol=2e-4, atol=5e-5)

    # G3: train mode (dropout 0), criterion, grads
    model.train()
    crit = build_oracle_criterion(10, 3, True, True)
    targets = synthetic_targets(2, 99, 10)
    o = model(x)
    ld, _ = crit(o, targets, None, slice(2))
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    total.backward()
    assert abs(total.item() - float(g[f'{name}_train_total'])) < 1e-3 * abs(float(g[f'{name}_train_total']))
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'{name}_train_loss_{k}'])) < 1e-3 * max(1.0, abs(v.item())), k
    params = dict(model.named_parameters())
    names = [str(n) for n in g[f'{name}_train_gradnames']]
    assert names == [n for n, p in model.named_parameters() if p.requires_grad]
    assert [str(n) for n in g[f'{name}_train_frozen']] == [n for n, p in model.named_parameters() if not p.requires_grad]
    gn = np.array([params[n].grad.norm().item() for n in names], dtype=np.float32)
    np.testing.assert_allclose(gn, g[f'{name}_train_gradnorm'], rtol=2e-3, atol=1e-5)
    for key in g.files:
        if key.startswith(f'{name}_train_grad::'):
            n = key.split('::')[1]
            ref = g[key]
            np.testing.assert_allclose(_digest(params[n].grad, 32), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())


def test_g5_criterion(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_criterion.npz'))
    crit = build_oracle_criterion(10, 3, True, True)
    gen = torch.Generator().manual_seed(55)
    B, Q = 6, 10
    outputs = {'pred_logits': torch.randn(B, Q, 11, generator=gen), 'pred_boxes': torch.rand(B, Q, 2, generator=gen) * 0.8 + 0.1,
               'at': torch.rand(B, 10, generator=gen),
               'aux_outputs': [{'pred_logits': torch.randn(B, Q, 11, generator=gen),
                                'pred_boxes': torch.rand(B, Q, 2, generator=gen) * 0.8 + 0.1} for _ in range(2)]}
    targets = synthetic_targets(B, 56, 10)
    idx, _ = crit.matcher({k: v for k, v in outputs.items() if k != 'aux_outputs'}, targets)
    np.testing.assert_array_equal(np.concatenate([i.numpy() for i, _ in idx]), g['match_src'])
    np.testing.assert_array_equal(np.concatenate([j.numpy() for _, j in idx]), g['match_tgt'])
    ld, _ = crit(outputs, targets, None, slice(B))
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'loss_{k}'])) < 1e-5 * max(1.0, abs(v.item())), k
    t2 = [dict(t) for t in targets]
    for t in t2[4:]:
        t['boxes'] = torch.zeros(0, 2)
    ld, _ = crit(outputs, t2, slice(4, 6), slice(4))
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'ws_loss_{k}'])) < 1e-5 * max(1.0, abs(v.item())), k
    ld, _ = crit(outputs, targets, None, slice(B), normalize=True)
    assert abs(ld['loss_ce'].item() - float(g['norm_loss_ce'])) < 1e-5


def test_g4_spsedt(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g4_spsedt.npz'))
    model = _load(O.build_oracle_model(1, 20, 6, 3, False, True, True, dropout=0.0, self_sup=True,
                                       train_backbone=False), 404)
    B, P = 2, 10
    x = torch.randn(B, 1, 496, 64, generator=torch.Generator().manual_seed(8))
    patches = torch.randn(B, P, 1, 128, 64, generator=torch.Generator().manual_seed(9))
    mask = torch.zeros(B, 496, 64, dtype=torch.bool)
    model.eval()
    with torch.no_grad():
        o = model((x, mask), patches)
    for k in ('pred_logits', 'pred_boxes', 'gt_feature'):
        np.testing.assert_allclose(o[k].numpy(), g[f'eval_{k}'], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(_digest(o['pred_feature'], 256), g['eval_pred_feature'], rtol=2e-4, atol=5e-5)
    model.train()
    o = model((x, mask), patches, query_mask=torch.from_numpy(g['train_query_mask']))
    for k in ('pred_logits', 'pred_boxes'):
        np.testing.assert_allclose(o[k].detach().numpy(), g[f'train_{k}'], rtol=2e-4, atol=5e-5)
    crit = build_oracle_criterion(1, 3, False, True, self_sup=True)
    targets = [{'labels': torch.zeros(P, dtype=torch.int64), 'boxes': torch.from_numpy(g['target_boxes'][i])}
               for i in range(B)]
    ld, _ = crit(o, targets, slice(B), slice(B))
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    total.backward()
    assert abs(total.item() - float(g['train_total'])) < 1e-3 * abs(float(g['train_total']))
    names = [str(n) for n in g['train_gradnames']]
    params = dict(model.named_parameters())
    assert names == [n for n, p in model.named_parameters() if p.requires_grad]
    gn = np.array([params[n].grad.norm().item() for n in names], dtype=np.float32)
    np.testing.assert_allclose(gn, g['train_gradnorm'], rtol=2e-3, atol=1e-5)


G16_MODES = ('max', 'avg', 'attn', 'weighted_sum', 'max_nomask')


@pytest.mark.parametrize('mode', G16_MODES)
def test_g16_pooling(golden_dir, mode):
    """--pooling variants (sedt.py:47-61, 96-119) and loss_weak_p (sedt.py:182-185) of the oracle against the reference's
    own outputs, losses and gradients on a strong | weak batch"""
    import sys
    sys.path.insert(0, golden_dir)
    import inputs as GI
    g = np.load(os.path.join(golden_dir, 'g16_pooling.npz'))
    c = GI.POOL
    ns, B = c['n_strong'], c['n_strong'] + c['n_weak']
    nomask = mode.endswith('_nomask')
    pooling = mode.split('_nomask')[0]
    i = 0 if nomask else c['modes'].index(mode)
    model = _load(O.build_oracle_model(10, 10, 3, 3, True, True, True, dropout=0.0, pooling=pooling), c['seed_w'] + i)
    x, targets = GI.pool_batch()
    if nomask:
        x, targets = x[:ns], targets[:ns]
    model.eval()
    with torch.no_grad():
        o = model(x)
    np.testing.assert_allclose(o['at_p'].numpy(), g[f'{mode}_eval_at_p'], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(o['at'].numpy(), g[f'{mode}_eval_at'], rtol=2e-4, atol=5e-5)
    model.train()
    crit = build_oracle_criterion(10, 3, True, True, pooling=pooling, weak_loss_p_coef=0.7)
    o = model(x)
    ld, _ = crit(o, targets, None if nomask else slice(ns, B), slice(ns))
    assert 'loss_weak_p' in ld
    total = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    total.backward()
    np.testing.assert_allclose(o['at_p'].detach().numpy(), g[f'{mode}_train_at_p'], rtol=2e-4, atol=2e-6)
    assert abs(total.item() - float(g[f'{mode}_train_total'])) < 1e-3 * abs(float(g[f'{mode}_train_total']))
    keys = {k[len(mode) + 12:] for k in g.files if k.startswith(f'{mode}_train_loss_')}
    assert set(ld) == keys
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'{mode}_train_loss_{k}'])) < 1e-3 * max(1.0, abs(v.item())), k
    params = dict(model.named_parameters())
    names = [str(n) for n in g[f'{mode}_train_gradnames']]
    assert names == [n for n, p in model.named_parameters() if p.requires_grad]
    gn = np.array([0.0 if params[n].grad is None else params[n].grad.norm().item() for n in names], dtype=np.float32)
    np.testing.assert_allclose(gn, g[f'{mode}_train_gradnorm'], rtol=2e-3, atol=1e-5)
    for key in g.files:
        if key.startswith(f'{mode}_train_grad::'):
            ref = g[key]
            np.testing.assert_allclose(_digest(params[key.split('::')[1]].grad, 32), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())


@pytest.mark.parametrize('name,act,pre', [('gelu_pre', 'gelu', True), ('gelu_post', 'gelu', False), ('relu_post', 'relu', False)])
def test_g17_activation_and_postnorm_backward(golden_dir, name, act, pre):
    """fixture G17 (reference transformer.py with activation='gelu', :423-431, and the post-norm stacks :177-190 / :240-261): the oracle's
    forward AND every parameter gradient under a linear loss - what pins the oracle for the HIP path's post-norm / GELU backward tests"""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import inputs as GI
    g = np.load(os.path.join(golden_dir, 'g17_activation_postnorm.npz'))
    src, pos, query, mask, w_hs, w_mem = GI.g17_inputs()
    m = _load(O.Transformer(256, 8, 3, 3, 2048, 0.0, pre, True, False, activation=act), 17)
    m.eval()
    with torch.no_grad():
        hs, mem = m(src, mask, query, pos)
    np.testing.assert_allclose(hs.numpy(), g[f'{name}_hs'], **TOL)
    np.testing.assert_allclose(mem.numpy(), g[f'{name}_mem'], **TOL)
    m.train()
    s_, q_ = src.clone().requires_grad_(True), query.clone().requires_grad_(True)
    hs, mem = m(s_, mask, q_, pos)
    loss = (hs * w_hs).sum() + (mem * w_mem).sum()
    loss.backward()
    assert abs(loss.item() - float(g[f'{name}_loss'])) <= 1e-4 * abs(float(g[f'{name}_loss']))
    names = [n for n, _ in m.named_parameters()]
    assert names == list(g[f'{name}_gradnames'])
    norms = np.array([p.grad.norm().item() if p.grad is not None else 0.0 for _, p in m.named_parameters()], np.float32)
    np.testing.assert_allclose(norms, g[f'{name}_gradnorm'], rtol=2e-4, atol=1e-6)
    for n, p in m.named_parameters():
        if p.grad is not None:
            ref = g[f'{name}_grad_{n}']
            np.testing.assert_allclose(_digest(p.grad, 32), ref, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    np.testing.assert_allclose(q_.grad.numpy(), g[f'{name}_dquery'], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(_digest(s_.grad, 256), g[f'{name}_dsrc'], rtol=2e-3, atol=2e-5)
