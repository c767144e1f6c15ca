"""CPU: the oracle's restatement of the criterion variants (fine_tune / normalize / focal loss / ratio), PostProcess,
get_pseudo_labels, the mean-teacher step, the input transforms and mixup reproduces what the REFERENCE returned for the same
seeded inputs (fixtures G9-G13, tests/golden/make_golden.py)."""
import os
import sys
from collections import Counter

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
import inputs as GI                                                                    # noqa: E402
from oracle import sedt_oracle as O                                                    # noqa: E402
from oracle import semi_oracle as S                                                    # noqa: E402
from oracle import transforms_oracle as TO                                             # noqa: E402
from oracle.criterion_oracle import build_oracle_criterion, PostProcess, synthetic_targets  # noqa: E402


def _rows(a):
    return [r[r >= 0] for r in a]


class replay_rand(object):
    """matcher.rand returns the recorded per-clip draws in order"""

    def __init__(self, rows):
        self.rows, self.i = rows, 0

    def __call__(self, n):
        r = torch.from_numpy(np.asarray(self.rows[self.i][:n], np.float32))
        self.i += 1
        assert len(r) == n
        return r


G9_CASES = {'ft': (True, False, False, 1.0), 'ft_eps3': (True, False, False, 3.0), 'ft_norm_eps3': (True, True, False, 3.0),
            'fl': (False, False, True, 1.0), 'fl_ft_eps3': (True, False, True, 3.0)}


@pytest.mark.parametrize('name', list(G9_CASES))
def test_g9_criterion_variants(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'g9_criterion_variants.npz'))
    ft, norm, fl, eps = G9_CASES[name]
    crit = build_oracle_criterion(epsilon=eps)
    outputs, targets, B, Q = GI.g9_inputs()
    crit.matcher.rand = replay_rand(_rows(g[f'{name}_rand']))
    ld, idx = crit(outputs, targets, None, slice(B), ft, norm, fl)
    for b, (i, j) in enumerate(idx):
        np.testing.assert_array_equal(i.numpy(), _rows(g[f'{name}_src'])[b])
        np.testing.assert_array_equal(j.numpy(), _rows(g[f'{name}_tgt'])[b])
    keys = {k[len(name) + 6:] for k in g.files if k.startswith(f'{name}_loss_')}
    assert set(ld) == keys
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'{name}_loss_{k}'])) <= 1e-5 * max(1.0, abs(v.item())), k
    if ft:                 # the fine-tune branch really changed the matching in at least one clip
        plain, _ = crit.matcher({k: v for k, v in outputs.items() if k != 'aux_outputs'}, targets, fl=fl)
        assert any(len(a[0]) != len(b[0]) or not torch.equal(a[0], b[0]) for a, b in zip(plain, idx))


def test_g9_focal_weak_split_and_ratio(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g9_criterion_variants.npz'))
    crit = build_oracle_criterion()
    outputs, targets, B, Q = GI.g9_inputs()
    t2 = [dict(t) for t in targets]
    for t in t2[4:]:
        t['boxes'] = torch.zeros(0, 2)
    ld, _ = crit(outputs, t2, slice(4, 6), slice(4), False, False, True)
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'fl_ws_loss_{k}'])) <= 1e-5 * max(1.0, abs(v.item())), k
    t3 = [dict(t) for t in targets]
    for t, r in zip(t3, _rows(g['ratio_values'])):
        if len(r):
            t['ratio'] = torch.from_numpy(r.copy())
    ld, _ = crit(outputs, t3, None, slice(B))
    for k, v in ld.items():
        assert abs(v.item() - float(g[f'ratio_loss_{k}'])) <= 1e-5 * max(1.0, abs(v.item())), k


@pytest.mark.parametrize('name,kw', [('none', dict(audio_tags=None)), ('m1', dict(at_m=1)), ('m2', dict(at_m=2)), ('m3', dict(at_m=3)),
                                     ('m2_t03', dict(at_m=2, threshold=0.3)), ('semi', dict(at_m=1, is_semi=True, threshold=None))])
def test_g10_postprocess(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, 'g10_postprocess.npz'))
    outputs, tags, sizes = GI.g10_inputs()
    kw = dict(kw)
    kw.setdefault('audio_tags', tags)
    r = PostProcess()({k: v.clone() for k, v in outputs.items()}, sizes, **kw)
    np.testing.assert_allclose(np.stack([x['scores'].numpy() for x in r]), g[f'{name}_scores'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(np.stack([x['labels'].numpy() for x in r]), g[f'{name}_labels'])
    np.testing.assert_allclose(np.stack([x['boxes'].numpy() for x in r]), g[f'{name}_boxes'], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('name,kw', [('nms', {}), ('raw', dict(del_overlap=False))])
def test_g11_pseudo_labels(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, 'g11_pseudo_labels.npz'))
    tea, thr, B = GI.g11_inputs()
    targets = [{'labels': torch.zeros(0, dtype=torch.int64), 'boxes': torch.zeros(0, 2), 'orig_size': torch.tensor(10.0)} for _ in range(B)]
    cnt = Counter()
    got = S.get_pseudo_labels(tea, PostProcess(), torch.full((B,), 10.0), targets, cnt, thr, **kw)
    np.testing.assert_array_equal([len(t['labels']) for t in got], g[f'{name}_count'])
    for b, t in enumerate(got):
        np.testing.assert_array_equal(t['labels'].numpy(), _rows(g[f'{name}_labels'])[b])
        np.testing.assert_array_equal(t['boxes'][:, 0].numpy(), _rows(g[f'{name}_centre'])[b])       # copied values: exact
        np.testing.assert_array_equal(t['boxes'][:, 1].numpy(), _rows(g[f'{name}_length'])[b])
    np.testing.assert_array_equal([cnt.get(c, 0) for c in range(10)], g[f'{name}_counter'])


def _semi_setup():
    c = GI.SEMI
    model = O.build_oracle_model(10, 20, 6, 3, True, True, True, dropout=0.0)
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), c['seed_w']))
    model.train()
    ema = S.EMA(model, 0.9)
    ema.register()
    gen = torch.Generator().manual_seed(5)
    for n in ema.shadow:
        ema.shadow[n] = ema.shadow[n] + 0.02 * ema.shadow[n].abs().mean() * torch.randn(ema.shadow[n].shape, generator=gen)
    groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
              {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-4}]
    opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)
    ns, nw, nu = c['n_strong'], c['n_weak'], c['n_unl']
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    return model, ema, build_oracle_criterion(10, 3, True, True), opt, masks


def test_g12_semi_step(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g12_semi_step.npz'))
    torch.set_num_threads(8)
    c = GI.SEMI
    thr = torch.full((10,), c['thr'])
    # (a) optimizer step withheld: losses, pseudo labels, every gradient norm
    model, ema, crit, opt, masks = _semi_setup()
    x_t, x_s, targets = GI.semi_batch()
    cnt = Counter()
    sup, unsup, total, pseudo = S.semi_step(model, ema, crit, opt, x_t, x_s, targets, classwise_threshold=thr, do_step=False,
                                            counter=cnt, **masks)
    assert abs(total.item() - float(g['total'])) <= 2e-5 * abs(float(g['total']))
    np.testing.assert_array_equal([len(t['labels']) for t in pseudo], g['pseudo_count'])
    for b, t in enumerate(pseudo):
        np.testing.assert_array_equal(t['labels'].numpy(), _rows(g['pseudo_labels'])[b])
        np.testing.assert_allclose(t['boxes'][:, 0].numpy(), _rows(g['pseudo_centre'])[b], rtol=1e-5)
    np.testing.assert_array_equal([cnt.get(k, 0) for k in range(10)], g['pseudo_counter'])
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == list(g['gradnames'])
    gn = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
    np.testing.assert_allclose(gn, g['gradnorm'], rtol=2e-3, atol=1e-6)
    # (b) the complete iteration: clip 0.1 + AdamW + EMA update
    model, ema, crit, opt, masks = _semi_setup()
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    shadow0 = {n: v.clone() for n, v in ema.shadow.items()}
    x_t, x_s, targets = GI.semi_batch()
    _, _, total, _ = S.semi_step(model, ema, crit, opt, x_t, x_s, targets, classwise_threshold=thr, **masks)
    assert abs(total.item() - float(g['step_total'])) <= 2e-5 * abs(float(g['step_total']))
    delta = np.array([(dict(model.named_parameters())[n].detach() - before[n]).norm().item() for n in names], np.float32)
    np.testing.assert_allclose(delta, g['step_delta'], rtol=2e-2, atol=1e-7)
    ed = np.array([(ema.shadow[n] - shadow0[n]).norm().item() for n in names], np.float32)
    np.testing.assert_allclose(ed, g['ema_delta'], rtol=1e-4, atol=1e-7)


def _check_target_rows(g, key, targets, exact=True):
    np.testing.assert_array_equal([len(t['labels']) for t in targets], g[f'{key}_nlabels'])
    np.testing.assert_array_equal([len(t['boxes']) for t in targets], g[f'{key}_nboxes'])
    for b, t in enumerate(targets):
        np.testing.assert_array_equal(t['labels'].numpy(), _rows(g[f'{key}_labels'])[b])
        cmp = np.testing.assert_array_equal if exact else (lambda a, b_: np.testing.assert_allclose(a, b_, rtol=1e-5))
        cmp(t['boxes'].reshape(-1, 2)[:, 0].numpy(), _rows(g[f'{key}_centre'])[b])
        cmp(t['boxes'].reshape(-1, 2)[:, 1].numpy(), _rows(g[f'{key}_length'])[b])
        r = t['ratio'].numpy() if 'ratio' in t else np.zeros(0, np.float32)
        np.testing.assert_allclose(r, _rows(g[f'{key}_ratio'])[b], rtol=1e-6)


def _digest(t, n=16):
    t = t.detach().float().flatten()
    idx = torch.linspace(0, t.numel() - 1, n).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item()], t[idx].numpy()]).astype(np.float32)


def mix_draws(np_seed, bs):
    """what np.random hands the reference's two mixups when seeded with np_seed before the iteration: mixup_data draws a Beta
    weight and shuffles arange(bs) (mixup.py:22-29), mixup_label_unlabel another Beta weight (mixup.py:141)"""
    np.random.seed(np_seed)
    lam1 = float(np.random.beta(1, 1))
    idx = np.asarray(list(range(bs)))
    np.random.shuffle(idx)
    lam2 = float(np.random.beta(1, 1))
    return lam1, idx, lam2


def test_g15_semi_step_with_mixup(golden_dir):
    """the oracle's mean-teacher iteration with mix-up == the reference's engine.semi_train(mix_up_ratio=0.6): what both mixups
    return (labels / boxes / ratios exact, features to rounding), the pseudo labels in between, total loss, gradient norms"""
    g = np.load(os.path.join(golden_dir, 'g15_mixup_steps.npz'))
    torch.set_num_threads(8)
    c = GI.SEMI_MIX
    ns, nw, nu = c['n_strong'], c['n_weak'], c['n_unl']
    model = O.build_oracle_model(10, 20, 6, 3, True, True, True, dropout=0.0)
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), c['seed_w']))
    model.train()
    ema = S.EMA(model, 0.9)
    ema.register()
    gen = torch.Generator().manual_seed(5)
    for n in ema.shadow:
        ema.shadow[n] = ema.shadow[n] + 0.02 * ema.shadow[n].abs().mean() * torch.randn(ema.shadow[n].shape, generator=gen)
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    x_t, x_s, targets = GI.semi_mix_batch()
    trace = {}
    sup, unsup, total, pseudo = S.semi_step(model, ema, build_oracle_criterion(10, 3, True, True), None, x_t, x_s, targets,
                                            classwise_threshold=torch.full((10,), c['thr']), do_step=False, mix_up_ratio=c['ratio'],
                                            mix_draws=mix_draws(c['np_seed'], ns + nw), trace=trace, **masks)
    md = trace['md']
    np.testing.assert_array_equal([md[2].stop, md[3].start, md[3].stop], g['semi_md_split'])
    _check_target_rows(g, 'semi_md', md[1])
    _check_target_rows(g, 'semi_pseudo', trace['pseudo'], exact=False)
    _check_target_rows(g, 'semi_lu', trace['lu'][1], exact=False)
    np.testing.assert_allclose(np.stack([_digest(md[0][i]) for i in range(ns + nw)]), g['semi_md_x_digest'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.stack([_digest(trace['lu'][0][i]) for i in range(nu)]), g['semi_lu_x_digest'], rtol=1e-5, atol=1e-6)
    assert abs(total.item() - float(g['semi_total'])) <= 2e-5 * abs(float(g['semi_total']))
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == list(g['semi_gradnames'])
    gn = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
    np.testing.assert_allclose(gn, g['semi_gradnorm'], rtol=2e-3, atol=1e-6)
    # the fixture exercises what it is meant to: mixed strong clips, a mixed weak clip, abandoned and mixed unlabelled clips
    assert (g['semi_md_ratio'][:ns] >= 0).any() and (g['semi_md_ratio'][ns:] >= 0).any()
    assert (g['semi_lu_ratio'] >= 0).any(1).sum() >= 2 and (g['semi_lu_nlabels'][:5] == g['semi_pseudo_nlabels'][:5]).any()


def test_g15_supervised_step_with_mixup(golden_dir):
    """engine.train(mix_up_ratio=0.6): the mixing moves a weak clip into the strong part (split 5|5 -> 6|4)"""
    g = np.load(os.path.join(golden_dir, 'g15_mixup_steps.npz'))
    torch.set_num_threads(8)
    c = GI.SUP_MIX
    ns, nw = c['n_strong'], c['n_weak']
    model = O.build_oracle_model(10, 20, 3, 3, True, True, True, dropout=0.0)
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), c['seed_w']))
    model.train()
    x, targets = GI.sup_mix_batch()
    trace = {}
    ld, total = S.train_step_mix(model, build_oracle_criterion(10, 3, True, True), None, x, targets, slice(ns), slice(ns, ns + nw),
                                 c['ratio'], mix_draws(c['np_seed'], ns + nw)[:2], do_step=False, trace=trace)
    md = trace['md']
    np.testing.assert_array_equal([md[2].stop, md[3].start, md[3].stop], g['sup_md_split'])
    assert md[2].stop != ns
    _check_target_rows(g, 'sup_md', md[1])
    np.testing.assert_allclose(np.stack([_digest(md[0][i]) for i in range(ns + nw)]), g['sup_md_x_digest'], rtol=1e-5, atol=1e-6)
    assert abs(total.item() - float(g['sup_total'])) <= 2e-5 * abs(float(g['sup_total']))
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert names == list(g['sup_gradnames'])
    gn = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
    np.testing.assert_allclose(gn, g['sup_gradnorm'], rtol=2e-3, atol=1e-6)


def test_g13_transforms(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g13_transforms_mixup.npz'))
    clips = list(g['in_db']) + [g['in_long']]
    for i, clip in enumerate(clips):
        p = g['params'][i]
        x = TO.pad_trunc(clip.astype(np.float32).copy(), 496)
        x = TO.time_mask(x, bool(p[0]), p[1], p[2])
        x = TO.freq_mask(x, bool(p[3]), p[4], p[5], "mean")
        x = TO.freq_shift(x, bool(p[6]), int(p[7]))
        y = TO.normalize(x.astype(np.float32)[None], g['scaler_mean'], g['scaler_std']).astype(np.float32)
        np.testing.assert_allclose(y, g['out'][i], rtol=1e-6, atol=1e-6)
    assert g['params'][:, 0].min() == 0 and g['params'][:, 0].max() == 1          # both branches of the time mask occur


def test_amplitude_to_db_properties():
    """librosa is absent (parity unpinned): check the published definition's invariants"""
    rng = np.random.RandomState(0)
    S = np.abs(rng.randn(64, 100)) + 1e-3
    d = TO.amplitude_to_db(S)
    np.testing.assert_allclose(d.max(), 20 * np.log10(S.max()), rtol=1e-6)
    assert d.min() >= d.max() - 80.0 - 1e-9
    np.testing.assert_allclose(TO.amplitude_to_db(10 * S, top_db=None), TO.amplitude_to_db(S, top_db=None) + 20.0, atol=1e-9)
    assert TO.amplitude_to_db(np.zeros((2, 2)), top_db=None).max() == -100.0     # amin = 1e-5


def test_g13_mixup(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g13_transforms_mixup.npz'))
    gen = torch.Generator().manual_seed(132)
    B = 6
    x = torch.randn(B, 1, 32, 8, generator=gen)
    tg = synthetic_targets(B, 133, 10)
    for t in tg[3:]:
        t['boxes'] = torch.zeros(0, 2)
    xm, ym, ms, mw = S.mixup_data(x, [dict(t) for t in tg], slice(3), slice(3, 6), float(g['mix_lam']), g['mix_index'], mix_up_ratio=0.67)
    np.testing.assert_allclose(xm.numpy(), g['mix_x'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal([ms.stop, mw.start, mw.stop], g['mix_masks'])
    np.testing.assert_array_equal([len(t['labels']) for t in ym], g['mix_nlabels'])
    np.testing.assert_array_equal([len(t['boxes']) for t in ym], g['mix_nboxes'])
    for b, t in enumerate(ym):
        np.testing.assert_array_equal(t['labels'].numpy(), _rows(g['mix_labels'])[b])
        r = t['ratio'].numpy() if 'ratio' in t else np.zeros(0, np.float32)
        np.testing.assert_allclose(r, _rows(g['mix_ratio'])[b], rtol=1e-6)
    x1, x2 = torch.randn(4, 1, 32, 8, generator=gen), torch.randn(4, 1, 32, 8, generator=gen)
    y1, y2 = synthetic_targets(4, 134, 10), synthetic_targets(4, 135, 10)
    xo, yo = S.mixup_label_unlabel(x1, x2, [dict(t) for t in y1], [dict(t) for t in y2], float(g['lu_lam']))
    np.testing.assert_allclose(xo.numpy(), g['lu_x'], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal([len(t['labels']) for t in yo], g['lu_nlabels'])
    for b, t in enumerate(yo):
        np.testing.assert_array_equal(t['labels'].numpy(), _rows(g['lu_labels'])[b])
        r = t['ratio'].numpy() if 'ratio' in t else np.zeros(0, np.float32)
        np.testing.assert_allclose(r, _rows(g['lu_ratio'])[b], rtol=1e-6)


# ------------------------------------------------------------------------------------------------- G14 SP-SEDT query patches
def test_g14_query_patches_oracle_matches_reference_and_pillow():
    """the oracle's restatement of DataLoadDf.get_random_patch + Query.transform_label against the fixture the reference's own
    code produced (boxes bit-exact, resized uint8 images bit-exact, float patches bit-exact), and its restatement of Pillow's
    bilinear resampling against the real Pillow on sizes that shrink, keep and enlarge"""
    from oracle import transforms_oracle as T
    from golden.inputs import QUERY, query_clips
    g = np.load(os.path.join(GOLDEN, 'g14_query_patches.npz'))
    for ci, data in enumerate(query_clips()):
        t = data.shape[1]
        for mode, fixed in (('free', False), ('fixed', True)):
            key = f'c{ci}_{mode}'
            if key + '_boxes' not in g.files:
                assert fixed and t < 128
                continue
            np.random.seed(900 + 10 * ci + int(fixed))
            boxes = T.random_patch_boxes(t, QUERY['num_patches'], fixed_patch_size=fixed)
            assert np.array_equal(np.asarray(boxes, np.float64), g[key + '_boxes'])
            b32 = np.asarray(boxes, np.float32)
            patches, codes, mm = T.query_patches(data.numpy(), b32, fixed)
            assert patches.shape == (QUERY['num_patches'], 1, 128, 64)
            assert np.array_equal(patches[:, 0, ::16, ::8], g[key + '_sample'])
            np.testing.assert_allclose([p.astype(np.float64).sum() for p in patches], g[key + '_sum'], rtol=1e-12)
            if not fixed:
                assert np.array_equal(np.asarray([T.patch_rows(b, t) for b in b32]), g[key + '_rows'])
                assert np.array_equal(codes, g[key + '_code'])
                assert np.array_equal(np.asarray(mm, np.float32), g[key + '_minmax'])
    from PIL import Image
    rng = np.random.RandomState(5)
    for h in (1, 2, 3, 7, 25, 64, 127, 128, 129, 200, 397, 496):
        img = rng.randint(0, 256, (h, 64)).astype(np.uint8)
        ref = np.array(Image.fromarray(img, mode='L').resize((64, 128), Image.BILINEAR))
        assert np.array_equal(T.pil_resize_rows(img, 128), ref), h
