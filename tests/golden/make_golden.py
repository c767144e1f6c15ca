#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the reference never
travels to the GPU box).  The fixtures hold inputs' seeds and the reference's
OUTPUTS only; weights and inputs are regenerated on both sides from seeds with
``oracle.sedt_oracle.seeded_state_dict`` / ``torch.randn(generator=...)``.

torchvision is not installed here, so ``sys.modules`` gets a minimal shim with the
four symbols the reference touches: ``models.resnet50`` (our
restatement of torchvision's published ResNet-50 v1.5 - third-party arithmetic,
"parity unpinned" by the reference), ``models._utils.IntermediateLayerGetter``,
``_is_tracing`` and ``ops.boxes.box_area``.  Everything else executed is the
reference's own code: FrozenBatchNorm2d, Backbone/Joiner, PositionEmbeddingSine,
Transformer (+ torch.nn.MultiheadAttention), SEDT, SPSEDT, HungarianMatcher,
SetCriterion.

usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import argparse
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import sedt_oracle as O              # noqa: E402
from oracle.criterion_oracle import synthetic_targets  # noqa: E402
sys.path.insert(0, HERE)
from inputs import g9_inputs as _g9_inputs, g10_inputs as _g10_inputs, g11_inputs as _g11_inputs, SEMI, semi_batch as _semi_batch, SEMI_MIX, SUP_MIX, semi_mix_batch as _semi_mix_batch, sup_mix_batch as _sup_mix_batch, POOL, pool_batch as _pool_batch, g17_inputs  # noqa: E402


# ----------------------------------------------------------------------------- shim
def install_torchvision_shim():
    tv = types.ModuleType('torchvision')
    models = types.ModuleType('torchvision.models')
    mutils = types.ModuleType('torchvision.models._utils')
    ops = types.ModuleType('torchvision.ops')
    boxes = types.ModuleType('torchvision.ops.boxes')

    class IntermediateLayerGetter(nn.ModuleDict):
        def __init__(self, model, return_layers):
            orig = dict(return_layers)
            remaining = dict(return_layers)
            layers = OrderedDict()
            for name, module in model.named_children():
                layers[name] = module
                remaining.pop(name, None)
                if not remaining:
                    break
            super().__init__(layers)
            self.return_layers = orig

        def forward(self, x):
            out = OrderedDict()
            for name, module in self.items():
                x = module(x)
                if name in self.return_layers:
                    out[self.return_layers[name]] = x
            return out

    def resnet50(replace_stride_with_dilation=None, pretrained=False, norm_layer=None, **kw):
        body = O.ResNet50Body(dilation=bool(replace_stride_with_dilation[2]), norm_layer=norm_layer)
        m = nn.Module()
        for name in ('conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'):
            m.add_module(name, getattr(body, name))
        m.add_module('avgpool', nn.AdaptiveAvgPool2d((1, 1)))
        m.add_module('fc', nn.Linear(2048, 1000))
        return m

    models.resnet50 = resnet50
    mutils.IntermediateLayerGetter = IntermediateLayerGetter
    models._utils = mutils
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ops.boxes = boxes
    tv.models, tv.ops = models, ops
    tv._is_tracing = lambda: False
    for name, mod in (('torchvision', tv), ('torchvision.models', models), ('torchvision.models._utils', mutils),
                      ('torchvision.ops', ops), ('torchvision.ops.boxes', boxes)):
        sys.modules[name] = mod


class _AbsentModule(types.ModuleType):
    """placeholder for third-party modules the reference imports at module level but that this image lacks (librosa,
    soundfile, dcase_util, sed_eval, psds_eval, torchvision.transforms).  It only lets ``import engine`` /
    ``import utilities.BoxTransforms`` succeed; NOTHING computed for a fixture comes from it (ApplyLog, the one transform
    that calls librosa, is never run: see oracle/transforms_oracle.py)."""

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        return type(k, (), {})


def import_reference_engine():
    """reference engine.py / utilities.mixup / utilities.BoxTransforms (after import_reference())"""
    for name in ('librosa', 'soundfile', 'dcase_util', 'dcase_util.data', 'sed_eval', 'psds_eval', 'torchvision.transforms'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _AbsentModule(name)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    import engine as rengine
    import utilities.mixup as rmixup
    import utilities.BoxTransforms as rbt
    rengine.to_cuda_if_available = lambda *a: a[0] if len(a) == 1 else list(a)
    return rengine, rmixup, rbt


def import_reference():
    install_torchvision_shim()
    sys.path.insert(0, REF)
    import utilities.utils as ru
    ru.to_cuda_if_available = lambda *a: a[0] if len(a) == 1 else list(a)
    import sedt as rsedt
    rsedt.to_cuda_if_available = ru.to_cuda_if_available
    torch.Tensor.cuda = lambda self, *a, **k: self          # spsedt.py:37-38 hard-codes .cuda()
    return rsedt, ru


def load_ref_transformer_module():
    spec = importlib.util.spec_from_file_location('ref_transformer', os.path.join(REF, 'sedt', 'transformer.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------- helpers
def ref_args(**over):
    a = argparse.Namespace(
        num_classes=10, lr_backbone=1e-4, backbone='resnet50', dilation=True, position_embedding='sine',
        enc_layers=3, dec_layers=3, dim_feedforward=2048, hidden_dim=256, dropout=0.1, nheads=8, num_queries=10,
        pre_norm=True, aux_loss=True, dec_at=True, pooling=None, self_sup=False, set_cost_class=1, set_cost_bbox=5,
        set_cost_giou=2, epsilon=1, alpha=1, ce_loss_coef=1, bbox_loss_coef=5, giou_loss_coef=2, eos_coef=0.1,
        weak_loss_coef=1, weak_loss_p_coef=1, feature_recon=True, query_shuffle=False, num_patches=10)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def digest(t, n=64):
    t = t.detach().float().flatten()
    idx = torch.linspace(0, t.numel() - 1, n).long()
    return np.concatenate([[t.mean().item(), t.abs().mean().item()], t[idx].numpy()]).astype(np.float32)


def npy(t):
    return t.detach().cpu().numpy()


def seeded_load(model, seed):
    sd = O.seeded_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    return model


def clip_input(b, t, seed):
    return torch.randn(b, 1, t, 64, generator=torch.Generator().manual_seed(seed))


# ----------------------------------------------------------------------------- fixtures
def g1_transformer(out):
    """G1: transformer-only, reference transformer.py loaded by path (pure torch)."""
    rt = load_ref_transformer_module()
    cases = {'pre_e3': dict(E=3, pre=True), 'post_e3': dict(E=3, pre=False), 'pre_e6': dict(E=6, pre=True)}
    res = {}
    for name, c in cases.items():
        torch.manual_seed(0)
        m = rt.Transformer(d_model=256, nhead=8, num_encoder_layers=c['E'], num_decoder_layers=3,
                           dim_feedforward=2048, dropout=0.1, normalize_before=c['pre'],
                           return_intermediate_dec=True, self_sup=False).eval()
        seeded_load(m, 11)
        g = torch.Generator().manual_seed(21)
        src = torch.randn(2, 256, 32, 4, generator=g)
        pos = torch.randn(2, 256, 32, 4, generator=g) * 0.5
        query = torch.randn(11, 256, generator=g)
        mask = torch.zeros(2, 32, 4, dtype=torch.bool)
        mask[1, 25:, :] = True                      # clip 1 has 7 padded time steps (G8-style key padding)
        with torch.no_grad():
            hs, mem = m(src, mask, query, pos)
        res[f'{name}_hs'], res[f'{name}_mem'] = npy(hs), npy(mem)
    # self-sup branch with block-diagonal decoder mask (transformer.py:49-60)
    m = rt.Transformer(256, 8, 3, 3, 2048, 0.1, normalize_before=True, return_intermediate_dec=True,
                       self_sup=True).eval()
    seeded_load(m, 12)
    g = torch.Generator().manual_seed(22)
    src = torch.randn(2, 256, 31, 4, generator=g)
    pos = torch.randn(2, 256, 31, 4, generator=g) * 0.5
    qe = torch.randn(20, 2, 256, generator=g)
    am = torch.ones(20, 20) * float('-inf')
    for i in range(10):
        am[2 * i:2 * i + 2, 2 * i:2 * i + 2] = 0
    with torch.no_grad():
        hs, mem = m(src, torch.zeros(2, 31, 4, dtype=torch.bool), qe, pos, decoder_mask=am)
    res['selfsup_hs'], res['selfsup_mem'] = npy(hs), npy(mem)
    np.savez_compressed(os.path.join(out, 'g1_transformer.npz'), **res)
    print('G1 ok', {k: v.shape for k, v in res.items()})


def g2_g3_sedt(out, rsedt):
    """G2 eval outputs + stage digests; G3 train-mode (dropout=0) losses and grads; G7 AdamW step."""
    res = {}
    for name, kw, T, B in (('urban', dict(enc_layers=3, num_queries=10), 500, 2),
                           ('dcase', dict(enc_layers=6, num_queries=20), 496, 2)):
        args = ref_args(dropout=0.0, **kw)
        model, criterion, _ = rsedt.build_model(args)
        seeded_load(model, 2020)
        x = clip_input(B, T, 7)
        model.eval()
        stages = {}
        body = model.backbone[0].body
        hooks = [body[n].register_forward_hook(lambda m, i, o, n=n: stages.__setitem__(n, o))
                 for n in ('bn1', 'layer1', 'layer2', 'layer3', 'layer4')]
        hooks.append(model.input_proj.register_forward_hook(lambda m, i, o: stages.__setitem__('input_proj', o)))
        for li, l in enumerate(model.transformer.encoder.layers):
            hooks.append(l.register_forward_hook(lambda m, i, o, li=li: stages.__setitem__(f'enc{li}', o)))
        for li, l in enumerate(model.transformer.decoder.layers):
            hooks.append(l.register_forward_hook(lambda m, i, o, li=li: stages.__setitem__(f'dec{li}', o)))
        with torch.no_grad():
            o = model(x)
        for h in hooks:
            h.remove()
        for k in ('pred_logits', 'pred_boxes', 'at'):
            res[f'{name}_eval_{k}'] = npy(o[k])
        for i, a in enumerate(o['aux_outputs']):
            res[f'{name}_eval_aux{i}_logits'], res[f'{name}_eval_aux{i}_boxes'] = npy(a['pred_logits']), npy(a['pred_boxes'])
        for k, v in stages.items():
            res[f'{name}_stage_{k}'] = digest(v)
        # G8: ragged batch -> non-trivial padding mask through mask-resize, pos-enc cumsum, key padding
        xs = [x[0], x[1][:, :T - 140, :]]
        with torch.no_grad():
            o = model(xs)
        for k in ('pred_logits', 'pred_boxes', 'at'):
            res[f'{name}_ragged_{k}'] = npy(o[k])

        # G3: train mode, dropout 0 -> deterministic; full criterion; grads
        model.train()
        targets = synthetic_targets(B, 99, 10)
        o = model(x)
        loss_dict, indices = criterion(o, targets, None, slice(B), False, False)
        wd = criterion.weight_dict
        total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
        model.zero_grad()
        total.backward()
        res[f'{name}_train_total'] = np.float32(total.item())
        for k, v in loss_dict.items():
            res[f'{name}_train_loss_{k}'] = np.float32(v.item())
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        res[f'{name}_train_gradnorm'] = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names],
                                                 dtype=np.float32)
        res[f'{name}_train_gradnames'] = np.array(names)
        res[f'{name}_train_frozen'] = np.array([n for n, p in model.named_parameters() if not p.requires_grad])
        for n in ('backbone.0.body.conv0.weight', 'backbone.0.body.conv0.bias',
                  'backbone.0.body.layer2.0.conv2.weight', 'backbone.0.body.layer4.2.conv3.weight',
                  'transformer.encoder.layers.0.self_attn.in_proj_weight',
                  'transformer.decoder.layers.2.multihead_attn.out_proj.weight', 'query_embed.weight',
                  'input_proj.bias', 'class_embed.weight'):
            res[f'{name}_train_grad::{n}'] = digest(dict(model.named_parameters())[n].grad, 32)
        if name == 'urban':
            # G7: clip 0.1 + one AdamW step with the reference's two param groups (train_sedt.py:234-240,269-270)
            groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
                      {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad],
                       "lr": 1e-4}]
            before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
            opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            opt.step()
            res['urban_step_total_gradnorm'] = np.float32(gn.item())
            res['urban_step_delta'] = np.array(
                [(dict(model.named_parameters())[n].detach() - before[n]).norm().item() for n in names], dtype=np.float32)
    np.savez_compressed(os.path.join(out, 'g2_g3_sedt.npz'), **res)
    print('G2/G3/G7/G8 ok', len(res), 'entries')


def g4_spsedt(out, rsedt):
    args = ref_args(enc_layers=6, num_queries=20, dec_at=False, self_sup=True, lr_backbone=0.0, dropout=0.0)
    model, criterion, _ = rsedt.build_model(args)
    seeded_load(model, 404)
    B, P = 2, 10
    x = clip_input(B, 496, 8)
    patches = torch.randn(B, P, 1, 128, 64, generator=torch.Generator().manual_seed(9))
    mask = torch.zeros(B, 496, 64, dtype=torch.bool)
    res = {}
    model.eval()
    with torch.no_grad():
        o = model((x, mask), patches)
    for k in ('pred_logits', 'pred_boxes', 'pred_feature', 'gt_feature'):
        res[f'eval_{k}'] = npy(o[k]) if k != 'pred_feature' else digest(o[k], 256)
    model.train()
    torch.manual_seed(31)
    qm = (torch.rand(20, B, 1) > 0.1).float()       # what spsedt.py:65 will draw after manual_seed(31)
    torch.manual_seed(31)
    o = model((x, mask), patches)
    res['train_query_mask'] = npy(qm)
    for k in ('pred_logits', 'pred_boxes'):
        res[f'train_{k}'] = npy(o[k])
    res['train_pred_feature'] = digest(o['pred_feature'], 256)
    # SP-SEDT targets: one box per patch, label 0 (DataLoad.py:57-77)
    g = torch.Generator().manual_seed(5)
    targets = []
    for _ in range(B):
        l = torch.rand(P, generator=g) * 0.3 + 0.05
        c = l / 2 + torch.rand(P, generator=g) * (1 - l)
        targets.append({'labels': torch.zeros(P, dtype=torch.int64), 'boxes': torch.stack([c, l], -1)})
    loss_dict, _ = criterion(o, targets, slice(B), slice(B), False, False)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    model.zero_grad()
    total.backward()
    res['train_total'] = np.float32(total.item())
    for k, v in loss_dict.items():
        res[f'train_loss_{k}'] = np.float32(v.item())
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    res['train_gradnames'] = np.array(names)
    res['train_gradnorm'] = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
    res['target_boxes'] = np.stack([npy(t['boxes']) for t in targets])
    np.savez_compressed(os.path.join(out, 'g4_spsedt.npz'), **res)
    print('G4 ok')


def g5_criterion(out, rsedt):
    """G5: matcher indices + criterion losses for fixed outputs/targets (host code pin)."""
    args = ref_args()
    _, criterion, _ = rsedt.build_model(args)
    g = torch.Generator().manual_seed(55)
    B, Q = 6, 10
    outputs = {'pred_logits': torch.randn(B, Q, 11, generator=g), 'pred_boxes': torch.rand(B, Q, 2, generator=g) * 0.8 + 0.1,
               'at': torch.rand(B, 10, generator=g),
               'aux_outputs': [{'pred_logits': torch.randn(B, Q, 11, generator=g),
                                'pred_boxes': torch.rand(B, Q, 2, generator=g) * 0.8 + 0.1} for _ in range(2)]}
    targets = synthetic_targets(B, 56, 10)
    res = {}
    idx, coef = criterion.matcher({k: v for k, v in outputs.items() if k != 'aux_outputs'}, targets)
    res['match_src'] = np.concatenate([npy(i) for i, _ in idx])
    res['match_tgt'] = np.concatenate([npy(j) for _, j in idx])
    res['match_sizes'] = np.array([len(i) for i, _ in idx])
    loss_dict, _ = criterion(outputs, targets, None, slice(B), False, False)
    for k, v in loss_dict.items():
        res[f'loss_{k}'] = np.float32(v.item())
    # weak+strong split (DCASE style: first 4 strong, last 2 weak-only with empty boxes)
    t2 = [dict(t) for t in targets]
    for t in t2[4:]:
        t['boxes'] = torch.zeros(0, 2)
    loss_dict, _ = criterion(outputs, t2, slice(4, 6), slice(4), False, False)
    for k, v in loss_dict.items():
        res[f'ws_loss_{k}'] = np.float32(v.item())
    # normalize=True coefficient path
    loss_dict, _ = criterion(outputs, targets, None, slice(B), False, True)
    res['norm_loss_ce'] = np.float32(loss_dict['loss_ce'].item())
    np.savez_compressed(os.path.join(out, 'g5_criterion.npz'), **res)
    print('G5 ok')


def g6_posenc(out, rsedt):
    import sedt.position_encoding as rpe
    import utilities.utils as ru
    pe = rpe.PositionEmbeddingSine(256, normalize=True)
    res = {}
    for h in (32, 31, 8):
        m = torch.zeros(1, h, 4, dtype=torch.bool)
        p = pe(ru.NestedTensor(torch.zeros(1, 2048, h, 4), m))
        res[f'pos_{h}'] = npy(p[0, :, :, 0].t())            # (H, 256)
    m = torch.zeros(1, 32, 4, dtype=torch.bool)
    m[0, 23:, :] = True
    p = pe(ru.NestedTensor(torch.zeros(1, 2048, 32, 4), m))
    res['pos_32_pad23'] = npy(p[0, :, :, 0].t())
    np.savez_compressed(os.path.join(out, 'g6_posenc.npz'), **res)
    print('G6 ok')


class recorded_rand(object):
    """context: torch.rand draws from a seeded generator and every draw is recorded (matcher.py:116 calls torch.rand)"""

    def __init__(self, seed):
        self.g, self.draws = torch.Generator().manual_seed(seed), []

    def __enter__(self):
        self.orig = torch.rand

        def fake(*a, **k):
            r = self.orig(*a, generator=self.g, **k)
            self.draws.append(r.clone())
            return r
        torch.rand = fake
        return self

    def __exit__(self, *e):
        torch.rand = self.orig
        return False


def _pad_rows(rows, width, fill):
    out = np.full((len(rows), width), fill, dtype=np.float32)
    for i, r in enumerate(rows):
        r = np.asarray(r, dtype=np.float32).reshape(-1)
        out[i, :len(r)] = r
    return out


def g9_criterion_variants(out, rsedt):
    """G9: fine_tune re-matching (two epsilons, +/- normalize) and the focal-loss branches, on fixed outputs/targets."""
    _, criterion, _ = rsedt.build_model(ref_args())
    outputs, targets, B, Q = _g9_inputs()
    res = {}
    cases = {'ft': dict(fine_tune=True, normalize=False, fl=False, eps=1.0),
             'ft_eps3': dict(fine_tune=True, normalize=False, fl=False, eps=3.0),
             'ft_norm_eps3': dict(fine_tune=True, normalize=True, fl=False, eps=3.0),
             'fl': dict(fine_tune=False, normalize=False, fl=True, eps=1.0),
             'fl_ft_eps3': dict(fine_tune=True, normalize=False, fl=True, eps=3.0)}
    for name, c in cases.items():
        criterion.matcher.epsilon = c['eps']
        with recorded_rand(900 + len(name)) as rr:
            loss_dict, indices = criterion(outputs, targets, None, slice(B), c['fine_tune'], c['normalize'], c['fl'])
        for k, v in loss_dict.items():
            res[f'{name}_loss_{k}'] = np.float32(v.item())
        res[f'{name}_src'] = _pad_rows([npy(i) for i, _ in indices], 2 * Q, -1)
        res[f'{name}_tgt'] = _pad_rows([npy(j) for _, j in indices], 2 * Q, -1)
        res[f'{name}_rand'] = _pad_rows([npy(r) for r in rr.draws], Q, -1) if rr.draws else np.zeros((0, Q), np.float32)
    # weak+strong split with focal loss (loss_weak -> weak_focal_loss)
    t2 = [dict(t) for t in targets]
    for t in t2[4:]:
        t['boxes'] = torch.zeros(0, 2)
    criterion.matcher.epsilon = 1.0
    loss_dict, _ = criterion(outputs, t2, slice(4, 6), slice(4), False, False, True)
    for k, v in loss_dict.items():
        res[f'fl_ws_loss_{k}'] = np.float32(v.item())
    # mixup-style targets with 'ratio' (positional coefficients, matcher.py:130-131)
    g = torch.Generator().manual_seed(93)
    t3 = [dict(t) for t in targets]
    for t in t3[:3]:
        t['ratio'] = torch.rand(len(t['labels']), generator=g) * 0.8 + 0.1
    loss_dict, _ = criterion(outputs, t3, None, slice(B), False, False, False)
    for k, v in loss_dict.items():
        res[f'ratio_loss_{k}'] = np.float32(v.item())
    res['ratio_values'] = _pad_rows([npy(t['ratio']) if 'ratio' in t else [] for t in t3], 9, -1)
    np.savez_compressed(os.path.join(out, 'g9_criterion_variants.npz'), **res)
    print('G9 ok', len(res))


def g10_postprocess(out, rsedt):
    """G10: PostProcess for every fusion mode, with and without tags, seconds and is_semi boxes."""
    pp = rsedt.PostProcess()
    outputs, tags, sizes = _g10_inputs()
    res = {}
    for name, kw in (('none', dict(audio_tags=None)), ('m1', dict(audio_tags=tags, at_m=1)), ('m2', dict(audio_tags=tags, at_m=2)),
                     ('m3', dict(audio_tags=tags, at_m=3)), ('m2_t03', dict(audio_tags=tags, at_m=2, threshold=0.3)),
                     ('semi', dict(audio_tags=tags, at_m=1, is_semi=True, threshold=None))):
        r = pp({k: v.clone() for k, v in outputs.items()}, sizes, **kw)
        res[f'{name}_scores'] = np.stack([npy(x['scores']) for x in r])
        res[f'{name}_labels'] = np.stack([npy(x['labels']) for x in r])
        res[f'{name}_boxes'] = np.stack([npy(x['boxes']) for x in r])
    np.savez_compressed(os.path.join(out, 'g10_postprocess.npz'), **res)
    print('G10 ok')


def g11_pseudo_labels(out, rsedt, rengine):
    """G11: engine.get_pseudo_labels on fixed teacher outputs (class-wise thresholds, min length, same-class overlap removal)."""
    from collections import Counter
    tea, thr, B = _g11_inputs()
    sizes = torch.full((B,), 10.0)
    res = {}
    for name, kw in (('nms', dict()), ('raw', dict(del_overlap=False))):
        targets = [{'labels': torch.zeros(0, dtype=torch.int64), 'boxes': torch.zeros(0, 2), 'orig_size': torch.tensor(10.0)}
                   for _ in range(B)]
        cnt = Counter()
        got = rengine.get_pseudo_labels({k: v.clone() for k, v in tea.items()}, {'bbox': rsedt.PostProcess()}, sizes, targets, cnt,
                                        classwise_threshold=thr, **kw)
        res[f'{name}_count'] = np.array([len(t['labels']) for t in got])
        res[f'{name}_labels'] = _pad_rows([npy(t['labels']) for t in got], 20, -1)
        res[f'{name}_centre'] = _pad_rows([npy(t['boxes'][:, 0]) for t in got], 20, -1)
        res[f'{name}_length'] = _pad_rows([npy(t['boxes'][:, 1]) for t in got], 20, -1)
        res[f'{name}_counter'] = np.array([cnt.get(c, 0) for c in range(10)])
    np.savez_compressed(os.path.join(out, 'g11_pseudo_labels.npz'), **res)
    print('G11 ok', res['nms_count'], res['raw_count'])


def g12_semi_step(out, rsedt, rengine, ru):
    """G12: one iteration of the reference's engine.semi_train (mean teacher, E=6, Q=20+1, mixup off): once with the
    optimizer step withheld (losses + every gradient norm), once complete (parameter and EMA-shadow deltas)."""
    import utilities.utils as rutils
    c = SEMI
    res = {}
    ns, nw, nu = c['n_strong'], c['n_weak'], c['n_unl']
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    for mode in ('grads', 'step'):
        args = ref_args(enc_layers=6, num_queries=20, dropout=0.0)
        model, criterion, post = rsedt.build_model(args)
        seeded_load(model, c['seed_w'])
        model.train()
        ema = rutils.EMA(model, 0.9)
        ema.register()
        g = torch.Generator().manual_seed(5)
        for n in ema.shadow:                                  # a teacher that differs from the student
            ema.shadow[n] = ema.shadow[n] + 0.02 * ema.shadow[n].abs().mean() * torch.randn(ema.shadow[n].shape, generator=g)
        shadow0 = {n: v.clone() for n, v in ema.shadow.items()}
        groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
                  {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-4}]
        opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)
        x_t, x_s, targets = _semi_batch()
        nt = lambda x: ru.NestedTensor(x, torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.bool))
        loader = [((nt(x_t), nt(x_s)), targets)]
        before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
        thr = torch.full((10,), c['thr'])
        value, counter = rengine.semi_train(loader, model, ema, criterion, opt, 0, 2 if mode == 'grads' else 1, 1, post,
                                            max_norm=0.1, classwise_threshold=thr, **masks)
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        if mode == 'grads':
            res['total'] = np.float32(value)
            res['gradnames'] = np.array(names)
            res['gradnorm'] = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
            res['pseudo_counter'] = np.array([counter.get(k, 0) for k in range(10)])
            res['pseudo_count'] = np.array([len(t['labels']) for t in targets[masks['mask_unlabel']]])
            res['pseudo_labels'] = _pad_rows([npy(t['labels']) for t in targets[masks['mask_unlabel']]], 20, -1)
            res['pseudo_centre'] = _pad_rows([npy(t['boxes'][:, 0]) for t in targets[masks['mask_unlabel']]], 20, -1)
            res['pseudo_length'] = _pad_rows([npy(t['boxes'][:, 1]) for t in targets[masks['mask_unlabel']]], 20, -1)
        else:
            res['step_total'] = np.float32(value)
            res['step_delta'] = np.array([(dict(model.named_parameters())[n].detach() - before[n]).norm().item() for n in names],
                                         np.float32)
            # semi_train updated the shadow AFTER the optimizer step: shadow1 = 0.9 shadow0 + 0.1 p_new
            res['ema_delta'] = np.array([(ema.shadow[n] - shadow0[n]).norm().item() for n in names], np.float32)
    np.savez_compressed(os.path.join(out, 'g12_semi_step.npz'), **res)
    print('G12 ok total', res['total'], 'pseudo', res['pseudo_count'])


class _HostPrefetcher(object):
    """engine.train pulls its batches through DataLoad.data_prefetcher, which opens a CUDA stream; on the CPU the same
    iteration protocol (next() -> (input, target), then (None, None)) without the stream"""

    def __init__(self, loader):
        self.it = iter(loader)

    def next(self):
        try:
            return next(self.it)
        except StopIteration:
            return None, None


def _target_rows(res, key, targets):
    res[f'{key}_nlabels'] = np.array([len(t['labels']) for t in targets])
    res[f'{key}_nboxes'] = np.array([len(t['boxes']) for t in targets])
    res[f'{key}_labels'] = _pad_rows([npy(t['labels']) for t in targets], 24, -1)
    res[f'{key}_centre'] = _pad_rows([npy(t['boxes'].reshape(-1, 2)[:, 0]) for t in targets], 24, -1)
    res[f'{key}_length'] = _pad_rows([npy(t['boxes'].reshape(-1, 2)[:, 1]) for t in targets], 24, -1)
    res[f'{key}_ratio'] = _pad_rows([npy(t['ratio']) if 'ratio' in t else [] for t in targets], 24, -1)


def g15_mixup_steps(out, rsedt, rengine, rmixup, ru):
    """G15: (a) one iteration of the reference's engine.semi_train with mix_up_ratio = 0.6 (train_ss_sedt.py's recipe): np.random
    seeded, so mixup_data's Beta draw + shuffle and mixup_label_unlabel's Beta draw are reproducible; recorded: what the two
    mixups returned (the calls are observed, not replaced), the total loss, every gradient norm, and - second run - the
    parameter / EMA deltas of the complete iteration.  (b) one iteration of engine.train with mix_up_ratio = 0.6 on a 5 strong +
    5 weak batch where the mixing moves a clip across the strong / weak boundary."""
    import utilities.utils as rutils
    res = {}
    # ------------------------------------------------------------------ (a) mean teacher
    c = SEMI_MIX
    ns, nw, nu = c['n_strong'], c['n_weak'], c['n_unl']
    masks = dict(mask_strong=slice(ns), mask_weak=slice(ns, ns + nw), mask_label=slice(ns + nw), mask_unlabel=slice(ns + nw, ns + nw + nu))
    seen = {}
    real_md, real_lu = rmixup.mixup_data, rmixup.mixup_label_unlabel

    def spy_md(x, y, *a, **k):
        r = real_md(x, y, *a, **k)
        seen['md'] = (r[0].tensors.clone(), [dict(t) for t in r[1]], r[2], r[3])
        return r

    def spy_lu(x1, x2, y1, y2, *a, **k):
        seen['pseudo'] = [dict(t) for t in y2]
        r = real_lu(x1, x2, y1, y2, *a, **k)
        seen['lu'] = (r[0].tensors.clone(), [dict(t) for t in r[1]])
        return r

    rengine.mixup_data, rengine.mixup_label_unlabel = spy_md, spy_lu
    try:
        for mode in ('grads', 'step'):
            args = ref_args(enc_layers=6, num_queries=20, dropout=0.0)
            model, criterion, post = rsedt.build_model(args)
            seeded_load(model, c['seed_w'])
            model.train()
            ema = rutils.EMA(model, 0.9)
            ema.register()
            g = torch.Generator().manual_seed(5)
            for n in ema.shadow:
                ema.shadow[n] = ema.shadow[n] + 0.02 * ema.shadow[n].abs().mean() * torch.randn(ema.shadow[n].shape, generator=g)
            shadow0 = {n: v.clone() for n, v in ema.shadow.items()}
            groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
                      {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-4}]
            opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)
            x_t, x_s, targets = _semi_mix_batch()
            nt = lambda x: ru.NestedTensor(x, torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.bool))
            loader = [((nt(x_t), nt(x_s)), targets)]
            before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
            thr = torch.full((10,), c['thr'])
            np.random.seed(c['np_seed'])
            value, counter = rengine.semi_train(loader, model, ema, criterion, opt, 0, 2 if mode == 'grads' else 1, 1, post,
                                                max_norm=0.1, classwise_threshold=thr, mix_up_ratio=c['ratio'], **masks)
            names = [n for n, p in model.named_parameters() if p.requires_grad]
            if mode == 'grads':
                res['semi_total'] = np.float32(value)
                res['semi_gradnames'] = np.array(names)
                res['semi_gradnorm'] = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
                res['semi_counter'] = np.array([counter.get(k, 0) for k in range(10)])
                res['semi_md_split'] = np.array([seen['md'][2].stop, seen['md'][3].start, seen['md'][3].stop])
                res['semi_md_x_digest'] = np.stack([digest(seen['md'][0][i], 16) for i in range(ns + nw)])
                _target_rows(res, 'semi_md', seen['md'][1])
                _target_rows(res, 'semi_pseudo', seen['pseudo'])
                res['semi_lu_x_digest'] = np.stack([digest(seen['lu'][0][i], 16) for i in range(nu)])
                _target_rows(res, 'semi_lu', seen['lu'][1])
            else:
                res['semi_step_total'] = np.float32(value)
                res['semi_step_delta'] = np.array([(dict(model.named_parameters())[n].detach() - before[n]).norm().item() for n in names],
                                                  np.float32)
                res['semi_ema_delta'] = np.array([(ema.shadow[n] - shadow0[n]).norm().item() for n in names], np.float32)
        # ------------------------------------------------------------------ (b) supervised step with mix-up (engine.train)
        c = SUP_MIX
        ns, nw = c['n_strong'], c['n_weak']
        rengine.data_prefetcher = _HostPrefetcher
        args = ref_args(enc_layers=3, num_queries=20, dropout=0.0)
        model, criterion, post = rsedt.build_model(args)
        seeded_load(model, c['seed_w'])
        model.train()
        groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
                  {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-4}]
        opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)
        x, targets = _sup_mix_batch()
        loader = [(ru.NestedTensor(x, torch.zeros(x.shape[0], x.shape[2], x.shape[3], dtype=torch.bool)), targets)]
        np.random.seed(c['np_seed'])
        value = rengine.train(loader, model, criterion, opt, 0, 2, mask_weak=slice(ns, ns + nw), mask_strong=slice(ns), max_norm=0.1,
                              mix_up_ratio=c['ratio'])
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        res['sup_total'] = np.float32(value)
        res['sup_gradnames'] = np.array(names)
        res['sup_gradnorm'] = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names], np.float32)
        res['sup_md_split'] = np.array([seen['md'][2].stop, seen['md'][3].start, seen['md'][3].stop])
        res['sup_md_x_digest'] = np.stack([digest(seen['md'][0][i], 16) for i in range(ns + nw)])
        _target_rows(res, 'sup_md', seen['md'][1])
    finally:
        rengine.mixup_data, rengine.mixup_label_unlabel = real_md, real_lu
    np.savez_compressed(os.path.join(out, 'g15_mixup_steps.npz'), **res)
    print('G15 ok semi total', res['semi_total'], 'split', res['semi_md_split'], 'pseudo', res['semi_pseudo_nlabels'], 'mixed unl',
          res['semi_lu_nlabels'], '| sup total', res['sup_total'], 'split', res['sup_md_split'])


def g13_transforms_mixup(out, rmixup, rbt):
    """G13: the reference's own transform classes (PadOrTrunc, TimeMask, FreqMask(mean), FreqShift, Normalize - NOT ApplyLog,
    which is librosa) with np.random seeded, the parameters they drew, and mixup_data / mixup_label_unlabel with their
    Beta / shuffle draws."""
    import utilities.Scaler as rscaler
    res = {}
    rng = np.random.RandomState(131)
    db = (rng.randn(3, 470, 64) * 12 - 40).astype(np.float64)          # log-mel-like clips, shorter than 496 frames
    db[2] = (rng.randn(520, 64) * 12 - 40)[:470]
    long_clip = (rng.randn(520, 64) * 12 - 40)
    sc = rscaler.Scaler()
    mean = rng.randn(64) * 3 - 40
    sc.load_state_dict({'mean_': mean.tolist(), 'mean_of_square_': (mean ** 2 + rng.rand(64) * 100 + 60).tolist()})
    res['scaler_mean'], res['scaler_std'] = sc.mean_.astype(np.float64), sc.std_.astype(np.float64)
    res['in_db'] = db.astype(np.float32)
    res['in_long'] = long_clip.astype(np.float32)
    outs, params = [], []
    for i, clip in enumerate(list(db) + [long_clip]):
        np.random.seed(1000 + i)
        x = rbt.pad_trunc_seq(clip.astype(np.float32).copy(), 496)
        tm, fm, fs = rbt.TimeMask(p=0.6), rbt.FreqMask(fill_mode="mean", p=0.6), rbt.FreqShift(p=0.6)
        x = tm.transform_data(x)
        x = fm.transform_data(x)
        x = fs.transform_data(x)
        t = rbt.ToTensor(unsqueeze_axis=0).transform_data(x)
        y = rbt.Normalize(sc).transform_data(t)
        outs.append(npy(y))
        params.append([float(tm.parameters['apply']), tm.parameters['t'], tm.parameters['t0'], float(fm.parameters['apply']),
                       fm.parameters['f'], fm.parameters['f0'], float(fs.parameters['apply']), float(fs.parameters['shift_size'])])
    res['out'] = np.stack(outs)
    res['params'] = np.asarray(params, np.float64)
    # ---- mixup_data (labelled batch: 3 strong + 3 weak) and mixup_label_unlabel
    import utilities.utils as rutils
    g = torch.Generator().manual_seed(132)
    B = 6
    x = torch.randn(B, 1, 32, 8, generator=g)
    tg = synthetic_targets(B, 133, 10)
    for t in tg[3:]:
        t['boxes'] = torch.zeros(0, 2)
    np.random.seed(77)
    lam = np.random.beta(1, 1)
    idx = np.asarray(list(range(B)))
    np.random.shuffle(idx)
    np.random.seed(77)
    nt = rutils.NestedTensor(x.clone(), torch.zeros(B, 32, 8, dtype=torch.bool))
    xm, ym, ms, mw = rmixup.mixup_data(nt, [dict(t) for t in tg], slice(3), slice(3, 6), mix_up_ratio=0.67, alpha=1)
    res['mix_lam'], res['mix_index'] = np.float64(lam), idx
    res['mix_x'] = npy(xm.tensors)
    res['mix_masks'] = np.array([ms.stop, mw.start, mw.stop])
    res['mix_nlabels'] = np.array([len(t['labels']) for t in ym])
    res['mix_nboxes'] = np.array([len(t['boxes']) for t in ym])
    res['mix_labels'] = _pad_rows([npy(t['labels']) for t in ym], 20, -1)
    res['mix_ratio'] = _pad_rows([npy(t['ratio']) if 'ratio' in t else [] for t in ym], 20, -1)
    res['mix_centre'] = _pad_rows([npy(t['boxes'].reshape(-1, 2)[:, 0]) for t in ym], 20, -1)
    # label/unlabel mixing
    x1, x2 = torch.randn(4, 1, 32, 8, generator=g), torch.randn(4, 1, 32, 8, generator=g)
    y1, y2 = synthetic_targets(4, 134, 10), synthetic_targets(4, 135, 10)
    np.random.seed(78)
    lam2 = np.random.beta(1, 1)
    np.random.seed(78)
    n1 = rutils.NestedTensor(x1.clone(), None)
    n2 = rutils.NestedTensor(x2.clone(), None)
    xo, yo = rmixup.mixup_label_unlabel(n1, n2, [dict(t) for t in y1], [dict(t) for t in y2], alpha=1)
    res['lu_lam'] = np.float64(lam2)
    res['lu_x'] = npy(xo.tensors)
    res['lu_nlabels'] = np.array([len(t['labels']) for t in yo])
    res['lu_labels'] = _pad_rows([npy(t['labels']) for t in yo], 20, -1)
    res['lu_ratio'] = _pad_rows([npy(t['ratio']) if 'ratio' in t else [] for t in yo], 20, -1)
    np.savez_compressed(os.path.join(out, 'g13_transforms_mixup.npz'), **res)
    print('G13 ok', res['params'][:, [0, 3, 6]].tolist(), res['mix_nlabels'], res['lu_nlabels'])


def install_pil_transforms_shim(rbt):
    """torchvision.transforms is absent from this image; Query (BoxTransforms.py:315-330) composes ToPILImage -> Resize((128, 64))
    -> ToTensor.  These stand-ins restate torchvision's documented behaviour of exactly those three calls on a (1, H, W) float
    tensor over the REAL Pillow (``pic.mul(255).byte()`` -> mode 'L' image; ``img.resize(size[::-1], BILINEAR)``; uint8 -> float
    / 255): the resampling arithmetic in the fixture is Pillow's own, the three wrapper lines are the unpinned part."""
    from PIL import Image

    class Compose(object):
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage(object):
        def __call__(self, pic):
            assert pic.dim() == 3 and pic.shape[0] == 1
            return Image.fromarray(pic.mul(255).byte().numpy()[0], mode='L')

    class Resize(object):
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            return img.resize(tuple(self.size[::-1]), Image.BILINEAR)

    class ToTensor(object):
        def __call__(self, img):
            a = np.array(img, np.uint8, copy=True)
            return torch.from_numpy(a).view(a.shape[0], a.shape[1], 1).permute(2, 0, 1).contiguous().float().div(255)

    tv = types.ModuleType('torchvision.transforms')
    tv.Compose, tv.ToPILImage, tv.Resize, tv.ToTensor = Compose, ToPILImage, Resize, ToTensor
    rbt.transforms = tv


def g14_query_patches(out, rbt):
    """G14: the reference's own SP-SEDT patch pipeline - DataLoadDf.get_random_patch (DataLoad.py:57-77) draws the boxes,
    Query.transform_label (BoxTransforms.py:332-360) crops / min-max normalises / resizes through Pillow / de-normalises.
    Stored per patch: the crop rows, (min, max), the resized uint8 image (the float patch is code / 255 * (max - min) + min,
    re-formed by the tests) and an f32 digest of the reference's float output."""
    import data_utils.DataLoad as rdl
    from inputs import QUERY, query_clips
    install_pil_transforms_shim(rbt)
    res = {}
    for ci, data in enumerate(query_clips()):
        t = data.shape[1]
        for mode, fixed in (('free', False), ('fixed', True)):
            if fixed and t < 128:
                continue
            fake = types.SimpleNamespace(fixed_patch_size=fixed, num_patches=QUERY['num_patches'], mu=0.2, sigma=0.26)
            np.random.seed(900 + 10 * ci + int(fixed))
            boxes = rdl.DataLoadDf.get_random_patch(fake, np.zeros((t, 64), np.float32))
            label = {'patches': None, 'boxes': torch.tensor(boxes, dtype=torch.float32)}
            _, lab = rbt.Query(fixed).transform_label((data.clone(), label))
            patches = lab['patches']
            assert patches.shape == (QUERY['num_patches'], 1, 128, 64), patches.shape
            key = f'c{ci}_{mode}'
            res[key + '_boxes'] = np.asarray(boxes, np.float64)
            res[key + '_sum'] = np.asarray([float(p.double().sum()) for p in patches])
            res[key + '_sample'] = npy(patches[:, 0, ::16, ::8])
            if fixed:
                continue
            rows, mm, codes = [], [], []
            for b, p in zip(label['boxes'], patches):
                c, l = b.numpy()
                s, e = c - l / 2, c + l / 2
                s_idx, e_idx = int(s * t), int(e * t)
                if s_idx >= e_idx:
                    s_idx, e_idx = max(0, s_idx - 1), min(t, e_idx + 1)
                crop = data[:, s_idx:e_idx, :]
                mn, mx = crop.min(), crop.max()
                code = torch.round((p[0] - mn) / (mx - mn) * 255).to(torch.uint8)
                assert torch.equal(code.float().div(255) * (mx - mn) + mn, p[0]), 'code does not reproduce the reference patch'
                rows.append([s_idx, e_idx])
                mm.append([float(mn), float(mx)])
                codes.append(code.numpy())
            res[key + '_rows'], res[key + '_minmax'], res[key + '_code'] = np.asarray(rows), np.asarray(mm, np.float32), np.stack(codes)
    np.savez_compressed(os.path.join(out, 'g14_query_patches.npz'), **res)
    print('G14 ok', {k: v.shape for k, v in res.items() if k.endswith('_rows')}, res['c2_free_rows'][:, 1] - res['c2_free_rows'][:, 0])


G16_GRADS = ('class_embed.weight', 'class_embed.bias', 'bbox_embed.layers.2.weight', 'weak_class_embed.weight',
             'transformer.decoder.layers.2.linear2.weight', 'attn_dense_softmax.weight', 'attn_dense_softmax.bias')


def g16_pooling(out, rsedt):
    """G16: the reference's SEDT with --pooling max / avg / attn / weighted_sum (dec_at model): the pooled clip-level output
    ``at_p`` in eval mode, and a train-mode (dropout 0) loss + backward through SetCriterion with a strong | weak split
    (loss_weak_p, sedt.py:182-185).  'max_nomask': weak_mask None on an all-strong batch (gt[None] / at_p[None])."""
    c = POOL
    ns, B = c['n_strong'], c['n_strong'] + c['n_weak']
    x, targets = _pool_batch()
    res = {}
    for i, mode in enumerate(c['modes'] + ('max_nomask',)):
        pooling = mode.split('_nomask')[0]
        args = ref_args(dropout=0.0, enc_layers=3, num_queries=10, pooling=pooling, weak_loss_p_coef=0.7)
        model, criterion, _ = rsedt.build_model(args)
        seeded_load(model, c['seed_w'] + (0 if mode == 'max_nomask' else i))
        nomask = mode.endswith('_nomask')
        xs, ts = (x[:ns], targets[:ns]) if nomask else (x, targets)
        model.eval()
        with torch.no_grad():
            o = model(xs)
        res[f'{mode}_eval_at_p'] = npy(o['at_p'])
        res[f'{mode}_eval_at'] = npy(o['at'])
        model.train()
        o = model(xs)
        loss_dict, _ = criterion(o, ts, None if nomask else slice(ns, B), slice(ns), False, False)
        wd = criterion.weight_dict
        total = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
        model.zero_grad()
        total.backward()
        res[f'{mode}_train_at_p'] = npy(o['at_p'])
        res[f'{mode}_train_total'] = np.float32(total.item())
        for k, v in loss_dict.items():
            res[f'{mode}_train_loss_{k}'] = np.float32(v.item())
        params = dict(model.named_parameters())
        names = [n for n, p in params.items() if p.requires_grad]
        res[f'{mode}_train_gradnames'] = np.array(names)
        res[f'{mode}_train_gradnorm'] = np.array([0.0 if params[n].grad is None else params[n].grad.norm().item() for n in names],
                                                 np.float32)
        for n in G16_GRADS:
            if n in params and params[n].grad is not None:
                res[f'{mode}_train_grad::{n}'] = digest(params[n].grad, 32)
    np.savez_compressed(os.path.join(out, 'g16_pooling.npz'), **res)
    print('G16 ok', {m: (float(res[f'{m}_train_loss_loss_weak_p']), float(res[f'{m}_train_total'])) for m in c['modes'] + ('max_nomask',)})


def g17_activation_postnorm(out):
    """G17: the reference transformer (transformer.py loaded by path) with activation='gelu' (transformer.py:423-431), pre- and
    post-norm, and the post-norm ReLU stack: forward (eval) AND every parameter's gradient under a linear loss in train mode with
    dropout 0 - pins the backward of forward_post (:177-190, :240-261), which G1 covers in the forward only."""
    rt = load_ref_transformer_module()
    cases = {'gelu_pre': dict(act='gelu', pre=True), 'gelu_post': dict(act='gelu', pre=False), 'relu_post': dict(act='relu', pre=False)}
    res = {}
    src, pos, query, mask, w_hs, w_mem = g17_inputs()
    for name, c in cases.items():
        torch.manual_seed(0)
        m = rt.Transformer(d_model=256, nhead=8, num_encoder_layers=3, num_decoder_layers=3, dim_feedforward=2048, dropout=0.0,
                           activation=c['act'], normalize_before=c['pre'], return_intermediate_dec=True, self_sup=False)
        seeded_load(m, 17)
        m.eval()
        with torch.no_grad():
            hs, mem = m(src, mask, query, pos)
        res[f'{name}_hs'], res[f'{name}_mem'] = npy(hs), npy(mem)
        m.train()
        m.zero_grad()
        s_, q_ = src.clone().requires_grad_(True), query.clone().requires_grad_(True)
        hs, mem = m(s_, mask, q_, pos)
        loss = (hs * w_hs).sum() + (mem * w_mem).sum()
        loss.backward()
        res[f'{name}_loss'] = np.float32(loss.item())
        names = [n for n, p_ in m.named_parameters()]
        res[f'{name}_gradnames'] = np.array(names)
        res[f'{name}_gradnorm'] = np.array([p_.grad.norm().item() if p_.grad is not None else 0.0 for _, p_ in m.named_parameters()], np.float32)
        for n, p_ in m.named_parameters():
            if p_.grad is not None:
                res[f'{name}_grad_{n}'] = digest(p_.grad, 32)
        res[f'{name}_dsrc'], res[f'{name}_dquery'] = digest(s_.grad, 256), npy(q_.grad)
    np.savez_compressed(os.path.join(out, 'g17_activation_postnorm.npz'), **res)
    print('G17 ok', len(res))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='', help='comma list of fixtures to (re)generate, e.g. g9,g12 (default: all)')
    ap.add_argument('--out', default=HERE)
    cli = ap.parse_args()
    want = set(w for w in cli.only.split(',') if w)
    on = lambda k: not want or k in want
    torch.set_num_threads(8)
    out = cli.out
    if on('g1'):
        g1_transformer(out)
    if on('g17'):
        g17_activation_postnorm(out)
    rsedt, ru = import_reference()
    if on('g6'):
        g6_posenc(out, rsedt)
    if on('g5'):
        g5_criterion(out, rsedt)
    if on('g2'):
        g2_g3_sedt(out, rsedt)
    if on('g4'):
        g4_spsedt(out, rsedt)
    if on('g9'):
        g9_criterion_variants(out, rsedt)
    if on('g10'):
        g10_postprocess(out, rsedt)
    if on('g16'):
        g16_pooling(out, rsedt)
    if want & {'g11', 'g12', 'g13', 'g14', 'g15'} or not want:
        rengine, rmixup, rbt = import_reference_engine()
        if on('g14'):
            g14_query_patches(out, rbt)
        if on('g11'):
            g11_pseudo_labels(out, rsedt, rengine)
        if on('g13'):
            g13_transforms_mixup(out, rmixup, rbt)
        if on('g12'):
            g12_semi_step(out, rsedt, rengine, ru)
        if on('g15'):
            g15_mixup_steps(out, rsedt, rengine, rmixup, ru)
