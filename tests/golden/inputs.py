"""Seeded INPUTS of the fixtures G9-G12, shared by make_golden.py (which feeds them to the reference) and by the tests
(which feed them to the oracle / the HIP path).  Data only: no reference code, no expected values."""
import torch

from oracle.criterion_oracle import synthetic_targets


def clip_input(b, t, seed):
    return torch.randn(b, 1, t, 64, generator=torch.Generator().manual_seed(seed))


def g9_inputs():
    g = torch.Generator().manual_seed(91)
    B, Q = 6, 10
    outputs = {'pred_logits': torch.randn(B, Q, 11, generator=g) * 2, 'pred_boxes': torch.rand(B, Q, 2, generator=g) * 0.6 + 0.2,
               'at': torch.rand(B, 10, generator=g),
               'aux_outputs': [{'pred_logits': torch.randn(B, Q, 11, generator=g) * 2,
                                'pred_boxes': torch.rand(B, Q, 2, generator=g) * 0.6 + 0.2} for _ in range(2)]}
    return outputs, synthetic_targets(B, 92, 10), B, Q


def g10_inputs():
    g = torch.Generator().manual_seed(101)
    B, Q, C = 5, 10, 10
    outputs = {'pred_logits': torch.randn(B, Q, C + 1, generator=g) * 2.5, 'pred_boxes': torch.rand(B, Q, 2, generator=g) * 0.5 + 0.1}
    tags = (torch.rand(B, C, generator=g) > 0.5).long()
    sizes = torch.tensor([10.0, 10.0, 7.5, 10.0, 4.0])
    return outputs, tags, sizes


def g11_inputs():
    g = torch.Generator().manual_seed(111)
    B, Q, C = 8, 20, 10
    logits = torch.randn(B, Q, C + 1, generator=g) * 3
    logits[..., -1] -= 1.0
    tea = {'pred_logits': logits, 'pred_boxes': torch.stack([torch.rand(B, Q, generator=g) * 0.7 + 0.15,
                                                             torch.rand(B, Q, generator=g) * 0.4], -1),
           'at': torch.rand(B, C, generator=g)}
    tea['pred_boxes'][0, :5, 1] = 0.01                   # shorter than 0.2 / 10 s: dropped
    thr = torch.rand(C, generator=g) * 0.25 + 0.3
    return tea, thr, B


SEMI = dict(n_strong=2, n_weak=2, n_unl=4, T=496, thr=0.115, seed_w=2021, seed_x=71, seed_t=72)


def semi_batch():
    c = SEMI
    B = c['n_strong'] + c['n_weak'] + c['n_unl']
    x_t = clip_input(B, c['T'], c['seed_x'])
    x_s = x_t.clone()
    x_s[c['n_strong'] + c['n_weak']:] += 0.1 * clip_input(c['n_unl'], c['T'], c['seed_x'] + 1)    # student view of the unlabelled clips
    targets = synthetic_targets(B, c['seed_t'], 10)
    for t in targets[c['n_strong']:]:
        t['boxes'] = torch.zeros(0, 2)
    for t in targets[c['n_strong'] + c['n_weak']:]:
        t['labels'] = torch.zeros(0, dtype=torch.int64)
    return x_t, x_s, targets


# G15: the same mean-teacher iteration with mix-up on (train_ss_sedt.py --mix_up_ratio 0.6) and a supervised mix-up step.
# 5 strong + 5 weak labelled clips: mixup_data mixes int(10 * 0.6) = 6 clips (five strong and one weak one),
# mixup_label_unlabel the first int(10 * 0.5) = 5 unlabelled clips with the (already mixed) strong ones.
SEMI_MIX = dict(n_strong=5, n_weak=5, n_unl=6, T=496, thr=0.115, seed_w=2023, seed_x=151, seed_t=152, np_seed=0, ratio=0.6)
SUP_MIX = dict(n_strong=5, n_weak=5, T=496, seed_w=2024, seed_x=161, seed_t=162, np_seed=3, ratio=0.6)


def sparse_targets(batch, seed, num_classes=10):
    """1-3 short events per clip: two such clips usually mix without a same-class overlap (mixup.py:84-93 abandons those)"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        n = int(torch.randint(1, 4, (1,), generator=g).item())
        l = torch.rand(n, generator=g) * 0.13 + 0.02
        c = l / 2 + torch.rand(n, generator=g) * (1 - l)
        out.append({'labels': torch.randint(0, num_classes, (n,), generator=g), 'boxes': torch.stack([c, l], dim=-1),
                    'orig_size': torch.tensor(10.0)})
    return out


def semi_mix_batch():
    c = SEMI_MIX
    nl = c['n_strong'] + c['n_weak']
    B = nl + c['n_unl']
    x_t = clip_input(B, c['T'], c['seed_x'])
    x_s = x_t.clone()
    x_s[nl:] += 0.1 * clip_input(c['n_unl'], c['T'], c['seed_x'] + 1)
    targets = sparse_targets(B, c['seed_t'])
    for t in targets[c['n_strong']:]:
        t['boxes'] = torch.zeros(0, 2)
    for t in targets[nl:]:
        t['labels'] = torch.zeros(0, dtype=torch.int64)
    return x_t, x_s, targets


def sup_mix_batch():
    c = SUP_MIX
    B = c['n_strong'] + c['n_weak']
    x = clip_input(B, c['T'], c['seed_x'])
    targets = sparse_targets(B, c['seed_t'])
    for t in targets[c['n_strong']:]:
        t['boxes'] = torch.zeros(0, 2)
    return x, targets


QUERY = dict(T=496, num_patches=10, seeds=(501, 502), short_clip_T=96)


def query_clips():
    """normalised (1, T, 64) clips the SP-SEDT patch cropper works on: two full-length clips and a short one (its longest
    boxes are UP-sampled to 128 frames)"""
    c = QUERY
    return [clip_input(1, c['T'], c['seeds'][0])[0] * 1.7 - 0.3, clip_input(1, c['T'], c['seeds'][1])[0] * 0.6 + 2.0,
            clip_input(1, c['short_clip_T'], 503)[0]]


# G16: the --pooling variants (sedt.py:47-61, 96-119, 182-185): 2 strong + 2 weak clips, dec_at model
POOL = dict(n_strong=2, n_weak=2, T=248, seed_w=1600, seed_x=163, seed_t=164, modes=('max', 'avg', 'attn', 'weighted_sum'))


def pool_batch():
    c = POOL
    B = c['n_strong'] + c['n_weak']
    x = clip_input(B, c['T'], c['seed_x'])
    targets = synthetic_targets(B, c['seed_t'], 10)
    for t in targets[c['n_strong']:]:
        t['boxes'] = torch.zeros(0, 2)
    return x, targets


def g17_inputs():
    """inputs of fixture G17 (also imported by the tests): src, pos, query, key-padding mask, and the two weight tensors of the linear
    loss  sum(hs * w_hs) + sum(mem * w_mem)"""
    g = torch.Generator().manual_seed(171)
    src = torch.randn(2, 256, 32, 4, generator=g)
    pos = torch.randn(2, 256, 32, 4, generator=g) * 0.5
    query = torch.randn(11, 256, generator=g)
    mask = torch.zeros(2, 32, 4, dtype=torch.bool)
    mask[1, 27:, :] = True
    w_hs = torch.randn(3, 2, 11, 256, generator=g)
    w_mem = torch.randn(2, 128, 256, generator=g) * 0.1
    return src, pos, query, mask, w_hs, w_mem
