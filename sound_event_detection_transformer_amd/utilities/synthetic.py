"""Synthetic inputs for benchmarks and smoke runs (there are no datasets or checkpoints offline).

These generators are deliberately bit-identical to the ones the test oracle uses (tests/test_synthetic_cpu.py checks
that), so a bench run, the golden fixtures and the parity tests all see the same weights and targets - but the product
never imports anything from oracle/."""
import math

import torch


def seeded_state_dict(template, seed):
    """deterministic random-init weights for a SEDT state_dict: SORTED keys, one CPU generator; scales keep activations
    O(1)-O(100) through the 16 FrozenBatchNorm residual blocks"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template.keys()):
        shape = tuple(template[k].shape)
        leaf = k.rsplit('.', 1)[-1]
        if ('.bn' in k or 'downsample.1' in k) and 'body' in k:        # FrozenBatchNorm2d buffers
            closing = 'bn3' in k or 'downsample.1' in k
            if leaf == 'weight':
                lo, hi = (0.3, 0.6) if closing else (0.7, 1.1)
                t = torch.rand(shape, generator=g) * (hi - lo) + lo
            elif leaf == 'running_var':
                t = torch.rand(shape, generator=g) + 0.5
            else:
                t = torch.randn(shape, generator=g) * 0.1
        elif 'norm' in k:
            t = torch.rand(shape, generator=g) * 0.4 + 0.8 if leaf == 'weight' else torch.randn(shape, generator=g) * 0.05
        elif k.endswith('query_embed.weight'):
            t = torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / math.prod(shape[1:]))
        else:
            t = torch.randn(shape, generator=g) * 0.05
        out[k] = t.to(template[k].dtype)
    return out


def synthetic_targets(batch, seed, num_classes=10):
    """URBAN-SED-shaped strong labels: n ~ clip(Poisson(4.5), 1, 9) events per clip, label ~ U{0..C-1},
    length ~ U(0.02, 0.5), centre ~ U(l/2, 1-l/2); boxes are (centre, length) in [0, 1]"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        n = int(torch.poisson(torch.tensor(4.5), generator=g).clamp(1, 9).item())
        length = torch.rand(n, generator=g) * 0.48 + 0.02
        centre = length / 2 + torch.rand(n, generator=g) * (1 - length)
        out.append({'labels': torch.randint(0, num_classes, (n,), generator=g),
                    'boxes': torch.stack([centre, length], dim=-1), 'orig_size': torch.tensor(10.0)})
    return out


def synthetic_batch(B, T, seed, device=None, num_classes=10):
    """(B,1,T,64) log-mel-shaped noise + strong targets"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, T, 64, generator=g)
    targets = synthetic_targets(B, seed + 1, num_classes)
    if device is not None:
        x = x.to(device)
        targets = [{k: v.to(device) for k, v in t.items()} for t in targets]
    return x, targets
