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


SEMI_SCALER = (0.0, 10.0)          # per-band (mean, std) in dB of the synthetic amplitudes below: normalised features ~ N(0, 1)


def synthetic_semi_raw(n_label, n_unl, T, seed, snr_db=30.0):
    """raw mel AMPLITUDES (B, T, 64) for the mean-teacher step, as the reference's semi loader sees them before its transform
    chain (train_ss_sedt.py:87-96): amplitudes 10^(N(0,1)/2), i.e. 10 dB of spread around 0 dB; the STUDENT copy of the
    unlabelled clips carries AugmentGaussianNoise's additive noise (BoxTransforms.py:136-159: per band std = sqrt(mean_t(x^2 *
    10^(-snr/10)))).  The two views then go through utilities.transforms.DeviceBoxTransform per step (teacher: FreqMask; student:
    TimeMask + FreqMask - the reference's TimeMask skips view 0, BoxTransforms.py:24-26)."""
    g = torch.Generator().manual_seed(seed)
    B = n_label + n_unl
    raw_t = torch.pow(10.0, 0.5 * torch.randn(B, T, 64, generator=g))
    raw_s = raw_t.clone()
    u = raw_t[n_label:]
    std = torch.sqrt((u * u * 10.0 ** (-snr_db / 10.0)).mean(dim=1, keepdim=True))
    raw_s[n_label:] = u + std * torch.randn(u.shape, generator=g)
    return raw_t, raw_s


def semi_view_transforms(T, device):
    """(teacher transform, student transform) of the mean-teacher recipe with --freq_mask --time_mask (train_ss_sedt.py:87-90)"""
    import numpy as np
    from .transforms import DeviceBoxTransform
    mean, std = np.full(64, SEMI_SCALER[0]), np.full(64, SEMI_SCALER[1])
    return (DeviceBoxTransform(T, mean, std, time_mask=False, freq_mask=True, device=device),
            DeviceBoxTransform(T, mean, std, time_mask=True, freq_mask=True, device=device))
