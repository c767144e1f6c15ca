"""GPU: does bf16 TRAINING track f32 training beyond one step?  (the headline is measured in bf16, whose outputs sit
1.3-2e-2 from the oracle; one-step evidence - loss 2.4e-4, gradient cosines >= 0.9983 - says nothing about a trajectory.)

100 captured C2 steps (URBAN-SED SEDT E = 3, Q = 10, dec_at, B = 64 of 500 frames; reference engine.py:56-80 per batch: forward ->
SetCriterion with Hungarian matching -> backward -> clip 0.1 -> AdamW 1e-4 / 1e-4) from the same seeded weights over the same cycle of four
synthetic batches, dropout 0 (so every mode sees the same function), device matching, in the three compute modes - and the CPU oracle's
first 10 steps of the same recipe (f32 torch autograd + torch.optim.AdamW + clip_grad_norm_ on the box's host cores).

What the curves can and cannot agree on.  The per-step loss is chaotic: AdamW's first updates are lr * sign-like (g / sqrt(v) with v ~ g^2),
so weights whose gradient is near zero move by +-lr on rounding noise, and a Hungarian assignment that flips moves a clip's loss at
once.  Two f32 runs whose initial weights differ by 1e-6 (relative, random) are 1e-2-scale apart after ten steps - as far as the f32 mode
is from the CPU oracle, and the scale of the bf16-to-f32 distance too.  The test therefore states:
  * oracle vs f32 mode: steps 1-2 within 1e-5, step 3 within 5e-4 (before the chaos has grown: the same arithmetic; measured 7e-8, 8e-7,
    4e-5), steps 1-10 within 5e-2 (measured 1.5e-2; the oracle's own rounding depends on the host's thread count);
  * the chaos floor: f32 against its 1e-6-perturbed twin, per step and on the 8-step moving average (two passes over the batch cycle);
  * bf16 / bf16x3 vs f32 on the moving average: maximum within 3 x the floor's maximum (and 8e-2), MEAN within 2 x the floor's mean + 5e-3;
    per step within 1.5e-1.  (Measured: floor 2.0e-2 max / 5.6e-3 mean; bf16 1.4e-2 / 6.1e-3 - indistinguishable from the floor;
    bf16x3 5.7e-2 / 1.0e-2 - another path through the same chaos, not a precision effect: it is the more exact mode);
  * every mode's loss drop over the 100 steps within 5 % of the f32 run's (the drop is what training is for).
Measured values are printed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS, ORACLE_STEPS, B, NBATCH = 100, 10, 64, 4


def _batches():
    from sound_event_detection_transformer_amd.utilities.synthetic import synthetic_batch
    return [synthetic_batch(B, 500, 5000 + 10 * i, torch.device('cpu')) for i in range(NBATCH)]


def _hip_curve(mode, steps, perturb=0.0):
    from sound_event_detection_transformer_amd import runtime
    from sound_event_detection_transformer_amd.engine import GraphedTrainStep, build_optimizer
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict
    runtime.set_compute_dtype(mode)
    try:
        dev = torch.device('cuda')
        model, crit, _ = build_model(default_args(dropout=0.0))
        model.load_state_dict(seeded_state_dict(model.state_dict(), 2020))
        if perturb:
            g_ = torch.Generator().manual_seed(1)
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.mul_(1.0 + perturb * torch.randn(p_.shape, generator=g_))
        model.to(dev).train()
        crit.to(dev)
        opt = build_optimizer(model)
        data = [(x.to(dev), t) for x, t in _batches()]
        g = GraphedTrainStep(model, crit, opt, data[0][0], data[0][1], None, slice(B), max_norm=0.1)
        curve = []
        for s in range(steps):
            x, t = data[s % NBATCH]
            total, _ = g(x, t)
            curve.append(float(total.item()))
        del g
        return np.array(curve)
    finally:
        runtime.set_compute_dtype('f32')


def _oracle_curve(steps):
    from oracle import sedt_oracle as O
    from oracle.criterion_oracle import build_oracle_criterion
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    model = O.build_oracle_model(10, 10, 3, 3, True, True, True, dropout=0.0)
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 2020))
    model.train()
    crit = build_oracle_criterion(10, 3, True, True)
    groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
              {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": 1e-4}]
    opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=1e-4)
    data = _batches()
    curve = []
    for s in range(steps):
        x, t = data[s % NBATCH]
        ld, _ = crit(model(x), t, None, slice(B))
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        curve.append(float(loss.item()))
    return np.array(curve)


def _smooth(c, w=2 * NBATCH):
    return np.convolve(c, np.ones(w) / w, mode='valid')


def test_bf16_and_bf16x3_training_track_f32_over_100_steps_and_the_oracle_over_10(capsys):
    f32 = _hip_curve('f32', STEPS)
    twin = _hip_curve('f32', STEPS, perturb=1e-6)
    x3 = _hip_curve('bf16x3', STEPS)
    bf = _hip_curve('bf16', STEPS)
    orc = _oracle_curve(ORACLE_STEPS)
    r = lambda a, b: np.abs(a - b) / np.abs(b)
    e_or = r(f32[:ORACLE_STEPS], orc)
    per = {k: r(c, f32) for k, c in (('twin', twin), ('bf16x3', x3), ('bf16', bf))}
    smo = {k: r(_smooth(c), _smooth(f32)) for k, c in (('twin', twin), ('bf16x3', x3), ('bf16', bf))}
    drop = lambda c: float(np.mean(c[:NBATCH]) - np.mean(c[-NBATCH:]))
    with capsys.disabled():
        print(f'\n[C2 training trajectory, B = 64, dropout 0, {STEPS} steps] loss f32 {f32[0]:.4f} -> {f32[-1]:.4f} (drop {drop(f32):.3f}); '
              f'f32 twin (+1e-6) -> {twin[-1]:.4f} (drop {drop(twin):.3f}); bf16 -> {bf[-1]:.4f} (drop {drop(bf):.3f}); bf16x3 -> {x3[-1]:.4f} '
              f'(drop {drop(x3):.3f}); oracle steps 1-{ORACLE_STEPS}: {orc[0]:.4f} -> {orc[-1]:.4f}\n  oracle vs f32 per step: '
              + ' '.join(f'{v:.1e}' for v in e_or) + '\n  relative distance to the f32 curve, per step max / mean | 8-step moving average max / mean:\n'
              + '\n'.join(f'    {k:7s} {per[k].max():.2e} / {per[k].mean():.2e} | {smo[k].max():.2e} / {smo[k].mean():.2e}' for k in per))
    for c in (f32, twin, x3, bf):
        assert np.all(np.isfinite(c))
    assert drop(f32) > 0.2 * f32[0], 'the f32 run did not train'
    assert e_or[:2].max() < 1e-5 and e_or[2] < 5e-4 and e_or.max() < 5e-2, e_or
    floor, floor_mean = max(smo['twin'].max(), 1e-3), smo['twin'].mean()
    for k in ('bf16', 'bf16x3'):
        assert smo[k].max() < max(3 * floor, 1e-2) and smo[k].max() < 8e-2, (k, smo[k].max(), floor)
        assert smo[k].mean() < 2 * floor_mean + 5e-3, (k, smo[k].mean(), floor_mean)
        assert per[k].max() < 1.5e-1, (k, per[k].max())
    for k, c in (('twin', twin), ('bf16', bf), ('bf16x3', x3)):
        assert abs(drop(c) - drop(f32)) < 0.05 * abs(drop(f32)), (k, drop(c), drop(f32))
