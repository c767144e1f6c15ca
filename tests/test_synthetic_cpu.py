"""CPU: the product's synthetic weight / target generators are bit-identical to the oracle's (so bench runs, golden
fixtures and parity tests share inputs without the product importing oracle/)."""
import torch

from oracle import sedt_oracle as O
from oracle.criterion_oracle import synthetic_targets as oracle_targets
from sound_event_detection_transformer_amd.utilities import synthetic as S


def test_seeded_state_dict_identical():
    tmpl = O.build_oracle_model(10, 10, 1, 1, True, True, True).state_dict()
    a, b = O.seeded_state_dict(tmpl, 123), S.seeded_state_dict(tmpl, 123)
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_synthetic_targets_identical():
    for seed in (1, 99):
        for ta, tb in zip(oracle_targets(16, seed, 10), S.synthetic_targets(16, seed, 10)):
            assert ta.keys() == tb.keys()
            for k in ta:
                assert torch.equal(ta[k], tb[k])


def test_transform_parameter_draws_follow_the_reference_order(golden_dir=None):
    """host logic of the device feature transform (no GPU): np.random is consumed exactly like the reference's TimeMask /
    FreqMask / FreqShift objects do (fixture G13 recorded what they drew after np.random.seed(1000 + i))"""
    import os
    import numpy as np
    from conftest import GOLDEN
    from sound_event_detection_transformer_amd.utilities.transforms import DeviceBoxTransform
    g = np.load(os.path.join(GOLDEN, 'g13_transforms_mixup.npz'))
    tf = DeviceBoxTransform(496, time_mask=True, freq_mask=True, freq_shift=True, device='cpu',
                            tm=(0.0, 0.1, 0.6), fm=(0.03, 0.4, 0.6), fs=(0.6, 4, 0, 2))
    for i, p in enumerate(g['params']):
        np.random.seed(1000 + i)
        r = tf.draw(470)
        assert (r['tm_t'], r['tm_t0']) == ((int(p[1] * 496), int(p[2] * 496)) if p[0] else (0, 0))
        assert (r['fm_on'], r['fm_f'], r['fm_f0']) == ((1, int(p[4] * 64), int(p[5] * 64)) if p[3] else (0, 0, 0))
        assert r['fs_shift'] == (int(p[7]) if p[6] else 0)


def test_box_transform_batch_draws_equal_the_per_clip_draws():
    """DeviceBoxTransform.draw_batch (the loader's fast path) consumes np.random exactly like draw() clip by clip - the order the
    reference's TimeMask / FreqMask / FreqShift objects draw in (utilities/BoxTransforms.py:380-383, 410-413, 437-443)"""
    import numpy as np
    from sound_event_detection_transformer_amd.utilities.transforms import DeviceBoxTransform
    for flags in ((True, True, True), (False, True, False), (True, False, True)):
        tf = DeviceBoxTransform(496, None, None, *flags, device='cpu')
        np.random.seed(3)
        a = np.stack([tf.draw(n) for n in (496, 431, 520, 496) * 8])
        sa = np.random.get_state()[1][:4].copy()
        np.random.seed(3)
        b = tf.draw_batch([496, 431, 520, 496] * 8)
        assert a.dtype == b.dtype and (a.view(np.int32) == b.view(np.int32)).all()
        assert (np.random.get_state()[1][:4] == sa).all()
