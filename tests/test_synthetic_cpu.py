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
