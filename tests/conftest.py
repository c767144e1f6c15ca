import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_addoption(parser):
    parser.addoption('--x3', action='store_true', default=False,
                     help="run every f32-mode test in the bf16x3 compute mode instead (runtime.set_compute_dtype('f32') selects 'bf16x3'): "
                          "the parity-grade fast mode has to pass the SAME fixture checks at the same tolerances (tests/test_x3_gpu.py)")


@pytest.fixture(autouse=True, scope='session')
def _x3_mode(request):
    if not request.config.getoption('--x3'):
        yield
        return
    from sound_event_detection_transformer_amd import runtime
    real = runtime.set_compute_dtype

    def patched(name):
        real('bf16x3' if name in ('f32', 'fp32', 'float32', 0) else name)
    runtime.set_compute_dtype = patched
    patched('f32')
    yield
    runtime.set_compute_dtype = real
