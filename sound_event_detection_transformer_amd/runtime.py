"""Process-wide runtime state of the HIP path: compute dtype and dropout seeds."""
import torch

from .lib import F32, BF16

_state = {'dtype': F32, 'seed': 0x5EDD0000}


def set_compute_dtype(name):
    """'f32' = parity mode (exact-f32 MFMA), 'bf16' = throughput mode (bf16 MFMA, f32 accumulate)."""
    _state['dtype'] = {'f32': F32, 'fp32': F32, 'float32': F32, 'bf16': BF16, 'bfloat16': BF16, F32: F32, BF16: BF16}[name]


def compute_dtype():
    return _state['dtype']


def torch_dtype():
    return torch.float32 if _state['dtype'] == F32 else torch.bfloat16


def manual_seed(seed):
    _state['seed'] = int(seed) & 0x7fffffff


def next_seed():
    """a fresh 32-bit seed per dropout site per forward (the kernels hash (seed, element index))"""
    _state['seed'] = (_state['seed'] * 1103515245 + 12345) & 0x7fffffff
    return _state['seed']


_seed_tensors = {}


def seed_ptr(device):
    """device-side seed word added to every dropout seed: a captured HIP graph draws fresh masks on each replay after
    ``bump_seed`` (the per-site seeds baked into the launches stay constant)."""
    key = str(device)
    if key not in _seed_tensors:
        _seed_tensors[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _seed_tensors[key]


def bump_seed(device):
    seed_ptr(device).add_(1)
