"""Process-wide runtime state of the HIP path: compute dtype and dropout seeds."""
import torch

from .lib import F32, BF16

_state = {'dtype': F32, 'seed': 0x5EDD0000}
_seed_tensors = {}


def set_compute_dtype(name):
    """'f32' = parity mode (exact-f32 MFMA), 'bf16' = throughput mode (bf16 MFMA, f32 accumulate), 'bf16x3' = parity-grade fast mode:
    f32 tensors everywhere as in 'f32', every contraction from bf16 hi / lo splits of its operands (three bf16 MFMAs per product
    block, f32 accumulate: ~2^-16 per product) - meets the 1e-3 tolerance at several times the exact-f32 rate."""
    from . import lib
    lib.GEMM_X3 = name in ('bf16x3', 'x3')
    if lib.GEMM_X3:
        name = 'f32'
    _state['dtype'] = {'f32': F32, 'fp32': F32, 'float32': F32, 'bf16': BF16, 'bfloat16': BF16, F32: F32, BF16: BF16}[name]


def compute_mode():
    """'f32' | 'bf16x3' | 'bf16'"""
    from . import lib
    return 'bf16' if _state['dtype'] == BF16 else ('bf16x3' if lib.GEMM_X3 else 'f32')


def compute_dtype():
    return _state['dtype']


def torch_dtype():
    return torch.float32 if _state['dtype'] == F32 else torch.bfloat16


def manual_seed(seed):
    """restart the dropout mask sequence: the host-side per-site seeds AND the device-side replay counters"""
    _state['seed'] = int(seed) & 0x7fffffff
    for t in _seed_tensors.values():
        t.zero_()


def seed_for_rank(rank):
    """data parallel: every rank draws its own dropout masks (the reference's DDP processes do: independent torch RNG
    streams); call once per process before capturing / running steps"""
    _state['seed'] = (_state['seed'] + 0x9E3779B1 * (int(rank) + 1)) & 0x7fffffff if rank else _state['seed']


def next_seed():
    """a fresh 32-bit seed per dropout site per forward (the kernels hash (seed, element index))"""
    _state['seed'] = (_state['seed'] * 1103515245 + 12345) & 0x7fffffff
    return _state['seed']


def seed_ptr(device):
    """device-side seed word added to every dropout seed: a captured HIP graph draws fresh masks on each replay after
    ``bump_seed`` (the per-site seeds baked into the launches stay constant)."""
    key = str(device)
    if key not in _seed_tensors:
        _seed_tensors[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _seed_tensors[key]


def bump_seed(device):
    seed_ptr(device).add_(1)


# ---------------------------------------------------------------------------------------------------------------------
# Asynchronous weight gradients.  In the backward pass the dgrad chain (layer n -> layer n-1) is the critical path; the
# weight gradients only feed the optimizer.  Inside an ``async_wgrad()`` scope every wgrad GEMM (+ its split-K
# reduction) is issued on a second HIP stream that forks from the main stream where its operands are ready, so the two
# run concurrently (and become parallel branches when the step is captured into a HIP graph).  The scope's exit - or an
# explicit ``join_wgrad()`` - makes the main stream wait for the side stream; only then may anything read the weight
# gradients.  Operands are kept referenced until the join, because the caching allocator would otherwise hand their
# memory to later main-stream allocations while the side stream is still reading it.
_aw = {'on': False, 'streams': {}, 'held': [], 'forked': False}


def async_wgrad_on():
    return _aw['on']


def side_stream(device):
    key = str(device)
    if key not in _aw['streams']:
        _aw['streams'][key] = torch.cuda.Stream(device=device)
    return _aw['streams'][key]


def join_wgrad():
    if _aw['forked']:
        for s in _aw['streams'].values():
            torch.cuda.current_stream(s.device).wait_stream(s)
        _aw['forked'] = False
    _aw['held'].clear()


class async_wgrad(object):
    """context manager: weight gradients computed inside are valid only after the scope exits (or join_wgrad())"""

    def __init__(self, enable=True):
        self.enable = enable

    def __enter__(self):
        self.prev = _aw['on']
        _aw['on'] = bool(self.enable)
        return self

    def __exit__(self, *exc):
        _aw['on'] = self.prev
        if not self.prev:
            join_wgrad()
        return False


class side(object):
    """``with runtime.side(t1, t2, ...):`` - run the body on the wgrad stream (when async wgrad is on), after everything
    issued so far on the current stream; t1.. are the tensors the body reads that were produced on the current stream."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.cm = None

    def __enter__(self):
        if not _aw['on']:
            return self
        dev = next(t.device for t in self.tensors if t is not None)
        s = side_stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        _aw['held'].extend(t for t in self.tensors if t is not None)
        _aw['forked'] = True
        self.cm = torch.cuda.stream(s)
        self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        if self.cm is not None:
            self.cm.__exit__(*exc)
        return False
