"""The per-batch body of the reference's ``engine.train`` loop (engine.py:56-80) as one function."""
import math

import os
import numpy as np
import torch

from . import ops
from .optim import FusedAdamW


class _Done(object):
    def wait(self):
        return True


def allreduce_mean(flat, async_op=False):
    """average one flat gradient buffer over the data-parallel group: RCCL AVG on GPUs; SUM + scale elsewhere
    (gloo has no AVG and, in this image, no GPU tensors: stage through the host - used by the CPU/gloo tests only).
    async_op (RCCL): returns the work handle; the collective runs on RCCL's stream after everything already queued on the
    current stream, and later kernels of the current stream overlap with it until handle.wait()."""
    dist = torch.distributed
    world = dist.get_world_size()
    if dist.get_backend() == 'nccl':
        work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)
        return work if async_op else flat
    if flat.numel() == 0:
        return _Done() if async_op else flat
    if flat.is_cuda:
        h = flat.float().cpu()                                # (a bf16 flat buffer is reduced in f32 on the host)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h.div_(world))
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
    return _Done() if async_op else flat


class _Upload(object):
    """small host -> device table refreshed before every replay of a graph: a ring of pinned staging buffers (a slot is reused
    only after the copy that last read it has finished), asynchronous copies on the current stream"""

    def __init__(self, nbytes, dev, slots=4):
        self.dev_buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self._pin = [torch.zeros(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self._ev = [None] * slots
        self._k = 0

    def send(self, raw):
        """raw: CPU uint8 tensor (<= nbytes)"""
        k = self._k
        self._k = (k + 1) % len(self._pin)
        if self._ev[k] is not None:
            self._ev[k].synchronize()
        n = raw.numel()
        self._pin[k][:n].copy_(raw)
        self.dev_buf[:n].copy_(self._pin[k][:n], non_blocking=True)
        self._ev[k] = torch.cuda.Event()
        self._ev[k].record()
        return self.dev_buf


_step_streams = {}


def train_stream(dev):
    """THE stream training work of a device runs on - eager steps, the warm-up steps of a capture and the captures themselves.
    A parameter's gradient accumulator (autograd's AccumulateGrad node) is tied to the stream of the forward that created it and
    lives as long as any autograd graph referencing it; if a later backward runs on another stream, autograd executes the
    accumulation of a second gradient (a parameter used by two forward passes, as in the mean-teacher step) on the OLD stream -
    outside a capture in progress on the new one, i.e. with garbage at replay.  One stream for everything removes the case."""
    key = str(dev)
    if key not in _step_streams:
        _step_streams[key] = torch.cuda.Stream(device=dev)
    return _step_streams[key]
# capture mode: only THIS thread's unsafe calls invalidate a capture.  With the default ("global") the RCCL watchdog thread of
# an initialised process group - it polls events with hipEventQuery - sporadically kills a capture in progress.
_CAPTURE = dict(capture_error_mode='thread_local')


def quiesce_collectives(dev=None):
    """call before a stream capture when an RCCL process group exists.  ProcessGroupNCCL's watchdog thread polls the completion event of
    every collective it still lists (hipEventQuery, every 100 ms) and HIP answers hipErrorCapturedEvent when the stream such an event
    was last recorded on is capturing at that moment - the watchdog then terminates the PROCESS ('operation not permitted on an event
    last recorded in a capturing stream').  The collectives of the warm-up steps (or of the replays before a second stepper is built)
    run on the training stream, the one the captures use: seen once in ~10 runs of tests/test_dp_gpu.py when a capture started
    inside the watchdog's polling period.  Draining the device and sleeping three polling periods lets the watchdog retire every
    finished work before the capture begins."""
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        import time
        torch.cuda.synchronize(dev)
        time.sleep(0.35)


def train_step(model, criterion, optimizer, batch_input, targets, mask_weak=None, mask_strong=None, max_norm=0.1,
               normalize=False, check_finite=True, patches=None, allreduce=False, fine_tune=False, fl=False, mix_up_ratio=0,
               do_step=True):
    """[mixup_data ->] forward -> SetCriterion -> weighted sum over weight_dict -> backward -> clip_grad_norm_(max_norm) ->
    step -> zero_grad.  Raises on a non-finite loss (the reference calls sys.exit(1), engine.py:70-73).

    mix_up_ratio > 0 (engine.py:50-53): the batch goes through utilities.mixup.mixup_data first (np.random draws as the
    reference's) and the criterion uses the strong | weak split it returns.  do_step=False withholds clip + optimizer step +
    zero_grad: gradients keep accumulating, as in the reference's loop while (i + 1) % accumrating_gradient_steps != 0
    (engine.py:76).

    When called on the default stream the step runs on a dedicated side stream (ordered after / before the caller's
    stream): a backward pass executed on the DEFAULT stream leaves the parameters' gradient accumulators tied to it, and
    a later HIP-graph capture of the backward (GraphedTrainStep) is then invalidated - ROCm 7.2 faults in
    hipStreamEndCapture instead of reporting it."""
    first = batch_input if torch.is_tensor(batch_input) else (getattr(batch_input, 'tensors', None) if not isinstance(
        batch_input, (tuple, list)) else (batch_input[0] if len(batch_input) and torch.is_tensor(batch_input[0]) else None))
    dev = first.device if torch.is_tensor(first) else None
    if dev is not None and dev.type == 'cuda' and torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
        side, cur = train_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = train_step(model, criterion, optimizer, batch_input, targets, mask_weak, mask_strong, max_norm, normalize,
                             check_finite, patches, allreduce, fine_tune, fl, mix_up_ratio, do_step)
        cur.wait_stream(side)
        return out
    if mix_up_ratio:
        from .utilities.mixup import mixup_data
        batch_input, targets, mask_strong, mask_weak = mixup_data(batch_input, targets, mask_strong, mask_weak, mix_up_ratio, alpha=1)
    outputs = model(batch_input, patches) if patches is not None else model(batch_input)
    loss_dict, _ = criterion(outputs, targets, mask_weak, mask_strong, fine_tune, normalize, fl)
    wd = criterion.weight_dict
    losses = getattr(criterion, 'last_total', None)      # the same weighted sum, pre-reduced as one dot product
    if losses is None:
        losses = sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)
    else:
        criterion.last_total = None                      # do not keep the autograd graph alive past this step
    if check_finite:
        v = losses.item()
        if not math.isfinite(v):
            raise FloatingPointError(f'Loss is {v}, stopping training: {loss_dict}')
    losses.backward()
    if not do_step:
        return losses.detach(), {k: v.detach() for k, v in loss_dict.items()}
    dp = allreduce and torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size() > 1
    if isinstance(optimizer, FusedAdamW) and dp:
        flat = optimizer.gather_grads()                  # plain data parallelism without the DDP wrapper
        allreduce_mean(flat)
        optimizer.step(max_norm=max_norm, from_flat=True)
    elif isinstance(optimizer, FusedAdamW):
        optimizer.step(max_norm=max_norm)                # clip + AdamW fused (three launches for all tensors)
    else:
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    # logging values only: detached, so that a caller holding on to them does not keep this step's autograd graph (and the
    # parameters' AccumulateGrad nodes of this stream) alive - a later HIP-graph capture of the backward would break on them
    return losses.detach(), {k: v.detach() for k, v in loss_dict.items()}


def build_optimizer(model, lr=1e-4, lr_backbone=1e-4, weight_decay=1e-4, fused=True):
    """AdamW with the reference's two parameter groups (train_sedt.py:234-240, 269-270)"""
    groups = [{"params": [p for n, p in model.named_parameters() if "backbone" not in n and p.requires_grad]},
              {"params": [p for n, p in model.named_parameters() if "backbone" in n and p.requires_grad], "lr": lr_backbone}]
    if fused:
        return FusedAdamW(groups, lr=lr, weight_decay=weight_decay)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay)


def _snapshot(model, optimizer, ema=None):
    """parameters + optimizer moments (+ EMA shadow) before the warm-up steps of a capture; _restore puts them back, so that
    constructing a graphed stepper leaves the training state untouched"""
    from . import runtime
    snap = {'p': [p.detach().clone() for p in model.parameters()], 'o': optimizer.snapshot(),
            'seed': {k: t.clone() for k, t in runtime._seed_tensors.items()}}       # device-side dropout replay counters
    if ema is not None:
        snap['e'] = {n: v.clone() for n, v in ema.shadow.items()}
    return snap


@torch.no_grad()
def _restore(model, optimizer, snap, ema=None):
    for p, v in zip(model.parameters(), snap['p']):
        p.copy_(v)
    optimizer.restore(snap['o'])
    from . import runtime
    for k, t in runtime._seed_tensors.items():
        if k in snap['seed']:
            t.copy_(snap['seed'][k])
        else:
            t.zero_()
    if ema is not None:
        for n, v in snap['e'].items():
            ema.shadow[n].copy_(v)


def dp_segment_plan(net, cuts='coarse'):
    """where the data-parallel steppers cut the backward: [(parameters, cut getter or None, cut name)] in the order autograd
    produces the gradients.  cuts: 'coarse' = transformer + heads | layer4 | layer3 | layer2 + stem (cut tensors: the outputs of
    layer4 / layer3 / layer2); 'fine' = additionally decoder + heads | encoder + input_proj (cut: the encoder output).  A model
    whose backbone is frozen (SP-SEDT, train_spsedt.py:50) has no backbone backward: its one cut is decoder + heads | encoder.
    Returns [] when the model cannot be cut (no SEDT backbone / transformer).  ``optimizer.set_segments([s[0] for s in plan])``
    gives any optimizer the same flat layout (same summation order of the gradient norm)."""
    if cuts == 'none' or not hasattr(net, 'transformer') or not hasattr(net, 'backbone'):
        return []
    body = getattr(net.backbone[0], 'body', None)
    if body is None or not hasattr(body, 'stage_out'):
        return []
    train = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    back = [(n, p) for n, p in train if n.startswith('backbone.')]
    enc = [p for n, p in train if n.startswith('transformer.encoder.') or n.startswith('input_proj.')]
    dec = [p for n, p in train if not n.startswith('backbone.') and not n.startswith('transformer.encoder.')
           and not n.startswith('input_proj.')]
    tr = net.transformer
    if back:
        segs = [(dec, lambda: tr.cut_memory, 'memory'), (enc, lambda: body.stage_out[3], 'stage_out3')] if cuts == 'fine' else \
               [(dec + enc, lambda: body.stage_out[3], 'stage_out3')]
        segs += [([p for n, p in back if '.layer4.' in n], lambda: body.stage_out[2], 'stage_out2'),
                 ([p for n, p in back if '.layer3.' in n], lambda: body.stage_out[1], 'stage_out1'),
                 ([p for n, p in back if '.layer4.' not in n and '.layer3.' not in n], None, 'tail')]
    else:
        segs = [(dec, lambda: tr.cut_memory, 'memory'), (enc, None, 'tail')]
    segs = [sg for sg in segs if sg[0]]
    if segs:
        segs[-1] = (segs[-1][0], None, 'tail')
    return segs


class _GraphedBase(object):
    """what every graphed stepper shares: the optimizer's private chunk tables, the learning-rate refresh before each
    replay, and the device-side non-finite-loss word (reference engine.py:70-73 / 167-169 abort on such a loss)."""
    check_every = 50

    def _init_common(self, criterion, optimizer, dev):
        self._tabname = f'graph{id(self)}'
        self._calls = 0
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)
        self._one = torch.ones((), dtype=torch.float32, device=dev)     # the backward's root gradient (autograd would fill one per step)
        criterion.nonfinite = self.nonfinite
        # the same word guards the update: with it raised (non-finite loss, or non-finite gradient norm) the captured clip + AdamW
        # (and the EMA update) leave parameters, moments, step count and teacher untouched, replay after replay, until the host
        # polls the word and raises - the training state stays at its last good value (the reference stops before the backward)
        optimizer.guard = self.nonfinite
        if getattr(self, 'ema', None) is not None:
            self.ema.guard = self.nonfinite
        # the dropout-seed word is advanced by the captured optimizer step itself (the final reduction of the gradient norm), one
        # launch less per replay - unless the optimizer does not run on every call (gradient accumulation) or does not clip
        from . import runtime as _rt
        self._host_bump = not (getattr(self, 'max_norm', 0) > 0 and getattr(self, 'accum_steps', 1) == 1)
        optimizer.seed_word = None if self._host_bump else _rt.seed_ptr(dev)

    def _release_shared(self):
        """the guard / seed words are baked into THIS stepper's captured launches (pointers); the shared optimizer / EMA objects must
        not keep them, or eager optimizer.step() / ema.update() calls - the eager fallback of a stepper whose construction failed,
        or eager steps between replays - would be silently skipped by a flag nothing polls"""
        opt = getattr(self, 'optimizer', None)
        if opt is not None:
            opt.guard = opt.seed_word = None
        ema = getattr(self, 'ema', None)
        if ema is not None and hasattr(ema, 'guard'):
            ema.guard = None

    def _before_replay(self):
        self.optimizer.refresh_hyperparams(self._tabname)      # StepLR / param_group['lr'] edits reach the captured upload

    # ---- the backward as SEGMENTS (data parallel: all-reduce segment k while the backward of segments k+1.. runs)
    def _plan_segments(self, net, optimizer, cuts):
        """decide where the backward is cut (dp_segment_plan) and tell the optimizer to lay its flat buffers out segment by
        segment.  Sets self.segs = [(parameters, cut getter or None), ...] or None (one segment: no cut)."""
        self.segs = None
        if cuts == 'none' or optimizer._static is not None:
            return
        segs = dp_segment_plan(net, cuts)
        if len(segs) < 2:
            return
        tr = net.transformer
        body = net.backbone[0].body
        body.keep_stage_out = any('stage_out' in sg[2] for sg in segs)
        tr.keep_cut = any(sg[2] == 'memory' for sg in segs)
        optimizer.set_segments([sg[0] for sg in segs])
        self.segs = [(sg[0], sg[1]) for sg in segs]

    def _grad_sink(self, accumulate=False):
        """scope in which the large weight gradients are written straight into their slots of the f32 flat gradient buffer
        (ops.grad_sink); a no-op scope when micro-batches accumulate there or the buffer is bf16"""
        views = None if accumulate else self.optimizer.flat_views()
        return ops.grad_sink(views)

    def _root_grad(self, root):
        return self._one if (root.dim() == 0 and root.dtype == torch.float32) else torch.ones_like(root)

    def _segment_backward(self, k, root=None, accumulate=False):
        """(inside the capture of graph k) gradients of segment k's parameters -> segment k of the flat buffer.  k = 0 starts
        from the loss ``root``; later segments continue from the gradient of the previous cut tensor."""
        params, cut = self.segs[k]
        cut_t = None if cut is None else cut()
        if cut is not None and cut_t is None:
            raise RuntimeError('the model did not keep the cut tensor of a data-parallel segment')
        want = ([cut_t] if cut_t is not None else []) + list(params)
        with self._grad_sink(accumulate):
            if k == 0:
                grads = torch.autograd.grad(root, want, grad_outputs=self._root_grad(root))
            else:
                grads = torch.autograd.grad(self._cut_t, want, grad_outputs=self._cut_g)
        if cut_t is not None:
            self._cut_t, self._cut_g = cut_t, grads[0]
            grads = grads[1:]
        else:
            self._cut_t = self._cut_g = None
        for p, g in zip(params, grads):
            p.grad = g
        return self.optimizer.gather_grads(k, accumulate=accumulate)

    def _after_replay(self, check_finite):
        self._calls += 1
        if check_finite or (self.check_every and self._calls % self.check_every == 0):
            if self.nonfinite.item():                          # one 4-byte copy every `check_every` steps
                raise FloatingPointError('Loss is not finite, stopping training (device flag raised by the criterion kernel)')


class GraphedTrainStep(_GraphedBase):
    """The training step as HIP graphs.

    device_matching=True (default): ONE graph holds the whole step - model forward, the Hungarian matching of every
    (decoder layer, clip) on the device (ops.match_targets), the fused loss kernel, backward and the fused clip/AdamW.
    Per batch the host only refreshes the static input and the flat target tables (asynchronous copies) and replays;
    nothing depends on a device->host copy, so the host runs ahead of the GPU and the GPU never waits for it.

    device_matching=False: the reference's split - graph A (forward), SetCriterion.prepare on the host (one D2H copy,
    batched C++ Hungarian, one H2D copy), graph B (loss + backward + optimizer).

    SP-SEDT (``example_patches`` given): the patches are a second static input, the query-patch mask is drawn inside the graph,
    the feature-reconstruction loss runs in its own fused kernel (reference engine.py:56-59, sedt/spsedt.py:34-91).

    ``fine_tune`` / ``normalize`` / ``fl``: the criterion variants of the reference's second training stage
    (train_sedt.py:309) - device matching only.

    Constructing one runs ``warmup`` eager steps (lazy initialisation of kernels, buffers, optimizer state) and then RESTORES
    parameters, optimizer moments and step count: the training state is the same before and after.  ``__call__`` returns
    the graph's static loss tensors (overwritten by the next call: ``.item()`` / ``.clone()`` what you keep).  A non-finite
    loss raises - checked on the host every ``check_every`` calls through a device flag (or at once with check_finite=True).

    Before constructing one, drop every tensor that still carries an autograd graph of an earlier EAGER backward on the
    default stream (loss dicts, model outputs): live AccumulateGrad nodes of another stream invalidate the capture of the
    backward (PyTorch warns 'AccumulateGrad node's stream does not match'; ROCm then faults in hipStreamEndCapture).

    Shapes are static: every batch must have the batch size, clip length and strong/weak split it was captured with (and,
    for the host split, the same per-clip target counts).  Dropout masks change on each replay through the device-side
    seed word (runtime.bump_seed); the Adam step count lives on the device too.  The optimizer's learning rates are re-read
    before every replay.  Eager ``optimizer.step()`` calls between replays are safe (the graph owns its pointer tables) - but call
    ``optimizer.zero_grad(set_to_none=True)`` before an eager BACKWARD: the captured backward leaves its static gradient tensors
    attached to the parameters, and autograd would add to them.

    Data parallel (world > 1): parameters are broadcast from rank 0 at construction; gradients are packed into ONE flat buffer
    (f32, or bf16 with ``grad_dtype=torch.bfloat16``) and averaged with RCCL between the backward graph(s) and an optimizer graph.
    With ``overlap_allreduce`` the backward is cut in segments (``dp_cuts``: 'coarse' = transformer + heads | layer4 | layer3 |
    layer2 + stem; 'fine' also splits decoder + heads | encoder; a frozen backbone leaves decoder + heads | encoder): every
    segment's graph ends by packing its gradients into its part of the flat buffer, whose all-reduce is launched asynchronously
    while the next segment's graph runs (engine.dp_segment_plan).

    ``mix_up_ratio`` > 0: mix-up inside the step (reference engine.py:50-53) - see ``__call__``.  ``accum_steps`` = k: gradients of
    k consecutive calls are added in the flat buffer and clip + AdamW run on every k-th call (engine.py:76).

    Two alternatives for the weight gradients are implemented and measured slower than the default (a layer's wgrads as ONE
    grouped launch on the main stream): async_wgrad=True issues them as a parallel branch of the graph (ROCm 7.2 pays
    50-100 us per cross-queue dependency: 7.6 vs 7.3 ms/step), coschedule=True lets them ride in the spare workgroup
    slots of later dgrad launches (ops.WgradPool; 7.3 vs 7.2 ms/step)."""

    def __init__(self, *args, **kw):
        try:
            self._construct(*args, **kw)
        finally:
            self._release_shared()

    def _construct(self, model, criterion, optimizer, example_input, example_targets, mask_weak=None, mask_strong=None,
                 max_norm=0.1, normalize=False, warmup=3, device_matching=True, max_targets=32, async_wgrad=False,
                 overlap_allreduce=True, coschedule=False, data_parallel=None, example_patches=None, fine_tune=False, fl=False,
                 ft_rand=None, mix_up_ratio=0.0, mix_alpha=1, max_events=20, accum_steps=1, dp_cuts='coarse', grad_dtype=None):
        import gc
        from . import runtime
        from .sedt import TargetTables
        if not isinstance(optimizer, FusedAdamW):
            raise RuntimeError('GraphedTrainStep needs FusedAdamW (device-side step count, one pointer table)')
        if (fine_tune or fl or mix_up_ratio) and not device_matching:
            raise NotImplementedError('fine_tune / fl / mix-up are built for the device-matching graph')
        if mix_up_ratio and example_patches is not None:
            raise NotImplementedError('mix-up is not part of the SP-SEDT recipe (train_spsedt.py has no --mix_up_ratio)')
        self.mix, self.mix_alpha, self.max_events = float(mix_up_ratio), mix_alpha, max_events
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.mw, self.ms, self.max_norm, self.normalize = mask_weak, mask_strong, max_norm, normalize
        self.fine_tune, self.fl, self.ft_rand = fine_tune, fl, ft_rand
        self.runtime = runtime
        self.async_wgrad = async_wgrad
        self.coschedule = (coschedule or ops._dev_env('SEDT_COSCHEDULE', '0') == '1') and not async_wgrad
        self.world = torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                            and torch.distributed.is_initialized()) else 1
        self.accum_steps, self._micro = int(accum_steps), 0
        if self.accum_steps < 1:
            raise ValueError('accum_steps >= 1')
        if self.accum_steps > 1 and not device_matching:
            raise NotImplementedError('gradient accumulation is built for the device-matching graph')
        if self.accum_steps > 1 and grad_dtype == torch.bfloat16:
            raise ValueError('accumulating micro-batch gradients in a bf16 flat buffer loses them: use the f32 buffer')
        # data_parallel=True forces the data-parallel schedule (flat gradients, all-reduce, optimizer graph) even in a
        # one-process group: lets a single GPU exercise the RCCL calls of the multi-GPU path
        self.dp = (self.world > 1) if data_parallel is None else bool(data_parallel)
        if self.dp and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError('data_parallel=True needs an initialised torch.distributed process group')
        net = getattr(model, 'module', model)
        if self.world > 1:
            broadcast_parameters(net)                         # replicas start identical (DDP does the same at wrap time)
            runtime.seed_for_rank(torch.distributed.get_rank())   # ... but drop different elements (seeds are baked at capture)
        # flat mode: gradients go through ONE flat buffer and the optimizer is its own graph (data parallel: the all-reduce sits
        # in between; accumulation: micro-batches add into the buffer and only every accum_steps-th call runs the optimizer)
        self.flat_mode = self.dp or self.accum_steps > 1
        if self.mix and mask_weak is None and self.flat_mode:
            # without a weak mask mixup_data DROPS clips on some draws (a merged pair without events, the unlabelled remainder:
            # utilities/mixup.py:104-122): such a batch cannot go through the captured schedule, and bailing out of it on ONE rank
            # in the middle of an epoch would leave the other ranks waiting in the collective.  Refused up front
            raise ValueError('mix-up under the data-parallel / accumulation schedule needs mask_weak (an empty slice(B, B) will do): '
                             'mixup_data then keeps the batch size on every draw')
        self.grad_dtype = grad_dtype
        self._plan_segments(net, optimizer, dp_cuts if (self.dp and overlap_allreduce and device_matching) else 'none')
        dev = example_input.device
        self.dev = dev
        self._init_common(criterion, optimizer, dev)
        self.static_x = example_input.clone()
        self.static_patches = None if example_patches is None else example_patches.clone()
        if self.mix:
            # mix-up (engine.py:50-53) inside the graph: the raw batch is a second static buffer, the feature mixing its first
            # kernel (job records refreshed per replay), the merged targets + the strong | weak split arrive as table data
            from .utilities.mixup import job_table
            self.static_raw = example_input.clone().float().contiguous()
            self.static_x = torch.empty_like(self.static_raw)
            self._jobs = _Upload(16 * self.static_raw.shape[0], dev)
            self._jobs.send(job_table([(i, 0, 1, 0.0) for i in range(self.static_raw.shape[0])]))     # identity until the first call
        snap = _snapshot(net, optimizer)
        side = train_stream(dev)
        self._capture = dict(stream=side, **_CAPTURE)
        side.wait_stream(torch.cuda.current_stream())
        # gradients another stepper's captured backward left attached to the parameters would be ACCUMULATED into by the warm-up's
        # eager backward (and the stack-level weight-gradient launches need fresh tensors: ops.defer_layer_wgrads)
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.stream(side), optimizer.table_set(self._tabname), ops.defer_layer_wgrads():
            for _ in range(warmup):                          # eager steps: lazy inits (LDS attributes, optimizer state)
                if device_matching:
                    self._eager_device_step(example_targets, max_targets)
                else:
                    train_step(model, criterion, optimizer, self.static_x, example_targets, mask_weak, mask_strong, max_norm,
                               normalize, check_finite=False, allreduce=True, patches=self.static_patches)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _restore(net, optimizer, snap)
        self.nonfinite.zero_()
        criterion.last_total = None      # drop the warm-up autograd graphs (their AccumulateGrad nodes belong to `side`)
        optimizer.zero_grad(set_to_none=True)
        if self.flat_mode:
            optimizer.enable_flat_grads(grad_dtype)          # pinned staging + flat buffer: not allocatable during capture
            optimizer._flat_g.zero_()
        gc.collect()
        quiesce_collectives(self.dev)
        self.device_matching = device_matching
        self.g_fwd = torch.cuda.CUDAGraph()
        self.g_bwd = self.g_opt = None
        self.g_seg, self.flat_parts = [], []
        acc = self.accum_steps > 1
        with optimizer.table_set(self._tabname), ops.defer_layer_wgrads():
            if device_matching:
                self.tables = self._make_tables(example_targets, max_targets)
                with torch.cuda.graph(self.g_fwd, **self._capture):
                    self.static_out = self._forward()
                    self.static_dense = criterion.prepare_device(self.static_out, self.tables, normalize=normalize,
                                                                 fine_tune=fine_tune, fl=fl, ft_rand=ft_rand)
                    if self.segs is None:
                        self._backward_and_step()
                    else:
                        self.static_losses = self.criterion.compute(self.static_out, self.static_dense, self.fl)
                        self.static_total = self.criterion.last_total
                        with self.runtime.async_wgrad(self.async_wgrad), ops.coschedule(self.coschedule):
                            self.flat_parts.append(self._segment_backward(0, self.static_total, acc))
                for k in range(1, len(self.segs or [])):          # one graph per further segment of the backward
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=self.g_fwd.pool(), **self._capture):
                        with self.runtime.async_wgrad(self.async_wgrad), ops.coschedule(self.coschedule):
                            self.flat_parts.append(self._segment_backward(k, None, acc))
                    self.g_seg.append(g)
            else:
                with torch.cuda.graph(self.g_fwd, **self._capture):
                    self.static_out = self._forward()
                dense, _ = criterion.prepare(self.static_out, example_targets, mask_weak, mask_strong, normalize)
                self.meta = dense['_meta']
                self.static_pack = dense['_pack'].clone()
                self.static_dense = criterion.dense_views(self.static_pack, self.meta)
                self.g_bwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool(), **self._capture):
                    self._backward_and_step()
            if self.flat_mode:
                # the RCCL all-reduce(s) of the flat gradient buffer sit between the graphs; then the fused clip + AdamW reads
                # the averaged (accumulated) gradients from the flat buffer
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt, pool=self.g_fwd.pool(), **self._capture):
                    optimizer.step(max_norm=max_norm, from_flat=True)
                    if acc:
                        optimizer._flat_g.zero_()            # the next micro-batch adds into an empty buffer
        torch.cuda.synchronize()
        optimizer.flush_uploads()

    # ------------------------------------------------------------------ pieces
    def _make_tables(self, example_targets, max_targets):
        from .sedt import TargetTables
        if self.ms is None or self.ms.start not in (None, 0) or self.ms.step not in (None, 1):
            raise NotImplementedError('strong_mask must be slice(0, n)')
        B = len(example_targets)
        ns = len(example_targets[self.ms])
        n_lab = self.mw.stop if self.mw is not None else self.ms.stop
        return TargetTables(B, ns, n_lab, self.dev, max_targets=max_targets, dynamic_split=bool(self.mix), weak_mask_none=self.mw is None,
                            with_ratio=bool(self.mix) or any('ratio' in t for t in example_targets)).load(example_targets)

    def _forward(self):
        if self.mix:
            ops.mixup(self.static_raw, self.static_raw, self._jobs.dev_buf, out=self.static_x)
        if self.static_patches is not None:
            from .utilities.utils import _no_padding_mask
            mask = _no_padding_mask(self.static_x.shape[0], self.static_x.shape[2], self.static_x.shape[3], self.dev)
            return self.model((self.static_x, mask), self.static_patches)
        return self.model(self.static_x)

    def _eager_device_step(self, targets, max_targets):
        """one un-captured step through exactly the code the capture will run (device matching, fused losses)"""
        tables = self._make_tables(targets, max_targets)
        out = self._forward()
        dense = self.criterion.prepare_device(out, tables, normalize=self.normalize, fine_tune=self.fine_tune, fl=self.fl,
                                              ft_rand=self.ft_rand)
        self.criterion.compute(out, dense, self.fl)
        total = self.criterion.last_total
        self.criterion.last_total = None
        torch.autograd.backward(total, grad_tensors=self._root_grad(total))
        if self.flat_mode:
            self.optimizer.enable_flat_grads(self.grad_dtype)
            flat = self.optimizer.gather_grads()
            if self.dp:
                allreduce_mean(flat)
            self.optimizer.step(max_norm=self.max_norm, from_flat=True)
        else:
            self.optimizer.step(max_norm=self.max_norm)
        self.optimizer.zero_grad(set_to_none=True)

    def _backward_and_step(self):
        self.static_losses = self.criterion.compute(self.static_out, self.static_dense, self.fl)
        self.static_total = self.criterion.last_total
        # weight gradients ride in the spare workgroup slots of the dgrad chain's launches; drained on exit
        with self.runtime.async_wgrad(self.async_wgrad), ops.coschedule(self.coschedule):
            torch.autograd.backward(self.static_total, grad_tensors=self._root_grad(self.static_total))
        if not self.flat_mode:
            self.optimizer.step(max_norm=self.max_norm)
        else:                                                # all gradients -> one flat buffer (one launch)
            self.flat_parts = [self.optimizer.gather_grads(accumulate=self.accum_steps > 1)]

    def __call__(self, batch_input, targets, check_finite=False, patches=None):
        """one step on (batch_input, targets).  With mix-up the batch arrives UNMIXED in the layout the stepper was built with
        (strong clips, then weak ones); the np.random draws, the label bookkeeping and the new split happen here on the host
        (utilities.mixup.plan_mixup_data), the feature mixing inside the graph.  Keep the targets on the host for that."""
        split = {}
        if self.mix:
            from .utilities.mixup import draw_mixup_data, plan_mixup_data, job_table
            import numpy as np
            rng = np.random.get_state()
            lam, index = draw_mixup_data(len(targets), self.mix_alpha)
            jobs, mixed, n_strong, n_weak = plan_mixup_data(targets, self.ms, self.mw, lam, index, self.mix, self.max_events)
            if len(mixed) != len(targets):
                # mixup_data dropped clips (utilities/mixup.py:104-122 without a weak mask: a merged pair with no events at all
                # and the unlabelled remainder leave the batch) - the captured shapes do not hold for this batch: it takes the
                # eager step, with the np.random stream rewound so that mixup_data makes the same draws
                np.random.set_state(rng)
                return self._eager_batch(batch_input, targets, patches)
            targets = mixed
            self.static_raw.copy_(batch_input, non_blocking=True)
            self._jobs.send(job_table(jobs))
            split = dict(ns=n_strong, n_lab=n_strong + n_weak)
        else:
            self.static_x.copy_(batch_input, non_blocking=True)
        if self.static_patches is not None:
            self.static_patches.copy_(patches, non_blocking=True)
        if self._host_bump:
            self.runtime.bump_seed(self.dev)
        self._before_replay()
        if self.device_matching:
            self.tables.load(targets, **split)
            self.g_fwd.replay()
        else:
            self.g_fwd.replay()
            dense, _ = self.criterion.prepare(self.static_out, targets, self.mw, self.ms, self.normalize)
            if dense['_meta'] != self.meta:
                raise RuntimeError(f'batch composition changed: captured {self.meta}, got {dense["_meta"]}')
            self.static_pack.copy_(dense['_pack'], non_blocking=True)
            self.g_bwd.replay()
        if self.g_opt is not None:
            self._micro += 1
            last = self._micro % self.accum_steps == 0        # (engine.py:76: step on every accum_steps-th batch)
            works = []
            if self.dp and last:                              # RCCL reduces segment k while the graphs of k+1.. run
                works.append(allreduce_mean(self.flat_parts[0], async_op=True))
            for k, g in enumerate(self.g_seg):
                g.replay()
                if self.dp and last:
                    works.append(allreduce_mean(self.flat_parts[k + 1], async_op=True))
            for w in works:
                w.wait()
            if last:
                self.g_opt.replay()
        self._after_replay(check_finite)
        return self.static_total, self.static_losses


    def _eager_batch(self, batch_input, targets, patches=None):
        """one batch the captured graph cannot take, through engine.train_step on the training stream (same optimizer state: the
        device-side step count and moments are shared)"""
        if self.flat_mode:
            raise NotImplementedError('a batch that mix-up shrinks cannot bypass the captured data-parallel / accumulation schedule: '
                                      'pass mask_weak (mixup_data then keeps the batch size) or use the eager train_step')
        # the captured backward leaves its (static) gradient tensors attached to the parameters: an eager backward would ADD to them
        self.optimizer.zero_grad(set_to_none=True)
        side, cur = train_stream(self.dev), torch.cuda.current_stream(self.dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = train_step(self.model, self.criterion, self.optimizer, batch_input, targets, self.mw, self.ms, self.max_norm,
                             self.normalize, check_finite=True, patches=patches, fine_tune=self.fine_tune, fl=self.fl,
                             mix_up_ratio=self.mix)
        cur.wait_stream(side)
        return out


def broadcast_parameters(model, src=0):
    """every rank starts from rank ``src``'s parameters and buffers (what DistributedDataParallel does when it wraps a
    model, reference train_spsedt.py:157-158)"""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if dist.get_backend() == 'nccl' or not t.is_cuda:
                dist.broadcast(t, src)
            else:                                             # gloo in this image has no GPU tensors: stage through the host
                h = t.cpu()
                dist.broadcast(h, src)
                t.copy_(h)


# ---------------------------------------------------------------------------------------------------------------------
# Mean-teacher (semi-supervised) step - counterpart of the per-batch body of reference engine.semi_train (engine.py:117-181)
# ---------------------------------------------------------------------------------------------------------------------
def _tensors(x):
    return x.tensors if hasattr(x, 'tensors') else x


# ---------------------------------------------------------------------------------------------------- validation / prediction
def predict_step(model, criterion, postprocessor, batch_input, targets, fusion_strategy=(1,), at=True, threshold=0.5):
    """The per-batch body of reference engine.get_sedt_predictions (engine.py:244-285) on the device: no-grad forward, the
    losses the reference logs (criterion with strong_mask = the whole batch), the thresholded audio tags, and
    ``postprocessors['bbox']`` once per fusion strategy.  Returns (loss_dict, audio_tags or None, {at_m: (scores [B,Q], labels
    [B,Q], boxes [B,Q,2] in seconds)}); the host-side decoding into event lists (decoder.decode_strong, pandas) stays the
    caller's, as in the reference."""
    with torch.no_grad():
        outputs = model(batch_input)
        B = outputs['pred_logits'].shape[0]
        loss_dict, _ = criterion(outputs, targets, None, slice(B))
        sizes = torch.stack([t['orig_size'] for t in targets], dim=0)
        audio_tags = (outputs['at'] > 0.5).long() if at else None
        if at:
            assert 'at' in outputs
        results = {m: postprocessor.batched(outputs, sizes, audio_tags=audio_tags, at_m=m, threshold=threshold) for m in fusion_strategy}
    return loss_dict, audio_tags, results


class GraphedPredictStep(object):
    """predict_step as ONE HIP graph (forward, device Hungarian matching + fused losses for the logged validation losses, audio
    tags, PostProcess for every fusion strategy): per batch the host refreshes the static input / target tables and replays;
    the outputs are static tensors (overwritten by the next call).  Shapes are static like GraphedTrainStep's.  Weight packs
    follow parameter-pointer swaps (an EMA teacher evaluated through ``ema.apply_shadow()`` gets its own pack plan)."""

    def __init__(self, model, criterion, postprocessor, example_input, example_targets, fusion_strategy=(1,), at=True, threshold=0.5,
                 max_targets=32, warmup=2):
        from .sedt import TargetTables
        self.model, self.criterion, self.post = model, criterion, postprocessor
        self.fusion, self.at, self.threshold = tuple(fusion_strategy), at, threshold
        dev = example_input.device
        self.static_x = example_input.clone()
        B = len(example_targets)
        self.tables = TargetTables(B, B, B, dev, max_targets=max_targets, with_ratio=False, weak_mask_none=True).load(example_targets)
        self.sizes = torch.stack([t['orig_size'] for t in example_targets], dim=0).to(dev).float().clone()
        stream = train_stream(dev)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        quiesce_collectives(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=stream, **_CAPTURE):
            self.out = self._body()
        torch.cuda.synchronize()

    def _body(self):
        with torch.no_grad():
            outputs = self.model(self.static_x)
            dense = self.criterion.prepare_device(outputs, self.tables)
            losses = self.criterion.compute(outputs, dense)
            tags = (outputs['at'] > 0.5).long() if self.at else None
            res = {m: self.post.batched(outputs, self.sizes, audio_tags=tags, at_m=m, threshold=self.threshold) for m in self.fusion}
        return losses, tags, res

    def __call__(self, batch_input, targets):
        self.static_x.copy_(batch_input, non_blocking=True)
        self.tables.load(targets)
        self.sizes.copy_(torch.stack([t['orig_size'] for t in targets], dim=0), non_blocking=True)
        self.graph.replay()
        return self.out


def pseudo_label_tables(tea_outputs, classwise_threshold, orig_size, tables, counter=None, del_overlap=True):
    """engine.py:300-348 without leaving the device: teacher outputs -> the flat target tables (sedt.TargetTables) that the
    on-device matching reads.  ``orig_size``: clip duration in seconds (the minimum event length is 0.2 / orig_size)."""
    ops.pseudo_labels(tea_outputs['pred_logits'], tea_outputs['pred_boxes'], tea_outputs.get('at'), classwise_threshold,
                      0.2 / float(orig_size), tables.as_dict(), counter, del_overlap)
    return tables


@torch.no_grad()
def get_pseudo_labels(tea_outputs, postprocessor, orig_unlabel_target_sizes, target_unlabeled, pseudo_labels_counter,
                      threshold=0.5, del_overlap=True, classwise_threshold=None):
    """the reference's signature and return value (engine.py:300-348: the list of target dicts with 'labels' / 'boxes'
    replaced by the pseudo events, the Counter updated) - computed by ONE kernel (sedt_pseudo_labels) and ONE device->host
    copy (clip offsets + class counts)."""
    from .sedt import TargetTables
    logits = tea_outputs['pred_logits']
    B, Q, C1 = logits.shape
    dev = logits.device
    thr = classwise_threshold.to(device=dev, dtype=torch.float32).contiguous()
    tables = TargetTables(B, B, B, dev, max_targets=Q)
    counter = torch.zeros(C1 - 1, dtype=torch.int32, device=dev)
    pseudo_label_tables(tea_outputs, thr, orig_unlabel_target_sizes[0].item(), tables, counter, del_overlap)
    host = torch.cat([tables.off[:B + 1], counter]).cpu().numpy()
    off, cnt = host[:B + 1], host[B + 1:]
    for i in range(B):
        target_unlabeled[i]['labels'] = tables.lab_cat[off[i]:off[i + 1]].clone()
        target_unlabeled[i]['boxes'] = tables.box_cat[off[i]:off[i + 1]].clone()
    if del_overlap and pseudo_labels_counter is not None:
        pseudo_labels_counter.update({int(c): int(n) for c, n in enumerate(cnt) if n})
    return target_unlabeled


def semi_train_step(model, ema, criterion, optimizer, batch_input_teacher, batch_input_student, targets, mask_strong, mask_weak,
                    mask_label, mask_unlabel, classwise_threshold, fine_tune=False, normalize=False, fl=False, max_norm=0.1,
                    counter=None, check_finite=True, do_step=True, do_ema=True, mix_up_ratio=0):
    """one iteration of the reference's semi_train (engine.py:117-181): supervised loss on the labelled part -> teacher
    (EMA weights swapped in, no grad) on the unlabelled part -> pseudo labels -> student on the augmented unlabelled part ->
    ONE backward over both graphs -> clip / AdamW -> EMA update.  Returns (sup dict, unsup dict, total, pseudo targets).

    mix_up_ratio > 0 (engine.py:128-133, 150-153; train_ss_sedt.py --mix_up_ratio): the labelled part goes through mixup_data
    before its forward, and the student's unlabelled view + the pseudo labels through mixup_label_unlabel - mixed with the
    ALREADY MIXED labelled batch, as the reference passes it on - between the teacher and the student forward.  The pseudo
    targets returned are then the mixed ones (what the student's criterion saw)."""
    dev = _tensors(batch_input_teacher).device
    if dev.type == 'cuda' and torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
        side, cur = train_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = semi_train_step(model, ema, criterion, optimizer, batch_input_teacher, batch_input_student, targets, mask_strong,
                                  mask_weak, mask_label, mask_unlabel, classwise_threshold, fine_tune, normalize, fl, max_norm,
                                  counter, check_finite, do_step, do_ema, mix_up_ratio)
        cur.wait_stream(side)
        return out
    xt, xs = _tensors(batch_input_teacher), _tensors(batch_input_student)
    wd = criterion.weight_dict
    x_lab, t_lab = xt[mask_label], targets[mask_label]
    if mix_up_ratio > 0:
        from .utilities.mixup import mixup_data, mixup_label_unlabel
        x_lab, t_lab, mask_strong, mask_weak = mixup_data(x_lab, t_lab, mask_strong, mask_weak, mix_up_ratio=mix_up_ratio, alpha=1)
    sup, _ = criterion(model(x_lab), t_lab, mask_weak, mask_strong, fine_tune, normalize, fl)
    sup_total = criterion.last_total
    unl = [dict(t) for t in targets[mask_unlabel]]
    ema.apply_shadow()
    with torch.no_grad():
        tea = model(xt[mask_unlabel])
        sizes = torch.stack([t['orig_size'] for t in unl], dim=0)
        pseudo = get_pseudo_labels(tea, None, sizes, unl, counter, classwise_threshold=classwise_threshold)
    ema.restore()
    x_u = xs[mask_unlabel]
    if mix_up_ratio > 0:
        x_u, pseudo = mixup_label_unlabel(x_lab, x_u, t_lab, pseudo, alpha=1)
    unsup, _ = criterion(model(x_u), pseudo, None, slice(x_u.shape[0]), fine_tune, normalize, fl)
    total = sup_total + criterion.last_total
    criterion.last_total = None
    if check_finite and not math.isfinite(total.item()):
        raise FloatingPointError('Loss is infinite, stopping training')
    total.backward()
    if do_step:
        if isinstance(optimizer, FusedAdamW):
            optimizer.step(max_norm=max_norm)
        else:
            if max_norm > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    if do_ema:
        ema.update()
    return ({k: v.detach() for k, v in sup.items()}, {k: v.detach() for k, v in unsup.items()}, total.detach(), pseudo)


def _slice_outputs(out, sl):
    """the clips `sl` of a model output dict (stacked head outputs included)"""
    o = {k: v[sl] for k, v in out.items() if torch.is_tensor(v)}
    if 'aux_outputs' in out:
        o['aux_outputs'] = [{k: v[sl] for k, v in a.items()} for a in out['aux_outputs']]
    if '_stacked' in out:
        o['_stacked'] = tuple(t[:, sl] for t in out['_stacked'])
        o['_q0'] = out.get('_q0', 0)
    return o


def _split_outputs(out, n):
    """(clips [0, n), clips [n, B)) of a model output dict for two criterion calls: the tensors the fused criterion reads - the stacked
    head outputs and the audio-tag output - are split by ONE launch into contiguous parts (functional.SplitClipsFn: one more launch merges
    their gradients), everything else is a view"""
    from . import functional as Fn
    if '_stacked' not in out or not out['_stacked'][0].is_cuda:
        return _slice_outputs(out, slice(0, n)), _slice_outputs(out, slice(n, None))
    la, ba = out['_stacked']
    ts, dims = [la, ba], [1, 1]
    if out.get('at') is not None:
        ts.append(out['at'])
        dims.append(0)
    parts = Fn.SplitClipsFn.apply(n, tuple(dims), *ts)
    k = len(ts)
    res = []
    for half, sl in ((parts[:k], slice(0, n)), (parts[k:], slice(n, None))):
        q0 = out.get('_q0', 0)
        lg, bx = half[0], half[1]
        Q = lg.shape[2] - q0
        o = {'pred_logits': lg[-1][:, q0:q0 + Q], 'pred_boxes': bx[-1][:, q0:q0 + Q], '_stacked': (lg, bx), '_q0': q0}
        if k == 3:
            o['at'] = half[2]
        o['aux_outputs'] = [{'pred_logits': lg[i][:, q0:q0 + Q], 'pred_boxes': bx[i][:, q0:q0 + Q]} for i in range(lg.shape[0] - 1)]
        for key, v in out.items():          # whatever else the model returned (e.g. at_p): plain views
            if torch.is_tensor(v) and key not in o:
                o[key] = v[sl]
        res.append(o)
    return res[0], res[1]


class GraphedSemiStep(_GraphedBase):
    """The mean-teacher step (reference engine.py:117-181) as ONE HIP graph: labelled forward + device matching + fused
    loss, teacher forward through the EMA weights (no grad), pseudo labels written by ``sedt_pseudo_labels`` straight into the
    flat target tables, student forward on the augmented unlabelled clips, device matching against those tables, one
    backward over both autograd graphs, fused clip + AdamW, fused EMA update.  No device->host copy anywhere.

    Live parameters without re-capture: ``EMA.apply_shadow`` / ``restore`` swap ``param.data`` between two FIXED
    sets of tensors (the student parameters, updated in place by the optimizer, and the EMA shadow, updated in place by
    ``EMA.update``).  The graph is captured over both pointer sets - each forward prepares its weights through the plan of the
    pointer set it sees (packing.PlanSet) - so replays always read the current student / teacher values.  Loading a
    state_dict copies INTO those tensors and is seen as well; only re-allocating parameters needs a new stepper.

    The thresholds (``classwise_threshold``, f32 [C] on the device) are read by the graph on every replay: update them in
    place (``stepper.threshold.copy_(...)``) when the driver adjusts them per epoch (train_ss_sedt.py:207).
    ``counter`` (int32 [C]) accumulates the pseudo events per class like the reference's pseudo_labels_counter."""

    def __init__(self, *args, **kw):
        try:
            self._construct(*args, **kw)
        finally:
            self._release_shared()

    def _construct(self, model, ema, criterion, optimizer, x_teacher, x_student, targets, mask_strong, mask_weak, mask_label,
                 mask_unlabel, classwise_threshold, orig_size=10.0, fine_tune=False, normalize=False, fl=False, max_norm=0.1,
                 warmup=2, max_targets=32, accumulating_ema_steps=1, fuse_student_forwards=True, mix_up_ratio=0.0, mix_alpha=1,
                 max_events=20, accum_steps=1, overlap_allreduce=True, dp_cuts='coarse', grad_dtype=None, data_parallel=None):
        import gc
        from . import runtime
        from .sedt import TargetTables
        if not isinstance(optimizer, FusedAdamW):
            raise RuntimeError('GraphedSemiStep needs FusedAdamW')
        self.accum_steps, self.ema_steps, self._micro = int(accum_steps), int(accumulating_ema_steps), 0
        if self.accum_steps < 1 or self.ema_steps < 1:
            raise ValueError('accum_steps / accumulating_ema_steps >= 1')
        if self.accum_steps > 1 and grad_dtype == torch.bfloat16:
            raise ValueError('accumulating micro-batch gradients in a bf16 flat buffer loses them: use the f32 buffer')
        self.model, self.ema, self.criterion, self.optimizer, self.runtime = model, ema, criterion, optimizer, runtime
        self.ms, self.mw, self.ml, self.mu = mask_strong, mask_weak, mask_label, mask_unlabel
        self.flags = dict(normalize=normalize, fine_tune=fine_tune, fl=fl)
        self.fl, self.max_norm, self.orig_size = fl, max_norm, orig_size
        # the labelled clips and the unlabelled (student view) clips go through the student weights: ONE forward over their
        # concatenation instead of two (clips are independent end to end - FrozenBatchNorm, per-clip attention - so every clip's
        # outputs are what its own forward gives; the GEMMs see twice the rows, the step half the launches)
        self.fuse = fuse_student_forwards
        self.mix, self.mix_alpha, self.max_events = float(mix_up_ratio), mix_alpha, max_events
        xt, xs = _tensors(x_teacher), _tensors(x_student)
        dev = xt.device
        self.dev = dev
        self._init_common(criterion, optimizer, dev)
        dist = torch.distributed
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.dp = (self.world > 1) if data_parallel is None else bool(data_parallel)
        if self.dp and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('data_parallel=True needs an initialised torch.distributed process group')
        # flat mode (see GraphedTrainStep): gradients through the flat buffer, optimizer (and EMA update) as graphs of their own
        self.flat_mode = self.dp or self.accum_steps > 1 or self.ema_steps > 1
        self.grad_dtype = grad_dtype
        self._plan_segments(model, optimizer, dp_cuts if (self.dp and overlap_allreduce and fuse_student_forwards) else 'none')
        if self.world > 1:
            broadcast_parameters(model)
            with torch.no_grad():
                for n, p in model.named_parameters():         # the teacher starts identical on every rank too
                    if n in ema.shadow:
                        h = ema.shadow[n] if dist.get_backend() == 'nccl' else ema.shadow[n].cpu()
                        dist.broadcast(h, 0)
                        ema.shadow[n].copy_(h)
            runtime.seed_for_rank(dist.get_rank())
        n_lab_clips = xt[mask_label].shape[0]
        self.x_cat = torch.cat([xt[mask_label], xs[mask_unlabel]])          # static input of the student forward(s)
        self.x_lab, self.x_stu = self.x_cat[:n_lab_clips], self.x_cat[n_lab_clips:]
        self.x_tea = xt[mask_unlabel].clone()
        self.threshold = classwise_threshold.to(device=dev, dtype=torch.float32).clone()
        lab_t = targets[mask_label]
        n_l, n_u = self.x_lab.shape[0], self.x_stu.shape[0]
        ns = len(lab_t[mask_strong])
        n_lab = mask_weak.stop if mask_weak is not None else mask_strong.stop
        Q = model.num_queries
        self.counter = torch.zeros(criterion.num_classes, dtype=torch.int32, device=dev)
        self.tab_l = TargetTables(n_l, ns, n_lab, dev, max_targets=max_targets, dynamic_split=bool(self.mix), weak_mask_none=mask_weak is None,
                                  with_ratio=bool(self.mix) or any('ratio' in t for t in lab_t)).load(lab_t)
        self.tab_u = TargetTables(n_u, n_u, n_u, dev, max_targets=max(Q, 1), weak_mask_none=True)      # engine.py:159: weak_mask None
        if self.mix:
            # mix-up inside the step (engine.py:128-133, 150-153): raw labelled / unlabelled-student clips are static inputs, the two
            # feature mixings are kernels of the graph writing the student's input x_cat; the labelled targets arrive mixed from
            # the host (plan_mixup_data), the unlabelled ones are merged ON the device with the pseudo labels the graph itself
            # produces (ops.mixup_targets) - nothing leaves the device between teacher and student forward
            from .utilities.mixup import job_table
            self.mix_num_u = int(n_l * 0.5)                   # mixup_label_unlabel's default mix_up_ratio (engine.py:150 passes none)
            if self.mix_num_u > n_u:
                raise ValueError(f'mixup_label_unlabel mixes {self.mix_num_u} unlabelled clips, the batch has {n_u}')
            self.x_lab_raw, self.x_stu_raw = self.x_lab.clone(), self.x_stu.clone()
            self.tab_p = self.tab_u                           # pseudo labels as the teacher gives them
            self.tab_u = TargetTables(n_u, n_u, n_u, dev, max_targets=max(max_targets, max_events, Q), with_ratio=True, weak_mask_none=True)
            self._jobs_l = _Upload(16 * n_l, dev)
            self._jobs_l.send(job_table([(i, 0, 1, 0.0) for i in range(n_l)]))
            self._lam_u = _Upload(8, dev)
            self._lam_u.send(torch.from_numpy(np.asarray([1.0, 0.0], np.float32).view(np.uint8).copy()))
            self.jobs_u = torch.zeros(16 * n_u, dtype=torch.uint8, device=dev)
        snap = _snapshot(model, optimizer, ema)
        side = train_stream(dev)
        self._capture = dict(stream=side, **_CAPTURE)
        side.wait_stream(torch.cuda.current_stream())
        optimizer.zero_grad(set_to_none=True)                 # (see GraphedTrainStep: no stale .grad under the warm-up's eager backward)
        with torch.cuda.stream(side), optimizer.table_set(self._tabname), ops.defer_layer_wgrads():
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _restore(model, optimizer, snap, ema)
        self.nonfinite.zero_()
        self.counter.zero_()
        # drop every handle on the warm-up autograd graphs: their AccumulateGrad nodes belong to `side`; kept alive, the
        # captured backward would accumulate the two forwards' gradients on that stream, outside the capture's ordering
        criterion.last_total = None
        self.sup = self.unsup = self.total = None
        optimizer.zero_grad(set_to_none=True)
        gc.collect()
        quiesce_collectives(self.dev)
        if self.flat_mode:
            optimizer.enable_flat_grads(grad_dtype)
            optimizer._flat_g.zero_()
        self.graph = torch.cuda.CUDAGraph()
        self.g_opt = self.g_ema = None
        self.g_seg, self.flat_parts = [], []
        with optimizer.table_set(self._tabname), ops.defer_layer_wgrads():
            with torch.cuda.graph(self.graph, **self._capture):
                self._body(part='fwd_bwd' if self.flat_mode else 'all')
            for k in range(1, len(self.segs or [])):          # data parallel: one graph per further segment of the backward
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.graph.pool(), **self._capture):
                    self.flat_parts.append(self._segment_backward(k, None, self.accum_steps > 1))
                self.g_seg.append(g)
            if self.flat_mode:                                # (RCCL all-reduces of the flat gradient segments in between)
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt, pool=self.graph.pool(), **self._capture):
                    self._body(part='update')
                self.g_ema = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_ema, pool=self.graph.pool(), **self._capture):
                    self.ema.update()
        torch.cuda.synchronize()
        optimizer.flush_uploads()

    def _body(self, part='all'):
        if part == 'update':
            self.optimizer.step(max_norm=self.max_norm, from_flat=True)
            if self.accum_steps > 1:
                self.optimizer._flat_g.zero_()
            return
        crit, model = self.criterion, self.model
        if self.mix:
            ops.mixup(self.x_lab_raw, self.x_lab_raw, self._jobs_l.dev_buf, out=self.x_lab)
        if not self.fuse:
            out_l = model(self.x_lab)
            self.sup = crit.compute(out_l, crit.prepare_device(out_l, self.tab_l, **self.flags), self.fl)
            total_l = crit.last_total
        self.ema.apply_shadow()
        try:
            with torch.no_grad():
                tea = model(self.x_tea)
        finally:
            self.ema.restore()
        if self.mix:
            pseudo_label_tables(tea, self.threshold, self.orig_size, self.tab_p, self.counter)
            ops.mixup_targets(self.tab_l, self.tab_p, self._lam_u.dev_buf.view(torch.float32), self.mix_num_u, self.tab_u, self.jobs_u,
                              self.max_events)
            ops.mixup(self.x_lab, self.x_stu_raw, self.jobs_u, out=self.x_stu)
        else:
            pseudo_label_tables(tea, self.threshold, self.orig_size, self.tab_u, self.counter)
        if self.fuse:
            out = model(self.x_cat)
            n = self.x_lab.shape[0]
            out_l, out_s = _split_outputs(out, n)
            self.sup = crit.compute(out_l, crit.prepare_device(out_l, self.tab_l, **self.flags), self.fl)
            total_l = crit.last_total
        else:
            out_s = model(self.x_stu)
        self.unsup = crit.compute(out_s, crit.prepare_device(out_s, self.tab_u, **self.flags), self.fl)
        self.total = total_l + crit.last_total
        crit.last_total = None
        if part == 'fwd_bwd' and self.segs is not None:
            self.flat_parts = [self._segment_backward(0, self.total, self.accum_steps > 1)]
            return
        torch.autograd.backward(self.total, grad_tensors=self._root_grad(self.total))
        if part == 'fwd_bwd':
            self.flat_parts = [self.optimizer.gather_grads(accumulate=self.accum_steps > 1)]
            return
        if self.flat_mode:                                    # (eager warm-up of the flat-buffer schedule)
            self.optimizer.enable_flat_grads(self.grad_dtype)
            flat = self.optimizer.gather_grads()
            if self.dp:
                allreduce_mean(flat)
            self.optimizer.step(max_norm=self.max_norm, from_flat=True)
        else:
            self.optimizer.step(max_norm=self.max_norm)
        self.optimizer.zero_grad(set_to_none=True)
        self.ema.update()

    def __call__(self, x_teacher, x_student, targets, check_finite=False):
        xt, xs = _tensors(x_teacher), _tensors(x_student)
        self.x_tea.copy_(xt[self.mu], non_blocking=True)
        if self.mix:
            from .utilities.mixup import draw_mixup_data, draw_mixup_label_unlabel, plan_mixup_data, job_table, lam_pair
            lab_t = targets[self.ml]
            lam, index = draw_mixup_data(len(lab_t), self.mix_alpha)            # np.random in the reference's order (mixup.py:22-29,
            lam_u = draw_mixup_label_unlabel(self.mix_alpha)                    # then :141; nothing else draws in between)
            jobs, lab_t, n_strong, n_weak = plan_mixup_data(lab_t, self.ms, self.mw, lam, index, self.mix, self.max_events)
            self.x_lab_raw.copy_(xt[self.ml], non_blocking=True)
            self.x_stu_raw.copy_(xs[self.mu], non_blocking=True)
            self._jobs_l.send(job_table(jobs))
            self._lam_u.send(torch.from_numpy(lam_pair(lam_u).view(np.uint8).copy()))
            self.tab_l.load(lab_t, ns=n_strong, n_lab=n_strong + n_weak)
        else:
            self.x_lab.copy_(xt[self.ml], non_blocking=True)
            self.x_stu.copy_(xs[self.mu], non_blocking=True)
            self.tab_l.load(targets[self.ml])
        if self._host_bump:
            self.runtime.bump_seed(self.dev)
        self._before_replay()
        self.graph.replay()
        if self.g_opt is not None:
            self._micro += 1
            last = self._micro % self.accum_steps == 0        # engine.py:174 / :180: optimizer and EMA on their own periods
            works = []
            if self.dp and last:
                works.append(allreduce_mean(self.flat_parts[0], async_op=True))
            for k, g in enumerate(self.g_seg):
                g.replay()
                if self.dp and last:
                    works.append(allreduce_mean(self.flat_parts[k + 1], async_op=True))
            for w in works:
                w.wait()
            if last:
                self.g_opt.replay()
            if self._micro % self.ema_steps == 0:
                self.g_ema.replay()
        self._after_replay(check_finite)
        return self.total, self.sup, self.unsup
