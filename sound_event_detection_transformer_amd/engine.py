"""The per-batch body of the reference's ``engine.train`` loop (engine.py:56-80) as one function."""
import math

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
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h.div_(world))
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
    return _Done() if async_op else flat


_step_streams = {}


def train_step(model, criterion, optimizer, batch_input, targets, mask_weak=None, mask_strong=None, max_norm=0.1,
               normalize=False, check_finite=True, patches=None, allreduce=False):
    """forward -> SetCriterion -> weighted sum over weight_dict -> backward -> clip_grad_norm_(max_norm) -> step ->
    zero_grad.  Raises on a non-finite loss (the reference calls sys.exit(1), engine.py:70-73).

    When called on the default stream the step runs on a dedicated side stream (ordered after / before the caller's
    stream): a backward pass executed on the DEFAULT stream leaves the parameters' gradient accumulators tied to it, and
    a later HIP-graph capture of the backward (GraphedTrainStep) is then invalidated - ROCm 7.2 faults in
    hipStreamEndCapture instead of reporting it."""
    first = batch_input if torch.is_tensor(batch_input) else (getattr(batch_input, 'tensors', None) if not isinstance(
        batch_input, (tuple, list)) else (batch_input[0] if len(batch_input) and torch.is_tensor(batch_input[0]) else None))
    dev = first.device if torch.is_tensor(first) else None
    if dev is not None and dev.type == 'cuda' and torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
        key = str(dev)
        if key not in _step_streams:
            _step_streams[key] = torch.cuda.Stream(device=dev)
        side, cur = _step_streams[key], torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            out = train_step(model, criterion, optimizer, batch_input, targets, mask_weak, mask_strong, max_norm, normalize,
                             check_finite, patches, allreduce)
        cur.wait_stream(side)
        return out
    outputs = model(batch_input, patches) if patches is not None else model(batch_input)
    loss_dict, _ = criterion(outputs, targets, mask_weak, mask_strong, False, normalize)
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


class GraphedTrainStep(object):
    """The training step as HIP graphs.

    device_matching=True (default): ONE graph holds the whole step - model forward, the Hungarian matching of every
    (decoder layer, clip) on the device (ops.match_targets), the fused loss kernel, backward and the fused clip/AdamW.
    Per batch the host only refreshes the static input and the flat target tables (asynchronous copies) and replays;
    nothing depends on a device->host copy, so the host runs ahead of the GPU and the GPU never waits for it.

    device_matching=False: the reference's split - graph A (forward), SetCriterion.prepare on the host (one D2H copy,
    batched C++ Hungarian, one H2D copy), graph B (loss + backward + optimizer).

    Before constructing one, drop every tensor that still carries an autograd graph of an earlier EAGER backward on the
    default stream (loss dicts, model outputs): live AccumulateGrad nodes of another stream invalidate the capture of the
    backward (PyTorch warns 'AccumulateGrad node's stream does not match'; ROCm then faults in hipStreamEndCapture).

    Shapes are static: every batch must have the batch size, clip length and strong/weak split it was captured with (and,
    for the host split, the same per-clip target counts).  Dropout masks change on each replay through the device-side
    seed word (runtime.bump_seed); the Adam step count lives on the device too.

    Data parallel (world > 1): gradients are packed into ONE flat f32 buffer and averaged with RCCL between the backward
    graph and an optimizer graph.  With a SEDT backbone the backward is cut after layer3 (overlap_allreduce=True): the
    first graph ends with the gradients of everything above the cut (transformer, heads, layer4, layer3 = 90 % of the
    bytes) packed into the head of the flat buffer, their all-reduce is launched asynchronously, and a second graph runs
    the backward of layer2 / layer1 / stem meanwhile; the small tail is reduced after it.

    Two alternatives for the weight gradients are implemented and measured slower than the default (a layer's wgrads as ONE
    grouped launch on the main stream): async_wgrad=True issues them as a parallel branch of the graph (ROCm 7.2 pays
    50-100 us per cross-queue dependency: 7.6 vs 7.3 ms/step), coschedule=True lets them ride in the spare workgroup
    slots of later dgrad launches (ops.WgradPool; 7.3 vs 7.2 ms/step)."""

    def __init__(self, model, criterion, optimizer, example_input, example_targets, mask_weak=None, mask_strong=None,
                 max_norm=0.1, normalize=False, warmup=3, device_matching=True, max_targets=32, async_wgrad=False,
                 overlap_allreduce=True, coschedule=False, data_parallel=None):
        import gc
        from . import runtime
        from .sedt import TargetTables
        if not isinstance(optimizer, FusedAdamW):
            raise RuntimeError('GraphedTrainStep needs FusedAdamW (device-side step count, one pointer table)')
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.mw, self.ms, self.max_norm, self.normalize = mask_weak, mask_strong, max_norm, normalize
        self.runtime = runtime
        self.async_wgrad = async_wgrad
        self.coschedule = coschedule and not async_wgrad
        self.world = torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                            and torch.distributed.is_initialized()) else 1
        # data-parallel overlap: parameters whose gradients come last (stem conv0 + layer2) go to the tail of the flat layout
        self.cut_body = None
        # data_parallel=True forces the data-parallel schedule (flat gradients, all-reduce, optimizer graph) even in a
        # one-process group: lets a single GPU exercise the RCCL calls of the multi-GPU path
        self.dp = (self.world > 1) if data_parallel is None else bool(data_parallel)
        if self.dp and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError('data_parallel=True needs an initialised torch.distributed process group')
        if self.dp and overlap_allreduce and device_matching:
            net = getattr(model, 'module', model)
            body = getattr(getattr(net, 'backbone', [None])[0], 'body', None) if hasattr(net, 'backbone') else None
            if body is not None and hasattr(body, 'stage_out') and optimizer._static is None:
                tail = [p for n, p in body.named_parameters() if p.requires_grad and (n.startswith('conv0.') or n.startswith('layer2.'))]
                if tail:
                    optimizer.set_tail_params(tail)
                    self.cut_body = body
        dev = example_input.device
        self.dev = dev
        self.static_x = example_input.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                          # eager steps: lazy inits (LDS attributes, optimizer state)
                train_step(model, criterion, optimizer, self.static_x, example_targets, mask_weak, mask_strong, max_norm,
                           normalize, check_finite=False, allreduce=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        criterion.last_total = None      # drop the warm-up autograd graphs (their AccumulateGrad nodes belong to `side`)
        if self.cut_body is not None:
            self.cut_body.keep_stage_out = True              # the capture below needs the layer2 output as the cut tensor
        optimizer.zero_grad(set_to_none=True)
        if self.dp:
            optimizer.enable_flat_grads()                    # pinned staging + flat buffer: not allocatable during capture
        gc.collect()
        self.device_matching = device_matching
        self.g_fwd = torch.cuda.CUDAGraph()
        self.g_bwd = self.g_opt = self.g_low = None
        if device_matching:
            if mask_strong is None or mask_strong.start not in (None, 0) or mask_strong.step not in (None, 1):
                raise NotImplementedError('strong_mask must be slice(0, n)')
            B = len(example_targets)
            ns = len(example_targets[mask_strong])
            n_lab = mask_weak.stop if mask_weak is not None else mask_strong.stop
            self.tables = TargetTables(B, ns, n_lab, dev, max_targets=max_targets,
                                       with_ratio=any('ratio' in t for t in example_targets)).load(example_targets)
            with torch.cuda.graph(self.g_fwd):
                self.static_out = model(self.static_x)
                self.static_dense = criterion.prepare_device(self.static_out, self.tables)
                if self.cut_body is None:
                    self._backward_and_step()
                else:
                    self._backward_above_cut()
            if self.cut_body is not None:
                self.g_low = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_low, pool=self.g_fwd.pool()):
                    self._backward_below_cut()
        else:
            with torch.cuda.graph(self.g_fwd):
                self.static_out = model(self.static_x)
            dense, _ = criterion.prepare(self.static_out, example_targets, mask_weak, mask_strong, normalize)
            self.meta = dense['_meta']
            self.static_pack = dense['_pack'].clone()
            self.static_dense = criterion.dense_views(self.static_pack, self.meta)
            self.g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool()):
                self._backward_and_step()
        if self.dp:
            # data parallel: ONE RCCL all-reduce of the flat gradient buffer between the graphs, then the fused
            # clip + AdamW reads the averaged gradients from the flat buffer
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_fwd.pool()):
                optimizer.step(max_norm=max_norm, from_flat=True)
        torch.cuda.synchronize()

    def _backward_above_cut(self):
        """loss + backward down to the output of layer2; gradients of all parameters above -> head of the flat buffer"""
        self.static_losses = self.criterion.compute(self.static_out, self.static_dense)
        self.static_total = self.criterion.last_total
        head, self._tail = self.optimizer.head_tail_params()
        self._cut = self.cut_body.stage_out[1]
        with self.runtime.async_wgrad(self.async_wgrad), ops.coschedule(self.coschedule):
            grads = torch.autograd.grad(self.static_total, [self._cut] + head)
        self._g_cut = grads[0]
        for p, g in zip(head, grads[1:]):
            p.grad = g
        self.flat_head = self.optimizer.gather_grads('head')

    def _backward_below_cut(self):
        with self.runtime.async_wgrad(self.async_wgrad), ops.coschedule(self.coschedule):
            grads = torch.autograd.grad(self._cut, self._tail, grad_outputs=self._g_cut)
        for p, g in zip(self._tail, grads):
            p.grad = g
        self.flat_tail = self.optimizer.gather_grads('tail')

    def _backward_and_step(self):
        self.static_losses = self.criterion.compute(self.static_out, self.static_dense)
        self.static_total = self.criterion.last_total
        # weight gradients ride in the spare workgroup slots of the dgrad chain's launches; drained on exit
        with self.runtime.async_wgrad(self.async_wgrad), ops.coschedule(self.coschedule):
            self.static_total.backward()
        if not self.dp:
            self.optimizer.step(max_norm=self.max_norm)
        else:
            self.flat_g = self.optimizer.gather_grads()      # all gradients -> one flat buffer (one launch)

    def __call__(self, batch_input, targets, check_finite=False):
        self.static_x.copy_(batch_input, non_blocking=True)
        self.runtime.bump_seed(self.dev)
        if self.device_matching:
            self.tables.load(targets)
            self.g_fwd.replay()
        else:
            self.g_fwd.replay()
            dense, _ = self.criterion.prepare(self.static_out, targets, self.mw, self.ms, self.normalize)
            if dense['_meta'] != self.meta:
                raise RuntimeError(f'batch composition changed: captured {self.meta}, got {dense["_meta"]}')
            self.static_pack.copy_(dense['_pack'], non_blocking=True)
            self.g_bwd.replay()
        if self.g_low is not None:
            w1 = allreduce_mean(self.flat_head, async_op=True)     # RCCL reduces the head while the tail's backward runs
            self.g_low.replay()
            w2 = allreduce_mean(self.flat_tail, async_op=True)
            w1.wait()
            w2.wait()
            self.g_opt.replay()
        elif self.g_opt is not None:
            allreduce_mean(self.flat_g)
            self.g_opt.replay()
        if check_finite:
            v = self.static_total.item()
            if not math.isfinite(v):
                raise FloatingPointError(f'Loss is {v}, stopping training')
        return self.static_total, self.static_losses
